"""GPU (-m gpu): the C ABI driven from a plain C program (tools/c_harness/vr_harness.c, built with gcc by
vegs_amd/build.py) -- no Python, no torch in the process that calls the library.  Outputs are compared with
the oracle exactly like the torch-side parity tests: images and radii bit-exact, gradients within tolerance."""
import os
import subprocess

import numpy as np
import pytest

from helpers import assert_grad_close, ill_conditioned, oracle_cam

pytestmark = pytest.mark.gpu


def test_plain_c_program_reproduces_the_oracle(tmp_path):
    import torch
    from oracle import oracle as orc
    from vegs_amd import build, scenes
    assert torch.cuda.is_available()
    exe = build.build_c_harness()
    sc, deg = scenes.scene_street(P=20_000, length=60.0, sh_degree=3, seed=4)
    cam = scenes.kitti_camera(0.0, 0.3, 688, 188)
    H, W, P, M = cam.image_height, cam.image_width, 20_000, 16
    rng = np.random.default_rng(8)
    gc, gq, gs = (rng.normal(size=(k, H, W)).astype(np.float32) for k in (3, 4, 3))
    case, outp = tmp_path / "case.bin", tmp_path / "out.bin"
    with open(case, "wb") as f:
        np.array([P, M, H, W, deg, 1], np.int32).tofile(f)
        np.array([cam.tanfovx, cam.tanfovy, 1.0], np.float32).tofile(f)
        for a in (np.zeros(3), cam.world_view_transform, cam.full_proj_transform, cam.camera_center,
                  sc["means3D"], sc["shs"], sc["opacities"], sc["scales"], sc["rotations"], gc, gq, gs):
            np.ascontiguousarray(a, np.float32).tofile(f)
    r = subprocess.run([exe, str(case), str(outp)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    N = H * W
    with open(outp, "rb") as f:
        rd = lambda n, dt=np.float32: np.fromfile(f, dt, n)
        color, depth, quat, scale, alpha = rd(3 * N), rd(N), rd(4 * N), rd(3 * N), rd(N)
        radii, (R, V) = rd(P, np.int32), rd(2, np.int64)
        grads = {"means3D": rd(3 * P), "means2D": rd(3 * P), "shs": rd(3 * M * P), "opacities": rd(P), "scales": rd(3 * P),
                 "rotations": rd(4 * P)}
    oc = oracle_cam(cam, [0, 0, 0], deg)
    o, st = orc.forward(oc, sc["means3D"], sc["shs"], None, sc["opacities"], sc["scales"], sc["rotations"], None)
    # (num_visible counts the Gaussians that HAVE list entries: with tight tile lists a few with radii > 0 reach no tile)
    assert np.array_equal(radii, o["radii"]) and V == int((st["tiles_touched"] > 0).sum()) and R == int(st["R"])
    assert V <= int((o["radii"] > 0).sum())
    for got, key in ((color, "color"), (depth, "depth"), (quat, "cov_quat"), (scale, "cov_scale"), (alpha, "alpha")):
        assert np.array_equal(got, o[key].ravel()), key               # bit-exact, as through the torch binding
    og = orc.backward(oc, st, gc, None, gq, gs, None)
    ill, explain = ill_conditioned(st)      # rows beyond 10x the allowance must be edge-on discs (conditioning printed)
    for k, g in grads.items():
        assert_grad_close(k, g.reshape(og[k].shape), og[k], explain=explain, ill=ill)


def test_plain_c_program_accumulates_in_place(tmp_path):
    """VR_FLAG_ACCUMULATE_GRADS through the C ABI alone: a second vr_backward of the same view on the same gradient arrays
    doubles every row the view renders, bit for bit (x + x is exact; both calls in the deterministic mode), leaves the rows of
    culled Gaussians as the first call wrote them (zeros), and overwrites dL_dmeans2D as always."""
    import torch
    from vegs_amd import build, scenes
    assert torch.cuda.is_available()
    exe = build.build_c_harness()
    P, M = 12_000, 16
    sc, deg = scenes.scene_street(P=P, length=60.0, sh_degree=3, seed=14)
    cam = scenes.kitti_camera(0.0, 0.3, 344, 94)
    H, W = cam.image_height, cam.image_width
    rng = np.random.default_rng(18)
    gc, gq, gs = (rng.normal(size=(k, H, W)).astype(np.float32) for k in (3, 4, 3))
    case, outp = tmp_path / "case.bin", tmp_path / "out.bin"
    with open(case, "wb") as f:
        np.array([P, M, H, W, deg, 2], np.int32).tofile(f)
        np.array([cam.tanfovx, cam.tanfovy, 1.0], np.float32).tofile(f)
        for a in (np.zeros(3), cam.world_view_transform, cam.full_proj_transform, cam.camera_center,
                  sc["means3D"], sc["shs"], sc["opacities"], sc["scales"], sc["rotations"], gc, gq, gs):
            np.ascontiguousarray(a, np.float32).tofile(f)
    r = subprocess.run([exe, str(case), str(outp)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    N = H * W
    with open(outp, "rb") as f:
        rd = lambda n, dt=np.float32: np.fromfile(f, dt, n)
        rd(12 * N)
        radii = rd(P, np.int32)
        rd(2, np.int64)
        sizes = (("means3D", 3), ("means2D", 3), ("shs", 3 * M), ("opacities", 1), ("scales", 3), ("rotations", 4))
        first = {k: rd(n * P).reshape(P, n) for k, n in sizes}
        second = {k: rd(n * P).reshape(P, n) for k, n in sizes}
    vis = radii > 0
    assert vis.sum() > 1000 and (~vis).sum() > 100
    for k, _ in sizes:
        if k == "means2D":
            assert np.array_equal(second[k], first[k])
            continue
        assert np.array_equal(second[k], first[k] + first[k]), k
        assert np.abs(first[k][vis]).max() > 0 and not second[k][~vis].any(), k
