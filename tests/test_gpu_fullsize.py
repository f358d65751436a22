"""GPU (-m gpu): BASELINE.json's full-size configurations.

C2 (500k Gaussians, 1376x376, SH 3) is still small enough for the oracle: forward bit-exact and
index-exact against it.  C3 (2M Gaussians, the headline config) is checked through size-independent
properties: run-to-run determinism of the forward, structure and sortedness of the tile lists,
exact background linearity, linearity of the backward in the upstream gradients.
"""
import numpy as np
import pytest
import torch

from helpers import OUT_NAMES, oracle_cam, rel_err
from test_gpu_parity import _export_binning, _run_hip, _settings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from vegs_amd import _capi
    _capi.load()
    return torch.device("cuda:0")


def _inputs(sc):
    return dict(means3D=sc["means3D"], shs=sc["shs"], colors_precomp=None, opacities=sc["opacities"],
                scales=sc["scales"], rotations=sc["rotations"], cov3D_precomp=None)


def test_c2_500k_forward_matches_oracle_bit_exact(dev):
    from oracle import oracle as orc
    from vegs_amd import scenes
    sc, deg = scenes.scene_street(P=500_000, length=120.0, sh_degree=3, seed=1)
    cam = scenes.kitti_camera(0.0, 0.3, 1376, 376)
    oc = oracle_cam(cam, [0, 0, 0], deg)
    o_out, st = orc.forward(oc, **_inputs(sc))
    gouts = [np.random.default_rng(5).normal(size=s).astype(np.float32) * 1e-3
             for s in [(3, 376, 1376), (1, 376, 1376), (4, 376, 1376), (3, 376, 1376), (1, 376, 1376)]]
    h_out, h_grads, res = _run_hip(_settings(cam, [0, 0, 0], deg, 1.0, dev), _inputs(sc), dev, gouts)
    assert np.array_equal(h_out["radii"], o_out["radii"])
    pl, rg = _export_binning(res, 376, 1376, dev)
    assert np.array_equal(rg, st["ranges"]) and np.array_equal(pl, st["point_list"])
    for n in OUT_NAMES:
        assert np.array_equal(h_out[n], o_out[n]), n
    o_grads = orc.backward(oc, st, *gouts)
    for k in ("means3D", "shs", "opacities", "scales", "rotations", "means2D"):
        assert rel_err(h_grads[k], o_grads[k]) < 1e-3, (k, rel_err(h_grads[k], o_grads[k]))


@pytest.fixture(scope="module")
def c3(dev):
    from vegs_amd import scenes
    sc, deg = scenes.scene_street(P=2_000_000, length=250.0, sh_degree=3, seed=2)
    cam = scenes.kitti_camera(20.0, -0.3, 1376, 376)
    T = {k: torch.tensor(v, device=dev) for k, v in sc.items()}
    return sc, deg, cam, T


def _fwd(T, cam, deg, bg, dev, requires_grad=False):
    from diff_gaussian_rasterization import GaussianRasterizer
    t = {k: v.clone().requires_grad_(requires_grad) for k, v in T.items()}
    m2d = torch.zeros(t["means3D"].shape[0], 3, device=dev, requires_grad=requires_grad)
    rast = GaussianRasterizer(raster_settings=_settings(cam, bg, deg, 1.0, dev))
    res = rast(means3D=t["means3D"], means2D=m2d, shs=t["shs"], opacities=t["opacities"], scales=t["scales"],
               rotations=t["rotations"])
    return res, t, m2d


def test_c3_forward_is_deterministic_and_bg_linear(c3, dev):
    sc, deg, cam, T = c3
    with torch.no_grad():
        a, *_ = _fwd(T, cam, deg, [0, 0, 0], dev)
        b, *_ = _fwd(T, cam, deg, [0, 0, 0], dev)
        c, *_ = _fwd(T, cam, deg, [0.25, 0.5, 1.0], dev)
    for x, y in zip(a, b):
        assert torch.equal(x, y)                    # bitwise run-to-run determinism (no atomics in forward)
    for i in (1, 2, 3, 4, 5):
        assert torch.equal(a[i], c[i])              # background only enters the colour image
    bg = torch.tensor([0.25, 0.5, 1.0], device=dev)[:, None, None]
    assert (c[0] - (a[0] + (1 - a[4]) * bg)).abs().max().item() < 1e-6
    assert 0.5 < a[4].mean().item() <= 1.0 and torch.isfinite(a[0]).all() and torch.isfinite(a[2]).all()


def test_c3_tile_lists_are_complete_and_depth_sorted(c3, dev):
    sc, deg, cam, T = c3
    t = {k: v.clone().requires_grad_(True) for k, v in T.items()}
    res, t, m2d = _fwd(T, cam, deg, [0, 0, 0], dev, requires_grad=True)
    pl, rg = _export_binning(res, 376, 1376, dev)
    R = res[0].grad_fn.num_rendered
    radii = res[5].cpu().numpy()
    assert R == len(pl) > 1_000_000
    nonempty = rg[rg[:, 1] > rg[:, 0]]
    assert (nonempty[:, 1] - nonempty[:, 0]).sum() == R          # ranges partition [0, R)
    order = np.argsort(nonempty[:, 0])
    assert nonempty[order][0, 0] == 0 and np.array_equal(nonempty[order][1:, 0], nonempty[order][:-1, 1])
    assert np.all(radii[pl] > 0)                                  # only visible Gaussians are listed
    # tiles_touched = rect area: each Gaussian appears once per touched tile
    counts = np.bincount(pl, minlength=len(radii))
    assert counts[radii == 0].sum() == 0
    # depth order inside every tile (float64 view depth; ties/rounding tolerance 1e-5 relative)
    V = cam.world_view_transform.astype(np.float64)
    z = sc["means3D"].astype(np.float64) @ V[:3, 2] + V[3, 2]
    zl = z[pl]
    tile_of = np.repeat(np.arange(len(rg)), rg[:, 1] - rg[:, 0])
    same = tile_of[1:] == tile_of[:-1]
    dz = zl[1:] - zl[:-1]
    assert np.all(dz[same] >= -1e-5 * np.abs(zl[1:][same]))
    tie = same & (dz == 0)
    assert np.all(pl[1:][tie] > pl[:-1][tie])


def test_c3_backward_is_linear_in_upstream_gradients(c3, dev):
    sc, deg, cam, T = c3
    rng = np.random.default_rng(9)
    shapes = [(3, 376, 1376), (4, 376, 1376), (3, 376, 1376)]
    g1 = [torch.tensor(rng.normal(size=s).astype(np.float32) * 1e-3, device=dev) for s in shapes]
    g2 = [torch.tensor(rng.normal(size=s).astype(np.float32) * 1e-3, device=dev) for s in shapes]

    def grads(gs):
        res, t, m2d = _fwd(T, cam, deg, [0, 0, 0], dev, requires_grad=True)
        torch.autograd.backward([res[0], res[2], res[3]], gs)
        return [t[k].grad for k in ("means3D", "shs", "opacities", "scales", "rotations")] + [m2d.grad]
    a, b = grads(g1), grads(g2)
    c = grads([x + y for x, y in zip(g1, g2)])
    for x, y, z in zip(a, b, c):
        assert torch.isfinite(z).all()
        assert rel_err((x + y).cpu().numpy(), z.cpu().numpy()) < 1e-3
    assert c[5][:, 2].abs().max().item() == 0.0
    radii = _fwd(T, cam, deg, [0, 0, 0], dev)[0][5]
    culled = radii == 0
    assert culled.any() and all(g[culled].abs().max().item() == 0.0 for g in c)   # dense, zero where culled
