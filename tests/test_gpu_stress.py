"""GPU (-m gpu): the binning under disturbance.  Its sorts and its emission scan are single-pass kernels whose workgroups
wait for sums posted by the workgroups before them (vegs_amd/csrc/binning.hip): correct only if nothing about the result
depends on WHO runs WHEN.  The parity tests run on a quiet GPU; here the same view is binned again and again while
another stream keeps the CUs, the L2s and the memory system busy, and while a second view is in flight on a second
stream -- the lists must come out bit for bit what the quiet run produced, every time, and no guard word may go up.
VEGS_STRESS_ROUNDS widens the loop for a campaign (default: what a round's test run can afford)."""
import os

import numpy as np
import pytest
import torch

import test_gpu_parity as tp
from test_gpu_parity import dev  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

ROUNDS = int(os.environ.get("VEGS_STRESS_ROUNDS", "60"))


def _view(P, W, H, seed, x_forward, dev):  # noqa: F811
    from vegs_amd import scenes
    sc, _ = scenes.scene_street(P=P, sh_degree=3, seed=seed)
    cam = scenes.kitti_camera(x_forward=x_forward, width=W, height=H)
    st = tp._settings(cam, [0.0, 0.0, 0.0], 3, 1.0, dev)
    t = {k: torch.tensor(v, device=dev) for k, v in sc.items()}
    return st, t


def _forward(st, t, dev):  # noqa: F811
    from diff_gaussian_rasterization import GaussianRasterizer
    P = t["means3D"].shape[0]
    m2d = torch.zeros(P, 3, device=dev, requires_grad=True)       # (a grad_fn: the lists are exported through its saved state)
    return GaussianRasterizer(raster_settings=st)(means3D=t["means3D"], means2D=m2d, shs=t["shs"], opacities=t["opacities"],
                                                  scales=t["scales"], rotations=t["rotations"])


def _lists(res, st, dev):  # noqa: F811
    return tp._export_binning(res, st.image_height, st.image_width, dev)


def test_lists_do_not_depend_on_what_else_the_gpu_is_doing(dev):  # noqa: F811
    from vegs_amd import _capi
    W, H = 1376, 376
    st, t = _view(1000000, W, H, 5, 10.0, dev)
    quiet = _forward(st, t, dev)
    pl0, rg0 = _lists(quiet, st, dev)
    img0 = quiet[0].clone()
    assert pl0.size > 1000000 and (t["means3D"].shape[0] >> 8) > 2048     # (sorts and scan: several resident sets of workgroups)
    before = _capi.load().vr_debug_rebinned()
    side = torch.cuda.Stream(dev)
    a = torch.randn(6144, 6144, device=dev)
    big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    rng = np.random.default_rng(0)
    for it in range(ROUNDS):
        with torch.cuda.stream(side):
            # a different mix every round: GEMMs hold CUs for ~ms, fills and copies hold the memory system
            for _ in range(int(rng.integers(1, 4))):
                a @ a
            if it % 2:
                big.fill_(it & 255)
            if it % 3 == 0:
                big[: 128 << 20].copy_(big[128 << 20:])
        res = _forward(st, t, dev)
        pl, rg = _lists(res, st, dev)
        assert np.array_equal(rg, rg0) and np.array_equal(pl, pl0), f"round {it}: the lists changed under load"
        assert torch.equal(res[0], img0), f"round {it}: the image changed under load"
    torch.cuda.synchronize()
    assert _capi.load().vr_debug_rebinned() == before


def test_two_views_in_flight_do_not_disturb_each_other(dev):  # noqa: F811
    """Two streams, two different views, no synchronisation between the forwards: each view's lists and image equal its
    quiet run's (separate guard words and status regions per forward in flight, ABI v9)."""
    views = [_view(300000, 1376, 376, 7, 5.0, dev), _view(200000, 1024, 320, 8, 25.0, dev)]
    quiet = []
    for st, t in views:
        r = _forward(st, t, dev)
        quiet.append((_lists(r, st, dev), r[0].clone()))
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    torch.cuda.synchronize()
    for it in range(max(ROUNDS // 3, 8)):
        res = []
        for (st, t), s in zip(views, streams):
            with torch.cuda.stream(s):
                res.append(_forward(st, t, dev))
        for (st, t), s, r, ((pl0, rg0), img0) in zip(views, streams, res, quiet):
            with torch.cuda.stream(s):
                pl, rg = _lists(r, st, dev)
                assert np.array_equal(rg, rg0) and np.array_equal(pl, pl0), f"round {it}: lists differ"
                assert torch.equal(r[0], img0), f"round {it}: image differs"
