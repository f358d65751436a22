"""GPU (-m gpu): the binning under disturbance.  Its sorts and its emission scan are single-pass kernels whose workgroups
wait for sums posted by the workgroups before them (vegs_amd/csrc/binning.hip): correct only if nothing about the result
depends on WHO runs WHEN.  The parity tests run on a quiet GPU; here the same view is binned again and again while
another stream keeps the CUs, the L2s and the memory system busy, and while a second view is in flight on a second
stream -- the lists must come out bit for bit what the quiet run produced, every time, and no guard word may go up.
The same for the render forward's chain mode (walker workgroups that follow segment products published by other workgroups
of the same launch, render_fwd.hip): the images of the disturbed runs equal the quiet run's, and the walkers' result
equals the three-round path's.
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


def test_headline_loop_under_the_depth_sorts_post_mortem():
    """>= 200 forwards + backwards of the headline scene's views, host running ahead as in bench.py, every forward followed
    by the depth sort's post-mortem (VEGS_DEBUG_BINNING=1, binning.hip: debug_verify_binning: permutation and pair integrity
    of both ping-pong buffers, digit totals, every posted status word, the last pass replayed on the host).  A finding fails
    the forward.  Round-5 verdict item 1: the run that once ended in a memory fault in k_emit_scan (garbage ids out of the
    depth sort) has this loop's shape; it has not been seen again on any box since, with or without the build that showed it
    (profiles/experiments/README.md, round 6) -- this test keeps looking."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VEGS_DEBUG_BINNING="1")
    n = max(200, ROUNDS * 4)
    r = subprocess.run([sys.executable, os.path.join(root, "profiles", "tools", "r06", "soak_postmortem.py"), str(n)],
                       capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0 and "clean" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


_BLIND_CHECK = r"""
import sys, torch
sys.path.insert(0, %r)
from vegs_amd import _capi, harness, scenes
_capi.load()
dev = torch.device("cuda:0")
sc, deg = scenes.scene_street(P=60000, length=40.0, sh_degree=1, seed=19)
cam = scenes.kitti_camera(0.0, 0.0, 344, 94)
T = {k: torch.tensor(v, device=dev) for k, v in sc.items()}
with torch.no_grad():
    harness.render(cam, T, deg, torch.zeros(3, device=dev))          # a sound view passes the post-mortem
    _capi.check(_capi.load().vr_debug_raise_guard(4))                # the next forward's sorted ids get one duplicate
    try:
        harness.render(cam, T, deg, torch.zeros(3, device=dev))
    except Exception as e:
        print("CAUGHT", str(e)[:200])
        sys.exit(0)
print("NOT CAUGHT")
sys.exit(1)
"""


def test_the_post_mortem_is_not_blind():
    """The instrument of the test above, shown an injected fault: vr_debug_raise_guard(4) overwrites one sorted id with its
    neighbour behind the depth sort; under VEGS_DEBUG_BINNING=1 that forward must FAIL with the post-mortem's finding (a
    duplicate id / a pair whose key is not its id's depth key), after a sound forward of the same view passed."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _BLIND_CHECK % root], capture_output=True, text=True,
                       env=dict(os.environ, VEGS_DEBUG_BINNING="1"), timeout=600)
    assert r.returncode == 0 and "CAUGHT" in r.stdout and "post-mortem" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    assert "duplicate ids" in r.stderr and "output of the last pass" in r.stderr, r.stderr[-3000:]


def test_two_views_in_flight_do_not_disturb_each_other(dev):  # noqa: F811
    """Two streams, two different views, no synchronisation between the forwards: each view's lists and image equal its
    quiet run's (separate guard words and status regions per forward in flight, ABI v9)."""
    # (the first view is large enough for the render forward's chain mode: walkers of one launch beside the other view's kernels)
    views = [_view(1000000, 1376, 376, 7, 5.0, dev), _view(200000, 1024, 320, 8, 25.0, dev)]
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


def test_chain_mode_and_rounds_agree_on_a_view_with_deep_tiles(dev):  # noqa: F811
    """A street view with vanishing-point tiles of hundreds of list segments: the default forward walks their chains inside
    k_seg_alpha's launch and skips the segments behind each chain's end (chain mode, vegs_amd/csrc/render_fwd.hip);
    VR_FLAG_ROUNDS_ON takes the three-round path without walkers.  Images, needed-segment counts and the deterministic
    backward's gradients must be bit-identical -- and stay so when the forward is repeated (which segments get skipped
    depends on timing; the result may not)."""
    import ctypes as C
    from diff_gaussian_rasterization import GaussianRasterizer
    from vegs_amd import _capi, rasterizer
    W, H = 1376, 376
    st, t = _view(1000000, W, H, 11, 30.0, dev)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    g = torch.Generator(device="cpu").manual_seed(3)
    gouts = [torch.randn(s, generator=g).to(dev) for s in [(3, H, W), (4, H, W), (3, H, W)]]

    def run(flags):
        leaves = {k: v.clone().requires_grad_(True) for k, v in t.items()}
        m2d = torch.zeros(leaves["means3D"].shape[0], 3, device=dev, requires_grad=True)
        old = rasterizer.needed_hints(False)
        try:
            with rasterizer.flags(flags | rasterizer.FLAG_DETERMINISTIC):
                res = GaussianRasterizer(raster_settings=st)(means3D=leaves["means3D"], means2D=m2d, shs=leaves["shs"],
                                                            opacities=leaves["opacities"], scales=leaves["scales"],
                                                            rotations=leaves["rotations"])
                need = torch.zeros(T, dtype=torch.int32, device=dev)
                saved = _capi.saved_of(res[0].grad_fn)
                _capi.check(_capi.load().vr_export_needed(C.byref(saved), H, W, need.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
                torch.autograd.backward([res[0], res[2], res[3]], gouts)
        finally:
            rasterizer.needed_hints(old)
        return [r.detach().clone() for r in res[:5]], need, {k: v.grad.clone() for k, v in leaves.items()}, m2d.grad.clone()

    img_r, need_r, grads_r, m2d_r = run(rasterizer.FLAG_ROUNDS_ON)
    assert int(need_r.max()) > 40          # (a chain longer than the prefix every tile computes up front)
    for rep in range(4):
        img_c, need_c, grads_c, m2d_c = run(0)
        assert torch.equal(need_c, need_r), rep
        for a, b in zip(img_c, img_r):
            assert torch.equal(a, b), rep
        assert torch.equal(m2d_c, m2d_r)
        for k in grads_r:
            assert torch.equal(grads_c[k], grads_r[k]), (rep, k)
    # walkers without patience (vr_debug_raise_guard(3): they leave at their first empty poll, as after their bounded wait on
    # a GPU that does not schedule the producers): k_seg_scan finishes their tiles behind the launch -- same result
    for rep in range(3):
        _capi.check(_capi.load().vr_debug_raise_guard(3))
        img_i, need_i, grads_i, m2d_i = run(0)
        assert torch.equal(need_i, need_r), rep
        for a, b in zip(img_i, img_r):
            assert torch.equal(a, b), rep
        assert torch.equal(m2d_i, m2d_r)
        for k in grads_r:
            assert torch.equal(grads_i[k], grads_r[k]), (rep, k)


_SWEEP = range(int(os.environ.get("VEGS_STRESS_ROUNDS", "60")) // 10)


@pytest.mark.parametrize("case", _SWEEP)
def test_chain_mode_sweep_against_the_rounds(case, dev):  # noqa: F811
    """Random street views around the sizes at which chain mode switches on (enough list segments, some tile deeper than the
    prefix, not more heavy tiles than walkers): the default forward against VR_FLAG_ROUNDS_ON -- five images and per-tile
    needed-segment counts bit-identical, whichever of the two the default turns out to be for the view."""
    import ctypes as C
    from diff_gaussian_rasterization import GaussianRasterizer
    from vegs_amd import _capi, rasterizer, scenes
    rng = np.random.default_rng(4100 + case)
    P = int(rng.choice([300000, 600000, 1000000, 1500000]))
    W, H = [(1376, 376), (1408, 376), (1000, 300), (704, 188)][int(rng.integers(0, 4))]
    disc = float(rng.choice([0.6, 1.0, 1.6, 2.5]))
    sc, _ = scenes.scene_street(P=P, length=float(rng.choice([60.0, 120.0, 250.0])), sh_degree=1, seed=500 + case)
    sc["scales"] = (sc["scales"] * disc).astype(np.float32)
    if case % 3 == 0:
        sc["opacities"] = np.clip(sc["opacities"] * 0.5, 0.0, 1.0).astype(np.float32)      # (long chains: more needed segments)
    cam = scenes.kitti_camera(x_forward=float(rng.uniform(0.0, 40.0)), y_left=float(rng.uniform(-2.0, 2.0)), width=W, height=H)
    st = tp._settings(cam, rng.uniform(0, 1, 3).astype(np.float32), 1, 1.0, dev)
    t = {k: torch.tensor(v, device=dev) for k, v in sc.items()}
    T = ((W + 15) // 16) * ((H + 15) // 16)

    def run(flags):
        m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
        old = rasterizer.needed_hints(False)
        try:
            with rasterizer.flags(flags):
                res = GaussianRasterizer(raster_settings=st)(means3D=t["means3D"], means2D=m2d, shs=t["shs"], opacities=t["opacities"],
                                                            scales=t["scales"], rotations=t["rotations"])
                need = torch.zeros(T, dtype=torch.int32, device=dev)
                saved = _capi.saved_of(res[0].grad_fn)
                _capi.check(_capi.load().vr_export_needed(C.byref(saved), H, W, need.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        finally:
            rasterizer.needed_hints(old)
        return [r.detach() for r in res[:5]], need

    img_r, need_r = run(rasterizer.FLAG_ROUNDS_ON)
    for rep in range(2):
        img_c, need_c = run(0)
        assert torch.equal(need_c, need_r), (case, rep)
        for a, b in zip(img_c, img_r):
            assert torch.equal(a, b), (case, rep)
