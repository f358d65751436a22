"""GPU (-m gpu): BASELINE.json's full-size configurations.

C2 (500k Gaussians, 1376x376, SH 3) is still small enough for the oracle: forward bit-exact and
index-exact against it.  C3 (2M Gaussians, the headline config) is checked through size-independent
properties: run-to-run determinism of the forward, structure and sortedness of the tile lists,
exact background linearity, linearity of the backward in the upstream gradients.
"""
import os

import numpy as np
import pytest
import torch

from helpers import (OUT_NAMES, assert_grad_close, grad_mismatch, ill_conditioned, oracle_cam, rel_err,
                     summation_sensitivity)
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


# Both list definitions at full size: 0 = the build's default, tight tile lists; FULL = the REFERENCE's emission rule, every
# tile of a Gaussian's rectangle -- the north star's "tile/sort indices bit-exact" is a statement about those.
FULL_LISTS = 32768          # VR_FLAG_FULL_TILE_LISTS
LIST_MODES = pytest.mark.parametrize("list_flags", [0, FULL_LISTS], ids=["tight-lists", "reference-lists"])


@LIST_MODES
def test_c2_500k_forward_matches_oracle_bit_exact(dev, list_flags):
    from oracle import oracle as orc
    from vegs_amd import scenes
    sc, deg = scenes.scene_street(P=500_000, length=120.0, sh_degree=3, seed=1)
    cam = scenes.kitti_camera(0.0, 0.3, 1376, 376)
    oc = oracle_cam(cam, [0, 0, 0], deg, flags=list_flags)
    o_out, st = orc.forward(oc, **_inputs(sc))
    gouts = [np.random.default_rng(5).normal(size=s).astype(np.float32) * 1e-3
             for s in [(3, 376, 1376), (1, 376, 1376), (4, 376, 1376), (3, 376, 1376), (1, 376, 1376)]]
    h_out, h_grads, res = _run_hip(_settings(cam, [0, 0, 0], deg, 1.0, dev), _inputs(sc), dev, gouts, flags=list_flags)
    assert res[0].grad_fn.num_rendered == st["R"]
    assert np.array_equal(h_out["radii"], o_out["radii"])
    pl, rg = _export_binning(res, 376, 1376, dev)
    assert np.array_equal(rg, st["ranges"]) and np.array_equal(pl, st["point_list"])
    for n in OUT_NAMES:
        assert np.array_equal(h_out[n], o_out[n]), n
    o_grads = orc.backward(oc, st, *gouts)
    ill, explain = ill_conditioned(st)
    for k in ("means3D", "shs", "opacities", "scales", "rotations", "means2D"):
        # default (atomic) mode: no row beyond 10x unless its conic is ill-conditioned (printed)
        assert_grad_close(k, h_grads[k], o_grads[k], rtol=1e-3, explain=explain, ill=ill)
    _oracle_parity("c2", _inputs(sc), deg, cam, dev, flags=list_flags)                     # deterministic mode: the tight comparison


@pytest.fixture(scope="module")
def c3(dev):
    from vegs_amd import scenes
    sc, deg = scenes.scene_street(P=2_000_000, length=250.0, sh_degree=3, seed=2)
    cam = scenes.kitti_camera(20.0, -0.3, 1376, 376)
    T = {k: torch.tensor(v, device=dev) for k, v in sc.items()}
    return sc, deg, cam, T


_C3_ILL = {}


def _c3_ill(c3):
    """Ill-conditioned rows of the c3 fixture's view (from the oracle's forward state, computed once)."""
    if "v" not in _C3_ILL:
        from oracle import oracle as orc
        sc, deg, cam, T = c3
        _, st = orc.forward(oracle_cam(cam, [0, 0, 0], deg), **_inputs(sc))
        _C3_ILL["v"] = ill_conditioned(st)
    return _C3_ILL["v"]


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


def test_c3_both_binning_paths_build_the_same_lists(c3, dev):
    """Headline scene: the single-launch radix passes (posted block sums; 403 and ~950 workgroups per pass, so all
    three levels are in use) and the scan-based passes (VR_FLAG_SCAN_BINNING) produce identical lists, ranges, images."""
    from vegs_amd import rasterizer
    sc, deg, cam, T = c3
    res_a, *_ = _fwd(T, cam, deg, [0, 0, 0], dev, requires_grad=True)
    with rasterizer.flags(rasterizer.FLAG_SCAN_BINNING):
        res_b, *_ = _fwd(T, cam, deg, [0, 0, 0], dev, requires_grad=True)
    pa, ra = _export_binning(res_a, 376, 1376, dev)
    pb, rb = _export_binning(res_b, 376, 1376, dev)
    assert len(pa) > 1_000_000 and np.array_equal(pa, pb) and np.array_equal(ra, rb)
    for x, y in zip(res_a, res_b):
        assert torch.equal(x, y)


def test_dense_scene_13m_entries_both_binning_paths(c3, dev):
    """The headline scene with every disc 3x larger, on the reference's full tile rectangles (VR_FLAG_FULL_TILE_LISTS): R ~ 13 M
    list entries = ~3200 workgroups per tile-sort pass (twelve complete groups on the third level of the posted sums).
    Lists, ranges and images of the single-launch passes equal those of the scan-based passes, and the lists have the
    structure the contract asks for.  Then the default, tight lists of the same view (a third of its pairs sit in rectangles
    of more than 64 tiles: cell masks, k_emit_big): both binning paths again, fewer than half the entries, the same image."""
    from vegs_amd import rasterizer
    sc, deg, cam, T = c3
    Td = dict(T)
    Td["scales"] = T["scales"] * 3.0
    out = {}
    for name, fl, lo, hi in (("full", rasterizer.FLAG_FULL_TILE_LISTS, 8_000_000, 10 ** 9), ("tight", 0, 3_000_000, 7_000_000)):
        with rasterizer.flags(fl):
            res_a, *_ = _fwd(Td, cam, deg, [0, 0, 0], dev, requires_grad=True)
        pa, ra = _export_binning(res_a, 376, 1376, dev)
        with rasterizer.flags(fl | rasterizer.FLAG_SCAN_BINNING):
            res_b, *_ = _fwd(Td, cam, deg, [0, 0, 0], dev, requires_grad=True)
        pb, rb = _export_binning(res_b, 376, 1376, dev)
        R = res_a[0].grad_fn.num_rendered
        assert R == len(pa) and lo < R < hi, (name, R)
        assert np.array_equal(pa, pb) and np.array_equal(ra, rb)
        for x, y in zip(res_a, res_b):
            assert torch.equal(x, y)
        nonempty = ra[ra[:, 1] > ra[:, 0]]
        assert (nonempty[:, 1] - nonempty[:, 0]).sum() == R
        radii = res_a[5].cpu().numpy()
        assert np.all(radii[pa] > 0)
        out[name] = (R, [x.detach() for x in res_a])
    assert out["tight"][0] < 0.5 * out["full"][0]
    assert torch.equal(out["tight"][1][5], out["full"][1][5])                   # radii
    # images: the same fragments in other partial sums (the 256-entry segments start at other entries) -- rounding, except
    # where a transmittance that differs in its last bits takes the 1e-4 stop test the other way: one more or one fewer
    # fragment of weight < 1e-4 in that pixel
    for x, y in zip(out["tight"][1][:5], out["full"][1][:5]):
        d, scale = (x - y).abs(), max(1.0, float(y.abs().max()))
        assert float(d.max()) <= 1.2e-4 * scale, float(d.max())
        assert float((d > 4e-6 * scale).float().mean()) < 1e-3


def test_c3_deterministic_backward_mode(c3, dev):
    """Headline scene, VR_FLAG_DETERMINISTIC: gradients bit-identical from run to run (the default mode's fp32 atomics
    are order-dependent) and within the per-row tolerance of the default mode."""
    from vegs_amd import rasterizer
    sc, deg, cam, T = c3
    rng = np.random.default_rng(19)
    gs = [torch.tensor(rng.normal(size=s).astype(np.float32) * 1e-3, device=dev) for s in [(3, 376, 1376), (4, 376, 1376), (3, 376, 1376)]]

    def grads(flags):
        with rasterizer.flags(flags):
            res, t, m2d = _fwd(T, cam, deg, [0, 0, 0], dev, requires_grad=True)
            torch.autograd.backward([res[0], res[2], res[3]], gs)
        return [t[k].grad for k in ("means3D", "shs", "opacities", "scales", "rotations")] + [m2d.grad]
    a, b = grads(rasterizer.FLAG_DETERMINISTIC), grads(rasterizer.FLAG_DETERMINISTIC)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    c = grads(0)
    ill, explain = _c3_ill(c3)
    for name, x, y in zip(("means3D", "shs", "opacities", "scales", "rotations", "means2D"), c, a):
        assert_grad_close("c3 atomic vs deterministic " + name, x.cpu().numpy(), y.cpu().numpy(), rtol=1e-3, floor=2e-6,
                          explain=explain, ill=ill)


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
    ill, explain = _c3_ill(c3)
    for x, y, z in zip(a, b, c):
        assert torch.isfinite(z).all()
        # (both sides carry fp32-atomic noise, and the preprocess backward amplifies it without bound for the
        # 1e-5-thin discs seen edge-on: only rows whose conic is ill-conditioned may leave the 10x cap; printed)
        assert_grad_close("linearity", (x + y).cpu().numpy(), z.cpu().numpy(), rtol=1e-3, floor=2e-6, explain=explain, ill=ill)
    assert c[5][:, 2].abs().max().item() == 0.0
    radii = _fwd(T, cam, deg, [0, 0, 0], dev)[0][5]
    culled = radii == 0
    assert culled.any() and all(g[culled].abs().max().item() == 0.0 for g in c)   # dense, zero where culled


# ---------------------------------------------------------------------------------------------------------------------
# The headline workloads under the oracle (round 3).  Forward: radii, tile lists, ranges and all five images BIT-EXACT.
# Gradients: the backward runs in VR_FLAG_DETERMINISTIC (per-(entry, region) partial sums added in list order: no
# order-dependent atomics) and is compared PER ROW with the oracle's double sums.
#   * means2D, opacities, shs are plain sums of per-fragment terms: rtol 2e-4, at most 0.001 % of the rows outside it
#     (measured: 0 or 1 row of 2-5 M -- a sum of mixed-sign terms that cancels), each explained as below.
#   * means3D, scales, rotations go on through conic -> cov2D -> cov3D (k_preprocess_bwd), which divides by det^2 of
#     the 2D covariance: for edge-on discs (VEGS initialises 1e-5-thin discs) that chain amplifies the fp32 rounding
#     of the sums it starts from without bound.  Which rows those are is MEASURED (helpers.summation_sensitivity): the
#     oracle's own fp32 preprocess backward is re-run on its per-Gaussian sums perturbed by 2e-6 x (sum of the absolute
#     per-fragment terms) -- what an fp32 summation of those terms carries; a row that moves by m allowances cannot be
#     held closer than that by any fp32 chain, the oracle's included.  EVERY row outside rtol 2e-4 must be within 8x of
#     its measured movement (asserted; the conic conditioning (A+C)^2/4det of the offenders is printed: 70x .. 3000x the
#     median); all other rows -- > 99.5 % -- pass at 2e-4, and at most 0.01 % of all rows are off by more than 3x.


# A row outside rtol 2e-4 must lie within EXPLAIN_FACTOR x what the summation-rounding model moves the checker's own result
# by (round 3: 8; round 4: the deterministic mode adds each Gaussian's slots in double and divides IEEE-exactly, so what is
# left to explain is the per-fragment fp32 arithmetic and the <= 64-pixel sums inside a slot); rows whose measured movement
# exceeds EXEMPT_FROM_CAP allowances are exempt from the 3x cap (not from the explanation).
EXPLAIN_FACTOR = float(os.environ.get("VEGS_EXPLAIN_FACTOR", "3.0"))
EXEMPT_FROM_CAP = float(os.environ.get("VEGS_EXEMPT_FROM_CAP", "1.0"))


def _oracle_parity(name, sc_inputs, deg, cam, dev, hip_runs=1, flags=0):
    """sc_inputs: op kwargs (numpy).  Renders `hip_runs` times through the operator (same camera tensors: with the hint
    cache on, the third run uses a warm needed-segment hint), checks every run bit-exact against the oracle's forward and
    the last run's deterministic-mode gradients per row against the oracle's backward."""
    from oracle import oracle as orc
    from vegs_amd import rasterizer
    H, W = cam.image_height, cam.image_width
    oc = oracle_cam(cam, [0, 0, 0], deg, flags=flags)       # (`flags`: on BOTH sides -- the list definition, a fork switch)
    o_out, st = orc.forward(oc, **sc_inputs)
    rng = np.random.default_rng(31)
    gouts = [rng.normal(size=s).astype(np.float32) * 1e-3 if m else None
             for s, m in zip([(3, H, W), (1, H, W), (4, H, W), (3, H, W), (1, H, W)], (1, 0, 1, 1, 0))]
    settings = _settings(cam, [0, 0, 0], deg, 1.0, dev)        # ONE set of camera tensors: the hint key
    old = rasterizer.needed_hints("always")
    try:
        rasterizer._NEEDED.clear()
        for run in range(hip_runs):
            last = run == hip_runs - 1
            h_out, h_grads, res = _run_hip(settings, sc_inputs, dev, gouts if last else None,
                                           flags=flags | (rasterizer.FLAG_DETERMINISTIC if last else 0))
            assert np.array_equal(h_out["radii"], o_out["radii"]), (name, run)
            pl, rg = _export_binning(res, H, W, dev)
            assert res[0].grad_fn.num_rendered == st["R"]
            assert np.array_equal(rg, st["ranges"]) and np.array_equal(pl, st["point_list"]), (name, run)
            for n in OUT_NAMES:
                assert np.array_equal(h_out[n], o_out[n]), (name, run, n, np.abs(h_out[n] - o_out[n]).max())
        if hip_runs >= 3:
            hint = list(rasterizer._NEEDED.values())[0]
            assert hint is not None and int(hint.max()) < 0x3FFFFFFF      # the last run really had a recorded hint
    finally:
        rasterizer.needed_hints(old)
    og = orc.backward(oc, st, *gouts, abs_sums=True)
    _, explain = ill_conditioned(st)
    names = ("means2D", "opacities", "shs", "means3D", "scales", "rotations")
    moved = summation_sensitivity(oc, st, og, names=names)
    for k in names:
        plain = k in ("means2D", "opacities", "shs")
        ill = moved[k] > EXEMPT_FROM_CAP
        assert ill.mean() < 5e-2, (name, k, "fraction of rows the oracle itself cannot resolve to a quarter allowance", float(ill.mean()))
        bad = assert_grad_close(f"{name} det {k}", h_grads[k], og[k], rtol=2e-4, floor=2e-7, outliers=1e-5 if plain else 1e-4,
                                near=1e-5 if plain else 1e-3, cap=3.0, explain=explain, ill=ill)
        # EVERY offender is explained: its error is within 8x of what the summation-rounding model moves the oracle's
        # own result by (plus the allowance) -- the model takes the largest of three 1-sigma draws (64 for the offenders), the actual
        # rounding may sit at 3 sigma
        _, ratio = grad_mismatch(h_grads[k], og[k], 2e-4, 2e-7)
        # (the offenders' movement re-measured over 64 directions: three draws undersample the one direction an edge-on disc
        # amplifies -- helpers.summation_sensitivity)
        mv = np.maximum(moved[k][bad], summation_sensitivity(oc, st, og, names=(k,), rows=bad, seeds=64)[k]) if len(bad) else moved[k][bad]
        unexplained = bad[ratio[bad] > 1.0 + EXPLAIN_FACTOR * mv]
        # how many rows needed an explanation at all, and the largest multiple of its measured movement any of them used
        used = float(((ratio[bad] - 1.0) / np.maximum(mv, 1e-9)).max()) if len(bad) else 0.0
        print(f"[{name}] {k}: {len(bad)} of {len(ratio)} rows outside rtol 2e-4 needed an explanation "
              f"(largest: {used:.2f} x its measured summation sensitivity; allowed {EXPLAIN_FACTOR})")
        assert len(unexplained) == 0, (name, k, "rows outside 2e-4 beyond what fp32 summation explains",
                                       [(int(i), float(ratio[i]), float(moved[k][i])) for i in unexplained[:8]])
    return st


def _bench_cams():
    from vegs_amd import scenes
    cams = []
    for s in range(8):                       # bench.py build_workload: stations every 10 m, stereo pair
        for y in (0.3, -0.3):
            cams.append(scenes.kitti_camera(10.0 * s, y, 1376, 376))
    return cams


@LIST_MODES
def test_c3_headline_views_match_oracle(c3, dev, list_flags):
    """BASELINE config C3 (2 M Gaussians, the headline): two of bench.py's 16 cameras under the oracle, one of them
    rendered three times so that its last forward runs with a recorded needed-segment hint -- on the build's tight tile
    lists and on the reference's full rectangles (radii / lists / ranges / images bit-exact in both)."""
    sc, deg, cam, T = c3
    cams = _bench_cams()
    st = _oracle_parity("c3 cam5", _inputs(sc), deg, cams[5], dev, hip_runs=3, flags=list_flags)
    # (tight tile lists: about 2.7 M entries; the reference's full rectangles: about 4.2 M)
    assert st["R"] > (3_500_000 if list_flags else 2_500_000) and (list_flags or st["R"] < 3_200_000)
    _oracle_parity("c3 cam10", _inputs(sc), deg, cams[10], dev, flags=list_flags)


def test_c3_dense_13m_entries_matches_oracle(c3, dev):
    """The dense variant of the headline scene (every disc 3x larger: ~13 M pairs on the reference's rectangles, 5.7 M list
    entries with the tight lists -- a third of the pairs sit in rectangles of more than 64 tiles: cell masks and k_emit_big
    at full size; the heaviest lists of the bench line) under the oracle."""
    sc, deg, cam, T = c3
    dense = dict(_inputs(sc))
    dense["scales"] = (sc["scales"] * 3.0).astype(np.float32)
    st = _oracle_parity("dense", dense, deg, _bench_cams()[0], dev)
    assert st["R"] > 4_000_000


_C5 = {}


def _c5_inputs(dev):
    """The concatenated op inputs of BASELINE config C5 (numpy), their SH degree and camera; cached."""
    if "inputs" not in _C5:
        from vegs_amd import harness, iteration, scenes
        P, NB = 5_000_000, 8
        sc, deg = scenes.scene_street(P=P, length=250.0, sh_degree=3, seed=3)
        cam = scenes.kitti_camera(10.0, 0.3, 1376, 376)
        with torch.no_grad():
            static = {k: torch.tensor(v, device=dev) for k, v in sc.items()}
            boxes = iteration.make_boxes(NB, dev)
            kw = harness.prepare_rasterization(static)
            for t, b2w in boxes:
                kw = harness.merge_kwargs(kw, harness.prepare_rasterization({k: v.detach() for k, v in t.items()}, b2w.detach()))
            inputs = {k: v.cpu().numpy() for k, v in kw.items()}
        del static, boxes, kw
        torch.cuda.empty_cache()
        assert inputs["means3D"].shape[0] == P + NB * 8196
        inputs.update(colors_precomp=None, cov3D_precomp=None)
        _C5.update(inputs=inputs, deg=deg, cam=cam)
    return _C5["inputs"], _C5["deg"], _C5["cam"]


def _c5_ill(dev):
    """Rows of C5's op inputs whose conic is ill-conditioned (oracle forward state; cached)."""
    if "ill" not in _C5:
        from oracle import oracle as orc
        inputs, deg, cam = _c5_inputs(dev)
        _, st = orc.forward(oracle_cam(cam, [0, 0, 0], deg), **inputs)
        _C5["ill"] = ill_conditioned(st)
    return _C5["ill"]


def test_c5_concatenated_inputs_match_oracle(dev):
    """BASELINE config C5's op call: 5 M static Gaussians + 8 box instances x 8,196 carried into the world frame and
    concatenated (gaussian_renderer/__init__.py:121-186, 263-333) -- the rasterizer on exactly those inputs under the
    oracle (forward bit-exact, gradients of the op inputs per row)."""
    P, NB = 5_000_000, 8
    inputs, deg, cam = _c5_inputs(dev)
    st = _oracle_parity("c5", inputs, deg, cam, dev)
    assert int((st["radii"][P:] > 0).sum()) > 1000          # the box instances are in view


def test_c5_full_step_5m_plus_box_instances(dev):
    """BASELINE config C5 on one GPU: 5 M static Gaussians + 8 box instances x 8,196 through the render_all-shaped
    composition, L1+SSIM + normal guidance, backward, densification statistics and Adam -- the counterpart of
    reference train.py:143-168,196,299-320 and gaussian_renderer/__init__.py:263-333 (vegs_amd/iteration.py).
    The fused step (instances N4 + losses N1 + Adam/statistics N2) must equal the same step built from the
    reference's op-by-op composition + ATen losses + torch.optim.Adam."""
    from vegs_amd import harness, iteration, scenes
    P, NB = 5_000_000, 8
    sc, deg = scenes.scene_street(P=P, length=250.0, sh_degree=3, seed=3)
    cam = scenes.kitti_camera(10.0, 0.3, 1376, 376)
    cam_t = harness.cam_tensors(cam, dev)
    rng = np.random.default_rng(11)
    gt = torch.tensor(rng.uniform(0, 1, (3, 376, 1376)).astype(np.float32), device=dev)
    normal = torch.tensor(rng.normal(size=(3, 376, 1376)).astype(np.float32), device=dev)
    bg = torch.zeros(3, device=dev)
    names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]

    out = {}
    for fused in (True, False):
        tr = iteration.Trainer(sc, dev, n_boxes=NB, fused=fused)
        before = {k: tr.p[k].detach().clone() for k in names}
        with torch.no_grad():                                       # forward determinism (no atomics in the forward)
            _, a = tr.forward_loss(cam, cam_t, deg, bg, gt, normal)
            _, b = tr.forward_loss(cam, cam_t, deg, bg, gt, normal)
            for k in ("render", "render_depth", "render_cov_quat_raw", "render_cov_scale", "alpha", "radii"):
                assert torch.equal(a[k], b[k]), k
            del a, b
        loss, pkg, grads = tr.step(cam, cam_t, deg, bg, gt, normal, keep_grads=True)
        radii = pkg["radii"]
        assert radii.shape == (P + NB * 8196,) and torch.isfinite(loss)
        assert int((radii[P:] > 0).sum()) > 1000                    # the box instances are in view
        culled = radii[:P] == 0
        assert culled.any()
        for k in names:
            g = grads[k]
            assert torch.isfinite(g).all(), k
            assert g[culled].abs().max().item() == 0.0, k           # dense gradients, exact zeros where culled
            assert torch.isfinite(tr.p[k]).all(), k
            assert torch.equal(tr.p[k][culled], before[k][culled]), k   # a zero gradient at step 1 moves nothing
        boxg = [(w.grad.clone(), {kk: t.grad.clone() for kk, t in b.items()}) for b, w in tr.boxes]
        vis = radii > 0
        assert torch.equal(tr.denom[:, 0], vis.float()) and torch.equal(tr.max_radii, radii.float() * vis)
        out[fused] = dict(loss=float(loss), grads={k: v.cpu().numpy() for k, v in grads.items()},
                          params={k: tr.p[k].detach().cpu().numpy() for k in names}, radii=radii.cpu().numpy(),
                          accum=tr.accum.cpu().numpy(), boxg=[(w.cpu().numpy(), {kk: t.cpu().numpy() for kk, t in d.items()}) for w, d in boxg])
        del tr, pkg, grads, boxg, before
        torch.cuda.empty_cache()

    A, B = out[True], out[False]
    assert np.array_equal(A["radii"], B["radii"])
    ill_all, explain = _c5_ill(dev)      # both variants carry fp32-atomic noise: rows beyond 10x must be ill-conditioned
    ill = ill_all[:P]
    assert abs(A["loss"] - B["loss"]) <= 1e-5 * abs(B["loss"]), (A["loss"], B["loss"])
    for k in names:
        assert_grad_close("c5 grad " + k, A["grads"][k], B["grads"][k], rtol=2e-3, floor=2e-6, outliers=2e-4,
                          explain=explain, ill=ill)
        # one Adam step moves every element by ~lr * sign(g): the two variants differ only where a noise-level
        # gradient changes sign (fp32 atomics are order-dependent), never by more than 2 lr
        pa, pb = A["params"][k], B["params"][k]
        tol = 1e-4 * np.abs(pb).max()
        frac = float((np.abs(pa - pb) > tol).mean())
        assert frac < 2e-3, (k, frac)
        assert np.abs(pa - pb).max() <= 2.01 * iteration.LRS[k] + 1e-6 * np.abs(pb).max(), k
    assert_grad_close("c5 densification accum", A["accum"], B["accum"], rtol=2e-3, floor=2e-6, outliers=2e-4)
    for (wa, da), (wb, db) in zip(A["boxg"], B["boxg"]):
        assert rel_err(wa, wb) < 5e-3                                # dL/d box2world: sums over 8,196 Gaussians
        for kk in ("means3D", "scales", "rotations", "opacities", "shs"):
            assert_grad_close("c5 box " + kk, da[kk], db[kk], rtol=2e-3, floor=2e-6, outliers=1e-3)
