"""Densification on the GPU (vegs_amd/csrc/densify.hip through vegs_amd.optim.densify_and_prune / reset_opacity, C ABI
include/vegs_optim.h) against (1) tests/golden/ref_densify.npz = the reference's own GaussianModel.densify_and_prune /
reset_opacity (scene/gaussian_model.py:215-218, 263-403) and (2) the numpy restatement oracle/densify_oracle.py at
sizes the fixture does not have."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_densify.npz")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


def _optimizer(par, m, v, dev, fused, step=3.0):
    """six named groups as scene/gaussian_model.py:159-168, with the given Adam moments as state (None: no state)"""
    from oracle import densify_oracle as do
    from vegs_amd import optim
    p = {k: torch.nn.Parameter(torch.tensor(par[k], device=dev)) for k in do.NAMES}
    groups = [{"params": [p[k]], "lr": 1e-3, "name": k} for k in do.NAMES]
    opt = (optim.Adam if fused else torch.optim.Adam)(groups, lr=0.0, eps=1e-15)
    if m is not None:
        for k in do.NAMES:
            opt.state[p[k]] = {"step": torch.tensor(step), "exp_avg": torch.tensor(m[k], device=dev),
                               "exp_avg_sq": torch.tensor(v[k], device=dev)}
    return opt, p


def _compare(opt, new, stats, want, want_m, want_v, exact_computed=False):
    from oracle import densify_oracle as do
    by_name = {g["name"]: g for g in opt.param_groups}
    for k in do.NAMES:
        q = by_name[k]["params"][0]
        assert q is new[k] and isinstance(q, torch.nn.Parameter) and q.requires_grad and q.is_leaf
        got = q.detach().cpu().numpy()
        assert got.shape == want[k].shape, (k, got.shape, want[k].shape)
        if k in ("xyz", "scaling") and not exact_computed:
            # the two computed columns of the split samples: float32 rounding of exp / log / a 3-term dot product
            np.testing.assert_allclose(got, want[k], rtol=3e-6, atol=3e-6, err_msg=k)
        else:
            assert np.array_equal(got, want[k]), k
        if want_m is not None:
            st = opt.state[q]
            assert np.array_equal(st["exp_avg"].cpu().numpy(), want_m[k]), k
            assert np.array_equal(st["exp_avg_sq"].cpu().numpy(), want_v[k]), k
        else:
            assert len(opt.state.get(q, {})) == 0
    assert len(opt.state) == (6 if want_m is not None else 0)          # the old parameters' entries are gone
    n = want["xyz"].shape[0]
    assert tuple(stats[0].shape) == (n, 1) and tuple(stats[1].shape) == (n, 1) and tuple(stats[2].shape) == (n,)
    assert all(float(s.abs().max()) == 0.0 for s in stats if s.numel())


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_densify_and_prune_reproduces_the_reference(tag, fused, dev):
    """Rows, their order, the carried parameters and Adam moments: exactly what the reference's methods left behind."""
    from test_densify_oracle import load_case
    from oracle import densify_oracle as do
    from vegs_amd import optim
    z = np.load(GOLD)
    par, m, v, acc, den, noise, (mg, mo, ext, pd, big) = load_case(z, tag)
    opt, p = _optimizer(par, m, v, dev, fused)
    new, stats = optim.densify_and_prune(opt, torch.tensor(acc, device=dev), torch.tensor(den, device=dev), mg, mo, ext,
                                         20 if big else None, pd, noise=torch.tensor(noise, device=dev))
    want = {k: z[f"{tag}_out_{k}"] for k in do.NAMES}
    _compare(opt, new, stats, want, {k: z[f"{tag}_out_m_{k}"] for k in do.NAMES}, {k: z[f"{tag}_out_v_{k}"] for k in do.NAMES})
    assert all(float(opt.state[new[k]]["step"]) == 3.0 for k in do.NAMES)
    # ... and training goes on: one optimizer step on the new tensors
    for k in do.NAMES:
        new[k].grad = torch.full_like(new[k], 1e-3)
    opt.step()
    assert all(float(opt.state[new[k]]["step"]) == 4.0 for k in do.NAMES)
    # reset_opacity on the densified model
    before = new["opacity"]
    q = optim.reset_opacity(opt)
    assert q is not before and {g["name"]: g for g in opt.param_groups}["opacity"]["params"][0] is q
    st = opt.state[q]
    assert float(st["exp_avg"].abs().max()) == 0.0 and float(st["exp_avg_sq"].abs().max()) == 0.0 and float(st["step"]) == 4.0
    np.testing.assert_allclose(q.detach().cpu().numpy(), do.reset_opacity(before.detach().cpu().numpy()), rtol=3e-6, atol=3e-6)
    assert float(torch.sigmoid(q.detach()).max()) <= 0.01 * (1 + 1e-5)


def _random_model(P, M, seed):
    rng = np.random.default_rng(seed)
    par = {"xyz": rng.normal(0, 3, (P, 3)), "f_dc": rng.normal(0, 1, (P, 1, 3)), "f_rest": rng.normal(0, .2, (P, M - 1, 3)),
           "opacity": rng.normal(-1.0, 3.0, (P, 1)), "scaling": rng.normal(np.log(0.03), 1.2, (P, 3)),
           "rotation": rng.normal(size=(P, 4))}
    par = {k: a.astype(np.float32) for k, a in par.items()}
    m = {k: rng.normal(0, 1e-3, a.shape).astype(np.float32) for k, a in par.items()}
    v = {k: (rng.normal(0, 1e-3, a.shape) ** 2).astype(np.float32) for k, a in par.items()}
    den = rng.integers(0, 40, (P, 1)).astype(np.float32)
    den[rng.random(P) < 0.1] = 0.0
    acc = (den * np.exp(rng.normal(np.log(2e-4), 1.0, (P, 1)))).astype(np.float32)
    return par, m, v, acc, den, rng


def _drop_borderline(par, acc, den, mg, mo, ext, pd):
    """rows whose classification hangs on the last bits of exp / sigmoid / the division (device and numpy libm may differ
    there) are moved off the thresholds; returns how many"""
    with np.errstate(divide="ignore", invalid="ignore"):
        g = np.nan_to_num(acc / den, nan=0.0).ravel()
    s = np.exp(par["scaling"].astype(np.float64))
    op = 1 / (1 + np.exp(-par["opacity"].astype(np.float64).ravel()))
    near = lambda x, t: np.abs(x - t) <= 1e-5 * abs(t)
    bad = near(g, mg) | near(s.max(1), pd * ext) | near(s.max(1), 0.1 * ext) | near(s.max(1) / 1.6, 0.1 * ext) | near(op, mo)
    par["scaling"][bad] -= 0.01
    par["opacity"][bad] += 0.01
    acc[bad] *= 1.001
    return int(bad.sum())


@pytest.mark.parametrize("P,M,big,state", [(300_001, 16, True, True), (70_000, 9, False, True), (40_000, 16, True, False),
                                           (1023, 4, True, True), (1, 16, True, True)])
def test_densify_equals_the_restated_sequence(P, M, big, state, dev):
    """Model sizes across several workgroups of the planner (and its edges), with and without optimizer state and the
    world-size prune: same rows in the same order as the restated reference sequence; copied columns and moments bit-exact."""
    from oracle import densify_oracle as do
    from vegs_amd import optim
    par, m, v, acc, den, rng = _random_model(P, M, seed=P)
    mg, mo, ext, pd = 2e-4, 0.005, 4.0, 0.01
    _drop_borderline(par, acc, den, mg, mo, ext, pd)
    src, kind, draw, S = do.plan(par["opacity"], par["scaling"], acc, den, mg, mo, ext, pd, big)
    noise = rng.normal(size=(2 * S, 3)).astype(np.float32)
    want, wm, wv, _, _ = do.densify_and_prune(par, m if state else None, v if state else None, acc, den, noise, mg, mo, ext, pd, big)
    opt, p = _optimizer(par, m if state else None, v if state else None, dev, fused=True)
    new, stats = optim.densify_and_prune(opt, torch.tensor(acc, device=dev), torch.tensor(den, device=dev), mg, mo, ext,
                                         20 if big else 0, pd, noise=torch.tensor(noise, device=dev))
    _compare(opt, new, stats, want, wm, wv)
    if P > 10000:
        assert (kind == 1).sum() > 100 and (kind == 2).sum() > 100 and (kind == 0).sum() < P


_SEEDS = range(*(int(v) for v in os.environ["VEGS_FUZZ_SEEDS"].split(":"))) if os.environ.get("VEGS_FUZZ_SEEDS") else range(8)


@pytest.mark.parametrize("seed", _SEEDS)
def test_densify_random_settings(seed, dev):
    """Random model sizes (around the planner's workgroup edges), thresholds, SH widths, with / without optimizer state and
    world-size prune, against the restated reference sequence (VEGS_FUZZ_SEEDS=a:b widens the sweep)."""
    from oracle import densify_oracle as do
    from vegs_amd import optim
    rng0 = np.random.default_rng(7000 + seed)
    P = int(rng0.choice([1, 2, 63, 255, 256, 257, 1023, 1024, 1025, 2047, 2049, 3000, 5000, 9000]))
    M = int(rng0.choice([1, 4, 9, 16]))
    big, state = bool(rng0.integers(0, 2)), bool(rng0.integers(0, 2))
    mg = float(rng0.choice([5e-5, 2e-4, 1e-3, 1e9]))
    mo = float(rng0.choice([0.005, 0.3, 0.9]))
    ext = float(rng0.choice([0.5, 4.0, 40.0]))
    pd = float(rng0.choice([0.01, 0.1]))
    par, m, v, acc, den, rng = _random_model(P, max(M, 2), seed=seed)
    if M == 1:
        par["f_rest"] = np.zeros((P, 0, 3), np.float32)
        m["f_rest"], v["f_rest"] = par["f_rest"].copy(), par["f_rest"].copy()
    _drop_borderline(par, acc, den, mg, mo, ext, pd)
    src, kind, draw, S = do.plan(par["opacity"], par["scaling"], acc, den, mg, mo, ext, pd, big)
    noise = rng.normal(size=(2 * S, 3)).astype(np.float32)
    want, wm, wv, _, _ = do.densify_and_prune(par, m if state else None, v if state else None, acc, den, noise, mg, mo, ext, pd, big)
    opt, p = _optimizer(par, m if state else None, v if state else None, dev, fused=bool(seed % 2))
    new, stats = optim.densify_and_prune(opt, torch.tensor(acc, device=dev), torch.tensor(den, device=dev), mg, mo, ext,
                                         7 if big else None, pd, noise=torch.tensor(noise, device=dev), empty_cache=False)
    _compare(opt, new, stats, want, wm, wv)


def test_densify_draws_its_own_noise_and_handles_empty_models(dev):
    from oracle import densify_oracle as do
    from vegs_amd import optim
    par, m, v, acc, den, rng = _random_model(20_000, 16, seed=3)
    opt, p = _optimizer(par, m, v, dev, fused=True)
    gen = torch.Generator(device=dev).manual_seed(11)
    new, _ = optim.densify_and_prune(opt, torch.tensor(acc, device=dev), torch.tensor(den, device=dev), 2e-4, 0.005, 4.0, 20, 0.01,
                                     generator=gen)
    src, kind, draw, S = do.plan(par["opacity"], par["scaling"], acc, den, 2e-4, 0.005, 4.0, 0.01, True)
    assert abs(new["xyz"].shape[0] - len(src)) <= 2 and S > 50          # (borderline rows not removed here)
    noise = torch.randn((2 * S, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(11)).cpu().numpy()
    want, _, _, _, _ = do.densify_and_prune(par, m, v, acc, den, noise, 2e-4, 0.005, 4.0, 0.01, True)
    if new["xyz"].shape[0] == len(src):
        np.testing.assert_allclose(new["xyz"].detach().cpu().numpy(), want["xyz"], rtol=3e-6, atol=3e-6)
    # everything pruned, then an empty model
    opt2, p2 = _optimizer(par, m, v, dev, fused=True)
    new2, stats2 = optim.densify_and_prune(opt2, torch.tensor(acc, device=dev), torch.tensor(den, device=dev), 1e9, 2.0, 4.0, 20, 0.01)
    assert new2["xyz"].shape == (0, 3) and new2["f_rest"].shape == (0, 15, 3) and stats2[0].shape == (0, 1)
    new3, _ = optim.densify_and_prune(opt2, stats2[0], stats2[1], 2e-4, 0.005, 4.0, 20, 0.01)
    assert new3["rotation"].shape == (0, 4)


def test_densify_argument_checks(dev):
    from vegs_amd import _capi, optim
    par, m, v, acc, den, rng = _random_model(500, 16, seed=9)
    opt, p = _optimizer(par, m, v, dev, fused=True)
    a, d = torch.tensor(acc, device=dev), torch.tensor(den, device=dev)
    with pytest.raises(ValueError, match="noise"):
        optim.densify_and_prune(opt, a, d, 2e-4, 0.005, 4.0, 20, 0.01, noise=torch.zeros(3, 3, device=dev))
    with pytest.raises(ValueError, match="one value per Gaussian"):
        optim.densify_and_prune(opt, a[:10], d, 2e-4, 0.005, 4.0, 20, 0.01)
    with pytest.raises(ValueError, match="GPU"):
        optim.densify_and_prune(opt, a.cpu(), d, 2e-4, 0.005, 4.0, 20, 0.01)
    bad = torch.optim.Adam([{"params": [p["xyz"]], "name": "xyz"}], lr=0.0)
    with pytest.raises(ValueError, match="missing"):
        optim.densify_and_prune(bad, a, d, 2e-4, 0.005, 4.0, 20, 0.01)
    lib = _capi.load()
    assert lib.vr_densify_plan(None, None, None, None, 5, None, None, None, None) == -1
    assert lib.vr_reset_opacity(None, None, None, 5, 0.01, None) == -1
    assert lib.vr_densify_plan_words(-1) == -1
    # nothing above touched the model
    assert {g["name"]: g for g in opt.param_groups}["xyz"]["params"][0] is p["xyz"]


def test_training_continues_after_densification(dev):
    """train.py:299-315 in sequence on a small street scene: iterations -> densify_and_prune -> iterations -> reset_opacity
    -> iterations; the model grows, every tensor and statistic follows, the loss stays finite."""
    from vegs_amd import harness, iteration, optim, scenes
    sc, deg = scenes.scene_street(P=60000, length=40.0, sh_degree=2, seed=7)
    cams = [scenes.kitti_camera(4.0 * s, 0.2, 344, 96) for s in range(4)]
    cam_ts = [harness.cam_tensors(c, dev) for c in cams]
    tr = iteration.Trainer(sc, dev, fused=True, lrs=iteration.REFERENCE_LRS)
    bg = torch.zeros(3, device=dev)
    with torch.no_grad():
        gts = [harness.render(c, {k: torch.tensor(v, device=dev) for k, v in sc.items()}, deg, bg, cam_t=t)["render"] * 0.9 + 0.05
               for c, t in zip(cams, cam_ts)]
    normal = torch.zeros(3, 96, 344, device=dev)
    normal[2] = 1.0
    for it in range(8):
        loss, _, _ = tr.step(cams[it % 4], cam_ts[it % 4], deg, bg, gts[it % 4], normal)
    P0 = tr.p["xyz"].shape[0]
    seen = int((tr.denom > 0).sum())
    assert seen > 1000
    thr = float((tr.accum / tr.denom.clamp_min(1)).flatten().sort().values[-P0 // 20])      # the top 5 % densify
    P1 = tr.densify_and_prune(thr, 0.005, 20.0, 20)
    assert P1 != P0 and P1 > P0 * 0.9 and tr.accum.shape == (P1, 1) and tr.max_radii.shape == (P1,)
    assert all(v.shape[0] == P1 for v in tr.p.values()) and float(tr.denom.max()) == 0.0
    for it in range(4):
        loss, pkg, _ = tr.step(cams[it % 4], cam_ts[it % 4], deg, bg, gts[it % 4], normal)
        assert torch.isfinite(loss) and pkg["radii"].shape[0] == P1
    tr.p["opacity"] = optim.reset_opacity(tr.opt)
    assert float(torch.sigmoid(tr.p["opacity"].detach()).max()) <= 0.0100001
    loss, _, _ = tr.step(cams[0], cam_ts[0], deg, bg, gts[0], normal)
    assert torch.isfinite(loss) and int((tr.denom > 0).sum()) > 500
