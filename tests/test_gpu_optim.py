"""GPU (-m gpu): the fused Adam + densification statistics (row N2) against the reference's own optimizer --
torch.optim.Adam on the CPU, built and edited the way scene/gaussian_model.py:159-168,263-331 does."""
import numpy as np
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (15, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,)}
LRS = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}


def _setup(P, device, cls, seed=0):
    rng = np.random.default_rng(seed)
    params = {k: torch.nn.Parameter(torch.tensor(rng.normal(size=(P,) + s).astype(np.float32), device=device))
              for k, s in NAMES.items()}
    groups = [{"params": [params[k]], "lr": LRS[k], "name": k} for k in NAMES]
    return params, cls(groups, lr=0.0, eps=1e-15)                     # scene/gaussian_model.py:168


def _grads(P, step, scale=1.0):
    rng = np.random.default_rng(1000 + step)
    g = {k: (rng.normal(size=(P,) + s) * scale * 10.0 ** rng.uniform(-6, 0)).astype(np.float32) for k, s in NAMES.items()}
    g["xyz"][::7] = 0.0                                                # invisible Gaussians: exact zero gradients
    return g


@pytest.mark.parametrize("P", [1, 1000, 4099])
def test_fused_adam_tracks_torch_adam(P):
    from vegs_amd.optim import Adam
    ref_p, ref_opt = _setup(P, "cpu", torch.optim.Adam)
    our_p, our_opt = _setup(P, DEV, Adam)
    for step in range(25):
        g = _grads(P, step)
        if step == 10:                                                 # update_learning_rate (gaussian_model.py:174-180)
            for opt in (ref_opt, our_opt):
                for grp in opt.param_groups:
                    if grp["name"] == "xyz":
                        grp["lr"] = 3.1e-5
        for k in NAMES:
            ref_p[k].grad = torch.tensor(g[k])
            our_p[k].grad = torch.tensor(g[k], device=DEV)
        if step == 5:
            our_p["rotation"].grad = ref_p["rotation"].grad = None    # a group without gradient is skipped
        ref_opt.step(); our_opt.step()
        ref_opt.zero_grad(set_to_none=True); our_opt.zero_grad(set_to_none=True)
    for k in NAMES:
        assert rel_err(our_p[k].detach().cpu().numpy(), ref_p[k].detach().numpy()) < 2e-6, k
        so, sr = our_opt.state[our_p[k]], ref_opt.state[ref_p[k]]
        assert float(so["step"]) == float(sr["step"])
        assert rel_err(so["exp_avg"].cpu().numpy(), sr["exp_avg"].numpy()) < 2e-6
        assert rel_err(so["exp_avg_sq"].cpu().numpy(), sr["exp_avg_sq"].numpy()) < 2e-6


def test_fused_adam_state_edits_as_the_reference_does_them():
    """prune + densify the optimizer state the way scene/gaussian_model.py:277-331 does, keep stepping, and
    round-trip state_dict() (scene/gaussian_model.py:68: capture())."""
    from vegs_amd.optim import Adam
    P = 500
    sets = [_setup(P, "cpu", torch.optim.Adam), _setup(P, DEV, Adam)]

    def run(steps, first):
        for step in range(first, first + steps):
            for params, opt in sets:
                n = params["xyz"].shape[0]
                g = _grads(n, step)
                for k in NAMES:
                    params[k].grad = torch.tensor(g[k], device=params[k].device)
                opt.step(); opt.zero_grad(set_to_none=True)

    run(4, 0)
    mask_np = np.random.default_rng(7).uniform(size=P) > 0.3
    for params, opt in sets:
        dev = params["xyz"].device
        mask = torch.tensor(mask_np, device=dev)
        for group in opt.param_groups:                                 # _prune_optimizer
            st = opt.state.get(group["params"][0], None)
            st["exp_avg"] = st["exp_avg"][mask]
            st["exp_avg_sq"] = st["exp_avg_sq"][mask]
            del opt.state[group["params"][0]]
            group["params"][0] = torch.nn.Parameter(group["params"][0][mask].requires_grad_(True))
            opt.state[group["params"][0]] = st
            params[group["name"]] = group["params"][0]
        for group in opt.param_groups:                                 # cat_tensors_to_optimizer
            ext = torch.full((37,) + NAMES[group["name"]], 0.25, device=dev)
            st = opt.state.get(group["params"][0], None)
            st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
            del opt.state[group["params"][0]]
            group["params"][0] = torch.nn.Parameter(torch.cat((group["params"][0], ext), dim=0).requires_grad_(True))
            opt.state[group["params"][0]] = st
            params[group["name"]] = group["params"][0]
    run(4, 4)
    sd = sets[1][1].state_dict()
    fresh = __import__("vegs_amd.optim", fromlist=["Adam"]).Adam(
        [{"params": [sets[1][0][k]], "lr": LRS[k], "name": k} for k in NAMES], lr=0.0, eps=1e-15)
    fresh.load_state_dict(sd)
    sets[1] = (sets[1][0], fresh)
    run(3, 8)
    for k in NAMES:
        assert sets[0][0][k].shape == sets[1][0][k].shape
        assert rel_err(sets[1][0][k].detach().cpu().numpy(), sets[0][0][k].detach().numpy()) < 2e-6, k


def test_densification_stats_match_the_reference_expressions():
    from vegs_amd.optim import add_densification_stats
    rng = np.random.default_rng(2)
    P = 10_000
    grad = rng.normal(size=(P, 3)).astype(np.float32)
    radii = (rng.integers(0, 40, P) * (rng.uniform(size=P) > 0.4)).astype(np.int32)
    accum, denom = rng.uniform(0, 1, (P, 1)).astype(np.float32), rng.integers(0, 5, (P, 1)).astype(np.float32)
    maxr = rng.integers(0, 30, P).astype(np.float32)
    # reference expressions on the CPU (scene/gaussian_model.py:411-413, train.py:299)
    t = {k: torch.tensor(v) for k, v in dict(grad=grad, radii=radii, accum=accum, denom=denom, maxr=maxr).items()}
    vis = t["radii"] > 0
    t["maxr"][vis] = torch.max(t["maxr"][vis], t["radii"][vis])
    t["accum"][vis] += torch.norm(t["grad"][vis, :2], dim=-1, keepdim=True)
    t["denom"][vis] += 1
    d = {k: torch.tensor(v, device=DEV) for k, v in dict(grad=grad, radii=radii, accum=accum, denom=denom, maxr=maxr).items()}
    add_densification_stats(d["grad"], d["radii"], d["accum"], d["denom"], d["maxr"])
    assert np.array_equal(d["denom"].cpu().numpy(), t["denom"].numpy())
    assert np.array_equal(d["maxr"].cpu().numpy(), t["maxr"].numpy())
    assert np.abs(d["accum"].cpu().numpy() - t["accum"].numpy()).max() < 1e-6
    with pytest.raises(ValueError):
        add_densification_stats(d["grad"], d["radii"].long(), d["accum"], d["denom"], d["maxr"])


def test_fused_adam_many_tensors_with_an_empty_one():
    """More tensors than one launch carries (8) with an EMPTY one among the first batch: every tensor must be updated
    exactly once per step (the batching once advanced by 8 although the empty tensor had taken no slot, and
    repeated a tensor)."""
    from vegs_amd.optim import Adam
    rng = np.random.default_rng(5)
    sizes = [5, 0, 7, 300, 1, 64, 33, 2, 1025, 17, 4, 9]          # 12 tensors, the second one empty
    ref = [torch.nn.Parameter(torch.tensor(rng.normal(size=(n, 3)).astype(np.float32))) for n in sizes]
    our = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref]
    ref_opt = torch.optim.Adam([{"params": [p], "lr": 1e-2 * (i + 1)} for i, p in enumerate(ref)], lr=0.0, eps=1e-15)
    our_opt = Adam([{"params": [p], "lr": 1e-2 * (i + 1)} for i, p in enumerate(our)], lr=0.0, eps=1e-15)
    for step in range(4):
        for p, q in zip(ref, our):
            g = rng.normal(size=tuple(p.shape)).astype(np.float32)
            p.grad, q.grad = torch.tensor(g), torch.tensor(g, device=DEV)
        ref_opt.step(); our_opt.step()
    for i, (p, q) in enumerate(zip(ref, our)):
        assert q.shape == p.shape
        if p.numel():
            assert rel_err(q.detach().cpu().numpy(), p.detach().numpy()) < 2e-6, i
            assert float(our_opt.state[q]["step"]) == 4.0


def test_step_many_is_all_or_nothing():
    """A batch of optimizers is validated BEFORE any of them is touched: when a later optimizer is rejected (weight decay,
    a CPU parameter -- there is no CPU path --, state of another shape), the earlier ones keep their step counters, get no
    freshly initialised state and no kernel runs (advisor finding, round 4: the counters used to advance first, which skews
    the bias correction of a retry and desynchronises the ranks of a distributed job)."""
    from vegs_amd import optim
    dev = torch.device("cuda:0")

    def fresh(cls=optim.Adam, device=dev, **kw):
        p = torch.nn.Parameter(torch.ones(4, 3, device=device))
        p.grad = torch.ones(4, 3, device=device)
        return cls([p], lr=1e-2, eps=1e-15, **kw), p

    good, pg = fresh()
    good.step()
    assert float(good.state[pg]["step"]) == 1.0
    before = pg.detach().clone()
    untouched, pu = fresh()                               # no state yet: must stay that way
    mism, pm = fresh()
    mism.state[pm] = {"step": torch.tensor(3.0), "exp_avg": torch.zeros(2, 3, device=dev), "exp_avg_sq": torch.zeros(4, 3, device=dev)}
    for bad, exc in ((fresh(torch.optim.Adam, weight_decay=0.1)[0], NotImplementedError), (fresh(device="cpu")[0], ValueError),
                     (mism, ValueError)):
        with pytest.raises(exc):
            optim.step_many([good, untouched, bad])
        torch.cuda.synchronize()
        assert float(good.state[pg]["step"]) == 1.0 and len(untouched.state.get(pu, {})) == 0
        assert torch.equal(pg.detach(), before) and torch.equal(pu.detach(), torch.ones(4, 3, device=dev))
    assert float(mism.state[pm]["step"]) == 3.0
    optim.step_many([good, untouched])                    # and the valid batch goes through
    assert float(good.state[pg]["step"]) == 2.0 and float(untouched.state[pu]["step"]) == 1.0
