"""Densification step (scene/gaussian_model.py:384-403) at the headline model size: the planned gather of
vegs_amd.optim.densify_and_prune against the reference's own sequence of ATen operations (cat / boolean-mask gathers per
tensor and Adam moment, written out below from the reference's description of the step), both on the GPU, same inputs.
    python profiles/tools/densify_bench.py [P] > profiles/<tag>_densify.json"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from vegs_amd import optim  # noqa: E402

NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


def model(P, dev, seed=1):
    g = torch.Generator(device=dev).manual_seed(seed)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    par = {"xyz": r(P, 3) * 3, "f_dc": r(P, 1, 3), "f_rest": r(P, 15, 3) * .2, "opacity": r(P, 1) * 3 - 1,
           "scaling": r(P, 3) * 1.2 + float(np.log(0.03)), "rotation": r(P, 4)}
    den = torch.randint(0, 40, (P, 1), device=dev, generator=g).float()
    acc = den * torch.exp(r(P, 1) + float(np.log(1.2e-4)))
    return par, acc, den


def optimizer(par, cls):
    p = {k: torch.nn.Parameter(v.clone()) for k, v in par.items()}
    opt = cls([{"params": [p[k]], "lr": 1e-3, "name": k} for k in NAMES], lr=0.0, eps=1e-15)
    for k in NAMES:
        p[k].grad = torch.full_like(p[k], 1e-3)
    opt.step()
    opt.zero_grad(set_to_none=True)
    return opt, p


def aten_sequence(opt, acc, den, max_grad, min_opacity, extent, max_screen_size, pd, noise):
    """the reference's order of operations: clone -> cat, split -> cat -> mask, prune -> mask, each over parameters and moments"""
    by = {g["name"]: g for g in opt.param_groups}
    cur = lambda k: by[k]["params"][0]

    def cat(ext):
        for k in NAMES:
            old = cur(k)
            st = opt.state.get(old)
            st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext[k])), 0)
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext[k])), 0)
            del opt.state[old]
            by[k]["params"][0] = torch.nn.Parameter(torch.cat((old, ext[k]), 0).requires_grad_(True))
            opt.state[cur(k)] = st

    def keep(mask):
        for k in NAMES:
            old = cur(k)
            st = opt.state.get(old)
            st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][mask], st["exp_avg_sq"][mask]
            del opt.state[old]
            by[k]["params"][0] = torch.nn.Parameter(old[mask].requires_grad_(True))
            opt.state[cur(k)] = st
    with torch.no_grad():
        grads = acc / den
        grads[grads.isnan()] = 0.0
        sel = (torch.norm(grads, dim=-1) >= max_grad) & (torch.exp(cur("scaling")).max(dim=1).values <= pd * extent)
        cat({k: cur(k)[sel] for k in NAMES})
        n = cur("xyz").shape[0]
        padded = torch.zeros(n, device=acc.device)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = (padded >= max_grad) & (torch.exp(cur("scaling")).max(dim=1).values > pd * extent)
        stds = torch.exp(cur("scaling")[sel]).repeat(2, 1)
        samples = stds * noise
        q = cur("rotation")[sel]
        q = q / q.norm(dim=1, keepdim=True)
        w, x, y, z = q.unbind(1)
        R = torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                         2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)), 1).reshape(-1, 3, 3)
        ext = {k: cur(k)[sel].repeat(2, *([1] * (cur(k).dim() - 1))) for k in NAMES}
        ext["xyz"] = torch.bmm(R.repeat(2, 1, 1), samples.unsqueeze(-1)).squeeze(-1) + ext["xyz"]
        ext["scaling"] = torch.log(stds / 1.6)
        cat(ext)
        keep(~torch.cat((sel, torch.zeros(2 * int(sel.sum()), device=acc.device, dtype=torch.bool))))
        mask = (torch.sigmoid(cur("opacity")) < min_opacity).squeeze()
        if max_screen_size:
            mask = mask | (torch.exp(cur("scaling")).max(dim=1).values > 0.1 * extent)
        keep(~mask)
    return {k: cur(k) for k in NAMES}


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    dev = torch.device("cuda", 0)
    par, acc, den = model(P, dev)
    cfg = (2e-4, 0.005, 4.0, 20, 0.01)
    res = {"P": P}
    # sizes + the draw (shared by both variants)
    opt, p = optimizer(par, optim.Adam)
    new, _ = optim.densify_and_prune(opt, acc, den, *cfg)
    n_out = new["xyz"].shape[0]
    probe_opt, _ = optimizer(par, torch.optim.Adam)
    import ctypes as C
    from vegs_amd import _capi
    lib = _capi.load()
    plan = torch.empty(int(lib.vr_densify_plan_words(P)), dtype=torch.int32, device=dev)
    counts = torch.empty(8, dtype=torch.int32, device=dev)
    st = _capi.VrDensifySettings(cfg[0], cfg[1], cfg[2], cfg[4], 1)
    _capi.check(lib.vr_densify_plan(p["opacity"].data_ptr(), p["scaling"].data_ptr(), acc.data_ptr(), den.data_ptr(), P, C.byref(st),
                                    plan.data_ptr(), counts.data_ptr(), torch.cuda.current_stream().cuda_stream))
    n_out2, nA, nB, nC, S = counts[:5].tolist()
    res.update(rows_out=n_out2, kept=nA, clones=nB, split_pairs=nC, split_originals=S)
    noise = torch.randn(2 * S, 3, device=dev)
    del opt, p, new, probe_opt

    def timed(fn, reps=5):
        ts = []
        for _ in range(reps):
            state = fn(None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(state)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            del state, out
            torch.cuda.empty_cache()
        return float(np.median(ts)), [round(t, 3) for t in ts]

    def mine(state):
        if state is None:
            return optimizer(par, optim.Adam)
        return optim.densify_and_prune(state[0], acc, den, *cfg, noise=noise, empty_cache=False)   # (the allocator flush of :403 is in neither variant)

    def aten(state):
        if state is None:
            return optimizer(par, torch.optim.Adam)
        return aten_sequence(state[0], acc, den, *cfg, noise=noise)
    res["planned_gather_ms"], res["planned_gather_runs"] = timed(mine)
    res["aten_sequence_ms"], res["aten_sequence_runs"] = timed(aten)
    # same result?
    o1, _ = mine(mine(None))
    o2 = aten(aten(None))
    res["same_rows"] = all(o1[k].shape == o2[k].shape for k in NAMES)
    res["max_abs_diff"] = {k: float((o1[k].detach() - o2[k].detach()).abs().max()) for k in NAMES} if res["same_rows"] else None
    bytes_moved = 2 * 3 * 59 * 4 * n_out2
    res["bytes_read_plus_written"] = bytes_moved
    res["planned_gather_GBps"] = bytes_moved / res["planned_gather_ms"] / 1e6
    print(json.dumps(res))


if __name__ == "__main__":
    main()
