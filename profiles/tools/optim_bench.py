"""Timing of the fused Adam step (row N2) at 2 M Gaussians x 59 floats next to torch.optim.Adam as the reference
builds it (scene/gaussian_model.py:159-168).  PYTHONPATH=. python profiles/tools/optim_bench.py"""
import json
import sys

import torch

from vegs_amd.optim import Adam

DEV = "cuda:0"
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
ONLY = len(sys.argv) > 2 and sys.argv[2] == 'only'   # only this repo's kernel
SHAPES = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (15, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,)}


def make(cls, **kw):
    torch.manual_seed(0)
    ps = {k: torch.nn.Parameter(torch.randn((P,) + s, device=DEV)) for k, s in SHAPES.items()}
    opt = cls([{"params": [p], "lr": 1e-3, "name": k} for k, p in ps.items()], lr=0.0, eps=1e-15, **kw)
    gs = {k: torch.randn_like(p) for k, p in ps.items()}
    return ps, opt, gs


def timeit(cls, iters=20, **kw):
    ps, opt, gs = make(cls, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(iters + 3):
        if it == 3:
            torch.cuda.synchronize(); e0.record()
        for k, p in ps.items():
            p.grad = gs[k]
        opt.step()
        opt.zero_grad(set_to_none=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


n = P * 59
out = {"gaussians": P, "elements": n, "alg_bytes": n * 28}
if not ONLY:
    out["torch_adam_ms"] = round(timeit(torch.optim.Adam), 4)
    out["torch_adam_fused_ms"] = round(timeit(torch.optim.Adam, fused=True), 4)
out["vegs_adam_ms"] = round(timeit(Adam), 4)
out["vegs_adam_GBps"] = round(out["alg_bytes"] / out["vegs_adam_ms"] / 1e6, 1)
out["frac_of_8TBps"] = round(out["vegs_adam_GBps"] / 8000, 3)
print(json.dumps(out))
