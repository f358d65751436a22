"""Timing of the fused instance transform + concatenation (row N4): a 2 M static model + 8 box instances of
8196 Gaussians (scene/gaussian_model.py:462), forward + backward of the op-input construction alone
(gaussian_renderer/__init__.py:274-303).  PYTHONPATH=. python profiles/tools/instances_bench.py"""
import json
import time

import numpy as np
import torch

from vegs_amd import harness, scenes
from vegs_amd.instances import prepare_and_merge

DEV = "cuda:0"
P, nb, n = 2_000_000, 8, 8196
rng = np.random.default_rng(0)
sc, deg = scenes.scene_random(P=P, sh_degree=3, seed=1)
static = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in sc.items()}
boxes, b2ws = [], []
for i in range(nb):
    b, _ = scenes.scene_random(P=n, sh_degree=3, seed=10 + i, extent=0.2)
    boxes.append({k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in b.items()})
    B = np.eye(4, dtype=np.float32)
    B[:3, :3] = harness.quaternion_to_matrix(torch.tensor(rng.normal(size=4))).numpy() * 1.3
    B[:3, 3] = rng.normal(size=3)
    b2ws.append(torch.tensor(B, device=DEV, requires_grad=True))
Ptot = P + nb * n
gout = {"means3D": torch.randn(Ptot, 3, device=DEV), "scales": torch.randn(Ptot, 3, device=DEV),
        "rotations": torch.randn(Ptot, 4, device=DEV), "shs": torch.randn(Ptot, 16, 3, device=DEV),
        "opacities": torch.randn(Ptot, 1, device=DEV)}


def op_by_op():
    kw = harness.prepare_rasterization(static)
    for t, b in zip(boxes, b2ws):
        kw = harness.merge_kwargs(kw, harness.prepare_rasterization(t, b))
    return kw


def fused():
    return prepare_and_merge(static, boxes, b2ws)


def timeit(fn, iters=20):
    def one():
        kw = fn()
        torch.autograd.backward([kw[k] for k in gout], [gout[k] for k in gout])
        for t in [static] + boxes:
            for v in t.values():
                v.grad = None
        for b in b2ws:
            b.grad = None
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


print(json.dumps({"static": P, "instances": nb, "gaussians_per_instance": n,
                  "op_by_op_ms": round(timeit(op_by_op), 3), "fused_ms": round(timeit(fused), 3)}))
