"""Timing of the fused per-pixel losses (row N1) on one KITTI-360 frame, next to the same losses written
with ATen ops the way the reference does (utils/loss_utils.py:18-79, loss/normal_guidance.py:3-22) -- i.e.
what VEGS would run on this GPU without the fused kernels.  PYTHONPATH=. python profiles/tools/loss_bench.py"""
import json
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

from vegs_amd import harness, losses, scenes

dev = "cuda:0"
H, W = 376, 1376
rng = np.random.default_rng(0)
x = torch.tensor(rng.uniform(0, 1, (3, H, W)).astype(np.float32), device=dev, requires_grad=True)
y = torch.tensor(rng.uniform(0, 1, (3, H, W)).astype(np.float32), device=dev)
q = torch.tensor(rng.normal(size=(4, H, W)).astype(np.float32), device=dev, requires_grad=True)
s = torch.tensor(rng.uniform(0.01, 0.3, (3, H, W)).astype(np.float32), device=dev, requires_grad=True)
n = torch.tensor(rng.normal(size=(3, H, W)).astype(np.float32), device=dev)
cam = types.SimpleNamespace(original_normal=n, R=scenes.R_KITTI)
g1 = torch.tensor([np.exp(-(i - 5) ** 2 / 4.5) for i in range(11)], dtype=torch.float32)
g1 = (g1 / g1.sum())
win = (g1[:, None] @ g1[None, :]).expand(3, 1, 11, 11).contiguous().to(dev)
Rw = torch.tensor(scenes.R_KITTI, dtype=torch.float32, device=dev)


def aten_losses():
    l1 = (x - y).abs().mean()
    mu1, mu2 = F.conv2d(x, win, padding=5, groups=3), F.conv2d(y, win, padding=5, groups=3)
    s1 = F.conv2d(x * x, win, padding=5, groups=3) - mu1 * mu1
    s2 = F.conv2d(y * y, win, padding=5, groups=3) - mu2 * mu2
    s12 = F.conv2d(x * y, win, padding=5, groups=3) - mu1 * mu2
    ss = (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()
    loss = 0.8 * l1 + 0.2 * (1 - ss)
    cs = s.permute(1, 2, 0).reshape(-1, 1, 3)
    Rm = harness.quaternion_to_matrix(q.permute(1, 2, 0).reshape(-1, 4))
    nw = (Rw @ n.reshape(3, -1)).t()[:, :, None].repeat(1, 1, 3)
    ng = 0.8 * (Rm * nw).sum(-2).abs().mean() + 0.2 * (Rm.detach() * cs * nw).sum(-2).abs().mean()
    return loss + 1e-3 * ng


def fused_losses():
    loss, _ = losses.photometric_loss(x, y, 0.2)
    return loss + 1e-3 * losses.loss_normal_guidance(cam, q, s)


def timeit(fn, iters=50):
    for _ in range(5):
        fn().backward()
    x.grad = q.grad = s.grad = None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn().backward()
        x.grad = q.grad = s.grad = None
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


a, f = aten_losses(), fused_losses()
out = {"frame": [3, H, W], "aten_ms": round(timeit(aten_losses), 4), "fused_ms": round(timeit(fused_losses), 4),
       "loss_aten": a.item(), "loss_fused": f.item()}
# algorithmic bytes of the fused path: photometric fwd reads 2 + writes 3 planes-per-channel, bwd reads 5 + writes 1
# (11 image-sized fp32 arrays of 3 channels); normal guidance fwd reads 10 planes, bwd reads 10 + writes 7
out["alg_bytes"] = 4 * H * W * (11 * 3 + 27)
out["fused_GBps"] = round(out["alg_bytes"] / out["fused_ms"] / 1e6, 1)
print(json.dumps(out))
