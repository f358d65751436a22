"""One VEGS training iteration (vegs_amd/iteration.py: the counterpart of train.py:143-168,196,299-320 -- render ->
L1+SSIM + normal guidance -> backward -> densification statistics -> Adam) on the C3 scene, per-view milliseconds:
  A  this rasterizer + the reference's ATen loss code + torch.optim.Adam      (VEGS unmodified on ROCm)
  B  this rasterizer + fused losses (N1) + fused Adam / statistics (N2) (+ fused instance transform N4 with --boxes)
PYTHONPATH=. python profiles/tools/iteration_bench.py [--gaussians 2000000] [--boxes 8]"""
import argparse
import json
import time

import numpy as np
import torch

from vegs_amd import harness, iteration, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=2_000_000)
ap.add_argument("--iters", type=int, default=32)
ap.add_argument("--boxes", type=int, default=0, help="dynamic box instances of 8196 Gaussians each (BASELINE config C5)")
ap.add_argument("--only", choices=["A", "B", "C"], default=None, help="run one mode only (for profiling)")
args = ap.parse_args()
dev = torch.device("cuda:0")
H, W = 376, 1376
sc, deg = scenes.scene_street(P=args.gaussians, length=250.0, sh_degree=3, seed=2)
cams = [scenes.kitti_camera(10.0 * s, y, W, H) for s in range(8) for y in (0.3, -0.3)]
cam_ts = [harness.cam_tensors(c, dev) for c in cams]
rng = np.random.default_rng(0)
gts = [torch.tensor(rng.uniform(0, 1, (3, H, W)).astype(np.float32), device=dev) for _ in range(4)]
normals = [torch.tensor(rng.normal(size=(3, H, W)).astype(np.float32), device=dev) for _ in range(4)]
bg = torch.zeros(3, device=dev)


def run(fused, factored=False):
    tr = iteration.Trainer(sc, dev, n_boxes=args.boxes, fused=fused, factored_sh=factored)

    def it(i):
        v = i % len(cams)
        return tr.step(cams[v], cam_ts[v], deg, bg, gts[i % 4], normals[i % 4])[0]

    for i in range(6):
        it(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(6, 6 + args.iters):
        last = it(i)
    torch.cuda.synchronize()
    out = (time.perf_counter() - t0) / args.iters * 1e3, float(last)
    del tr                          # (a third 5 M model on top of two cached ones measured 10 % slow: start every variant clean)
    torch.cuda.empty_cache()
    return out


if args.only:
    ms, loss = run(args.only != "A", factored=args.only == "C")
    print(json.dumps({"mode": args.only, "ms": round(ms, 3), "loss": loss, "gaussians": args.gaussians, "boxes": args.boxes}))
    raise SystemExit(0)
a_ms, a_loss = run(False)
b_ms, b_loss = run(True)
res = {"gaussians": args.gaussians, "boxes": args.boxes, "frame": [H, W], "iters": args.iters,
       "A_rasterizer_only_ms": round(a_ms, 3), "B_all_fused_ms": round(b_ms, 3),
       "A_iter_per_s": round(1e3 / a_ms, 1), "B_iter_per_s": round(1e3 / b_ms, 1), "loss_A": a_loss, "loss_B": b_loss}
# C: B + factored SH gradient (the static model's dense [P,16,3] gradient is never written or read)
c_ms, c_loss = run(True, factored=True)
res.update({"C_fused_factored_sh_ms": round(c_ms, 3), "C_iter_per_s": round(1e3 / c_ms, 1), "loss_C": c_loss})
print(json.dumps(res))
