"""How far do the per-tile needed-segment counts move over one training epoch (drift_by_training below: 300 iterations
at the reference's learning rates + one prune/clone)?  Sizes the margin of the needed-segment hints.
    gpurun -- 'python profiles/tools/epoch_drift.py'"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import numpy as np
import torch


# (moved here from bench.py in round 4, when the stale-hint variant left the bench line: hints are an evaluation-only
# feature now, vegs_amd/rasterizer.py)
def drift_by_training(sc, deg, cams, cam_ts, gouts, device, iters, seed=9):
    """What lies between two visits of a camera in the reference's loop (train.py:126-128 pops cameras without
    replacement: one visit per epoch): `iters` iterations of the counterpart of train.py:143-168,196,292-320 -- render,
    L1+SSIM + normal guidance against fixed random targets, backward, densification statistics, fused Adam at the
    REFERENCE's learning rates -- followed by one prune (opacity < 0.005, scene/gaussian_model.py:397-407) and one clone
    of the Gaussians with the largest accumulated screen-space gradient (top 1 %, :365-395).  Runs with the hint cache
    suspended (the cache keeps what the cameras' LAST visit before this epoch recorded).  Returns the new scene dict."""
    from vegs_amd import iteration, rasterizer
    rng = np.random.default_rng(seed)
    tr = iteration.Trainer(sc, device, fused=True, lrs=iteration.REFERENCE_LRS)
    bg = torch.zeros(3, device=device)
    H, W = cams[0].image_height, cams[0].image_width
    gts = [torch.tensor(rng.uniform(0, 1, (3, H, W)).astype(np.float32), device=device) for _ in range(4)]
    normals = [torch.tensor(rng.normal(size=(3, H, W)).astype(np.float32), device=device) for _ in range(4)]
    was = rasterizer._use_hints
    rasterizer._use_hints = False          # suspended, NOT cleared
    try:
        for it in range(iters):
            v = int(rng.integers(len(cams)))
            tr.step(cams[v], cam_ts[v], deg, bg, gts[it % 4], normals[it % 4])
    finally:
        rasterizer._use_hints = was
    T = iteration.op_inputs(tr.p)
    grad = (tr.accum / tr.denom.clamp(min=1)).reshape(-1)
    keep = T["opacities"].reshape(-1) >= 0.005
    thr = torch.quantile(grad[torch.randperm(grad.numel(), device=device)[:1_000_000]], 0.99)
    clone = keep & (grad >= thr)
    idx = torch.cat((torch.nonzero(keep).reshape(-1), torch.nonzero(clone).reshape(-1)))
    out = {k: v[idx].contiguous().cpu().numpy() for k, v in T.items()}
    info = {"iterations": iters, "pruned": int((~keep).sum()), "cloned": int(clone.sum()), "gaussians_after": int(idx.numel())}
    del tr
    torch.cuda.empty_cache()
    return out, info

from vegs_amd import harness, rasterizer, scenes
dev = torch.device("cuda:0")
sc, deg = scenes.scene_street(P=2_000_000, length=250.0, sh_degree=3, seed=2)
cams = [scenes.kitti_camera(10.0 * s, y, 1376, 376) for s in range(8) for y in (0.3, -0.3)]
cam_ts = [harness.cam_tensors(c, dev) for c in cams]
bg = torch.zeros(3, device=dev)


def needed(scd):
    T = {k: torch.tensor(v, device=dev) for k, v in scd.items()}
    out = []
    rasterizer.needed_hints(True)
    with torch.no_grad():
        for c, ct in zip(cams, cam_ts):
            rasterizer._NEEDED.clear()
            harness.render(c, T, deg, bg, cam_t=ct)
            harness.render(c, T, deg, bg, cam_t=ct)
            out.append(list(rasterizer._NEEDED.values())[0].clone().cpu().numpy().astype(np.int64))
    return np.stack(out)


a = needed(sc)
rasterizer.needed_hints(False)
sc2, info = drift_by_training(sc, deg, cams, cam_ts, None, dev, 300)
b = needed(sc2)
print(info)
print("needed segments per view: before", a.sum(1).mean(), "after", b.sum(1).mean())
for name, lim in (("h+2+h/8", a + 2 + a // 8), ("h+3+h/4", a + 3 + a // 4), ("h+4+h/2", a + 4 + a // 2), ("h+4+h", a + 4 + a)):
    short = b > lim
    print(f"margin {name}: short tiles per view {short.sum(1).mean():.1f}; segments beyond the limit {np.maximum(b - lim, 0).sum(1).mean():.0f}; "
          f"computed up front {lim.sum(1).mean():.0f} (needed {b.sum(1).mean():.0f})")
big = a >= 20
r = b[big] / a[big]
print("tiles with >= 20 needed segments: growth ratio percentiles", {q: round(float(np.percentile(r, q)), 2) for q in (5, 25, 50, 75, 95, 99)})
