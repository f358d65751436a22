"""How far do the per-tile needed-segment counts move over one training epoch (bench.py drift_by_training: 300 iterations
at the reference's learning rates + one prune/clone)?  Sizes the margin of the needed-segment hints.
    gpurun -- 'python profiles/tools/epoch_drift.py'"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
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
sc2, info = bench.drift_by_training(sc, deg, cams, cam_ts, None, dev, 300)
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
