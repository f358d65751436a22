import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vegs_amd import harness, scenes, rasterizer
dev = torch.device('cuda:0')
sc, deg = scenes.scene_street(P=2_000_000, length=250.0, sh_degree=3, seed=2)
T = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in sc.items()}
bg = torch.zeros(3, device=dev)
for scale in (0.25, 1.0):
  rng = np.random.default_rng(5)
  Tp = {k: v.detach().clone() for k, v in T.items()}
  Tp["means3D"] += torch.tensor(rng.normal(0, 0.02*scale, tuple(Tp["means3D"].shape)).astype(np.float32), device=dev)
  lo = torch.logit(Tp["opacities"].clamp(1e-4, 1-1e-4)) + torch.tensor(rng.normal(0, 0.3*scale, tuple(Tp["opacities"].shape)).astype(np.float32), device=dev)
  Tp["opacities"] = torch.sigmoid(lo)
  Tp["scales"] = Tp["scales"] * torch.tensor(np.exp(rng.normal(0, 0.05*scale, tuple(Tp["scales"].shape))).astype(np.float32), device=dev)
  for x in (0.0, 40.0):
    cam = scenes.kitti_camera(x, 0.3, 1376, 376)
    ct = harness.cam_tensors(cam, dev)
    rasterizer._NEEDED.clear()
    with torch.no_grad():
        harness.render(cam, Tp, deg, bg, cam_t=ct)
        harness.render(cam, Tp, deg, bg, cam_t=ct)      # (a key's hint array exists from its second sighting on)
        a = list(rasterizer._NEEDED.values())[0].clone().cpu().numpy().astype(np.int64)
        rasterizer._NEEDED.clear()
        harness.render(cam, T, deg, bg, cam_t=ct)
        harness.render(cam, T, deg, bg, cam_t=ct)
        b = list(rasterizer._NEEDED.values())[0].clone().cpu().numpy().astype(np.int64)
    d = b - a
    out = d > 1
    print(f"drift x{scale} view {x}: tiles {len(a)} needed sum old {a.sum()} new {b.sum()}; outgrown by >1: {out.sum()} tiles; "
          f"growth of those: abs p50 {np.percentile(d[out],50) if out.any() else 0} p90 {np.percentile(d[out],90) if out.any() else 0} max {d.max()}; "
          f"relative (d/a) p90 {np.percentile((d[out]/np.maximum(a[out],1)),90) if out.any() else 0:.2f} max {(d[out]/np.maximum(a[out],1)).max() if out.any() else 0:.2f}")
    for marg in ("1", "2+12%", "2+25%"):
        if marg == "1": k = a + 1
        elif marg == "2+12%": k = a + 2 + a // 8
        else: k = a + 2 + a // 4
        miss = np.maximum(b - k, 0)
        print(f"   margin {marg}: tiles with fallback {int((miss>0).sum())}, fallback segments total {int(miss.sum())} max per tile {int(miss.max())}, extra computed {int((np.minimum(k, 10**9) - b).clip(0).sum())}")
