import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vegs_amd import harness, scenes
dev = torch.device('cuda:0')
sc, deg = scenes.scene_street(P=2_000_000, length=250.0, sh_degree=3, seed=2)
T = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in sc.items()}
dc = T["shs"][:, :1].detach().clone().requires_grad_(True); rest = T["shs"][:, 1:].detach().clone().requires_grad_(True)
cams = [scenes.kitti_camera(10.0 * s, y, 1376, 376) for s in range(8) for y in (0.3, -0.3)]
cam_ts = [harness.cam_tensors(c, dev) for c in cams]
bg = torch.zeros(3, device=dev); H, W = 376, 1376
g = [torch.randn(3, H, W, device=dev) * 1e-6, torch.randn(4, H, W, device=dev) * 1e-6, torch.randn(3, H, W, device=dev) * 1e-6]
for split in (False, True):
    t = dict(T)
    if split: t["shs"] = (dc, rest)
    def step(i):
        pkg = harness.render(cams[i % 16], t, deg, bg, cam_t=cam_ts[i % 16])
        torch.autograd.backward([pkg["render"], pkg["render_cov_quat"], pkg["render_cov_scale"]], g)
        for p in list(T.values()) + [dc, rest]: p.grad = None
    for i in range(8): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(64): step(i)
    torch.cuda.synchronize()
    print("split" if split else "cat-free single tensor", (time.perf_counter() - t0) / 64 * 1e3, "ms/view")
