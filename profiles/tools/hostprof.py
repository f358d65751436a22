import sys, time, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vegs_amd import _capi, harness, scenes
dev = torch.device('cuda:0')
sc, deg = scenes.scene_street(P=2_000_000, length=250.0, sh_degree=3, seed=2)
T = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in sc.items()}
params = list(T.values())
cams = [scenes.kitti_camera(10.0 * s, y, 1376, 376) for s in range(8) for y in (0.3, -0.3)]
cam_ts = [harness.cam_tensors(c, dev) for c in cams]
bg = torch.zeros(3, device=dev)
H, W = 376, 1376
g = [torch.randn(3, H, W, device=dev) * 1e-6, torch.randn(4, H, W, device=dev) * 1e-6, torch.randn(3, H, W, device=dev) * 1e-6]
def step(i):
    v = i % 16
    pkg = harness.render(cams[v], T, deg, bg, cam_t=cam_ts[v])
    torch.autograd.backward([pkg["render"], pkg["render_cov_quat"], pkg["render_cov_scale"]], g)
    for p in params: p.grad = None
for i in range(8): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(64): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host-side per step %.3f ms ; incl. final drain %.3f ms" % ((t1 - t0) / 64 * 1e3, (t2 - t0) / 64 * 1e3))
pr = cProfile.Profile(); pr.enable()
for i in range(64): step(i)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18); print(s.getvalue()[:3500])
