import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vegs_amd import harness, scenes
dev = torch.device('cuda:0')
cam = scenes.camera_c1(256, 256)
for name, P, spread, opac, scale in [("one tile opaque", 300_000, 0.01, 0.9, 0.01), ("one tile faint", 300_000, 0.01, 0.006, 0.01),
                                      ("screen-filling x20k", 20_000, 0.3, 0.05, 2.0), ("one pixel 1M", 1_000_000, 1e-4, 0.02, 1e-3)]:
    rng = np.random.default_rng(0)
    sc, deg = scenes.scene_random(P=P, sh_degree=0, seed=1, scale=scale, extent=spread)
    sc["opacities"][:] = opac
    t = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in sc.items()}
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pkg = harness.render(cam, t, deg, torch.zeros(3, device=dev))
        pkg["render"].sum().backward()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fn = pkg["render"].grad_fn
    print(f"{name}: P={P} R={fn.num_rendered} fwd+bwd {dt*1e3:.1f} ms  finite={bool(torch.isfinite(t['means3D'].grad).all())}", flush=True)
