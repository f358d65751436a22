"""Soak run: 3000 fused training iterations on a model that grows like a densifying one (P +3 % every 100
iterations, views of different sizes interleaved); prints time per iteration and allocator statistics per block.
PYTHONPATH=. python profiles/tools/soak.py"""
import time
import types

import numpy as np
import torch

from vegs_amd import harness, losses, optim, scenes

dev = torch.device("cuda:0")
P = 150_000
sc, deg = scenes.scene_street(P=P, length=120.0, sh_degree=3, seed=3)
cams = [scenes.kitti_camera(8.0 * s, y, w, h) for s in range(6) for y in (0.3, -0.3) for (w, h) in ((1376, 376), (688, 188))]
bg = torch.zeros(3, device=dev)
rng = np.random.default_rng(0)


def make(sc):
    t = {k: torch.tensor(v, device=dev) for k, v in sc.items()}
    p = {"xyz": t["means3D"].clone(), "f_dc": t["shs"][:, :1].contiguous(), "f_rest": t["shs"][:, 1:].contiguous(),
         "opacity": torch.logit(t["opacities"].clamp(1e-4, 1 - 1e-4)), "scaling": torch.log(t["scales"]), "rotation": t["rotations"].clone()}
    p = {k: torch.nn.Parameter(v) for k, v in p.items()}
    return p, optim.Adam([{"params": [p[k]], "lr": 1e-4, "name": k} for k in p], lr=0.0, eps=1e-15)


p, opt = make(sc)
gts = {(c.image_height, c.image_width): torch.rand(3, c.image_height, c.image_width, device=dev) for c in cams}
nrm = {(c.image_height, c.image_width): torch.randn(3, c.image_height, c.image_width, device=dev) for c in cams}
t0 = time.perf_counter()
for it in range(3000):
    cam = cams[it % len(cams)]
    key = (cam.image_height, cam.image_width)
    t = {"means3D": p["xyz"], "shs": (p["f_dc"], p["f_rest"]), "opacities": torch.sigmoid(p["opacity"]),
         "scales": torch.exp(p["scaling"]), "rotations": torch.nn.functional.normalize(p["rotation"])}
    pkg = harness.render(cam, t, deg, bg)
    q = pkg["render_cov_quat"]
    q = torch.where((q.detach() ** 2).sum(0, keepdim=True) > 0, q, torch.ones_like(q))
    loss, _ = losses.photometric_loss(pkg["render"], gts[key], 0.2)
    loss = loss + 1e-3 * losses.loss_normal_guidance(types.SimpleNamespace(original_normal=nrm[key], R=scenes.R_KITTI), q, pkg["render_cov_scale"])
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    if (it + 1) % 100 == 0:                      # "densify": clone 3 % of the Gaussians, rebuild the optimizer state
        n = p["xyz"].shape[0]
        idx = torch.randint(0, n, (n * 3 // 100,), device=dev)
        for group in opt.param_groups:
            old = group["params"][0]
            st = opt.state.pop(old)
            new = torch.nn.Parameter(torch.cat((old.detach(), old.detach()[idx]), 0))
            st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(old.detach()[idx])), 0)
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(old.detach()[idx])), 0)
            group["params"][0] = new
            opt.state[new] = st
            p[group["name"]] = new
    if (it + 1) % 500 == 0:
        torch.cuda.synchronize()
        ms = torch.cuda.memory_stats(dev)
        print(f"it {it + 1}: P={p['xyz'].shape[0]} loss={loss.item():.4f} {1e3 * (time.perf_counter() - t0) / 500:.2f} ms/it "
              f"allocated={ms['allocated_bytes.all.current'] / 2**20:.0f} MiB reserved={ms['reserved_bytes.all.current'] / 2**20:.0f} MiB "
              f"hipMallocs={ms['num_device_alloc']}", flush=True)
        assert torch.isfinite(loss)
        t0 = time.perf_counter()
print("soak ok")
