"""Soak run: 3000 fused training iterations with the real densification schedule of train.py:299-315 (statistics every
iteration, densify_and_prune every 100 with a threshold that densifies the top 3 %, reset_opacity at 1500; views of
different sizes interleaved); prints time per iteration and allocator statistics per block.
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
accum = torch.zeros(P, 1, device=dev)
denom = torch.zeros(P, 1, device=dev)
max_radii = torch.zeros(P, device=dev)
t0 = time.perf_counter()
for it in range(3000):
    cam = cams[it % len(cams)]
    key = (cam.image_height, cam.image_width)
    t = {"means3D": p["xyz"], "shs": (p["f_dc"], p["f_rest"]), "opacities": torch.sigmoid(p["opacity"]),
         "scales": torch.exp(p["scaling"]), "rotations": torch.nn.functional.normalize(p["rotation"])}
    pkg = harness.render(cam, t, deg, bg)
    loss, _ = losses.training_loss(pkg["render"], gts[key], types.SimpleNamespace(original_normal=nrm[key], R=scenes.R_KITTI),
                                   pkg["render_cov_quat"], pkg["render_cov_scale"], 0.2, 1e-3, guard_empty=True)
    loss.backward()
    with torch.no_grad():
        optim.add_densification_stats(pkg["viewspace_points"].grad, pkg["radii"], accum, denom, max_radii)
    opt.step()
    opt.zero_grad(set_to_none=True)
    if (it + 1) % 100 == 0:                      # train.py:303-312
        g = (accum / denom.clamp_min(1)).flatten()
        thr = float(g.sort().values[-max(g.numel() * 3 // 100, 1)])
        new, (accum, denom, max_radii) = optim.densify_and_prune(opt, accum, denom, max(thr, 1e-12), 0.005, 30.0, 20, 0.01)
        p = new
    if (it + 1) == 1500:                         # train.py:314-315
        p["opacity"] = optim.reset_opacity(opt)
    if (it + 1) % 500 == 0:
        torch.cuda.synchronize()
        ms = torch.cuda.memory_stats(dev)
        print(f"it {it + 1}: P={p['xyz'].shape[0]} loss={loss.item():.4f} {1e3 * (time.perf_counter() - t0) / 500:.2f} ms/it "
              f"allocated={ms['allocated_bytes.all.current'] / 2**20:.0f} MiB reserved={ms['reserved_bytes.all.current'] / 2**20:.0f} MiB "
              f"hipMallocs={ms['num_device_alloc']}", flush=True)
        assert torch.isfinite(loss) and all(v.shape[0] == p["xyz"].shape[0] for v in p.values())
        t0 = time.perf_counter()
print("soak ok")
