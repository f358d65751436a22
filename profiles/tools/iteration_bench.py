"""One VEGS training iteration (the counterpart of train.py:143-168,196,299-320: render -> L1+SSIM + normal
guidance -> backward -> densification statistics -> Adam) on the C3 scene, per-view milliseconds:
  A  this rasterizer + the reference's ATen loss code + torch.optim.Adam      (VEGS unmodified on ROCm)
  B  this rasterizer + fused losses (N1) + fused Adam / statistics (N2)
PYTHONPATH=. python profiles/tools/iteration_bench.py [--gaussians 2000000]"""
import argparse
import json
import time
import types

import numpy as np
import torch
import torch.nn.functional as F

from vegs_amd import harness, losses, optim, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=2_000_000)
ap.add_argument("--iters", type=int, default=32)
ap.add_argument("--boxes", type=int, default=0, help="dynamic box instances of 8196 Gaussians each (BASELINE config C5)")
args = ap.parse_args()
dev = torch.device("cuda:0")
H, W = 376, 1376
sc, deg = scenes.scene_street(P=args.gaussians, length=250.0, sh_degree=3, seed=2)
cams = [scenes.kitti_camera(10.0 * s, y, W, H) for s in range(8) for y in (0.3, -0.3)]
cam_ts = [harness.cam_tensors(c, dev) for c in cams]
rng = np.random.default_rng(0)
gts = [torch.tensor(rng.uniform(0, 1, (3, H, W)).astype(np.float32), device=dev) for _ in range(4)]
normals = [torch.tensor(rng.normal(size=(3, H, W)).astype(np.float32), device=dev) for _ in range(4)]
g1 = torch.tensor([np.exp(-(i - 5) ** 2 / 4.5) for i in range(11)], dtype=torch.float32)
g1 = g1 / g1.sum()
win = (g1[:, None] @ g1[None, :]).expand(3, 1, 11, 11).contiguous().to(dev)
Rw = torch.tensor(scenes.R_KITTI, dtype=torch.float32, device=dev)
bg = torch.zeros(3, device=dev)


def model():
    """raw parameters + optimizer groups as scene/gaussian_model.py:145-168"""
    t = {k: torch.tensor(v, device=dev) for k, v in sc.items()}
    p = {"xyz": t["means3D"].clone(), "f_dc": t["shs"][:, :1].contiguous(), "f_rest": t["shs"][:, 1:].contiguous(),
         "opacity": torch.logit(t["opacities"].clamp(1e-4, 1 - 1e-4)), "scaling": torch.log(t["scales"]),
         "rotation": t["rotations"].clone()}
    p = {k: torch.nn.Parameter(v.requires_grad_(True)) for k, v in p.items()}
    lrs = {"xyz": 1.6e-6, "f_dc": 2.5e-4, "f_rest": 2.5e-4 / 20, "opacity": 5e-3, "scaling": 5e-4, "rotation": 1e-4}
    return p, [{"params": [p[k]], "lr": lrs[k], "name": k} for k in p]


BOX = []
if args.boxes:
    brng = np.random.default_rng(5)
    for i in range(args.boxes):
        b, _ = scenes.scene_random(P=8196, sh_degree=3, seed=100 + i, extent=1.0, scale=0.05)
        B = np.eye(4, dtype=np.float32)
        ang = brng.uniform(0, 6.28)
        B[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32) * 1.5
        B[:3, 3] = [10.0 + 12.0 * i, brng.uniform(-3, 3), -0.8]
        BOX.append(({k: torch.tensor(v, device=dev, requires_grad=True) for k, v in b.items()},
                    torch.tensor(B, device=dev, requires_grad=True)))


def render(p, v, fused):
    t = {"means3D": p["xyz"], "shs": None if (fused and not BOX) else torch.cat((p["f_dc"], p["f_rest"]), dim=1),
         "opacities": torch.sigmoid(p["opacity"]),
         "scales": torch.exp(p["scaling"]), "rotations": F.normalize(p["rotation"])}          # gaussian_model.py:100-120
    if not BOX:
        if fused:   # the model's two SH tensors as they are: no torch.cat, no slicing copies in the backward
            t["shs"] = (p["f_dc"], p["f_rest"])
        return harness.render(cams[v], t, deg, bg, cam_t=cam_ts[v])
    return harness.render_all(cams[v], t, [b for b, _ in BOX], [w for _, w in BOX], deg, bg, cam_t=cam_ts[v], fused=fused)


def aten_loss(pkg, gt, normal):
    x, q, s = pkg["render"], pkg["render_cov_quat"], pkg["render_cov_scale"]
    l1 = (x - gt).abs().mean()
    mu1, mu2 = F.conv2d(x, win, padding=5, groups=3), F.conv2d(gt, win, padding=5, groups=3)
    s1 = F.conv2d(x * x, win, padding=5, groups=3) - mu1 * mu1
    s2 = F.conv2d(gt * gt, win, padding=5, groups=3) - mu2 * mu2
    s12 = F.conv2d(x * gt, win, padding=5, groups=3) - mu1 * mu2
    ss = (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()
    Rm = harness.quaternion_to_matrix(q.permute(1, 2, 0).reshape(-1, 4))
    nw = (Rw @ normal.reshape(3, -1)).t()[:, :, None].repeat(1, 1, 3)
    ng = 0.8 * (Rm * nw).sum(-2).abs().mean() + 0.2 * (Rm.detach() * s.permute(1, 2, 0).reshape(-1, 1, 3) * nw).sum(-2).abs().mean()
    return 0.8 * l1 + 0.2 * (1 - ss) + 1e-3 * ng


def fused_loss(pkg, gt, normal):
    loss, _ = losses.photometric_loss(pkg["render"], gt, 0.2)
    cam = types.SimpleNamespace(original_normal=normal, R=scenes.R_KITTI)
    return loss + 1e-3 * losses.loss_normal_guidance(cam, pkg["render_cov_quat"], pkg["render_cov_scale"])


def run(fused):
    p, groups = model()
    opt = (optim.Adam if fused else torch.optim.Adam)(groups, lr=0.0, eps=1e-15)
    P = p["xyz"].shape[0] + 8196 * len(BOX)     # statistics over the concatenated op inputs
    accum, denom, maxr = (torch.zeros(P, 1, device=dev), torch.zeros(P, 1, device=dev), torch.zeros(P, device=dev))

    def it(i):
        v = i % len(cams)
        pkg = render(p, v, fused)
        # NaN guard for pixels no Gaussian covers (the reference's loss would be NaN there): same in both variants
        q = pkg["render_cov_quat"]
        pkg["render_cov_quat"] = torch.where((q.detach() * q.detach()).sum(0, keepdim=True) > 0, q, torch.ones_like(q))
        loss = (fused_loss if fused else aten_loss)(pkg, gts[i % 4], normals[i % 4])
        loss.backward()
        with torch.no_grad():
            vis, radii, vsp = pkg["visibility_filter"], pkg["radii"], pkg["viewspace_points"]
            if fused:
                optim.add_densification_stats(vsp.grad, radii, accum, denom, maxr)
            else:
                maxr[vis] = torch.max(maxr[vis], radii[vis].float())                       # train.py:299
                accum[vis] += torch.norm(vsp.grad[vis, :2], dim=-1, keepdim=True)          # gaussian_model.py:411-413
                denom[vis] += 1
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    for i in range(6):
        it(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(6, 6 + args.iters):
        last = it(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.iters * 1e3, float(last)


a_ms, a_loss = run(False)
b_ms, b_loss = run(True)
print(json.dumps({"gaussians": args.gaussians, "boxes": args.boxes, "frame": [H, W], "iters": args.iters,
                  "A_rasterizer_only_ms": round(a_ms, 3), "B_all_fused_ms": round(b_ms, 3),
                  "A_iter_per_s": round(1e3 / a_ms, 1), "B_iter_per_s": round(1e3 / b_ms, 1),
                  "loss_A": a_loss, "loss_B": b_loss}))
