#!/usr/bin/env python
"""bench.py -- headline benchmark of the rasterizer hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one view of the synthetic KITTI-360-shaped corridor rendered forward AND backward
through the drop-in operator (diff_gaussian_rasterization surface -> C ABI -> HIP kernels), with all
12 output channels produced and the three gradients VEGS' losses feed back (colour, cov_quat,
cov_scale; reference train.py:162-168).  Inputs are resident in HBM before the timed region.
For N > 1 (launched by torch.distributed.run, one rank per GPU) views are sharded one camera per
rank and every step ends with the all-reduce of the 59-float/Gaussian gradients over RCCL.

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job views/s, plus
Mfragments/s, the roofline object of the dominant kernel (HIP-event timed inside the timed region)
and the CPU baseline (the oracle -- a port, the reference has no CPU rasterizer -- timed on the host
cores on a bounded sample: one view of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from vegs_amd import _capi, dist as vdist, harness, scenes  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (guide: MI355X_MICROARCH.md chip table)


def build_workload(args):
    if args.workload == "c3":
        P, length, seed = args.gaussians or 2_000_000, 250.0, 2
    else:  # c2
        P, length, seed = args.gaussians or 500_000, 120.0, 1
    sc, deg = scenes.scene_street(P=P, length=length, sh_degree=3, seed=seed)
    cams = []
    n_stations = 8
    for s in range(n_stations):               # stations every 10 m, stereo pair (baseline 0.6 m)
        for y in (0.3, -0.3):
            cams.append(scenes.kitti_camera(10.0 * s, y, args.width, args.height))
    return sc, deg, cams, P


def upstream_grads(pkg, cam, rng, device):
    """dL/dout shaped like VEGS' losses: L1 against a fixed random target on colour, and the
    normal-guidance loss (reference loss/normal_guidance.py:3-22) on cov_quat / cov_scale."""
    H, W = cam.image_height, cam.image_width
    target = torch.tensor(rng.uniform(0, 1, (3, H, W)).astype(np.float32), device=device)
    normal = torch.tensor(rng.normal(size=(3, H, W)).astype(np.float32), device=device)
    q = pkg["render_cov_quat"].detach().clone().requires_grad_(True)
    s = pkg["render_cov_scale"].detach().clone().requires_grad_(True)
    qs = torch.where((q * q).sum(0, keepdim=True) > 0, q, torch.ones_like(q))
    r, i, j, k = torch.unbind(qs.permute(1, 2, 0).reshape(-1, 4), -1)
    two_s = 2.0 / (r * r + i * i + j * j + k * k)
    rot = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                       two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                       two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1).reshape(-1, 3, 3)
    cs = s.permute(1, 2, 0).reshape(-1, 1, 3)
    Rm = torch.tensor(cam.R, dtype=torch.float32, device=device)
    nw = (Rm @ normal.reshape(3, -1)).t()[:, :, None].expand(-1, 3, 3)
    loss = 1e-3 * (0.8 * (rot * nw).sum(dim=-2).abs().mean() + 0.2 * ((rot.detach() * cs) * nw).sum(dim=-2).abs().mean())
    gq, gs = torch.autograd.grad(loss, [q, s])
    gc = torch.sign(pkg["render"].detach() - target) / target.numel()
    return gc.contiguous(), gq.contiguous(), gs.contiguous()


def cpu_baseline(sc, deg, cams, gouts, views):
    """Oracle (oracle/vr_oracle.c, OpenMP over the host cores) fwd+bwd on a few views of the workload."""
    from oracle import oracle as orc
    orc.build()
    dt, frags = 0.0, 0
    for v in views:
        cam = cams[v]
        oc = orc.make_cam(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, [0, 0, 0], 1.0,
                          cam.world_view_transform, cam.full_proj_transform, cam.camera_center, deg, 16)
        g = [x.cpu().numpy() for x in gouts[v]]
        t0 = time.perf_counter()
        out, st = orc.forward(oc, sc["means3D"], sc["shs"], None, sc["opacities"], sc["scales"], sc["rotations"], None)
        orc.backward(oc, st, g[0], None, g[1], g[2], None)
        dt += time.perf_counter() - t0
        frags += int(st["n_contrib"].sum())
    return dt, frags, os.cpu_count()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", choices=["c3", "c2"], default="c3")
    ap.add_argument("--gaussians", type=int, default=0)
    ap.add_argument("--width", type=int, default=1376)
    ap.add_argument("--height", type=int, default=376)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stages", action="store_true", help="also print a per-stage time breakdown to stderr")
    ap.add_argument("--views-per-step", type=int, default=1,
                    help="views each rank renders per step; their gradients accumulate locally and are exchanged once")
    args = ap.parse_args()

    rank, world, local = vdist.init_from_env()
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the rasterizer has no CPU path)")
    device = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(device)
    _capi.load()

    sc, deg, cams, P = build_workload(args)
    T = {k: torch.tensor(v, device=device, requires_grad=True) for k, v in sc.items()}
    params = [T["means3D"], T["shs"], T["opacities"], T["scales"], T["rotations"]]
    bg = torch.zeros(3, device=device)
    cam_ts = [harness.cam_tensors(c, device) for c in cams]
    n_views = len(cams)
    H, W = args.height, args.width
    N = H * W
    K = (deg + 1) ** 2

    # ---- untimed setup pass over every view: upstream gradients + work counters (P_z, V, R, F)
    rng = np.random.default_rng(1234)
    gouts, counters = [], []
    from vegs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    for v, cam in enumerate(cams):
        pkg = harness.render(cam, T, deg, bg, cam_t=cam_ts[v])
        gouts.append(upstream_grads(pkg, cam, rng, device))
        c = _capi.counters()
        c["F"] = _capi.count_fragments(pkg["render"].grad_fn, H, W, device)
        rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, bg, 1.0, cam_ts[v]["viewmatrix"],
                                           cam_ts[v]["projmatrix"], deg, cam_ts[v]["campos"], False, False)
        c["Pz"] = int(GaussianRasterizer(rs).markVisible(T["means3D"]).sum().item())
        counters.append(c)
        del pkg
    torch.cuda.synchronize()

    vps = max(1, args.views_per_step)

    def step(i):
        done = []
        for k in range(vps):
            v = vdist.view_for_rank(i * vps + k, rank, world, n_views)
            pkg = harness.render(cams[v], T, deg, bg, cam_t=cam_ts[v])
            gc, gq, gs = gouts[v]
            torch.autograd.backward([pkg["render"], pkg["render_cov_quat"], pkg["render_cov_scale"]], [gc, gq, gs])
            done.append(v)
        if world > 1:
            vdist.allreduce_grads(params, world)
        for p in params:
            p.grad = None
        return done

    for i in range(args.warmup):
        step(i)
    _capi.profile_level(2 if args.stages else 1)
    _capi.profile_collect()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    mallocs0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
    reserved0 = torch.cuda.memory_reserved(device)
    t0 = time.perf_counter()
    views_done = []
    for i in range(args.warmup, args.warmup + args.steps):
        views_done.extend(step(i))
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    stage = _capi.profile_collect()
    _capi.profile_level(0)

    frag_local = float(sum(counters[v]["F"] for v in views_done))
    if world > 1:
        t = torch.tensor([elapsed, frag_local], dtype=torch.float64, device=device)
        tmax = t.clone()
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
        elapsed, frag_total = float(tmax[0]), float(t[1])
    else:
        frag_total = frag_local

    if rank != 0:
        return
    views = args.steps * world * vps
    value = views / elapsed
    mean = {k: float(np.mean([counters[v][k] for v in views_done])) for k in ("Pz", "V", "R", "F")}
    # algorithmic bytes per view, SURVEY.md section 8(d)
    b_alg = 32 * P + 28 * mean["Pz"] + (294 + 24 * K) * mean["V"] + 188 * mean["R"] + 112 * N + (56 + 12 * K) * P
    # dominant kernel: k_seg_bwd (gradients of one (tile, segment)), timed with HIP events recorded by the
    # library on the stream it launches on, inside the timed region.  Its algorithmic bytes are those of the
    # whole render-backward step (72R + 56N + 68V), of which it is the only heavy kernel.
    kern = "k_seg_bwd"
    ms_k = stage[kern][0] / max(stage[kern][1], 1)
    bytes_k = 72 * mean["R"] + 56 * N + 68 * mean["V"]
    achieved = bytes_k / (ms_k * 1e-3) / 1e9 if ms_k > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(kern)
        except Exception:
            traffic = None
    res = {
        "metric": "rasterizer fwd+bwd views/sec + Mfragments/sec, 2M Gaussians @1376x376",
        "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "mfragments_per_s": round(frag_total / elapsed / 1e6, 2),
        "config": {"workload": f"{args.workload}: {P} street Gaussians (VEGS disc init), SH deg {deg}, {W}x{H} "
                               f"KITTI-360 intrinsics, {n_views} views cycled, 12 output channels + colour/quat/scale grads",
                   "gaussians": P, "width": W, "height": H, "views_per_step_per_gpu": vps,
                   "parallelism": f"view-sharded x{world}" + (" + RCCL grad all-reduce (59 f32/Gaussian)" if world > 1 else ""),
                   "mean_counters": {k: round(v, 1) for k, v in mean.items()}},
        "roofline": {"bound": "hbm", "kernel": kern, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "alg_bytes_per_launch": round(bytes_k), "avg_launch_ms": round(ms_k, 4),
                     "stage_ms": {k: round(stage[k][0] / max(stage[k][1], 1), 4) for k in ("render_fwd", "render_bwd")
                                  if stage[k][1] > 0},
                     "whole_view_alg_bytes": round(b_alg),
                     "whole_view_frac": round(b_alg / (elapsed / (args.steps * vps)) / 1e9 / HBM_PEAK_GBS, 5)},
    }
    if args.stages:
        print("hipMalloc calls inside the timed region:",
              torch.cuda.memory_stats(device).get("num_device_alloc", 0) - mallocs0, "reserved MB before/after:",
              reserved0 >> 20, torch.cuda.memory_reserved(device) >> 20, file=sys.stderr)
        print("stage breakdown (ms per step; k_seg_bwd is part of render_bwd):",
              {k: round(v[0] / args.steps, 4) for k, v in stage.items()},
              file=sys.stderr)
    if world == 1 and not args.no_cpu_baseline:
        cpu_views = [0, 5, 10, 15]
        dt, frags, cores = cpu_baseline(sc, deg, cams, gouts, cpu_views)
        res["cpu_baseline"] = {"value": round(len(cpu_views) / dt, 5), "unit": "views/s", "cores": cores, "kind": "port",
                               "sample": f"{len(cpu_views)} views (0,5,10,15) of the same workload, oracle fwd+bwd on all "
                                         f"host cores, {dt:.1f} s",
                               "mfragments_per_s": round(frags / dt / 1e6, 2)}
    print(json.dumps(res))


def _finish():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    finally:
        _finish()
