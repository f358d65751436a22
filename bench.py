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

from vegs_amd import _capi, dist as vdist, harness, scenes, so3  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (guide: MI355X_MICROARCH.md chip table)
VALU_PEAK_GINST = 1228.8   # wave64 VALU instructions per ns: 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles (same guide)
VALU_MEASURED_GINST = 924.0   # ... what a stream of independent v_fma_f32 reaches at 8 waves per SIMD (2.66 cycles each;
                              # profiles/r05_ubench_pk_rate.txt) -- v_pk_fma_f32 takes two such slots, v_rcp / v_exp ~3.5


def _trace(msg):
    """Progress markers on stderr (VEGS_BENCH_TRACE=1): where a run was when something outside Python ended it."""
    if os.environ.get("VEGS_BENCH_TRACE") == "1":
        torch.cuda.synchronize()
        print(f"[bench trace] {msg}", file=sys.stderr, flush=True)


def profile_figures(kern, sources):
    """(HBM traffic bytes per launch, VALU wave-instructions per launch, provenance) of `kern` from the committed PMC
    profile (profiles/pmc_traffic.json) -- or (None, None, reason) when a source file of that kernel has changed since
    the counters were collected: a stale figure must not ride along silently."""
    import hashlib
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(tpath))
    except Exception as e:
        return None, None, f"unavailable ({e.__class__.__name__})"
    rec = d.get("_sources") or {}
    for f in sources:
        try:
            h = hashlib.sha256(open(os.path.join(ROOT, "vegs_amd", "csrc", f), "rb").read()).hexdigest()[:16]
        except OSError:
            h = None
        if rec.get(f) != h:
            print(f"warning: profiles/pmc_traffic.json was collected on another version of {f}; roofline.traffic and "
                  "roofline.secondary.valu are reported as null until the counters are re-collected "
                  "(profiles/tools/collect_round.sh)", file=sys.stderr)
            return None, None, f"stale: {f} changed since {d.get('_collected')}"
    return d.get(kern), (d.get("_valu_insts") or {}).get(kern), d.get("_collected")


def whole_view_traffic():
    """HBM bytes of one whole view (forward + backward) from the committed PMC profile: the sum over the view's kernels of
    counter bytes per launch x launches per view (profiles/make_traffic.py: `_whole_view`) -- or None when ANY kernel source
    has changed since the counters were collected."""
    import hashlib
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return None
    for f, want in (d.get("_sources") or {}).items():
        try:
            h = hashlib.sha256(open(os.path.join(ROOT, "vegs_amd", "csrc", f), "rb").read()).hexdigest()[:16]
        except OSError:
            h = None
        if h != want:
            return None
    return (d.get("_whole_view") or {}).get("bytes")


def build_workload(args):
    if args.workload == "c3":
        P, length, seed = args.gaussians or 2_000_000, 250.0, 2
    else:  # c2
        P, length, seed = args.gaussians or 500_000, 120.0, 1
    sc, deg = scenes.scene_street(P=P, length=length, sh_degree=3, seed=seed)
    if getattr(args, "disc_scale", 1.0) != 1.0:      # (profiling aid: the "dense" variant's scene as the main workload)
        sc["scales"] = (sc["scales"] * args.disc_scale).astype(np.float32)
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
    rot = so3.quaternion_to_matrix(qs.permute(1, 2, 0).reshape(-1, 4))
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


def prepare(sc, deg, cams, device, rng, count=True, cam_ts=None):
    """Scene tensors + per-view upstream gradients and work counters (one untimed pass over every view)."""
    T = {k: torch.tensor(v, device=device, requires_grad=True) for k, v in sc.items()}
    bg = torch.zeros(3, device=device)
    cam_ts = cam_ts if cam_ts is not None else [harness.cam_tensors(c, device) for c in cams]
    gouts, counters = [], []
    from vegs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    for v, cam in enumerate(cams):
        H, W = cam.image_height, cam.image_width
        pkg = harness.render(cam, T, deg, bg, cam_t=cam_ts[v])
        gouts.append(upstream_grads(pkg, cam, rng, device))
        c = _capi.counters()
        c["R_lists"], c["V_lists"] = c["R"], c["V"]      # entries of the build's own (tight) tile lists; Gaussians that have any
        if count:
            c["F_lists"] = _capi.count_fragments(pkg["render"].grad_fn, H, W, device)
            c["B"] = _capi.count_blended(pkg["render"].grad_fn, H, W, device)
            c["A"] = _capi.count_flushes(pkg["render"].grad_fn, H, W, device)
            # R and F as BASELINE.md / SURVEY 8(d) define them: on the REFERENCE's tile rectangles (every tile of the
            # rectangle an entry, every entry a pixel walks a fragment) -- one more, untimed render with
            # VR_FLAG_FULL_TILE_LISTS; the build's default lists leave out the pairs that cannot reach any pixel
            from vegs_amd import rasterizer as _r
            with _r.flags(_r.get_flags() | _r.FLAG_FULL_TILE_LISTS):
                ref = harness.render(cam, T, deg, bg, cam_t=cam_ts[v])
                c["R"], c["V"] = _capi.counters()["R"], _capi.counters()["V"]
                c["F"] = _capi.count_fragments(ref["render"].grad_fn, H, W, device)
            del ref
            rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, bg, 1.0, cam_ts[v]["viewmatrix"],
                                               cam_ts[v]["projmatrix"], deg, cam_ts[v]["campos"], False, False)
            c["Pz"] = int(GaussianRasterizer(rs).markVisible(T["means3D"]).sum().item())
        counters.append(c)
        del pkg
    torch.cuda.synchronize()
    return dict(T=T, params=[T[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")], bg=bg, cams=cams,
                cam_ts=cam_ts, gouts=gouts, counters=counters, deg=deg)


def make_step(wl, rank, world, vps, factored=False, exchange="dense", mode="train", exchange_on=True, streams=1,
              direct=None, keep_grads=False):
    """One step = `vps` views forward + backward on this rank, then (N > 1) the gradient exchange.
    exchange "dense": all-reduce of the 59-float/Gaussian gradients.  "factored": the op returns the 3-float factor of
    the SH gradient, the ranks all-gather the factors (12 B) and all-reduce the other 11 floats (44 B), and every rank
    rebuilds the averaged dense SH gradient locally (inside the timed region) -- the same tensors on every rank at the
    end of the step as with "dense", for 256 instead of 826 MB over the links at N = 8.
    mode "train": through harness.render, the counterpart of the reference's render() glue (a fresh zeros
    screenspace_points + 0 with retain_grad per view, gaussian_renderer/__init__.py:27-32); "noglue": the same op call
    with one persistent means2D leaf (what the ATen glue costs is the difference); "forward": forward only under
    torch.no_grad(), as the reference's evaluation and video paths call it (train.py:338-508, render_video.py:162,202)."""
    T, cams, cam_ts, gouts, params, deg, bg = (wl[k] for k in ("T", "cams", "cam_ts", "gouts", "params", "deg", "bg"))
    n_views = len(cams)
    fact_x = world > 1 and exchange in ("factored", "direct") and vps == 1
    # several views per rank and step at N > 1 (round 6): what iteration.Trainer does -- every view's SH gradient stays its
    # 3-float factor, the other 11 floats are ACCUMULATED IN PLACE over the rank's views, and the step ends with ONE all-gather
    # of the k factors per rank + ONE all-reduce of the 11 floats, instead of an all-reduce of 59 dense floats (DESIGN section 8:
    # the configuration priced at ~78 % efficiency).  (`direct` has no multi-view gather: RCCL carries it.)
    fact_multi = world > 1 and exchange in ("factored", "direct") and vps > 1
    factored = factored or fact_x or fact_multi
    others = [T[k] for k in ("means3D", "opacities", "scales", "rotations")]
    from vegs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    m2d = torch.zeros_like(T["means3D"], requires_grad=True) if mode == "noglue" else None
    xch = None
    if fact_x and exchange_on:
        if exchange == "direct":           # hand-written peer-to-peer exchange over hipIpc mappings (vegs_amd/xgmi.py)
            from vegs_amd import xgmi
            xch = direct
            if xch is None:
                xch = xgmi.DirectExchange(rank, world, T["means3D"].device)
                P_ = T["means3D"].shape[0]
                xch.reserve(11 * P_ + 64, 3 * P_ + 64)    # (collective allocation: outside the timed region)
        else:
            xch = vdist.FactorExchange(world)
    if fact_x and not exchange_on:
        fact_x = False             # (measurement aid: the same step without any collective)

    def direct(v):
        cam, ct = cams[v], cam_ts[v]
        rs = GaussianRasterizationSettings(int(cam.image_height), int(cam.image_width), cam.tanfovx, cam.tanfovy, bg, 1.0,
                                           ct["viewmatrix"], ct["projmatrix"], deg, ct["campos"], False, False)
        return GaussianRasterizer(rs)(means3D=T["means3D"], means2D=m2d if m2d is not None else T["means3D"],
                                      shs=T["shs"], opacities=T["opacities"], scales=T["scales"], rotations=T["rotations"])

    # streams > 1: the views of a step alternate between side streams (vegs_amd/views.py: a view's forward, loss
    # gradient and backward stay on ONE stream, in order); two views in flight fill each other's latency-bound stretches.
    # The views of a step are independent by construction (same parameters, gradients summed): no result changes.
    from vegs_amd import views as vviews
    vs = vviews.ViewStreams(T["means3D"].device, streams)

    def one_view(v):
        if mode == "forward":
            with torch.no_grad():
                direct(v)
            return
        gc, gq, gs = gouts[v]
        if mode == "noglue":
            out = direct(v)
            torch.autograd.backward([out[0], out[2], out[3]], [gc, gq, gs])
            m2d.grad = None
            return
        sink = torch.zeros_like(T["means3D"], requires_grad=True) if factored else None
        if factored and not fact_x and vps > 1:
            sinks.append((sink, cam_ts[v]["campos"]))        # (fact_multi included: its factors are gathered at the step's end)
        pkg = harness.render(cams[v], T, deg, bg, cam_t=cam_ts[v], sh_color_grad=sink)
        if fact_x:          # overlapped exchange: the factors start travelling between the backward's two halves
            with xch.armed(cam_ts[v]["campos"]):
                torch.autograd.backward([pkg["render"], pkg["render_cov_quat"], pkg["render_cov_scale"]], [gc, gq, gs])
        else:
            torch.autograd.backward([pkg["render"], pkg["render_cov_quat"], pkg["render_cov_scale"]], [gc, gq, gs])

    sinks = []

    def step(i):
        done = []
        sinks.clear()
        from vegs_amd import rasterizer as _rast
        old_acc = _rast.accumulate_grads(True) if fact_multi else None
        try:
            for k in range(vps):
                v = vdist.view_for_rank(i * vps + k, rank, world, n_views)
                vs.run(one_view, v)
                done.append(v)
            vs.join()
        finally:
            if fact_multi:
                _rast.accumulate_grads(old_acc)
        if mode == "forward":
            return done
        if fact_x:
            from vegs_amd import optim
            F, Cc = xch.finish(others)
            T["shs"].grad = optim.sh_grad_from_factors(T["means3D"].detach(), Cc, F, deg, T["shs"].shape[1], 1.0 / world)
        elif fact_multi:
            from vegs_amd import optim
            F = torch.stack([s_.grad for s_, _ in sinks])
            Cc = torch.stack([c.reshape(3) for _, c in sinks]).to(F.device, torch.float32)
            if exchange_on:
                F, Cc = vdist.all_gather_views(F, Cc, world)          # [world * vps, P, 3], [world * vps, 3]
                vdist.allreduce_grads(others, world)
            T["shs"].grad = optim.sh_grad_from_factors(T["means3D"].detach(), Cc, F, deg, T["shs"].shape[1], 1.0 / world)
            sinks.clear()
        elif world > 1 and exchange_on:
            vdist.allreduce_grads(params, world)
        if sinks:
            # a multi-view step with the factored SH gradient: the dense [P,16,3] gradient of the STEP is rebuilt once from
            # the views' 3-float factors (inside the timed region: the step ends with the same tensors as the dense batch)
            from vegs_amd import optim
            F = torch.stack([s_.grad for s_, _ in sinks])
            Cc = torch.stack([c.reshape(3) for _, c in sinks]).to(F.device, torch.float32)
            T["shs"].grad = optim.sh_grad_from_factors(T["means3D"].detach(), Cc, F, deg, T["shs"].shape[1], 1.0)
        if not keep_grads:          # (keep_grads: tests compare the step's final gradients across exchange schemes)
            for p in params:
                p.grad = None
        return done
    return step


def timed(step, warmup, steps, world, first=None):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides.  Returns (seconds, views)."""
    first = warmup if first is None else first
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    views_done = []
    for i in range(first, first + steps):
        views_done.extend(step(i))
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, views_done


def _max_over_ranks(x, world):
    if world <= 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=torch.device("cuda") if torch.distributed.get_backend() == "nccl" else "cpu")
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t[0])


def _agree_min(x, world):
    """min over ranks of an integer (the parents' own group: RCCL in production, gloo in the single-GPU tests)."""
    if world <= 1:
        return int(x)
    t = torch.tensor([int(x)], dtype=torch.int64, device=torch.device("cuda") if torch.distributed.get_backend() == "nccl" else "cpu")
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
    return int(t[0])


def _share_from_rank0(x, rank, world):
    if world <= 1:
        return int(x)
    t = torch.tensor([int(x) if rank == 0 else 0], dtype=torch.int64,
                     device=torch.device("cuda") if torch.distributed.get_backend() == "nccl" else "cpu")
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return int(t[0])


def direct_in_children(args, rank, world, local):
    """The direct hipIpc exchange has never run across devices (no multi-GPU node was available to the build): the job
    that prints the bench line must not be the first to try.  Every rank starts a SACRIFICIAL CHILD -- this same script
    with --exchange direct, the children forming their own process group over gloo (handles and barriers only; the data
    moves through the hipIpc windows) on the parents' devices -- which sets the windows up, verifies an exchange and times
    the headline regions under the same contract.  The parents wait, GPUs idle, with a hard wall-clock limit and kill what
    is left (their own children's process groups, by pid); they never map a window themselves.  A child that faults, hangs
    or fails its set-up costs nothing but the direct measurement.  Returns the record for `exchange.auto.direct`:
    {"ok": all children exited 0, "child_rc": this rank's, "line": the children's bench line (rank 0 only) ...}."""
    import signal
    import socket
    import subprocess
    port = 0
    if rank == 0:
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
    port = _share_from_rank0(port, rank, world)
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VEGS_DIST_BACKEND="gloo", VEGS_BENCH_CHILD="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "GROUP_RANK", "ROLE_RANK",
              "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--repeats", str(args.repeats), "--exchange", "direct", "--workload", args.workload,
           "--gaussians", str(args.gaussians), "--width", str(args.width), "--height", str(args.height),
           "--views-per-step", str(args.views_per_step), "--no-variants", "--no-cpu-baseline"]
    t0 = time.perf_counter()
    rec = {"ran_in": "sacrificial child processes (one per rank, own gloo rendezvous); the parents never map a peer window",
           "timeout_s": args.probe_timeout}
    try:
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    except OSError as e:
        proc = None
        rec["spawn_error"] = str(e)[:200]
    out, err, rc = "", "", -999
    if proc is not None:
        try:
            out, err = proc.communicate(timeout=args.probe_timeout)
            rc = proc.returncode
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)        # (the session this parent started for its own child)
            except OSError:
                proc.kill()
            out, err = proc.communicate()
            rc = -998
            rec["timed_out"] = True
    rec["child_rc"] = rc                 # (-998: killed at the time limit, -999: could not be started)
    if -900 < rc < 0:
        rec["child_signal"] = -rc        # ended by a signal: 11 = SIGSEGV, 6 = SIGABRT (a GPU memory fault aborts the process)
    ok = _agree_min(1 if rc == 0 else 0, world) == 1
    rec["ok"] = ok
    rec["elapsed_s"] = round(time.perf_counter() - t0, 1)
    if rank == 0:
        line = None
        for ln in out.splitlines():
            if ln.startswith("{"):
                try:
                    line = json.loads(ln)
                except ValueError:
                    pass
        if ok and line is not None:
            rec["line"] = {k: line.get(k) for k in ("value", "ms_per_step", "ms_per_step_regions", "steps", "warmup", "repeats")}
            rec["line"]["exchange"] = line.get("exchange")
        elif ok:
            rec["ok"] = False
            rec["note"] = "children exited 0 but printed no line"
        if not rec["ok"]:
            rec["stderr_tail"] = (err or "")[-600:]
    return rec


def timed_median(step, steps, world, repeats, first=0):
    """`repeats` timed regions of exactly K steps each (same cameras every time), each bracketed as in timed(); the
    MEDIAN region is the one reported (K steps are only ~40 ms: a single region is at the mercy of one scheduling hiccup).
    Returns (median seconds, views of one region, [seconds of every region])."""
    runs = []
    views = None
    for _ in range(max(1, repeats)):
        dt, done = timed(step, 0, steps, world, first=first)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda") if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t[0])
        runs.append(dt)
        views = done
    return float(np.median(runs)), views, runs


def stage_profile(step, steps):
    """Per-stage milliseconds per view (HIP events around every stage: level 2), measured OUTSIDE the timed region
    because every event pair costs a few microseconds of bubble."""
    _capi.profile_level(2)
    _capi.profile_collect()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    st = _capi.profile_collect()
    _capi.profile_level(0)
    return {k: round(v[0] / max(v[1], 1), 4) for k, v in st.items() if v[1] > 0}


def warm_hints(step, n_views):
    """Two passes over every camera: a key's hint array exists from its second sighting on (vegs_amd/rasterizer.py)."""
    for i in range(2 * n_views):
        step(i)


def variant(name, sc, deg, cams, device, steps, warmup, factored=False, hints="off", mode="train", repeats=3, vps=1,
            streams=1, flags=0, accumulate=False):
    """A few steps of another scene / camera / operator configuration, reported next to the headline (N = 1 only).
    hints: "off" = per-camera needed-segment hints disabled, as in the headline; "warm" (forward-only modes: the cache
    serves forwards under no_grad only, vegs_amd/rasterizer.py) = every camera was rendered before with the SAME model --
    evaluation of fixed cameras.  mode: see make_step."""
    from vegs_amd import rasterizer
    old = rasterizer.needed_hints(hints != "off")
    old_flags = rasterizer.set_flags(rasterizer.get_flags() | flags)
    old_acc = rasterizer.accumulate_grads(accumulate)
    extra = {}
    try:
        wl = prepare(sc, deg, cams, device, np.random.default_rng(77))
        step = make_step(wl, 0, 1, vps, factored, mode=mode, streams=streams)
        if hints == "warm":
            warm_hints(step, len(cams))
        for i in range(warmup):
            step(i)
        dt, done, runs = timed_median(step, steps, 1, repeats)
    finally:
        rasterizer.needed_hints(old)
        rasterizer.set_flags(old_flags)
        rasterizer.accumulate_grads(old_acc)
    cn = wl["counters"]
    mean = {k: float(np.mean([cn[v][k] for v in done])) for k in ("V", "R", "R_lists", "F", "F_lists", "B")}
    nv = len(done)                      # steps x views per step
    res = {"workload": name, "views_per_s": round(nv / dt, 2), "ms_per_view": round(dt / nv * 1e3, 4),
           "ms_per_view_runs": [round(r / nv * 1e3, 4) for r in runs],
           "mfragments_per_s": round(sum(cn[v]["F_lists"] for v in done) / dt / 1e6, 1),
           "mfragments_per_s_reference_lists": round(sum(cn[v]["F"] for v in done) / dt / 1e6, 1),
           "blended_mfragments_per_s": round(sum(cn[v]["B"] for v in done) / dt / 1e6, 1),
           "mean_counters": {k: round(v, 1) for k, v in mean.items()}}
    res.update(extra)
    return res


def cpu_plan(cores):
    """the CPU restatement on C1 (10 k random Gaussians, 256x256, SH 0) and on a down-scaled C2
    (50 k street Gaussians, 1376x376, SH 3), forward + backward, median of 5 after one warm-up."""
    from oracle import oracle as orc
    out = {}
    sc1, d1 = scenes.scene_random(P=10000, sh_degree=0, seed=0)
    sc2, d2 = scenes.scene_street(P=50000, length=120.0, sh_degree=3, seed=1)
    for tag, sc, deg, cam in (("c1_10k_256x256_sh0", sc1, d1, scenes.camera_c1(256, 256)),
                              ("c2_50k_1376x376_sh3", sc2, d2, scenes.kitti_camera(0.0, 0.3, 1376, 376))):
        H, W = cam.image_height, cam.image_width
        oc = orc.make_cam(H, W, cam.tanfovx, cam.tanfovy, [0, 0, 0], 1.0, cam.world_view_transform,
                          cam.full_proj_transform, cam.camera_center, deg, 16)
        rng = np.random.default_rng(3)
        g = [rng.normal(size=(k, H, W)).astype(np.float32) for k in (3, 4, 3)]
        ts, frags = [], 0
        for it in range(6):
            t0 = time.perf_counter()
            o, st = orc.forward(oc, sc["means3D"], sc["shs"], None, sc["opacities"], sc["scales"], sc["rotations"], None)
            orc.backward(oc, st, g[0], None, g[1], g[2], None)
            if it:
                ts.append(time.perf_counter() - t0)
            frags = int(st["n_contrib"].sum())
        med = float(np.median(ts))
        out[tag] = {"views_per_s": round(1.0 / med, 3), "mfragments_per_s": round(frags / med / 1e6, 2), "runs": 5}
    return out


def bench_c5(args, rank, world, device):
    """BASELINE config C5: 5 M static Gaussians + 8 dynamic box instances x 8,196 Gaussians, the FULL training step --
    render_all-shaped forward (instance transform + concatenation), L1 + SSIM + normal guidance as one node, backward,
    gradient exchange (N > 1), densification statistics, Adam over the static model, every instance model and every
    BoxModel, BoxModel.regularize (reference train.py:143-168,196,254-320) -- one view per rank per iteration
    (vegs_amd.iteration.Trainer).  Prints the bench line of that workload."""
    from vegs_amd import iteration
    P = args.gaussians or 5_000_000
    sc, deg = scenes.scene_street(P=P, length=250.0, sh_degree=3, seed=2)
    cams = [scenes.kitti_camera(10.0 * s_, y, args.width, args.height) for s_ in range(8) for y in (0.3, -0.3)]
    cam_ts = [harness.cam_tensors(c, device) for c in cams]
    rng = np.random.default_rng(99)
    H, W = args.height, args.width
    gts = [torch.tensor(rng.uniform(0, 1, (3, H, W)).astype(np.float32), device=device) for _ in range(4)]
    normals = [torch.tensor(rng.normal(size=(3, H, W)).astype(np.float32), device=device) for _ in range(4)]
    bg = torch.zeros(3, device=device)
    tr = iteration.Trainer(sc, device, n_boxes=args.boxes, fused=True, factored_sh=True, lrs=iteration.REFERENCE_LRS,
                           optimise_boxes=True, world=world, rank=rank,
                           exchange="factored" if args.exchange == "auto" else args.exchange)

    def step(i):
        v = vdist.view_for_rank(i, rank, world, len(cams))
        tr.step(cams[v], cam_ts[v], deg, bg, gts[i % 4], normals[i % 4])
        return [v]
    for i in range(args.warmup):
        step(i)
    tr.sh_adam_events = []
    elapsed, views_done, region_s = timed_median(step, args.steps, world, args.repeats, first=args.warmup)
    events, tr.sh_adam_events = tr.sh_adam_events, None
    torch.cuda.synchronize()
    if rank != 0:
        return
    ms_k = float(np.median([a.elapsed_time(b) for a, b in events])) if events else 0.0
    rows = tr._rows()
    # dominant kernel: the SH Adam straight from the factors (k_sh_factors<true>): reads and writes param / exp_avg /
    # exp_avg_sq of the 48 SH floats per Gaussian (24 B per float) + the factors of the N views and the means
    bytes_k = 24.0 * 48 * P + 12.0 * P * (world + 1)
    achieved = bytes_k / (ms_k * 1e-3) / 1e9 if ms_k > 0 else 0.0
    res = {"metric": "rasterizer fwd+bwd views/sec + Mfragments/sec, 2M Gaussians @1376x376",
           "value": round(args.steps * world / elapsed, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "repeats": len(region_s),
           "ms_per_step_regions": [round(r / args.steps * 1e3, 4) for r in region_s],
           "config": {"workload": f"c5 (NOT the headline workload): {P} street Gaussians + {args.boxes} box instances x 8196 Gaussians "
                                  f"with learnable BoxModel poses, SH deg {deg}, {W}x{H}; one view per GPU per iteration; FULL training "
                                  "step: render_all-shaped forward, L1+SSIM + normal guidance, backward, exchange, densification "
                                  "statistics, Adam over static + instance models + BoxModels (one launch), BoxModel.regularize",
                      "gaussians": P, "rows_rendered": rows, "boxes": args.boxes, "width": W, "height": H,
                      "parallelism": f"view-sharded x{world}" + ("" if world == 1 else f" + {tr.exchange} exchange")},
           "roofline": {"bound": "hbm", "kernel": "k_sh_factors<true> (Adam of the 48 SH floats per Gaussian straight from the "
                                                  "views' 3-float factors)", "achieved": round(achieved, 2),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                        "alg_bytes_per_launch": round(bytes_k), "avg_launch_ms": round(ms_k, 4),
                        "alg_bytes_note": "(24 x 48 + 12 (views + 1)) bytes per Gaussian: parameter and both moments of the 48 SH "
                                          "floats read and written, the views' factors and the means read",
                        "launches_timed": len(events)},
           "cpu_baseline": None}
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", choices=["c3", "c2", "c5"], default="c3",
                    help="c3 = the headline (BASELINE.json); c2 = 500 k Gaussians; c5 = 5 M + box instances, the FULL training step")
    ap.add_argument("--boxes", type=int, default=8, help="c5: dynamic box instances in frame")
    ap.add_argument("--gaussians", type=int, default=0)
    ap.add_argument("--width", type=int, default=1376)
    ap.add_argument("--height", type=int, default=376)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra 1408x376 / dense-scene measurements")
    ap.add_argument("--stages", action="store_true", help="also print a per-stage time breakdown to stderr")
    ap.add_argument("--exchange", choices=["auto", "factored", "dense", "direct"], default="auto",
                    help="N > 1: gradient exchange scheme (auto = the headline is timed with `factored`; `direct` is then set up, "
                         "verified and timed by sacrificial child processes and reported beside it; factored = RCCL all-gather of the rank-1 SH factors + all-reduce of "
                         "the other 11 floats; dense = RCCL all-reduce of all 59 floats per Gaussian; direct = the factored "
                         "scheme over hand-written peer-to-peer kernels: every rank pushes 1/N shards into all peers' hipIpc "
                         "windows at once, vegs_amd/csrc/xgmi.hip)")
    ap.add_argument("--probe-timeout", type=float, default=120.0,
                    help="N > 1, --exchange auto: wall-clock limit in seconds for the child processes that set up, verify and "
                         "time the direct exchange; what is left of them is killed")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median is reported "
                    "(the driver fixes --steps 20 = 26 ms per region: five regions make the median robust)")
    ap.add_argument("--disc-scale", type=float, default=1.0,
                    help="multiply every Gaussian's scales (3.0 = the 'dense' variant's scene); a profiling aid, changes the workload")
    ap.add_argument("--streams", type=int, default=1,
                    help="with --views-per-step > 1: HIP streams the views of a step alternate between (2 = two views in flight)")
    ap.add_argument("--accumulate", action="store_true",
                    help="with --views-per-step > 1: accumulate the views' gradients IN PLACE (rasterizer.accumulate_grads; "
                         "not the headline configuration)")
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
    if args.workload == "c5":
        return bench_c5(args, rank, world, device)

    sc, deg, cams, P = build_workload(args)
    H, W = args.height, args.width
    N = H * W
    K = (deg + 1) ** 2
    n_views = len(cams)
    from vegs_amd import rasterizer
    rasterizer.needed_hints(False)
    rasterizer.accumulate_grads(bool(args.accumulate))
    _trace("workload built")
    wl = prepare(sc, deg, cams, device, np.random.default_rng(1234))
    _trace("prepared")
    counters, gouts = wl["counters"], wl["gouts"]
    vps = max(1, args.views_per_step)
    # --exchange auto (the default): THIS process times the headline regions with the RCCL exchange (factored, overlapped)
    # and keeps that number; the direct hipIpc exchange is then set up, verified and timed by sacrificial child processes
    # (direct_in_children, after the parents' own measurements) and reported beside it.  Nothing the untested peer-to-peer
    # path does -- a fault, a hang, a failed mapping -- can cost the record.
    auto_rec, direct_x = None, None
    auto = args.exchange == "auto"
    if auto:
        args.exchange = "factored"
    child = os.environ.get("VEGS_BENCH_CHILD") == "1"
    ranks_seen = None
    if world > 1:
        # who is in the job, as the communicator delivers it: every rank's device, and a sum of ones over the ranks
        mine = {"rank": rank, "local_rank": local, "device": device.index, "name": torch.cuda.get_device_name(device),
                "pid": os.getpid()}
        try:
            mine["pci_bus_id"] = torch.cuda.get_device_properties(device).pci_bus_id
        except Exception:
            pass
        everyone = [None] * world
        torch.distributed.all_gather_object(everyone, mine)
        ones = torch.ones(1, device=device if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(ones)
        ranks_seen = {"backend": torch.distributed.get_backend() + (" (= RCCL on ROCm)" if torch.distributed.get_backend() == "nccl" else ""),
                      "ranks_seen": int(ones.item()), "ranks": everyone}
    step = make_step(wl, rank, world, vps, exchange=args.exchange, streams=args.streams, direct=direct_x)
    if child and world > 1 and args.exchange == "direct":
        # test hook of the parents' safety net: the child dies / hangs AFTER its windows are mapped (tests/test_gpu_dist.py)
        fault = os.environ.get("VEGS_XGMI_PROBE_FAULT", "")
        if fault and rank == world - 1:
            step(0)
            torch.cuda.synchronize()
            if fault == "segv":
                import signal
                os.kill(os.getpid(), signal.SIGSEGV)
            elif fault == "hang":
                time.sleep(1e6)

    for i in range(args.warmup):
        step(i)
    _trace("warm")
    _capi.profile_level(1)          # level 1: HIP events around the roofline kernel only (one launch in four), inside the timed region
    _capi.profile_collect()
    mallocs0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
    reserved0 = torch.cuda.memory_reserved(device)
    # `repeats` regions of exactly K steps (barrier + synchronize on both sides of each, max over ranks per region);
    # the median region is the one reported
    elapsed, views_done, region_s = timed_median(step, args.steps, world, args.repeats, first=args.warmup)
    stage = _capi.profile_collect()
    _capi.profile_level(0)
    _trace("timed")
    mallocs1 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)

    frag_local = float(sum(counters[v]["F"] for v in views_done))
    frag_own_local = float(sum(counters[v]["F_lists"] for v in views_done))
    blend_local = float(sum(counters[v]["B"] for v in views_done))
    if world > 1:
        # host tensors with gloo (the single-GPU test transport), device tensors with RCCL
        red_dev = device if torch.distributed.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([frag_local, blend_local, frag_own_local], dtype=torch.float64, device=red_dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
        frag_total, blend_total, frag_own_total = float(t[0]), float(t[1]), float(t[2])
    else:
        frag_total, blend_total, frag_own_total = frag_local, blend_local, frag_own_local

    exchange = None
    if world > 1:
        # what the exchange costs: the same K steps once more WITHOUT any collective (every rank; outside the headline's
        # timed regions).  exposed = ms per step with - without: the part of the exchange that compute does not hide.
        fact = args.exchange in ("factored", "direct") and vps == 1
        fact_m = args.exchange in ("factored", "direct") and vps > 1
        step0 = make_step(wl, rank, world, vps, factored=fact, exchange=args.exchange, exchange_on=False, direct=direct_x)
        dt0, _, _ = timed_median(step0, args.steps, world, 1, first=args.warmup)
        scheme = args.exchange if fact else ("factored" if fact_m else "dense")
        exchange = {"scheme": scheme + (" (all-gather of the SH factors started between the backward's two halves, "
                                        "all-reduce of the other 11 floats after it; one wait)" if fact else "")
                              + (f" ({vps} views per rank: their SH factors gathered and the in-place accumulated 11 floats "
                                 "all-reduced once, at the step's end)" if fact_m else "")
                              + (" -- peer-to-peer over hipIpc windows, no RCCL" if scheme == "direct" else ""),
                    "exchange_bytes_per_rank": vdist.exchange_bytes_per_rank(P, world, "factored" if (fact or fact_m) else "dense")
                                               + ((vps - 1) * (world - 1) * 12 * P if fact_m else 0),
                    "expected": vdist.exchange_model(P, world, "dense" if scheme == "dense" else "factored", vps),
                    "ms_per_step_without_exchange": round(dt0 / args.steps * 1e3, 4),
                    "exchange_exposed_ms": round((elapsed - dt0) / args.steps * 1e3, 4),
                    "communicator": ranks_seen}
        if auto and fact:
            # the headline above IS the RCCL measurement; now -- everything of this process measured and kept -- the direct
            # exchange, in children
            del step0
            torch.cuda.synchronize()
            rec = direct_in_children(args, rank, world, local)
            mine_ms = elapsed / args.steps * 1e3
            exchange["auto"] = {"headline": "factored (RCCL), timed by this process before anything touched the direct path",
                                "factored_ms_per_step": round(mine_ms, 4), "direct": rec}
            if rec.get("ok") and rec.get("line"):
                d_ms = rec["line"]["ms_per_step"]
                exchange["auto"]["direct_ms_per_step"] = d_ms
                exchange["auto"]["direct_views_per_s"] = rec["line"]["value"]
                exchange["auto"]["direct_exchange_exposed_ms"] = (rec["line"].get("exchange") or {}).get("exchange_exposed_ms")
                exchange["auto"]["faster"] = "direct" if d_ms < mine_ms else "factored"
    if rank != 0:
        return
    stage_ms = stage_profile(step, 8) if world == 1 else {}
    _trace("stages profiled")
    views = args.steps * world * vps
    value = views / elapsed
    mean = {k: float(np.mean([counters[v][k] for v in views_done])) for k in ("Pz", "V", "R", "R_lists", "F", "F_lists", "B")}
    # algorithmic bytes per view, SURVEY.md section 8(d)
    # (R term on the lists the kernels walk; `..._ref`: on the reference's full rectangles)
    b_alg = 32 * P + 28 * mean["Pz"] + (294 + 24 * K) * mean["V"] + 188 * mean["R_lists"] + 112 * N + (56 + 12 * K) * P
    b_alg_ref = b_alg + 188 * (mean["R"] - mean["R_lists"])
    # dominant kernel: k_seg_bwd (gradients of one (tile, segment)), timed with HIP events recorded by the
    # library on the stream it launches on, inside the timed region.  Its algorithmic bytes are those of the
    # whole render-backward step (72R + 56N + 68V), of which it is the only heavy kernel --
    # priced on the units the launch actually PROCESSES: R_lists, the entries of the build's own tile lists.  The
    # reference's rectangles hold a third more entries (R: one per tile of the rectangle), which this
    # kernel never touches; that figure is kept as `frac_on_reference_lists` for comparison with earlier rounds.
    kern = "k_seg_bwd"
    ms_k = stage[kern][0] / max(stage[kern][1], 1)
    bytes_k = 72 * mean["R_lists"] + 56 * N + 68 * mean["V"]
    bytes_k_ref = 72 * mean["R"] + 56 * N + 68 * mean["V"]
    achieved = bytes_k / (ms_k * 1e-3) / 1e9 if ms_k > 0 else 0.0
    # (the kernel's own sources AND everything upstream that shapes its work: lists, masks, records)
    traffic, valu_insts, traffic_from = profile_figures(kern, ("render_bwd.hip", "vr_segment.h", "vr_device.h", "render_fwd.hip",
                                                               "binning.hip", "preprocess.hip"))
    # secondary ceilings (reported, not graded): VALU issue -- wave-level VALU instructions of the kernel (SQ
    # counters of the committed profile) against 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction -- and the L2
    # atomics of the backward: 17 fp32 atomics per (list entry, 8x8 region) flush, counted on the device per view
    flushes = float(np.mean([counters[v]["A"] for v in views_done])) if "A" in counters[views_done[0]] else None
    secondary = {"valu": None if not valu_insts or ms_k <= 0 else {
                     "kernel": kern, "wave_instructions_per_launch": valu_insts,
                     "achieved_ginst_per_s": round(valu_insts / (ms_k * 1e-3) / 1e9, 1), "peak_ginst_per_s": VALU_PEAK_GINST,
                     "frac": round(valu_insts / (ms_k * 1e-3) / 1e9 / VALU_PEAK_GINST, 4),
                     "measured_issue_peak_ginst_per_s": VALU_MEASURED_GINST,
                     "frac_of_measured_issue_peak": round(valu_insts / (ms_k * 1e-3) / 1e9 / VALU_MEASURED_GINST, 4),
                     "note": "instructions, not issue slots: a packed fp32 instruction takes two slots and a transcendental "
                             "~3.5 (profiles/tools/ubench/pk_rate.hip); with ~40 % of this kernel's instructions packed and four "
                             "reciprocals per trip its stream is ~230 us of pure issue (HISTORY.md section 14)"},
                 "l2_atomics": None if flushes is None else {
                     "kernel": kern, "flushes_per_view": round(flushes), "atomics_per_view": round(17 * flushes),
                     "achieved_gatomics_per_s": round(17 * flushes / (ms_k * 1e-3) / 1e9, 2) if ms_k > 0 else None,
                     "peak": None, "note": "fp32 global atomics (hardware, -munsafe-fp-atomics), 17 per (entry, region) flush; "
                                           "no published L2 atomic peak for gfx950"}}
    # per-stage roofline fractions (level-2 stage timers of 8 views OUTSIDE the timed region): SURVEY 8(a)'s algorithmic
    # bytes of every row on the units this build processes (R_lists) against the stage's time and the HBM peak
    Rl, Vm, Pz = mean["R_lists"], mean["V"], mean["Pz"]
    stage_bytes = {
        "preprocess": 12 * P + 8 * P + 28 * Pz + (4 + 12 * K + 67) * Vm,                 # K1
        "binning": 8 * P + (4 * P + 16 * Vm + 12 * Rl) + 24 * Rl + 8 * Rl,                 # K2 + K3 + K4 (floor) + K5
        "render_fwd": 72 * Rl + 56 * N,                                                   # K6
        "render_bwd": 72 * Rl + 56 * N + 68 * Vm,                                         # K7
        "preprocess_bwd": (68 + 4 + 12 * K + 67) * Vm + (56 + 12 * K) * P,                # K8 (SURVEY's figure: incl. an SH re-read this build does not do)
    }
    stage_frac = {}
    if stage_ms:
        t_bin = sum(stage_ms.get(k, 0.0) for k in ("compact", "depth_sort", "emit", "tile_sort", "ranges"))
        for k, by in stage_bytes.items():
            t = t_bin if k == "binning" else stage_ms.get(k, 0.0)
            if t > 0:
                stage_frac[k] = {"ms": round(t, 4), "alg_bytes": round(by), "frac": round(by / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    t_view = elapsed / (args.steps * vps)
    res = {
        "metric": "rasterizer fwd+bwd views/sec + Mfragments/sec, 2M Gaussians @1376x376",
        "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        # `repeats` regions of K steps each were timed; value / ms_per_step are those of the MEDIAN region
        "repeats": len(region_s), "ms_per_step_regions": [round(r / args.steps * 1e3, 4) for r in region_s],
        # Fragments = list entries TRAVERSED by the forward blend loop (sum of n_contrib).  `mfragments_per_s`: the entries of
        # the build's own tile lists, i.e. what its loops actually walk; `..._reference_lists`: the same views counted on the
        # REFERENCE's full tile rectangles (BASELINE.md's definition; a third of those pairs cannot reach any pixel and are
        # not listed by default) -- multiplied by THIS build's view rate, so comparable with upstream's figure of merit but
        # not a statement about work done; B = (pixel, splat) pairs actually BLENDED (alpha >= 1/255 before the stop)
        "mfragments_per_s": round(frag_own_total / elapsed / 1e6, 2),
        "mfragments_per_s_reference_lists": round(frag_total / elapsed / 1e6, 2),
        "blended_mfragments_per_s": round(blend_total / elapsed / 1e6, 2),
        "exchange": exchange,
        "config": {"workload": f"{args.workload}: {P} street Gaussians (VEGS disc init), SH deg {deg}, {W}x{H} "
                               f"KITTI-360 intrinsics, {n_views} views cycled, 12 output channels + colour/quat/scale grads; tile lists "
                               f"without the (Gaussian, tile) pairs that cannot reach a pixel (R_lists, F_lists; R, F = the same views on the reference's full rectangles); "
                               + (f"EVERY DISC x{args.disc_scale} (--disc-scale: not the headline workload); " if args.disc_scale != 1.0 else "")
                               + "every view is rendered as a camera's first visit (no per-camera state carried between views)",
                   "hints": "off",
                   "gaussians": P, "width": W, "height": H, "views_per_step_per_gpu": vps, "accumulate_in_place": bool(args.accumulate),
                   "parallelism": f"view-sharded x{world}" + ("" if world == 1 else
                                                              " + direct hipIpc all-gather of SH factors (3 f32) + reduce-scatter/all-gather (11 f32) per Gaussian"
                                                              if args.exchange == "direct" and vps == 1 else
                                                              " + RCCL all-gather of SH factors (3 f32) + all-reduce (11 f32) per Gaussian"
                                                              if args.exchange == "factored" and vps == 1 else
                                                              f" + RCCL all-gather of {vps} SH factors per rank (3 f32 each) + all-reduce (11 f32, accumulated in place over the rank's views) per Gaussian"
                                                              if args.exchange in ("factored", "direct") and vps > 1 else
                                                              " + RCCL grad all-reduce (59 f32/Gaussian)"),
                   "mean_counters": {k: round(v, 1) for k, v in mean.items()}},
        "roofline": {"bound": "hbm", "kernel": kern, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "traffic_collected": traffic_from, "secondary": secondary,
                     "alg_bytes_per_launch": round(bytes_k), "avg_launch_ms": round(ms_k, 4),
                     "alg_bytes_note": "72 R_lists + 56 N + 68 V: SURVEY 8(a) row K7 per unit x the units one launch processes "
                                       "(R_lists = entries of the build's own tile lists, the ones the kernel walks)",
                     "frac_on_reference_lists": round(bytes_k_ref / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms_k > 0 else 0.0,
                     "launches_timed": int(stage[kern][1]),
                     "stage_ms": stage_ms,
                     "stages": stage_frac,
                     "whole_view_alg_bytes": round(b_alg),
                     # (counter bytes of every kernel of one view, committed profile; null while any kernel source is newer)
                     "whole_view_traffic": whole_view_traffic(),
                     "whole_view_frac": round(b_alg / t_view / 1e9 / HBM_PEAK_GBS, 5),
                     "whole_view_frac_on_reference_lists": round(b_alg_ref / t_view / 1e9 / HBM_PEAK_GBS, 5)},
    }
    if args.stages:
        print("hipMalloc calls inside the timed region:", mallocs1 - mallocs0, "reserved MB before/after:",
              reserved0 >> 20, torch.cuda.memory_reserved(device) >> 20, file=sys.stderr)
        print("stage breakdown (ms per view; k_seg_bwd is part of render_bwd):", stage_ms, file=sys.stderr)
    if world == 1 and not args.no_variants:
        del wl, step
        torch.cuda.empty_cache()
        cams_w = []
        for s_ in range(4):
            for y in (0.3, -0.3):
                cams_w.append(scenes.kitti_camera(10.0 * s_, y, 1408, 376))
        dense = dict(sc)
        dense["scales"] = (sc["scales"] * 3.0).astype(np.float32)
        res["variants"] = [
            variant("same scene at 1408x376 (the resolution the reference's comments use)", sc, deg, cams_w, device, 16, 4),
            variant("dense: same scene with every disc 3x larger (R ~ an order of magnitude up), 1376x376", dense, deg,
                    cams[:8], device, 8, 2),
            variant("headline scene, SH gradient returned as its 3-float factor (sh_color_grad) instead of [P,16,3]", sc,
                    deg, cams[:8], device, 16, 4, factored=True),
            variant("headline scene, FORWARD ONLY under no_grad with per-camera needed-segment hints WARM (the same fixed "
                    "cameras of a static model rendered again: the one place the hint cache is offered -- in training it lost "
                    "1.5 % once the hints were an epoch old, BENCH_r03, and a differentiated forward no longer gets one)", sc,
                    deg, cams, device, 16, 4, hints="warm", mode="forward"),
            variant("headline scene, op called without the reference's render() glue (one persistent means2D leaf instead "
                    "of zeros_like + 0 / retain_grad per view): the difference to the headline is ATen glue", sc, deg, cams,
                    device, 16, 4, mode="noglue"),
            variant("headline scene, FORWARD ONLY under torch.no_grad() (evaluation / video rendering, train.py:338-508, "
                    "render_video.py:162,202)", sc, deg, cams, device, 16, 4, mode="forward"),
            variant("headline scene, FORWARD ONLY, TWO VIEWS IN FLIGHT on two HIP streams (vegs_amd/views.py: frames of an "
                    "evaluation / video loop are independent)", sc, deg, cams, device, 2, 1, mode="forward", vps=16, streams=2),
            variant("headline scene, a BATCH OF 8 VIEWS per iteration on ONE stream (gradients of the views summed by "
                    "autograd: + a dense accumulate per view)", sc, deg, cams, device, 4, 1, vps=8),
            variant("headline scene, a batch of 8 views per iteration, TWO VIEWS IN FLIGHT on two HIP streams", sc, deg,
                    cams, device, 4, 1, vps=8, streams=2),
            variant("headline scene, a batch of 8 views per iteration on ONE stream, gradients ACCUMULATED IN PLACE "
                    "(rasterizer.accumulate_grads / VR_FLAG_ACCUMULATE_GRADS: from the second view on the backward adds its "
                    "visible rows into the leaves' .grad instead of writing dense arrays for autograd to add)", sc, deg, cams,
                    device, 4, 1, vps=8, accumulate=True),
            variant("headline scene, a batch of 8 views per iteration, gradients accumulated in place, TWO VIEWS IN FLIGHT "
                    "on two HIP streams", sc, deg, cams, device, 4, 1, vps=8, streams=2, accumulate=True),
            variant("headline scene, a batch of 8 views per iteration on ONE stream, the 11 non-SH floats accumulated in place "
                    "and the SH gradient kept FACTORED per view (3 floats), the step's dense [P,16,3] gradient rebuilt once "
                    "from the 8 factors inside the timed region", sc, deg, cams, device, 4, 1, vps=8, accumulate=True, factored=True),
            variant("headline scene, a batch of 8 views, in-place accumulation + factored SH, TWO VIEWS IN FLIGHT", sc, deg,
                    cams, device, 4, 1, vps=8, streams=2, accumulate=True, factored=True),
            variant("headline scene with VR_FLAG_FULL_TILE_LISTS: every tile of the reference's rectangles a list entry (the "
                    "build's default leaves out the third of them whose tile the splat cannot reach: same radii, images and "
                    "gradients to rounding) -- what the tight lists buy", sc, deg, cams, device, 16, 4, flags=rasterizer.FLAG_FULL_TILE_LISTS),
            variant("headline scene with VR_FLAG_FAST_EXP: the compositing's 2^x by v_exp_f32 in forward AND backward (lists "
                    "bit-exact, images within 1e-5 of the bit-exact mode but for threshold fragments; NOT the headline mode)",
                    sc, deg, cams, device, 16, 4, flags=rasterizer.FLAG_FAST_EXP),
        ]
        noglue = [v for v in res["variants"] if "without the reference's render() glue" in v["workload"]][0]
        full = [v for v in res["variants"] if "VR_FLAG_FULL_TILE_LISTS" in v["workload"]][0]
        # the north star asks for the reference's tile / sort indices: the same views with the reference's full tile
        # rectangles as the lists (VR_FLAG_FULL_TILE_LISTS) -- the headline `value` runs the default, tighter lists
        res["views_per_s_reference_tile_lists"] = full["views_per_s"]
        res["ms_per_view_reference_tile_lists"] = full["ms_per_view"]
        res["aten_glue_ms_per_view"] = round(res["ms_per_step"] / vps - noglue["ms_per_view"], 4)
    if world == 1 and not args.no_cpu_baseline:
        cpu_views = [0, 2, 4, 6, 8, 10, 12, 14]           # ~1.6 s each on the box's host cores: 10 ... 15 s of CPU work
        dt, frags, cores = cpu_baseline(sc, deg, cams, gouts, cpu_views)
        res["cpu_baseline"] = {"value": round(len(cpu_views) / dt, 5), "unit": "views/s", "cores": cores, "kind": "port",
                               "sample": f"{len(cpu_views)} views (every second one) of the same workload, oracle fwd+bwd on all "
                                         f"host cores, {dt:.1f} s",
                               "mfragments_per_s": round(frags / dt / 1e6, 2),
                               "baseline_md_plan": cpu_plan(cores)}
    print(json.dumps(res))


def _finish():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    finally:
        _finish()
