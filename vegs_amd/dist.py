"""View-sharded multi-GPU helper: one process per GPU, one camera per rank, replicated Gaussians,
one gradient exchange per iteration.

The reference trains one view per iteration on one GPU (train.py:126-150) and has no distributed
code at all; batching views is an extension defined so that one view per iteration reproduces the
reference exactly:  loss = mean over the iteration's views of the per-view loss.

Exchange step = all-reduce(SUM) of the gradients of the op's Gaussian inputs -- 3+48+1+3+4 = 59
floats per Gaussian (the parameter groups of scene/gaussian_model.py:159-166) -- scaled by 1/n_views,
plus the densification statistics of scene/gaussian_model.py:411-413 / train.py:300:
SUM of per-view ||grad_xy|| * visible and of `visible`, MAX of radii.  Works on any
torch.distributed backend: "nccl" (= RCCL over xGMI on MI355X) on GPUs, "gloo" in the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  No-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # VEGS_DIST_BACKEND=gloo lets the N>1 path be exercised on a single-GPU box (several ranks
            # sharing one device, which RCCL refuses); production default is nccl (= RCCL over xGMI)
            backend = os.environ.get("VEGS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def view_for_rank(step, rank, world, n_views):
    """Camera index rendered by `rank` at iteration `step`: consecutive views go to consecutive ranks."""
    return (step * world + rank) % n_views


_AVG_OK = {}


def _avg_supported(group, device):
    """ReduceOp.AVG exists on RCCL/NCCL only; probed once per (backend, group) with a one-element collective
    (every rank takes the same branch, so the probe cannot desynchronise the ranks)."""
    backend = dist.get_backend(group)
    key = (backend, id(group))
    if key not in _AVG_OK:
        ok = False
        if backend == "nccl":
            try:
                probe = torch.ones(1, device=device)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=group)
                ok = abs(float(probe.item()) - 1.0) < 1e-6
            except (RuntimeError, ValueError):
                ok = False
        _AVG_OK[key] = ok
    return _AVG_OK[key]


def _all_reduce(t, op, group=None, async_op=False):
    """dist.all_reduce that also serves GPU tensors on a backend without device support (gloo: the hook that lets the
    N > 1 path run with several ranks on ONE GPU, which RCCL refuses) by staging through the host.  Returns a work
    handle or None."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.detach().cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
        return None
    return dist.all_reduce(t, op=op, group=group, async_op=async_op)


def allreduce_grads(tensors, world=None, group=None, flat_bucket_bytes=1 << 20):
    """In-place mean over ranks of `tensor.grad` for every tensor in `tensors` (same shapes on all
    ranks).  Gradients of at least `flat_bucket_bytes` are reduced in place, one collective each, with no
    staging copies (they are already contiguous [P,k] blocks written by the rasterizer's backward; at
    2 M Gaussians that is all five: 8 ... 384 MB); smaller ones are packed into one flat bucket so the launch
    count stays low.  On RCCL/NCCL the 1/world scaling rides in the collective (ReduceOp.AVG) instead of
    a separate pass over the 236 bytes per Gaussian; backends without AVG (gloo) sum and scale."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    grads = [t.grad for t in tensors if t.grad is not None]
    if world <= 1 or not grads:
        return
    avg = _avg_supported(group, grads[0].device)
    op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
    big = [g for g in grads if g.numel() * g.element_size() >= flat_bucket_bytes and g.is_contiguous()]
    small = [g for g in grads if not (g.numel() * g.element_size() >= flat_bucket_bytes and g.is_contiguous())]
    works = []
    for g in big:
        works.append(_all_reduce(g, op, group, async_op=True))
    flat = None
    if small:
        flat = torch.cat([g.reshape(-1) for g in small])
        works.append(_all_reduce(flat, op, group, async_op=True))
    for w in works:
        if w is not None:
            w.wait()
    inv = 1.0 / world
    if not avg:
        for g in big:
            g.mul_(inv)
    if flat is not None:
        if not avg:
            flat.mul_(inv)
        off = 0
        for g in small:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n


def allreduce_densification_stats(viewspace_grad, visibility, radii, group=None):
    """Per-view statistics -> statistics of the whole view batch, identical on every rank.
    Returns (grad_norm_sum [P,1], denom [P,1], max_radii [P]):
      grad_norm_sum = sum over views of ||viewspace_grad[:, :2]|| where visible   (NOT the norm of the sum)
      denom         = number of views in which the Gaussian was visible
      max_radii     = max over views of radii."""
    vis = visibility.to(viewspace_grad.dtype).unsqueeze(1)
    stats = torch.cat([torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True) * vis, vis], dim=1)
    mr = radii.clone()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        _all_reduce(stats, dist.ReduceOp.SUM, group)
        _all_reduce(mr, dist.ReduceOp.MAX, group)
    return stats[:, 0:1], stats[:, 1:2], mr


def exchange_factored(tensors, sh_factor, campos, world=None, group=None):
    """The view-sharded exchange with FACTORED SH gradients (vegs_amd.rasterizer: sh_color_grad; include/vegs_rast.h:
    VrInGrads.dL_dcolors_sh).  Per view, dL/dshs is the rank-1 product basis(dir(camera, mean)) x dL/d(colour); every
    rank holds the means and can be told all cameras, so only the 3-float factor has to travel:
      * `tensors` (means3D, opacities, scales, rotations: 11 floats per Gaussian): all-reduce mean, as allreduce_grads;
      * `sh_factor` [P,3] (this rank's factor) and `campos` [3] (this rank's camera centre): ALL-GATHER.
    Per rank and view 12 + 44 instead of 236 bytes per Gaussian cross the links (at N = 8: 8 x 24 MB gathered + 88 MB
    reduced instead of 472 MB reduced, for 2 M Gaussians), and the 384 MB dense SH gradient is never materialised:
    returns (factors [N,P,3], campos [N,3]) for vegs_amd.optim.adam_step_sh_factored(..., scale=1/N) (or
    sh_grad_from_factors).  With a single process it returns the inputs as a batch of one view."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    f = sh_factor.detach().contiguous()
    c = campos.detach().to(f.device, torch.float32).reshape(3).contiguous()
    if world <= 1:
        return f[None], c[None]
    allreduce_grads(tensors, world, group)
    if dist.get_backend(group) == "nccl":     # RCCL: one collective straight into the [N,P,3] / [N,3] blocks
        out_f = torch.empty((world,) + tuple(f.shape), dtype=f.dtype, device=f.device)
        out_c = torch.empty((world, 3), dtype=c.dtype, device=c.device)
        dist.all_gather_into_tensor(out_f, f, group=group)
        dist.all_gather_into_tensor(out_c, c, group=group)
        return out_f, out_c
    staged = f.is_cuda                          # gloo (tests, single-GPU box): through the host
    src_f, src_c = (f.cpu(), c.cpu()) if staged else (f, c)
    fs = [torch.empty_like(src_f) for _ in range(world)]
    cs = [torch.empty_like(src_c) for _ in range(world)]
    dist.all_gather(fs, src_f, group=group)
    dist.all_gather(cs, src_c, group=group)
    return torch.stack(fs).to(f.device), torch.stack(cs).to(f.device)
