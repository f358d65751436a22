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


def allreduce_grads(tensors, world=None, group=None, flat_bucket_bytes=64 << 20):
    """In-place mean over ranks of `tensor.grad` for every tensor in `tensors` (same shapes on all
    ranks).  Large gradients are reduced in place, one collective each (they are already contiguous
    [P,k] blocks written by the rasterizer's backward); small ones are packed into one flat bucket so
    the launch count stays low.  Returns the list of async work handles already waited for."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    grads = [t.grad for t in tensors if t.grad is not None]
    if world <= 1 or not grads:
        return
    big = [g for g in grads if g.numel() * g.element_size() >= flat_bucket_bytes]
    small = [g for g in grads if g.numel() * g.element_size() < flat_bucket_bytes]
    works = []
    for g in big:
        works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True))
    flat = None
    if small:
        flat = torch.cat([g.reshape(-1) for g in small])
        works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    inv = 1.0 / world
    for g in big:
        g.mul_(inv)
    if flat is not None:
        off = 0
        for g in small:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            g.mul_(inv)
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
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(mr, op=dist.ReduceOp.MAX, group=group)
    return stats[:, 0:1], stats[:, 1:2], mr
