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
        _shared_device_precautions(world)
    return rank, world, local


def _shared_device_precautions(world):
    """Several ranks on ONE GPU (the single-GPU test configuration; production is one process per GPU): the single-launch
    radix passes of the binning make a workgroup wait for lower-indexed workgroups of the same launch, which is safe as
    long as that launch's workgroups are dispatched in order onto free CUs -- not when eight processes' launches
    oversubscribe the chip and wait on each other's CU slots (observed: 8 ranks x 2 M Gaussians on one MI355X: waits
    timing out, views failed).  Those ranks take the multi-launch passes without inter-workgroup waits instead
    (VR_FLAG_SCAN_BINNING: same lists, bit for bit)."""
    if not torch.cuda.is_available():
        return
    per_node = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if per_node > torch.cuda.device_count():
        from . import rasterizer
        rasterizer.set_flags(rasterizer.get_flags() | rasterizer.FLAG_SCAN_BINNING)


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
        off, parts = 0, []
        for g in small:
            n = g.numel()
            parts.append(flat[off:off + n].view_as(g))
            off += n
        torch._foreach_copy_(small, parts)          # one multi-tensor launch instead of one copy per gradient


def all_gather_views(factors, campos, world=None, group=None):
    """All-gather of the ranks' SH factors [n_local, rows, 3] and camera centres [n_local, 3] -> ([world * n_local, rows,
    3], [world * n_local, 3]), rank-major (rank r's views at r * n_local ...).  RCCL: one collective each; gloo (tests,
    several ranks on one GPU): staged through the host."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    f = factors.detach().contiguous()
    c = campos.detach().to(f.device, torch.float32).contiguous()
    if world <= 1:
        return f, c
    if dist.get_backend(group) == "nccl":
        out_f = torch.empty((world * f.shape[0],) + tuple(f.shape[1:]), dtype=f.dtype, device=f.device)
        out_c = torch.empty((world * c.shape[0], 3), dtype=c.dtype, device=c.device)
        dist.all_gather_into_tensor(out_f, f, group=group)
        dist.all_gather_into_tensor(out_c, c, group=group)
        return out_f, out_c
    staged = f.is_cuda
    src_f, src_c = (f.cpu(), c.cpu()) if staged else (f, c)
    fs = [torch.empty_like(src_f) for _ in range(world)]
    cs = [torch.empty_like(src_c) for _ in range(world)]
    dist.all_gather(fs, src_f, group=group)
    dist.all_gather(cs, src_c, group=group)
    return torch.cat(fs).to(f.device), torch.cat(cs).to(f.device)


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


# ---------------------------------------------------------------------------------------------------------------------
# Overlapped exchange (round 3).  The SH factors are complete when the RENDER backward is; k_preprocess_bwd (0.14 ms at
# 2 M Gaussians) still has to run before the other 11 floats per Gaussian exist.  With the backward split in two calls
# (include/vegs_rast.h: vr_backward_render / vr_backward_preprocess; vegs_amd.rasterizer.set_backward_split_hook) the
# all-gather of the factors is started between the two and travels while the second half computes; the all-reduce of the
# remaining gradients follows as soon as they exist and runs CONCURRENTLY with whatever is left of the all-gather (both
# asynchronous; one wait at the end).  Optionally the all-reduce only carries the rows some rank actually rendered.

def allreduce_rows(tensors, radii, world=None, group=None, threshold=0.7):
    """Mean over ranks of `t.grad` for t in `tensors` ([P, ...] each), exchanging only the rows that are visible
    (radii > 0) on AT LEAST ONE rank: a byte mask is OR-ed over the ranks (all-reduce MAX, P bytes), the rows of the
    union are packed into one [n, k] block, reduced, and scattered back -- rows outside the union are exactly zero on
    every rank (the op writes zeros for culled Gaussians) and stay zero.  Falls back to the dense all-reduce when the
    union covers more than `threshold` of the rows (packing would cost more than it saves).  Returns the number of rows
    exchanged (P for the dense path)."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    grads = [t.grad for t in tensors if t.grad is not None]
    if world <= 1 or not grads:
        return 0
    P = grads[0].shape[0]
    if radii.numel() != P:
        raise ValueError(f"allreduce_rows: radii has {radii.numel()} rows, the gradients {P} (op-level radii that include "
                         "instance rows must be sliced to the model's rows first)")
    vis = (radii > 0).to(torch.uint8)
    _all_reduce(vis, dist.ReduceOp.MAX, group)
    idx = torch.nonzero(vis).reshape(-1)
    n = int(idx.numel())
    if n > threshold * P:
        allreduce_grads(tensors, world, group)
        return P
    widths = [g.reshape(P, -1).shape[1] for g in grads]
    pack = torch.cat([g.reshape(P, -1).index_select(0, idx) for g in grads], dim=1).contiguous()
    avg = _avg_supported(group, pack.device)
    _all_reduce(pack, dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group)
    if not avg:
        pack.mul_(1.0 / world)
    off = 0
    for g, wdt in zip(grads, widths):
        g.reshape(P, -1).index_copy_(0, idx, pack[:, off:off + wdt])
        off += wdt
    return n


class FactorExchange:
    """One iteration's overlapped exchange with factored SH gradients:

        ex = FactorExchange(world)              # once
        ex.begin(campos)                        # before loss.backward(): arms the hook between the backward's halves
        loss.backward()                         #   -> the all-gather of the factors starts after the render backward
        F, C = ex.finish(others, radii)         # all-reduce of the other gradients, then ONE wait for both collectives

    F [N,P,3] / C [N,3] are what vegs_amd.optim.sh_grad_from_factors / adam_step_sh_factored take (scale = 1/N).
    With a backend that cannot run device collectives asynchronously (gloo in the single-GPU tests) the same calls
    are made in the same order, synchronously: the arithmetic is identical, only nothing overlaps."""

    def __init__(self, world=None, group=None, sparse_rows=False):
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.group, self.sparse_rows = group, sparse_rows
        self._work, self._out_f, self._out_c, self._campos, self._f = [], None, None, None, None
        self._armed, self._old_hook = False, None
        self.rows_exchanged = None

    def begin(self, campos):
        from . import rasterizer
        if self._armed:
            raise RuntimeError("FactorExchange.begin(): already armed (finish() or abort() the previous iteration first)")
        self._campos = campos
        self._work, self._out_f, self._out_c, self._f = [], None, None, None
        self._old_hook = rasterizer.set_backward_split_hook(self._on_factors)
        self._armed = True

    def abort(self):
        """Take the hook out again without exchanging anything (the backward raised): later, unrelated backward passes
        must not start stale collectives."""
        from . import rasterizer
        if self._armed:
            rasterizer.set_backward_split_hook(self._old_hook)
            self._armed = False

    def armed(self, campos):
        """`with ex.armed(campos): loss.backward()` -- begin(), and abort() if the backward raises."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            self.begin(campos)
            try:
                yield self
            except BaseException:
                self.abort()
                raise
        return scope()

    def _on_factors(self, factor):
        """Called by the op's backward between its two halves: `factor` [P,3] is complete in stream order."""
        f = factor.detach()
        self._f = f
        if self.world <= 1:
            return
        c = self._campos.detach().to(f.device, torch.float32).reshape(3).contiguous()
        if dist.get_backend(self.group) == "nccl":
            self._out_f = torch.empty((self.world,) + tuple(f.shape), dtype=f.dtype, device=f.device)
            self._out_c = torch.empty((self.world, 3), dtype=c.dtype, device=c.device)
            self._work.append(dist.all_gather_into_tensor(self._out_f, f, group=self.group, async_op=True))
            self._work.append(dist.all_gather_into_tensor(self._out_c, c, group=self.group, async_op=True))
        else:                                   # host-staged transport: gathered in finish()
            self._c = c

    def finish(self, tensors, radii=None):
        self.abort()                             # (the hook is needed during the backward only)
        if self._f is None:
            raise RuntimeError("FactorExchange.finish(): the backward did not deliver a factor "
                               "(was the op called with sh_color_grad?)")
        if self.world <= 1:
            c = self._campos.detach().to(self._f.device, torch.float32).reshape(1, 3)
            return self._f[None], c
        if self.sparse_rows and radii is not None:
            self.rows_exchanged = allreduce_rows(tensors, radii, self.world, self.group)
        else:
            allreduce_grads(tensors, self.world, self.group)
            self.rows_exchanged = tensors[0].shape[0]
        if self._out_f is not None:
            for w in self._work:
                w.wait()
            return self._out_f, self._out_c
        f, c = self._f.contiguous(), self._c
        staged = f.is_cuda
        src_f, src_c = (f.cpu(), c.cpu()) if staged else (f, c)
        fs = [torch.empty_like(src_f) for _ in range(self.world)]
        cs = [torch.empty_like(src_c) for _ in range(self.world)]
        dist.all_gather(fs, src_f, group=self.group)
        dist.all_gather(cs, src_c, group=self.group)
        return torch.stack(fs).to(f.device), torch.stack(cs).to(f.device)


def exchange_bytes_per_rank(P, world, scheme, rows=None):
    """Bytes one rank SENDS per iteration (ring / direct algorithms send as much as they receive): dense = all-reduce of
    59 floats per Gaussian; factored = all-gather of 3 floats (every rank receives (N-1) blocks) + all-reduce of 11."""
    if world <= 1:
        return 0
    ar = lambda nbytes: 2.0 * (world - 1) / world * nbytes
    if scheme == "dense":
        return int(ar(236 * P))
    rows = P if rows is None else rows
    return int((world - 1) * 12 * P + ar(44 * rows))


# xGMI on an 8 x MI355X node (MI355X_MICROARCH.md): every GPU has a private link to each of its 7 peers, ~153 GB/s per
# direction and link; RCCL's rings on 8 fully connected GPUs reach ~300 GB/s of bus bandwidth.
XGMI_LINK_GBS = 153.0
RCCL_RING_BUS_GBS = 300.0


def exchange_model(P, world, scheme, views_per_rank=1, link_gbs=XGMI_LINK_GBS, ring_bus_gbs=RCCL_RING_BUS_GBS):
    """What one step's gradient exchange is EXPECTED to move and cost on an xGMI node -- the prediction a first real
    multi-GPU run is read against (bench.py emits it beside the measured `exchange_exposed_ms`; no link has been measured by
    this build).  Per Gaussian: dense = all-reduce of 59 floats; factored = all-gather of 3 floats per local view +
    all-reduce of 11 floats.  Two transports per collective: RCCL's ring (bus bandwidth) and a direct one-hop algorithm that
    drives every link at once (what vegs_amd/xgmi.py does; RCCL may or may not).  Returns a dict of bytes and milliseconds."""
    if world <= 1:
        return None
    links = min(world - 1, 7)
    k = max(1, int(views_per_rank))
    ar_bytes = (236 if scheme == "dense" else 44) * P              # the all-reduced block
    ag_block = 0 if scheme == "dense" else 12 * P * k              # this rank's all-gathered block
    ring = lambda nbytes: 2.0 * (world - 1) / world * nbytes / (ring_bus_gbs * 1e9) * 1e3
    # direct two-shot all-reduce: every rank pushes (N-1)/N of the block over its N-1 links, and the reduced shards come back
    ar_link = 2.0 * ar_bytes / world
    ag_link = float(ag_block)                                        # one copy of the block to every peer, one per link
    return {
        "scheme": scheme, "views_per_rank": k, "links_per_gpu": links, "link_GBs_per_direction": link_gbs,
        "rccl_ring_bus_GBs_assumed": ring_bus_gbs,
        "all_reduce": {"bytes": int(ar_bytes), "sent_per_rank_ring": int(2.0 * (world - 1) / world * ar_bytes),
                       "bytes_per_link_direct": int(ar_link), "ms_ring": round(ring(ar_bytes), 4),
                       "ms_direct": round(ar_link / (link_gbs * 1e9) * 1e3, 4)},
        "all_gather": None if scheme == "dense" else {
            "block_bytes": int(ag_block), "received_per_rank": int((world - 1) * ag_block), "bytes_per_link_direct": int(ag_link),
            "ms_ring": round((world - 1) / world * world * ag_block / (ring_bus_gbs * 1e9) * 1e3, 4),
            "ms_direct": round(ag_link / (link_gbs * 1e9) * 1e3, 4)},
        "note": "predictions, not measurements: ms_ring = RCCL ring at the assumed bus bandwidth, ms_direct = one hop over all links "
                "at once; with the overlapped scheme the all-gather starts ~0.14 ms before the backward ends (under "
                "k_preprocess_bwd) and the all-reduce follows it, so exchange_exposed_ms should land between "
                "ms_direct(all_reduce) and ms_ring(all_reduce) + what the all-gather's tail adds",
    }
