"""Host side of the DIRECT gradient exchange (include/vegs_xgmi.h, vegs_amd/csrc/xgmi.hip): every rank pushes its 1/N
shards into ALL peers' windows at once through hipIpc mappings -- all 7 xGMI links of a GPU busy, where RCCL's ring keeps
one link pair busy per step -- and the reduced gradients / gathered factors are READ WHERE THEY LAND: the tensors this
module hands back are views of the window (no un-bucketing copy).

    ex = DirectExchange(rank, world, device, group)             # once; windows are (re)sized collectively on demand
    ex.begin_gather([factor, campos])                           # as soon as the factors exist (between the backward's halves)
    grads = ex.allreduce_mean([g_xyz, g_opacity, ...], scale)   # -> list of tensors = views of the window's result[]
    F, C = ex.finish_gather([(n_local, rows, 3), (n_local, 3)]) # -> [world * n_local, rows, 3], [world * n_local, 3]

torch.distributed is used for ONE thing: all-gathering the 64-byte IPC handles when a window is created (and the
agreement on its size).  PyTorch is plumbing: the window is wrapped as a tensor through __cuda_array_interface__.
RCCL stays the default exchange of the trainer and of bench.py until a multi-GPU node has measured both
(`--exchange direct`); several processes on ONE GPU run the same code (tests/test_gpu_xgmi.py).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _capi


class _Window:
    """The whole window as a float32 tensor (zero-copy) via the CUDA array interface."""

    def __init__(self, ptr, floats):
        self.__cuda_array_interface__ = {"shape": (int(floats),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class DirectExchange:
    def __init__(self, rank, world, device, group=None, headroom=1.25):
        self.rank, self.world, self.device, self.group, self.headroom = int(rank), int(world), torch.device(device), group, headroom
        self.ctx = None
        self.reduce_cap = self.gather_cap = 0
        self.win = None
        self.parity = 0
        self._pending = None
        self._side = None
        self.wait_bound = None          # seconds; None = the library's default (10 s)

    def set_wait_bound(self, seconds):
        """Bound of every device-side wait for a peer (include/vegs_xgmi.h: FAILURE CONTRACT); applies to this and every
        later window of the object."""
        self.wait_bound = float(seconds)
        if self.ctx is not None:
            _capi.check(_capi.load().vr_xgmi_set_wait_bound(self.ctx, self.wait_bound))

    def failed(self):
        """True once a wait on the current window has timed out, as far as the host can see WITHOUT synchronising (the
        pinned mirror of the window's error word).  Every exchange call tests the same word first and raises."""
        return self.ctx is not None and bool(_capi.load().vr_xgmi_failed(self.ctx))

    # ---- window management (collective: every rank must make the same calls with the same sizes)
    def _ensure(self, reduce_floats, gather_floats, agree=False):
        """Collective (re)allocation when a capacity is exceeded.  Views handed out earlier (gradients, gathered blocks)
        die with the old window: callers hold them for one iteration only.  The NEW window is created and mapped by
        everybody while the old one still exists (two live allocations cannot be mistaken for each other by the IPC
        layer), then the old one is retired: unmapped everywhere, barrier, freed."""
        if self.ctx is not None and reduce_floats <= self.reduce_cap and gather_floats <= self.gather_cap:
            return
        if self._pending is not None:
            raise RuntimeError("the window would have to grow while a gather is in flight: reserve() both capacities first")
        lib = _capi.load()
        multi = self.world > 1
        old = self.ctx
        if old is not None and multi:
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)              # every rank idle: nobody reads or writes an old window any more
        if reduce_floats > self.reduce_cap:
            self.reduce_cap = int(reduce_floats * self.headroom) + 64
        if gather_floats > self.gather_cap:
            self.gather_cap = int(gather_floats * self.headroom) + 64
        HB = 72
        ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            handle = (C.c_ubyte * HB)()
            try:
                _capi.check(lib.vr_xgmi_create(self.rank, self.world, self.reduce_cap, self.gather_cap, C.byref(ctx)))
                _capi.check(lib.vr_xgmi_handle(ctx, handle))
                made = True
            except Exception:
                if not agree:
                    raise
                made = False
            if agree and not self._agree(made):       # (try_setup: nobody goes on to the handle exchange alone)
                if ctx.value:
                    lib.vr_xgmi_destroy(ctx)
                self.reduce_cap = self.gather_cap = 0
                return
            if multi:
                mine = torch.tensor(list(handle), dtype=torch.uint8)
                if dist.get_backend(self.group) == "nccl":
                    allh = torch.empty((self.world, HB), dtype=torch.uint8, device=self.device)
                    dist.all_gather_into_tensor(allh, mine.to(self.device), group=self.group)
                    allh = allh.cpu()
                else:
                    parts = [torch.empty(HB, dtype=torch.uint8) for _ in range(self.world)]
                    dist.all_gather(parts, mine, group=self.group)
                    allh = torch.stack(parts)
                buf = (C.c_ubyte * (HB * self.world))(*allh.reshape(-1).tolist())
                try:
                    _capi.check(lib.vr_xgmi_attach(ctx, buf))
                    mapped = True
                except Exception:
                    if not agree:
                        raise
                    mapped = False
                if agree and not self._agree(mapped):
                    torch.cuda.synchronize(self.device)
                    lib.vr_xgmi_destroy(ctx)
                    self.reduce_cap = self.gather_cap = 0
                    return
                dist.barrier(group=self.group)          # nobody pushes before everybody has mapped everybody
        self.win = None
        if old is not None:
            _capi.check(lib.vr_xgmi_detach(old))
            if multi:
                dist.barrier(group=self.group)          # nobody frees a window that a peer still has mapped
            _capi.check(lib.vr_xgmi_destroy(old))
        self.ctx = ctx
        if self.wait_bound is not None:
            _capi.check(lib.vr_xgmi_set_wait_bound(ctx, self.wait_bound))
        lay = _capi.VrXgmiLayout()
        _capi.check(lib.vr_xgmi_layout(ctx, C.byref(lay)))
        self.lay = lay
        self.win = torch.as_tensor(_Window(lib.vr_xgmi_window(ctx), lay.total_floats), device=self.device)
        self.parity = 0
        self._pending = None

    def _agree(self, ok):
        """True iff `ok` on EVERY rank (one tiny collective: ranks must take the same branch after a step that may fail
        locally -- a rank that went on alone would wait for the others forever)."""
        if self.world <= 1 or not dist.is_initialized():
            return bool(ok)
        dev = self.device if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    def try_setup(self, reduce_floats, gather_floats, verify=True):
        """Collective, failure-tolerant set-up for callers that have a fall-back (bench.py --exchange auto): allocates and
        maps the windows and -- `verify` -- runs one small exchange whose result every rank checks against what it must be.
        Returns True on every rank or False on every rank (after tearing down whatever was built); never raises for
        device / IPC errors."""
        try:
            self._ensure(int(reduce_floats), int(gather_floats), agree=True)
            ok = self.ctx is not None
        except Exception:
            ok = False
        if not self._agree(ok):
            self._drop()
            return False
        if not verify:
            return True
        try:
            n = 4099
            g = torch.full((n,), float(self.rank + 1), device=self.device)
            f = torch.full((1, 33, 3), float(self.rank + 1), device=self.device)
            c = torch.full((1, 3), float(10 * self.rank), device=self.device)
            self.begin_gather([f, c])
            (r,) = self.allreduce_mean([g], 1.0)
            F, Cc = self.finish_gather()
            self.check()
            want = float(self.world * (self.world + 1) // 2)
            ok = bool((r == want).all()) and all(bool((F[j] == float(j + 1)).all()) and bool((Cc[j] == float(10 * j)).all())
                                                  for j in range(self.world))
        except Exception:
            ok = False
        if not self._agree(ok):
            self._drop()
            return False
        return True

    def _drop(self):
        """Tear down without collectives (the peers are doing the same, or never got this far)."""
        try:
            if self.ctx is not None:
                torch.cuda.synchronize(self.device)
                self.win = None
                _capi.load().vr_xgmi_destroy(self.ctx)
        except Exception:
            pass
        self.ctx, self.win, self._pending = None, None, None
        self.reduce_cap = self.gather_cap = 0

    def close(self):
        if self.ctx is not None:
            multi = self.world > 1 and dist.is_initialized()
            if multi:
                torch.cuda.synchronize(self.device)
                dist.barrier(group=self.group)          # peers may still be reading / writing this window
            self.win = None
            _capi.check(_capi.load().vr_xgmi_detach(self.ctx))
            if multi:
                dist.barrier(group=self.group)          # nobody frees a window that a peer still has mapped
            _capi.check(_capi.load().vr_xgmi_destroy(self.ctx))
            self.ctx = None

    def __del__(self):
        try:
            if self.ctx is not None:
                _capi.load().vr_xgmi_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    def reserve(self, reduce_floats, gather_floats):
        """Size the window (collective).  Called implicitly by the exchange calls; explicit use keeps the (re)allocation
        and its barrier out of a timed region."""
        self._ensure(int(reduce_floats), int(gather_floats))

    @staticmethod
    def _segments(tensors):
        keep = []
        for t in tensors:
            if not t.is_cuda or t.dtype != torch.float32:
                raise ValueError("the direct exchange moves float32 GPU tensors")
            keep.append(t.detach().contiguous())
        arr = (_capi.VrXgmiSegment * len(keep))(*[_capi.VrXgmiSegment(_capi.ptr(t), t.numel()) for t in keep])
        return keep, arr

    # ---- all-gather (one-shot push), split in two so that the push can start as early as the data exists
    def begin_gather(self, tensors, stream=None):
        """Push this rank's block (the tensors, back to back) into slot `rank` of every peer's gather buffer."""
        keep, arr = self._segments(tensors)
        total = sum((t.numel() + 3) // 4 * 4 for t in keep)
        self._ensure(0, total)
        offs = (C.c_int64 * len(keep))()
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _capi.check(_capi.load().vr_xgmi_allgather_begin(self.ctx, arr, len(keep), self.parity, offs, s.cuda_stream))
        self._pending = (self.parity, [int(o) for o in offs], [tuple(t.shape) for t in keep], keep, s)
        self.parity ^= 1

    def finish_gather(self):
        """Wait (on the current stream) for every peer's block; returns one tensor per pushed tensor, shaped
        [world, *shape] -- views of the window: rank j's block at index j."""
        if self._pending is None:
            raise RuntimeError("finish_gather() without begin_gather()")
        parity, offs, shapes, keep, s = self._pending
        self._pending = None
        cur = torch.cuda.current_stream(self.device)
        if s != cur:
            cur.wait_stream(s)
        with torch.cuda.device(self.device):
            _capi.check(_capi.load().vr_xgmi_allgather_wait(self.ctx, parity, cur.cuda_stream))
        base, slot = int(self.lay.gather_offset[parity]), int(self.lay.gather_slot)
        region = self.win[base:base + self.world * slot].view(self.world, slot)
        out = []
        for off, shape in zip(offs, shapes):
            n = 1
            for d in shape:
                n *= d
            out.append(region[:, off:off + n].view((self.world,) + shape))
        return out

    # ---- all-reduce (two-shot push), result read in place
    def allreduce_mean(self, tensors, scale=None):
        """scale * (sum over the ranks, in rank order) of every tensor; returns tensors shaped like the inputs that are
        VIEWS of the window's result buffer (valid until the next allreduce_mean)."""
        keep, arr = self._segments(tensors)
        total = sum(t.numel() for t in keep)
        self._ensure(total, 0)
        offs = (C.c_int64 * len(keep))()
        scale = (1.0 / self.world) if scale is None else float(scale)
        with torch.cuda.device(self.device):
            _capi.check(_capi.load().vr_xgmi_allreduce(self.ctx, arr, len(keep), scale, offs,
                                                       torch.cuda.current_stream(self.device).cuda_stream))
        base = int(self.lay.result_offset)
        return [self.win[base + int(o):base + int(o) + t.numel()].view(t.shape) for o, t in zip(offs, keep)]

    def check(self):
        with torch.cuda.device(self.device):
            _capi.check(_capi.load().vr_xgmi_check(self.ctx, torch.cuda.current_stream(self.device).cuda_stream))

    # ---- overlapped use, like vegs_amd.dist.FactorExchange: the factors start travelling between the backward's halves
    def armed(self, campos):
        """`with ex.armed(campos): loss.backward()` -- the op's backward hands its SH factor [rows,3] to begin_gather on a
        SIDE stream as soon as the render backward has produced it, so the pushes to the peers run under
        k_preprocess_bwd.  Size the window first (reserve()): nothing may be re-allocated while a gather is in flight."""
        import contextlib
        from . import rasterizer

        @contextlib.contextmanager
        def scope():
            if self._side is None:
                self._side = torch.cuda.Stream(self.device)
            c = campos.detach().to(self.device, torch.float32).reshape(1, 3).contiguous()

            def on_factors(factor):
                cur = torch.cuda.current_stream(self.device)
                self._side.wait_stream(cur)
                f = factor.detach()
                f.record_stream(self._side)
                self.begin_gather([f[None], c], stream=self._side)
            old = rasterizer.set_backward_split_hook(on_factors)
            try:
                yield self
            finally:
                rasterizer.set_backward_split_hook(old)
        return scope()

    def finish(self, tensors, n_local=1):
        """After the armed backward: all-reduce (mean) of the tensors' gradients -- left as views of the window -- and the
        gathered (F [world, rows, 3], C [world, 3])."""
        if self._pending is None:
            raise RuntimeError("DirectExchange.finish(): the backward did not deliver a factor (was the op called with sh_color_grad?)")
        grads = [t for t in tensors if t.grad is not None]
        out = self.allreduce_mean([t.grad for t in grads], 1.0 / self.world)
        for t, g in zip(grads, out):
            t.grad = g
        F, Cc = self.finish_gather()
        return F.reshape((self.world * n_local,) + tuple(F.shape[2:])), Cc.reshape(self.world * n_local, 3)

    # ---- the trainer's exchange (vegs_amd.iteration.Trainer, exchange="direct")
    def exchange(self, tensors, Fv, Cv, n_local):
        """tensors: parameters whose .grad holds this rank's sum over its n_local views; Fv [n_local, rows, 3] / Cv
        [n_local, 3]: this rank's SH factors and camera centres.  On return every .grad is the mean over all views (a
        view of the window) and (F, C) are the gathered [world * n_local, ...] blocks."""
        grads = [t for t in tensors if t.grad is not None]
        # both regions sized in ONE collective (re)allocation, before anything is in flight
        self._ensure(sum(t.grad.numel() for t in grads), Fv.numel() + Cv.numel() + 8)
        self.begin_gather([Fv, Cv])
        out = self.allreduce_mean([t.grad for t in grads], 1.0 / self.world)
        for t, g in zip(grads, out):
            t.grad = g
        F, Cc = self.finish_gather()
        return F.reshape((self.world * n_local,) + tuple(Fv.shape[1:])), Cc.reshape(self.world * n_local, 3)
