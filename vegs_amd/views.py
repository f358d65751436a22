"""Several views in flight on one GPU: independent views alternate between HIP streams.

One view is a strict chain (preprocess -> lists -> compositing [-> loss gradient -> backward]) in which a third of
the time is spent in short, latency-bound launches (the five radix passes, the per-tile chains) that leave most of
the 256 CUs idle.  Two independent views on two streams fill each other's idle stretches: on the headline scene
forward-only rendering goes from 0.73 to 0.58 ms per view, a batch of 8 training views from 1.56 to 1.36 ms per
view (profiles/tools/streams_probe.py).  Nothing inside a view changes -- its kernels run in the same order on ONE
stream -- so every image, and every view's own gradient, is bit-identical to the one-stream result.  What is NOT fixed
is the order in which autograd adds the gradients of views whose backwards ran on different streams into a shared leaf's
`.grad`: the sums of two runs can differ by an fp32 rounding (profiles/tools/r06/accum_diag.py).  With
`rasterizer.accumulate_grads(True)` the library adds in place and orders consecutive accumulations itself: bit-equal to
autograd's accumulation on one stream (tests/test_gpu_accumulate.py).

Who has independent views: the reference's evaluation and video loops (train.py:338-508, render_video.py:162,202:
one render() per camera under no_grad, nothing carried from frame to frame) and this build's view batches (several
views per iteration with summed gradients; the reference itself trains on one view per
iteration, train.py:126-150, and that case has nothing to overlap).

The library side needs no switch: every call takes the caller's stream, workspaces are per call, the host thread's
mailbox hands out one ticket per forward.  The contract stays "a view's forward and its backward run on the same
stream"; PyTorch's autograd already replays a backward on its forward's stream.
"""
import collections

import torch


class ViewStreams:
    """Round-robin over `n` side streams of `device`.

    run(fn, *args) calls fn on the next side stream and returns (result, event recorded behind fn's work).  A side
    stream waits for the caller's stream ONCE, at its first view after construction or after a join(): what the caller
    enqueued before the batch (parameter updates) is visible to every view; what lands on the caller's stream while the
    batch runs is not waited for -- autograd puts the gradient accumulation of view k there, and view k+2 waiting for it
    would wait for view k+1, i.e. serialise the batch (measured: no gain at all).
    join() makes the caller's stream wait for all side streams -- call it before touching what the views produced
    (accumulated gradients, images) from the caller's stream.  n = 1 runs everything on the caller's stream.
    The side streams are HIGH-PRIORITY streams, not for the priority: HIP multiplexes streams onto a few hardware queues,
    and a normal-priority side stream that lands on the caller's queue sits behind the barrier packets autograd's
    accumulation puts there (they wait for the other side stream) -- one ViewStreams in eight serialised completely;
    high-priority streams get queues of their own (profiles/tools/stream_queues.py)."""

    def __init__(self, device, n=2, priority=-1):
        self.device = torch.device(device)
        self.n = max(1, int(n))
        self.side = [torch.cuda.Stream(self.device, priority=priority) for _ in range(self.n)] if self.n > 1 else []
        self._next = 0
        self._fresh = set(range(len(self.side)))

    def run(self, fn, *args, **kwargs):
        if not self.side:
            return fn(*args, **kwargs), None
        k = self._next % self.n
        st = self.side[k]
        self._next += 1
        if k in self._fresh:
            self._fresh.discard(k)
            st.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(st):
            out = fn(*args, **kwargs)
            ev = torch.cuda.Event()
            ev.record(st)
        return out, ev

    def join(self):
        if not self.side:
            return
        main = torch.cuda.current_stream(self.device)
        for st in self.side:
            main.wait_stream(st)
        self._fresh = set(range(len(self.side)))


def _tensors_of(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_of(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_of(v)


def render_sequence(cams, render_fn, device, streams=2):
    """Generator over render_fn(cam) for cam in cams, with up to `streams` views in flight (evaluation / video:
    train.py:338-508, render_video.py:162,202).  Each result is handed over on the CALLER's stream: the caller's
    stream waits for that view only (the following views keep running on their streams), and the result's tensors
    are marked as used on the caller's stream so the caching allocator does not recycle them early.  Call it under
    torch.no_grad(); results come in the order of `cams`."""
    vs = ViewStreams(device, streams)
    main = torch.cuda.current_stream(vs.device) if vs.side else None
    pending = collections.deque()

    def hand_over(item):
        out, ev = item
        if ev is not None:
            main.wait_event(ev)
            for t in _tensors_of(out):
                if t.is_cuda:
                    t.record_stream(main)
        return out

    for cam in cams:
        pending.append(vs.run(render_fn, cam))
        if len(pending) >= vs.n:
            yield hand_over(pending.popleft())
    while pending:
        yield hand_over(pending.popleft())


def view_batch(views, fwd_bwd_fn, device, streams=2):
    """Runs fwd_bwd_fn(view) -- forward, loss gradient and backward of ONE view -- for every view of a batch, the views
    alternating between `streams` streams, then joins: when it returns, the gradients every view accumulated into
    the shared parameters are complete as far as the caller's stream is concerned.  Returns the list of results."""
    vs = ViewStreams(device, streams)
    outs = [vs.run(fwd_bwd_fn, v)[0] for v in views]
    vs.join()
    if vs.side:      # results were allocated on the side streams: the caller's stream uses them from here on
        main = torch.cuda.current_stream(vs.device)
        for t in _tensors_of(outs):
            if t.is_cuda:
                t.record_stream(main)
    return outs
