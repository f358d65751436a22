"""Host-side mirror of the reference's BoxModel (model/boxmodel.py:4-57): the learnable correction of a dynamic
instance's annotated pose,

    d_box2world = [[diag(delta_s) @ quaternion_to_matrix(delta_r), delta_t], [0 0 0 1]]         (:30-38)
    adjustbox2world() = box2world @ d_box2world                                                  (:40-42)

optimised per instance with torch.optim.Adam([delta_r, delta_s, delta_t], lr=boxmodel_lr) (:13; stepped at
train.py:270-274) and pulled back to the identity by regularize() (:44-49).  Same attribute names and methods, so code
written against the reference's class reads the same; what differs is how MANY instances are handled at once:

    adjust_all(box_models)        every in-frame instance's adjustbox2world() in one launch, [n,4,4]; the backward -- one
                                  launch -- deposits the gradients on each model's delta_r / delta_s / delta_t and applies
                                  train.py:199-205 (a NaN in delta_r.grad or delta_s.grad zeroes the three gradients)
    regularize_all(box_models)    regularize() of every model: one launch for the regularizer's gradients, one
                                  multi-tensor Adam launch (vegs_amd.optim.step_many), zero_grad

(vegs_amd/csrc/instances.hip: k_box_fwd / k_box_bwd / k_box_reg, C ABI include/vegs_instances.h.)  There is no CPU path: the
op-by-op ATen statement of the same class -- what the CPU tests pin against the reference's own class
(tests/golden/ref_boxmodel.npz) and what the GPU tests compare the kernels with -- is test infrastructure and lives in
oracle/boxmodel_oracle.py (BoxModelOpByOp).  adjust_all / regularize_all accept such objects beside BoxModels (any object
with adjustbox2world() / regularize() and `fused == False` is simply called), so that a comparison run can mix them in.
"""
import torch

from . import _capi


class BoxModel:
    fused = True

    def __init__(self, box2world, lr=0.005, lambda_reg=0.001, device=None, fused=True):
        """box2world: the annotated pose, 4x4 (obj_box2world, model/boxmodel.py:16-21: [R | T] of the annotation).
        lr / lambda_reg: arguments/__init__.py:116-117 (boxmodel_lr, boxmodel_lambda_reg)."""
        if not fused:
            raise ValueError("vegs_amd.boxmodel.BoxModel is the HIP path; the op-by-op composition is test infrastructure: "
                             "oracle.boxmodel_oracle.BoxModelOpByOp")
        from . import optim
        device = torch.device(device) if device is not None else torch.as_tensor(box2world).device
        self.lr, self.lambda_reg = lr, lambda_reg
        self.delta_r = torch.tensor([1., 0., 0., 0.], device=device, requires_grad=True)
        self.delta_s = torch.tensor([1., 1., 1.], device=device, requires_grad=True)
        self.delta_t = torch.tensor([0., 0., 0.], device=device, requires_grad=True)
        self.optimizer = optim.Adam([self.delta_r, self.delta_s, self.delta_t], lr=lr)
        self.box2world = torch.as_tensor(box2world, dtype=torch.float32).to(device).contiguous()

    def adjustbox2world(self):
        return adjust_all([self])[0]

    def regularize(self, iteration=None):
        return regularize_all([self])

    def get_deltas(self):
        with torch.no_grad():
            r, s, t = (x.detach().cpu() for x in (self.delta_r, self.delta_s, self.delta_t))
            return [torch.linalg.vector_norm(r - torch.tensor([1., 0., 0., 0.])).item(),
                    torch.linalg.vector_norm(s - 1.0).item(), torch.linalg.vector_norm(t).item()]


def _table(box_models):
    return (_capi.VrBoxModel * len(box_models))(*[
        _capi.VrBoxModel(b.box2world.data_ptr(), b.delta_r.data_ptr(), b.delta_s.data_ptr(), b.delta_t.data_ptr())
        for b in box_models])


class _Adjust(torch.autograd.Function):
    """(delta_r_0, delta_s_0, delta_t_0, delta_r_1, ...) -> adjusted [n,4,4]"""

    @staticmethod
    def forward(ctx, box_models, nan_guard, *deltas):
        n = len(box_models)
        dev = deltas[0].device
        for b in box_models:
            for name, t, k in (("delta_r", b.delta_r, 4), ("delta_s", b.delta_s, 3), ("delta_t", b.delta_t, 3),
                               ("box2world", b.box2world, 16)):
                if not t.is_cuda or t.device != dev or t.dtype != torch.float32 or t.numel() != k or not t.is_contiguous():
                    raise ValueError(f"BoxModel.{name} must be a contiguous float32 GPU tensor of {k} values on {dev} "
                                     "(there is no CPU path)")
        out = torch.empty((n, 4, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _capi.load().vr_boxmodel_forward(_table(box_models), n, out.data_ptr(),
                                                  torch.cuda.current_stream(dev).cuda_stream)
        _capi.check(rc)
        ctx.box_models, ctx.nan_guard = box_models, bool(nan_guard)
        return out

    @staticmethod
    def backward(ctx, g):
        box_models = ctx.box_models
        n = len(box_models)
        dev = g.device
        g = g.contiguous()
        buf = torch.empty((n, 10), dtype=torch.float32, device=dev)       # [delta_r | delta_s | delta_t] per instance
        base, grads, items = buf.data_ptr(), [], []
        for i in range(n):
            row = base + 40 * i
            items.append(_capi.VrBoxModelGrads(row, row + 16, row + 28))
            grads += [buf[i, 0:4], buf[i, 4:7], buf[i, 7:10]]
        with torch.cuda.device(dev):
            rc = _capi.load().vr_boxmodel_backward(_table(box_models), (_capi.VrBoxModelGrads * n)(*items), n, g.data_ptr(),
                                                   1 if ctx.nan_guard else 0, torch.cuda.current_stream(dev).cuda_stream)
        _capi.check(rc)
        return (None, None) + tuple(grads)


def adjust_all(box_models, nan_guard=True):
    """adjustbox2world() of every model, [n,4,4] (row i = box_models[i]); differentiable w.r.t. each model's three
    deltas.  nan_guard: apply train.py:199-205 inside the backward."""
    if not box_models:
        raise ValueError("no box models")
    if not all(getattr(b, "fused", False) for b in box_models):       # op-by-op checker objects in the list: called one by one
        return torch.stack([b.adjustbox2world() for b in box_models])
    flat = []
    for b in box_models:
        flat += [b.delta_r, b.delta_s, b.delta_t]
    return _Adjust.apply(list(box_models), nan_guard, *flat)


@torch.no_grad()
def regularize_all(box_models):
    """BoxModel.regularize (model/boxmodel.py:44-49) of every model: loss = lambda_reg (|delta_r - (1,0,0,0)| +
    |delta_s - 1| + |delta_t|), backward, optimizer.step(), zero_grad -- as two launches for all models."""
    if not box_models:
        return
    if not all(getattr(b, "fused", False) for b in box_models):
        for b in box_models:
            with torch.enable_grad():
                b.regularize()
        return
    from . import optim
    dev = box_models[0].delta_r.device
    by_lambda = {}
    for b in box_models:
        by_lambda.setdefault(float(b.lambda_reg), []).append(b)
    for lam, group in by_lambda.items():
        n = len(group)
        buf = torch.empty((n, 10), dtype=torch.float32, device=dev)
        items = []
        for i, b in enumerate(group):
            row = buf.data_ptr() + 40 * i
            items.append(_capi.VrBoxModelGrads(row, row + 16, row + 28))
            # (the reference's regularize() ADDS to whatever .grad holds; train.py:272-274 has zeroed it just before)
            b.delta_r.grad, b.delta_s.grad, b.delta_t.grad = buf[i, 0:4], buf[i, 4:7], buf[i, 7:10]
        with torch.cuda.device(dev):
            rc = _capi.load().vr_boxmodel_regularizer_grad(_table(group), (_capi.VrBoxModelGrads * n)(*items), n, lam,
                                                           torch.cuda.current_stream(dev).cuda_stream)
        _capi.check(rc)
    optim.step_many([b.optimizer for b in box_models])
    for b in box_models:
        b.optimizer.zero_grad(set_to_none=True)
