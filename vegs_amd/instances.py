"""Fused counterpart of the reference's per-instance `prepare_rasterization(..., box2world=...)` +
`merge_kwargs` (gaussian_renderer/__init__.py:121-186) as render_all / render_dyn use them (:188-333):
ONE HIP launch carries the means / scales / rotations of every box instance into the world frame and writes
them, together with the static model's, straight into the concatenated op inputs; one launch (+ a tiny
reduction) produces all gradients including dL/d(box2world) (vegs_amd/csrc/instances.hip, C ABI
include/vegs_instances.h).  shs / opacities need no arithmetic and are concatenated by torch.cat as in the
reference.  GPU tensors only; there is no CPU path.
"""
import ctypes as C

import torch

from . import _capi


def _check(name, t, cols, device):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name} must be a GPU tensor (there is no CPU path)")
    if t.device != device or t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != cols:
        raise ValueError(f"{name} must be float32 [n,{cols}] on {device} (got {t.dtype} {tuple(t.shape)} on {t.device})")


class _TransformConcat(torch.autograd.Function):
    """inputs: (box2world_all | None, means_0, scales_0, rot_0, box2world_0 | row index | None, means_1, ...)
    -> (means, scales, rotations).  With `box2world_all` [n,4,4] (what vegs_amd.boxmodel.adjust_all returns) an instance
    names its pose by ROW INDEX: one gradient tensor for all poses instead of a select/accumulate chain per instance."""

    @staticmethod
    def forward(ctx, b_all, *flat):
        lib = _capi.load()
        n_inst = len(flat) // 4
        device = flat[0].device
        if b_all is not None:
            if not b_all.is_cuda or b_all.dtype != torch.float32 or b_all.dim() != 3 or tuple(b_all.shape[1:]) != (4, 4):
                raise ValueError("batched box2world must be a float32 GPU tensor [n,4,4]")
            b_all = b_all.contiguous()
        items, offset, keep = [], 0, []
        for i in range(n_inst):
            m, s, r, b = flat[4 * i:4 * i + 4]
            if isinstance(b, int):
                if b_all is None or not 0 <= b < b_all.shape[0]:
                    raise ValueError("box2world row index without (or outside) the batched box2world tensor")
                b = b_all[b]
            _check("means3D", m, 3, device); _check("scales", s, 3, device); _check("rotations", r, 4, device)
            if s.shape[0] != m.shape[0] or r.shape[0] != m.shape[0]:
                raise ValueError("means3D, scales and rotations of an instance must have the same number of rows")
            m, s, r = m.contiguous(), s.contiguous(), r.contiguous()
            if b is not None:
                if not b.is_cuda or b.shape != (4, 4) or b.dtype != torch.float32:
                    raise ValueError("box2world must be a float32 GPU tensor of shape (4, 4)")
                b = b.contiguous()
            keep.append((m, s, r, b))
            items.append(_capi.VrInstance(_capi.ptr(m), _capi.ptr(s), _capi.ptr(r), None if b is None else b.data_ptr(),
                                          m.shape[0], offset))
            offset += m.shape[0]
        out_m = torch.empty((offset, 3), dtype=torch.float32, device=device)
        out_s = torch.empty((offset, 3), dtype=torch.float32, device=device)
        out_r = torch.empty((offset, 4), dtype=torch.float32, device=device)
        arr = (_capi.VrInstance * n_inst)(*items)
        with torch.cuda.device(device):
            rc = lib.vr_instances_forward(arr, n_inst, _capi.ptr(out_m), _capi.ptr(out_s), _capi.ptr(out_r),
                                          torch.cuda.current_stream(device).cuda_stream)
        _capi.check(rc)
        ctx.keep = keep
        ctx.device = device
        ctx.b_all = b_all
        ctx.b_rows = [b if isinstance(b, int) else None for b in flat[3::4]]
        return out_m, out_s, out_r

    @staticmethod
    def backward(ctx, g_m, g_s, g_r):
        lib = _capi.load()
        device = ctx.device
        g_m, g_s, g_r = g_m.contiguous(), g_s.contiguous(), g_r.contiguous()
        items, gitems, grads, offset = [], [], [], 0
        # (rows of the batched pose tensor that no instance uses get a zero gradient)
        db_all = None if ctx.b_all is None else torch.zeros_like(ctx.b_all)
        for (m, s, r, b), row in zip(ctx.keep, ctx.b_rows):
            n = m.shape[0]
            if b is None:                       # static model: its gradients ARE rows of the concatenated gradients
                grads += [g_m[offset:offset + n], g_s[offset:offset + n], g_r[offset:offset + n], None]
                gi = _capi.VrInstanceGrads(None, None, None, None)
            else:
                dm, ds, dr = torch.empty_like(m), torch.empty_like(s), torch.empty_like(r)
                if row is None:
                    db = torch.empty((4, 4), dtype=torch.float32, device=device)
                    grads += [dm, ds, dr, db]
                else:
                    db = db_all[row]
                    grads += [dm, ds, dr, None]
                gi = _capi.VrInstanceGrads(_capi.ptr(dm), _capi.ptr(ds), _capi.ptr(dr), db.data_ptr())
            items.append(_capi.VrInstance(_capi.ptr(m), _capi.ptr(s), _capi.ptr(r), None if b is None else b.data_ptr(),
                                          n, offset))
            gitems.append(gi)
            offset += n
        arr = (_capi.VrInstance * len(items))(*items)
        garr = (_capi.VrInstanceGrads * len(items))(*gitems)
        arena = _capi.Arena(device)
        cb = arena.callback()
        with torch.cuda.device(device):
            rc = lib.vr_instances_backward(arr, garr, len(items), _capi.ptr(g_m), _capi.ptr(g_s), _capi.ptr(g_r), cb, None,
                                           torch.cuda.current_stream(device).cuda_stream)
        del cb
        arena.release_scratch()
        if arena.error is not None:
            raise arena.error
        _capi.check(rc)
        return (db_all,) + tuple(grads)


def prepare_and_merge(static, boxes, box2worlds):
    """Op inputs for a frame with dynamic instances: the result of
        kw = prepare_rasterization(static); for each box: kw = merge_kwargs(kw, prepare_rasterization(box, box2world))
    (gaussian_renderer/__init__.py:274-303).  `static` / `boxes[i]`: dicts with means3D, shs, opacities, scales,
    rotations; `box2worlds[i]`: differentiable 4x4 tensors -- or ONE tensor [n,4,4] for all instances (what
    vegs_amd.boxmodel.adjust_all returns: a single gradient tensor for all poses).  `static` may be None (render_dyn)."""
    if len(boxes) != len(box2worlds):
        raise ValueError("one box2world per box instance")
    b_all = box2worlds if isinstance(box2worlds, torch.Tensor) else None
    poses = list(range(len(boxes))) if b_all is not None else list(box2worlds)
    models = ([] if static is None else [(static, None)]) + list(zip(boxes, poses))
    if not models:
        raise ValueError("nothing to render")
    flat = []
    for t, b in models:
        flat += [t["means3D"], t["scales"], t["rotations"], b]
    means, scales, rotations = _TransformConcat.apply(b_all, *flat)
    cat = (lambda k: models[0][0][k]) if len(models) == 1 else (lambda k: torch.cat([t[k] for t, _ in models], 0))
    # SH: the instances' rows go behind the static model's as an SH TAIL (the rasterizer reads the static model's tensor(s)
    # where they are: whole [P0,M,3] or the pair (features_dc, features_rest)); merge_kwargs' torch.cat would copy the whole
    # static model to append a few thousand rows -- 2 x 0.96 GB per view at 5 M Gaussians, forward and backward
    if static is not None and boxes:
        head = static["shs"]
        dc, rest = head if isinstance(head, (tuple, list)) else (head, None)
        M = dc.shape[1] + (rest.shape[1] if rest is not None else 0)
        tail = torch.cat([b["shs"] for b in boxes], 0) if len(boxes) > 1 else boxes[0]["shs"]
        if (3 * M) % 4 == 0 and M <= 16 and tail.shape[1] == M:
            shs = (dc, rest, tail)
        else:   # the kernels' tail path needs whole float4 rows: fall back to one concatenated tensor
            whole = dc if rest is None else torch.cat((dc, rest), 1)
            shs = torch.cat((whole, tail), 0)
    else:
        shs = cat("shs")
        if isinstance(models[0][0]["shs"], (tuple, list)) and len(models) > 1:
            raise ValueError("split SH storage is supported for the static model only")
    return {"means3D": means, "shs": shs, "opacities": cat("opacities"), "scales": scales, "rotations": rotations}


class _Activate(torch.autograd.Function):
    """(_opacity [P,1], _scaling [P,3], _rotation [P,4]) -> (sigmoid, exp, F.normalize) in one launch each way."""

    @staticmethod
    def forward(ctx, raw_opacity, raw_scaling, raw_rotation):
        lib = _capi.load()
        device = raw_rotation.device
        P = raw_rotation.shape[0]
        _check("_scaling", raw_scaling, 3, device); _check("_rotation", raw_rotation, 4, device)
        if not raw_opacity.is_cuda or raw_opacity.dtype != torch.float32 or raw_opacity.numel() != P or raw_scaling.shape[0] != P:
            raise ValueError("_opacity must be a float32 GPU tensor with one value per Gaussian, like _scaling and _rotation")
        ro, rs, rr = raw_opacity.contiguous(), raw_scaling.contiguous(), raw_rotation.contiguous()
        opacity, scales, rot = torch.empty_like(ro), torch.empty_like(rs), torch.empty_like(rr)
        with torch.cuda.device(device):
            rc = lib.vr_activations_forward(_capi.ptr(ro), _capi.ptr(rs), _capi.ptr(rr), P, _capi.ptr(opacity),
                                            _capi.ptr(scales), _capi.ptr(rot), torch.cuda.current_stream(device).cuda_stream)
        _capi.check(rc)
        ctx.save_for_backward(opacity, scales, rr)
        ctx.set_materialize_grads(False)
        return opacity, scales, rot

    @staticmethod
    def backward(ctx, g_o, g_s, g_r):
        lib = _capi.load()
        opacity, scales, rr = ctx.saved_tensors
        device = rr.device
        need = ctx.needs_input_grad
        g_o, g_s, g_r = (None if g is None else g.contiguous() for g in (g_o, g_s, g_r))
        d_o = torch.empty_like(opacity) if need[0] else None
        d_s = torch.empty_like(scales) if need[1] else None
        d_r = torch.empty_like(rr) if need[2] else None
        with torch.cuda.device(device):
            rc = lib.vr_activations_backward(_capi.ptr(opacity), _capi.ptr(scales), _capi.ptr(rr), rr.shape[0],
                                             _capi.ptr(g_o), _capi.ptr(g_s), _capi.ptr(g_r), _capi.ptr(d_o), _capi.ptr(d_s),
                                             _capi.ptr(d_r), torch.cuda.current_stream(device).cuda_stream)
        _capi.check(rc)
        return d_o, d_s, d_r


def activate(raw_opacity, raw_scaling, raw_rotation):
    """The model's three activations (scene/gaussian_model.py:98-120: get_opacity, get_scaling, get_rotation) for
    every Gaussian in one launch; gradients to the raw parameters in one launch.  Returns (opacity, scales, rotations)
    shaped like the inputs."""
    return _Activate.apply(raw_opacity, raw_scaling, raw_rotation)
