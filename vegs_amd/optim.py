"""Host-side mirror of the per-Gaussian update that follows loss.backward() in a VEGS iteration, computed
by fused HIP kernels (vegs_amd/csrc/optim.hip, C ABI include/vegs_optim.h):

  Adam(params, lr, betas, eps)          drop-in for torch.optim.Adam as scene/gaussian_model.py:159-168 builds it
                                        (six named groups, eps=1e-15) and train.py:319-320 steps it
  add_densification_stats(...)          scene/gaussian_model.py:411-413 + the max_radii2D line of train.py:299
  densify_and_prune(optimizer, ...)     scene/gaussian_model.py:384-403 with its clone / split / prune and the
                                        optimizer-state surgery (:278-351), as one planned gather
  reset_opacity(optimizer)              scene/gaussian_model.py:215-218

`Adam` is a torch.optim.Optimizer: param_groups (with the reference's extra "name" keys) and the per-parameter
state dict {"step", "exp_avg", "exp_avg_sq"} have torch's layout, because the reference's densification code
edits them in place (scene/gaussian_model.py:263-331: _prune_optimizer, cat_tensors_to_optimizer,
replace_tensor_to_optimizer) and checkpoints them with state_dict().  One kernel launch updates all groups.
GPU tensors only; there is no CPU path.
"""
import ctypes as C
import math

import torch

from . import _capi


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("the fused Adam covers the reference's configuration: no weight decay, no amsgrad")
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        step_many([self])
        return loss


def _validate(optimizer):
    """Everything _collect would raise for, checked WITHOUT touching the optimizer: step_many validates every optimizer of
    its batch before the first step counter moves, so an exception leaves all of them exactly as they were (a half-advanced
    batch would skew the bias correction of a retry and desynchronise the ranks of a distributed job)."""
    for group in optimizer.param_groups:
        if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
            raise NotImplementedError("the fused Adam covers the reference's configuration: no weight decay, no amsgrad")
        for p in group["params"]:
            if p.grad is None:
                continue
            if not p.is_cuda:
                raise ValueError("fused Adam expects GPU parameters (there is no CPU path)")
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                raise ValueError("fused Adam expects dense float32 parameters and gradients")
            if not p.is_contiguous():
                raise ValueError("fused Adam expects contiguous parameters")
            st = optimizer.state.get(p) or {}
            if len(st):
                m, v = st.get("exp_avg"), st.get("exp_avg_sq")
                if (m is None or v is None or "step" not in st or m.shape != p.shape or v.shape != p.shape
                        or not m.is_contiguous() or not v.is_contiguous()):
                    raise ValueError("optimizer state does not match its parameter")


def _collect(optimizer, by_cfg, keep):
    """The VrAdamTensor entries of one VALIDATED optimizer (vegs_amd.optim.Adam or torch.optim.Adam: same state layout), its
    step counters advanced as torch.optim.Adam.step does; parameters without a gradient are skipped."""
    for group in optimizer.param_groups:
        b1, b2 = group["betas"]
        for p in group["params"]:
            if p.grad is None:
                continue
            st = optimizer.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            m, v = st["exp_avg"], st["exp_avg_sq"]
            st["step"] += 1
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            keep.append(g)
            by_cfg.setdefault((p.device, float(b1), float(b2)), []).append(
                _capi.VrAdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                   float(group["lr"]), int(st["step"].item()), float(group["eps"])))


@torch.no_grad()
def step_many(optimizers):
    """optimizer.step() of SEVERAL Adam optimizers as one launch (per 64 tensors): what train.py:254-275 does one optimizer
    after the other when dynamic objects are in frame -- the static model's six groups, six more per instance model
    (densification_and_optimization(..., box=True)) and the three pose corrections of every BoxModel
    (box_model.optimizer.step(), model/boxmodel.py:13) -- with every tensor's own learning rate, step count and eps in
    the kernel's block->tensor table.  Optimizers may be vegs_amd.optim.Adam or torch.optim.Adam (same state layout).
    All-or-nothing: every optimizer is validated before any state (step counters, fresh moments) is touched."""
    lib = _capi.load()
    optimizers = list(optimizers)
    for opt in optimizers:
        _validate(opt)
    by_cfg, keep = {}, []
    for opt in optimizers:
        _collect(opt, by_cfg, keep)
    for (device, b1, b2), items in by_cfg.items():
        arr = (_capi.VrAdamTensor * len(items))(*items)
        with torch.cuda.device(device):
            rc = lib.vr_adam_step(arr, len(items), b1, b2, -1.0, torch.cuda.current_stream(device).cuda_stream)
        _capi.check(rc)


def add_densification_stats(viewspace_point_grad, radii, xyz_gradient_accum, denom, max_radii2D):
    """In place, for every Gaussian with radii > 0 (the reference's visibility_filter, gaussian_renderer/__init__.py:117):
    xyz_gradient_accum += ||viewspace_point_grad[:, :2]||, denom += 1 (scene/gaussian_model.py:411-413) and
    max_radii2D = max(max_radii2D, radii) (train.py:299)."""
    P = radii.shape[0]
    for name, t, shape, dt in (("viewspace_point_grad", viewspace_point_grad, (P, 3), torch.float32),
                               ("radii", radii, (P,), torch.int32),
                               ("xyz_gradient_accum", xyz_gradient_accum, (P, 1), torch.float32),
                               ("denom", denom, (P, 1), torch.float32), ("max_radii2D", max_radii2D, (P,), torch.float32)):
        if not t.is_cuda:
            raise ValueError(f"{name} must be a GPU tensor (there is no CPU path)")
        if tuple(t.shape) != shape or t.dtype != dt or not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous {dt} {shape} (got {t.dtype} {tuple(t.shape)})")
    lib = _capi.load()
    with torch.cuda.device(radii.device):
        rc = lib.vr_densify_stats(_capi.ptr(viewspace_point_grad), _capi.ptr(radii), P, _capi.ptr(xyz_gradient_accum),
                                  _capi.ptr(denom), _capi.ptr(max_radii2D),
                                  torch.cuda.current_stream(radii.device).cuda_stream)
    _capi.check(rc)


# ---------------------------------------------------------------------------------------------------------------------
# Densification (csrc/densify.hip).  The model's tensors are found the way the reference finds them: by the "name" of their
# optimizer group (scene/gaussian_model.py:159-166: xyz, f_dc, f_rest, opacity, scaling, rotation; one tensor per group).
DENSIFY_NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
_ROLES = {"xyz": 1, "scaling": 2}


def _named_groups(optimizer):
    groups = {}
    for group in optimizer.param_groups:
        if len(group["params"]) != 1:
            raise ValueError("densification expects one tensor per optimizer group (scene/gaussian_model.py:313)")
        groups[group.get("name")] = group
    missing = [n for n in DENSIFY_NAMES if n not in groups]
    if missing:
        raise ValueError(f"optimizer groups named {missing} are missing (scene/gaussian_model.py:159-166)")
    return groups


def _model_tensor(name, p, P, width=None):
    if not p.is_cuda:
        raise ValueError(f"{name} must be a GPU tensor (there is no CPU path)")
    if p.dtype != torch.float32 or p.dim() < 2 or p.shape[0] != P or not p.is_contiguous():
        raise ValueError(f"{name} must be contiguous float32 [{P}, ...] (got {p.dtype} {tuple(p.shape)})")
    if width is not None and math.prod(p.shape[1:]) != width:
        raise ValueError(f"{name} must have {width} values per Gaussian (got {tuple(p.shape)})")


def densify_and_prune(optimizer, xyz_gradient_accum, denom, max_grad, min_opacity, extent, max_screen_size,
                      percent_dense, noise=None, generator=None, empty_cache=True, prune=True):
    """GaussianModel.densify_and_prune (scene/gaussian_model.py:384-403) on the tensors of `optimizer`'s six named groups:
    clone the small Gaussians whose mean screen-space gradient `xyz_gradient_accum / denom` reaches `max_grad`, replace
    the large ones by two samples, prune by opacity (and, if `max_screen_size` is truthy, by world size -- the
    reference's screen-size test never fires, see include/vegs_optim.h), carrying the Adam moments of the survivors
    along and starting the new rows' at zero.  As in the reference the optimizer ends up with NEW nn.Parameter objects
    (state re-keyed, "step" untouched); they are returned as {name: parameter} like its `optimizable_tensors`, together
    with the fresh statistics (xyz_gradient_accum, denom, max_radii2D) = zeros of the new length (:349-351).

    One planned gather instead of the reference's two concatenations and two mask passes per tensor: the result has
    the same rows in the same order.  `noise` [2 S, 3] is the unit-normal draw behind torch.normal (:367), S = number
    of split Gaussians; drawn here with `generator` when not given -- pass it to make the step reproducible.
    `empty_cache`: finish with torch.cuda.empty_cache() as the reference does (:403).  `prune=False` is the reference's
    argument of that name (:384, :397): clone and split only, nothing is pruned by opacity or size."""
    groups = _named_groups(optimizer)
    par = {n: groups[n]["params"][0] for n in DENSIFY_NAMES}
    P = par["xyz"].shape[0]
    dev = par["xyz"].device
    for n, w in (("xyz", 3), ("opacity", 1), ("scaling", 3), ("rotation", 4), ("f_dc", None), ("f_rest", None)):
        _model_tensor(n, par[n], P, w)
    for n, t in (("xyz_gradient_accum", xyz_gradient_accum), ("denom", denom)):
        if not t.is_cuda or t.dtype != torch.float32 or t.numel() != P or not t.is_contiguous():
            raise ValueError(f"{n} must be a contiguous float32 GPU tensor with one value per Gaussian")
    lib = _capi.load()
    settings = _capi.VrDensifySettings(float(max_grad), float(min_opacity), float(extent), float(percent_dense),
                                       2 if not prune else (1 if max_screen_size else 0))
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev), torch.no_grad():
        plan = torch.empty(max(int(lib.vr_densify_plan_words(P)), 1), dtype=torch.int32, device=dev)
        counts = torch.empty(8, dtype=torch.int32, device=dev)
        _capi.check(lib.vr_densify_plan(_capi.ptr(par["opacity"]), _capi.ptr(par["scaling"]), _capi.ptr(xyz_gradient_accum),
                                        _capi.ptr(denom), P, C.byref(settings), plan.data_ptr(), counts.data_ptr(), stream))
        n_out, _, _, _, n_split = (int(c) for c in counts[:5].tolist())       # the one host round trip of the step
        if noise is None:
            noise = torch.randn((2 * n_split, 3), dtype=torch.float32, device=dev, generator=generator)
        if not noise.is_cuda or noise.dtype != torch.float32 or tuple(noise.shape) != (2 * n_split, 3):
            raise ValueError(f"noise must be a float32 GPU tensor [{2 * n_split}, 3] (two samples per split Gaussian)")
        noise = noise.contiguous()
        new, moments, items = {}, {}, []
        for n in DENSIFY_NAMES:
            p = par[n]
            st = optimizer.state.get(p, None)
            have = st is not None and "exp_avg" in st
            new[n] = torch.empty((n_out,) + tuple(p.shape[1:]), dtype=torch.float32, device=dev)
            m_src = v_src = m_dst = v_dst = None
            if have:
                m_src, v_src = st["exp_avg"], st["exp_avg_sq"]
                if m_src.shape != p.shape or v_src.shape != p.shape or not m_src.is_contiguous() or not v_src.is_contiguous():
                    raise ValueError(f"optimizer state of {n} does not match its parameter")
                m_dst, v_dst = torch.empty_like(new[n]), torch.empty_like(new[n])
                moments[n] = (m_dst, v_dst)
            if math.prod(p.shape[1:]) == 0:      # f_rest of a degree-0 model is [P,0,3]: nothing to move
                continue
            items.append(_capi.VrDensifyTensor(_capi.ptr(p), _capi.ptr(new[n]), _capi.ptr(m_src), _capi.ptr(m_dst),
                                               _capi.ptr(v_src), _capi.ptr(v_dst), math.prod(p.shape[1:]), _ROLES.get(n, 0)))
        if n_out > 0 and items:
            arr = (_capi.VrDensifyTensor * len(items))(*items)
            _capi.check(lib.vr_densify_apply(plan.data_ptr(), n_out, n_split, arr, len(items), _capi.ptr(par["scaling"]),
                                             _capi.ptr(par["rotation"]), _capi.ptr(noise), stream))
    out = {}
    for n in DENSIFY_NAMES:                      # _prune_optimizer / cat_tensors_to_optimizer's re-keying (:278-331)
        group, old = groups[n], par[n]
        stored = optimizer.state.get(old, None)
        param = torch.nn.Parameter(new[n].requires_grad_(True))
        if stored is not None:
            if n in moments:
                stored["exp_avg"], stored["exp_avg_sq"] = moments[n]
            del optimizer.state[old]
            optimizer.state[param] = stored
        group["params"][0] = param
        out[n] = param
    stats = (torch.zeros((n_out, 1), dtype=torch.float32, device=dev), torch.zeros((n_out, 1), dtype=torch.float32, device=dev),
             torch.zeros((n_out,), dtype=torch.float32, device=dev))
    if empty_cache:
        # as the reference does at the end of the step (:403): every per-Gaussian buffer changes size now, and the caching
        # allocator would otherwise keep the old generation's blocks (a 3000-iteration soak: 10 GB reserved for 0.6 GB in use)
        del new, moments, items, plan, counts, par
        torch.cuda.empty_cache()
    return out, stats


def reset_opacity(optimizer, cap=0.01):
    """GaussianModel.reset_opacity (scene/gaussian_model.py:215-218): opacity = inverse_sigmoid(min(sigmoid(opacity), 0.01))
    as a NEW nn.Parameter of the "opacity" group with zeroed Adam moments (replace_tensor_to_optimizer, :263-276); one
    launch.  Returns the parameter."""
    group = None
    for g in optimizer.param_groups:
        if g.get("name") == "opacity":
            group = g
    if group is None or len(group["params"]) != 1:
        raise ValueError('the optimizer has no single-tensor group named "opacity"')
    old = group["params"][0]
    if not old.is_cuda or old.dtype != torch.float32 or not old.is_contiguous():
        raise ValueError("opacity must be a contiguous float32 GPU tensor (there is no CPU path)")
    stored = optimizer.state.get(old, None)
    with torch.no_grad():
        new = old.detach().clone()
        m = v = None
        if stored is not None and "exp_avg" in stored:
            m, v = torch.empty_like(new), torch.empty_like(new)
        with torch.cuda.device(old.device):
            _capi.check(_capi.load().vr_reset_opacity(_capi.ptr(new), _capi.ptr(m), _capi.ptr(v), new.numel(), float(cap),
                                                      torch.cuda.current_stream(old.device).cuda_stream))
    param = torch.nn.Parameter(new.requires_grad_(True))
    if stored is not None:
        if m is not None:
            stored["exp_avg"], stored["exp_avg_sq"] = m, v
        del optimizer.state[old]
        optimizer.state[param] = stored
    group["params"][0] = param
    return param


# ---------------------------------------------------------------------------------------------------------------------
# Factored SH gradients.  The rasterizer can return, instead of dL/dshs [P,M,3], its rank-1 factor: the clamp-masked
# dL/d(colour) [P,3] of the view (GaussianRasterizer(..., sh_color_grad=sink)); dL/dshs[i,k,c] = basis_k(dir_i) *
# factor[i,c].  A view-sharded job all-gathers those 3 floats per Gaussian and view (vegs_amd.dist.exchange_factored)
# and every rank rebuilds -- or, better, consumes -- the sum over views locally.

def _check_factor_inputs(means3D, campos, factors):
    if not (means3D.is_cuda and campos.is_cuda and factors.is_cuda):
        raise ValueError("means3D, campos and factors must be GPU tensors (there is no CPU path)")
    P = means3D.shape[0]
    if means3D.dtype != torch.float32 or tuple(means3D.shape) != (P, 3):
        raise ValueError("means3D must be float32 [P,3]")
    if campos.dtype != torch.float32 or campos.dim() != 2 or campos.shape[1] != 3:
        raise ValueError("campos must be float32 [n_views,3]")
    n = campos.shape[0]
    if factors.dtype != torch.float32 or tuple(factors.shape) != (n, P, 3):
        raise ValueError(f"factors must be float32 [{n},{P},3] (got {tuple(factors.shape)})")
    # the P rows may be the HEAD of longer per-view blocks (the static model's rows of an all-gathered
    # [n_views, P + instance rows, 3]): passed with its view stride instead of a copy
    if not (factors.stride(2) == 1 and factors.stride(1) == 3 and (n == 1 or factors.stride(0) >= 3 * P)):
        factors = factors.contiguous()
    stride = factors.stride(0) if n > 1 else 0
    return P, n, means3D.contiguous(), campos.contiguous(), factors, stride


def sh_grad_from_factors(means3D, campos, factors, sh_degree, M, scale=1.0, split=False):
    """Dense dL/dshs from the factors of `n_views` views: scale * sum_v basis(dir(means3D, campos[v])) x factors[v].
    Returns [P,M,3], or (dc [P,1,3], rest [P,M-1,3]) with split=True (the model's own storage)."""
    P, n, means3D, campos, factors, fstride = _check_factor_inputs(means3D, campos, factors)
    dev = means3D.device
    if split:
        out = torch.empty((P, 1, 3), dtype=torch.float32, device=dev)
        rest = torch.empty((P, M - 1, 3), dtype=torch.float32, device=dev)
    else:
        out, rest = torch.empty((P, M, 3), dtype=torch.float32, device=dev), None
    with torch.cuda.device(dev):
        rc = _capi.load().vr_sh_grad_from_factors(_capi.ptr(means3D), P, _capi.ptr(campos), _capi.ptr(factors), n, fstride,
                                                  int(sh_degree), int(M), float(scale), _capi.ptr(out), _capi.ptr(rest),
                                                  torch.cuda.current_stream(dev).cuda_stream)
    _capi.check(rc)
    return (out, rest) if split else out


def _sh_state(opt, p):
    for group in opt.param_groups:
        for q in group["params"]:
            if q is p:
                st = opt.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                # the same checks torch.optim.Adam's step makes: densification code replaces these tensors in place
                # (scene/gaussian_model.py:263-331) and a mismatch would be written out of bounds by the kernel
                for name in ("exp_avg", "exp_avg_sq"):
                    t = st[name]
                    if t.shape != p.shape or t.dtype != torch.float32 or not t.is_cuda or t.device != p.device \
                            or not t.is_contiguous():
                        raise ValueError(f"optimizer state {name} must be a contiguous float32 tensor shaped like its "
                                         f"parameter {tuple(p.shape)} on {p.device} (got {tuple(t.shape)}, {t.dtype}, {t.device})")
                return group, st
    raise ValueError("parameter is not in the optimizer")


@torch.no_grad()
def adam_step_sh_factored(opt, features_dc, features_rest, means3D, campos, factors, sh_degree, scale=1.0):
    """One Adam step of the SH parameters (features_dc [P,1,3] + features_rest [P,M-1,3], or the whole [P,M,3] tensor with
    features_rest=None) of optimizer `opt` (vegs_amd.optim.Adam or torch.optim.Adam: same state layout) straight from
    the factors -- the dense gradient is built per Gaussian in registers and never written.  Equivalent to setting
    .grad = sh_grad_from_factors(...) on the two parameters and stepping only them."""
    P, n, means3D, campos, factors, fstride = _check_factor_inputs(means3D, campos, factors)
    M = features_dc.shape[1] + (features_rest.shape[1] if features_rest is not None else 0)
    items, states, betas, eps = [], [], None, None
    for p in (features_dc, features_rest):
        if p is None:
            items.append(None)
            continue
        if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
            raise ValueError("SH parameters must be contiguous float32 GPU tensors")
        group, st = _sh_state(opt, p)
        if betas is None:
            betas, eps = tuple(group["betas"]), float(group["eps"])
        elif betas != tuple(group["betas"]) or eps != float(group["eps"]):
            raise ValueError("features_dc and features_rest must share betas and eps")
        items.append(_capi.VrShAdamTensor(p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                          float(group["lr"]), int(st["step"].item()) + 1))
        states.append(st)
    K = (int(sh_degree) + 1) ** 2
    if not 0 <= int(sh_degree) <= 3 or M < K:
        raise ValueError(f"sh_degree {sh_degree} needs {K} coefficients, the parameters hold {M}")
    dev = means3D.device
    with torch.cuda.device(dev):
        rc = _capi.load().vr_sh_adam_step(_capi.ptr(means3D), P, _capi.ptr(campos), _capi.ptr(factors), n, fstride, int(sh_degree),
                                          int(M), float(scale), C.byref(items[0]),
                                          C.byref(items[1]) if items[1] is not None else None, betas[0], betas[1], eps,
                                          torch.cuda.current_stream(dev).cuda_stream)
    _capi.check(rc)
    for st in states:          # the step counters advance only once every check and the call itself have passed
        st["step"] += 1
