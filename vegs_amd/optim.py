"""Host-side mirror of the per-Gaussian update that follows loss.backward() in a VEGS iteration, computed
by fused HIP kernels (vegs_amd/csrc/optim.hip, C ABI include/vegs_optim.h):

  Adam(params, lr, betas, eps)          drop-in for torch.optim.Adam as scene/gaussian_model.py:159-168 builds it
                                        (six named groups, eps=1e-15) and train.py:319-320 steps it
  add_densification_stats(...)          scene/gaussian_model.py:411-413 + the max_radii2D line of train.py:299

`Adam` is a torch.optim.Optimizer: param_groups (with the reference's extra "name" keys) and the per-parameter
state dict {"step", "exp_avg", "exp_avg_sq"} have torch's layout, because the reference's densification code
edits them in place (scene/gaussian_model.py:263-331: _prune_optimizer, cat_tensors_to_optimizer,
replace_tensor_to_optimizer) and checkpoints them with state_dict().  One kernel launch updates all groups.
GPU tensors only; there is no CPU path.
"""
import ctypes as C

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
        lib = _capi.load()
        by_cfg = {}
        keep = []                                            # keeps contiguous gradient copies alive until launch
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise ValueError("fused Adam expects GPU parameters (there is no CPU path)")
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                    raise ValueError("fused Adam expects dense float32 parameters and gradients")
                if not p.is_contiguous():
                    raise ValueError("fused Adam expects contiguous parameters")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if m.shape != p.shape or v.shape != p.shape or not m.is_contiguous() or not v.is_contiguous():
                    raise ValueError("optimizer state does not match its parameter")
                keep.append(g)
                by_cfg.setdefault((p.device, float(b1), float(b2), float(group["eps"])), []).append(
                    _capi.VrAdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                       float(group["lr"]), int(st["step"].item())))
        for (device, b1, b2, eps), items in by_cfg.items():
            arr = (_capi.VrAdamTensor * len(items))(*items)
            with torch.cuda.device(device):
                rc = lib.vr_adam_step(arr, len(items), b1, b2, eps, torch.cuda.current_stream(device).cuda_stream)
            _capi.check(rc)
        return loss


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
    return P, n, means3D.contiguous(), campos.contiguous(), factors.contiguous()


def sh_grad_from_factors(means3D, campos, factors, sh_degree, M, scale=1.0, split=False):
    """Dense dL/dshs from the factors of `n_views` views: scale * sum_v basis(dir(means3D, campos[v])) x factors[v].
    Returns [P,M,3], or (dc [P,1,3], rest [P,M-1,3]) with split=True (the model's own storage)."""
    P, n, means3D, campos, factors = _check_factor_inputs(means3D, campos, factors)
    dev = means3D.device
    if split:
        out = torch.empty((P, 1, 3), dtype=torch.float32, device=dev)
        rest = torch.empty((P, M - 1, 3), dtype=torch.float32, device=dev)
    else:
        out, rest = torch.empty((P, M, 3), dtype=torch.float32, device=dev), None
    with torch.cuda.device(dev):
        rc = _capi.load().vr_sh_grad_from_factors(_capi.ptr(means3D), P, _capi.ptr(campos), _capi.ptr(factors), n,
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
    P, n, means3D, campos, factors = _check_factor_inputs(means3D, campos, factors)
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
        rc = _capi.load().vr_sh_adam_step(_capi.ptr(means3D), P, _capi.ptr(campos), _capi.ptr(factors), n, int(sh_degree),
                                          int(M), float(scale), C.byref(items[0]),
                                          C.byref(items[1]) if items[1] is not None else None, betas[0], betas[1], eps,
                                          torch.cuda.current_stream(dev).cuda_stream)
    _capi.check(rc)
    for st in states:          # the step counters advance only once every check and the call itself have passed
        st["step"] += 1
