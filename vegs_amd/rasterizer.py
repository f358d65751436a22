"""Python operator surface of the rasterizer -- the drop-in for the reference's
`diff_gaussian_rasterization` package (un-vendored submodule, reference .gitmodules:7-9).

Same names, arguments and return values as the reference's call sites use:
  * GaussianRasterizationSettings(...)  12 keyword fields   gaussian_renderer/__init__.py:38-51
  * GaussianRasterizer(raster_settings=...)(means3D=, means2D=, shs=, colors_precomp=, opacities=,
    scales=, rotations=, cov3D_precomp=) -> (color[3,H,W], depth[1,H,W], cov_quat[4,H,W],
    cov_scale[3,H,W], alpha[1,H,W], radii[P] int32)          gaussian_renderer/__init__.py:86-94
  * GaussianRasterizer.markVisible(positions) -> bool[P]     utils/norminit_utils.py:55,179
  * means2D receives the screen-space gradient ([P,3], z = 0) that
    scene/gaussian_model.py:411-413 accumulates for densification.
All computation happens in libvegsrast.so (hand-written HIP, gfx950) through the C ABI of
include/vegs_rast.h; there is no PyTorch/CPU fallback.
"""
import contextlib
import ctypes as C
import os
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _capi


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# ---- switches for the assumptions about the un-vendored fork (include/vegs_rast.h VrFlags) and the
# deterministic backward mode.  GaussianRasterizationSettings keeps the reference's 12 fields, so the flags live
# beside it: process-wide default from the environment (VEGS_RAST_FLAGS=<int>), changed with set_flags() or, for a
# block of code, `with flags(...)`.  A forward captures the flags in force; its backward uses the same ones.
FLAG_SCALE_MODIFIED = _capi.FLAG_SCALE_MODIFIED            # cov_scale blends scale_modifier * scales
FLAG_DEPTH_NORMALIZED = _capi.FLAG_DEPTH_NORMALIZED        # depth = sum(w z) / alpha
FLAG_EXTRA_NO_ALPHA_GRAD = _capi.FLAG_EXTRA_NO_ALPHA_GRAD  # depth/quat/scale: gradients to the attributes only
FLAG_FILL_EMPTY = _capi.FLAG_FILL_EMPTY                    # cov_quat += T_final * (1,0,0,0)
FLAG_DETERMINISTIC = _capi.FLAG_DETERMINISTIC              # backward without atomics (bit-reproducible gradients)
FLAG_SCAN_BINNING = _capi.FLAG_SCAN_BINNING                # binning without inter-workgroup waits (multi-launch passes)
FLAG_ROUNDS_OFF = _capi.FLAG_ROUNDS_OFF                    # forward: every list segment at once, whatever the list density
FLAG_ROUNDS_ON = _capi.FLAG_ROUNDS_ON                      # forward: segment rounds, whatever the list density (default: by density)
FLAG_RAW_PARAMS = _capi.FLAG_RAW_PARAMS                    # opacities / scales / rotations are the model's raw parameters: activated in the kernel
FLAG_FULL_TILE_LISTS = _capi.FLAG_FULL_TILE_LISTS          # tile lists hold the reference's full rectangles (default: tiles a splat cannot reach are left out)
FLAG_VERIFY_BINNING = _capi.FLAG_VERIFY_BINNING            # forward waits for the binning guard; a tripped view is re-binned without waits (opt-in)
FLAG_FAST_EXP = _capi.FLAG_FAST_EXP                        # 2^x by v_exp_f32 in the compositing kernels, forward and backward (images ~1e-6 off the bit-exact mode)
FLAG_ACCUMULATE_GRADS = _capi.FLAG_ACCUMULATE_GRADS        # backward adds into the caller's gradient arrays (set per call by accumulate_grads(), not by hand)
_flags = int(os.environ.get("VEGS_RAST_FLAGS", "0"), 0)


def set_flags(value):
    """Set the process-wide VrFlags; returns the previous value."""
    global _flags
    old, _flags = _flags, int(value)
    return old


def get_flags():
    return _flags


@contextlib.contextmanager
def flags(value):
    old = set_flags(value)
    try:
        yield
    finally:
        set_flags(old)


def _dev_f32(t, device):
    return t.to(device=device, dtype=torch.float32).contiguous()


def _prep(t, name, device, cols=None):
    """None/empty -> None; otherwise a contiguous fp32 tensor on `device` (validated)."""
    if t is None or t.numel() == 0:
        return None
    if not t.is_cuda:
        raise ValueError(f"{name} must be a GPU tensor (the rasterizer has no CPU path)")
    if t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device}")
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32 (got {t.dtype})")
    return t.contiguous()


def _settings_struct(rs, device, keep, flag_bits=0):
    bg = _dev_f32(rs.bg, device)
    view = _dev_f32(rs.viewmatrix, device)
    proj = _dev_f32(rs.projmatrix, device)
    campos = _dev_f32(rs.campos, device)
    if bg.numel() != 3 or view.numel() != 16 or proj.numel() != 16 or campos.numel() != 3:
        raise ValueError("bg/campos must hold 3 values and viewmatrix/projmatrix 16")
    keep.extend([bg, view, proj, campos])
    return _capi.VrSettings(int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy),
                            float(rs.scale_modifier), int(rs.sh_degree), int(bool(rs.prefiltered)),
                            int(bool(rs.debug)), bg.data_ptr(), view.data_ptr(), proj.data_ptr(), campos.data_ptr(),
                            int(flag_bits))


def _inputs_struct(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, sh_rest=None, sh_tail=None):
    P = means3D.shape[0]
    M = (sh.shape[1] if sh is not None else 0) + (sh_rest.shape[1] if sh_rest is not None else 0)
    return _capi.VrInputs(P, M, _capi.ptr(means3D), _capi.ptr(sh), _capi.ptr(colors_precomp), _capi.ptr(opacities),
                          _capi.ptr(scales), _capi.ptr(rotations), _capi.ptr(cov3Ds_precomp), _capi.ptr(sh_rest),
                          _capi.ptr(sh_tail), sh.shape[0] if sh_tail is not None else 0)


def _cpu_args_copy(args):
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


# largest number of tile-list entries seen per (device, image size, model size): handed to the library as a capacity hint so that
# it can request its R-sized buffers before the forward's single host synchronisation.  A running
# maximum (not the last value) keeps every request the same size from step to step, so PyTorch's
# caching allocator serves them from its pool instead of going back to hipMalloc.
_LAST_R = {}

# Needed-segment hints (include/vegs_rast.h: VrSaved.needed_hint): per camera, the per-tile number of list segments its
# previous forward needed.  Keyed by the identity of the camera's matrices (VEGS' Camera objects keep their
# world_view_transform / full_proj_transform tensors for their lifetime, scene/cameras.py:76-87) and the image size.  A hint
# is never trusted -- a wrong one (a reused address, a scene that changed) only costs time, the kernels redo what it
# missed -- so no invalidation is needed.  A key's hint array is created on its SECOND sighting (value None until then):
# a camera has to come back before anything is spent on it.
# An EVALUATION-ONLY feature, OFF by default: a hint pays only while the model stands still between two visits of a
# camera (re-rendering fixed views of a trained model under torch.no_grad(): -3 % of the forward); in training a camera
# comes back once per epoch, and hints one epoch old COST 1.5 ... 4 % (HISTORY.md section 12, BENCH_r03) -- so a forward
# that will be differentiated never gets one: needed_hints(True) / VEGS_RAST_HINTS=1 turn the cache on for forwards under
# no_grad only.  ("always" -- tests: the hinted kernels' catch-up rounds under a backward -- lifts that restriction.)
_NEEDED = {}
_NEEDED_MAX = 4096


def _hint_mode(v):
    if v in (False, None, 0, "0", "", "off"):
        return False
    return "always" if v == "always" else "eval"


_use_hints = _hint_mode(os.environ.get("VEGS_RAST_HINTS", "0"))


def needed_hints(enabled):
    """Per-camera needed-segment hints for forwards under no_grad (True), for every forward ("always": tests), or off
    (False, the default); returns the previous setting."""
    global _use_hints
    old, _use_hints = _use_hints, _hint_mode(enabled)
    if not _use_hints:
        _NEEDED.clear()
    return old


# Hook between the two halves of the backward (include/vegs_rast.h: vr_backward_render / vr_backward_preprocess).  When
# set and the forward asked for the factored SH gradient, the backward calls `hook(sh_factor [P,3])` after the render
# backward -- the factor is complete in stream order -- and before the preprocess backward: vegs_amd.dist starts the
# all-gather of the factors there, so that it travels while the second half computes.
_split_hook = None


def set_backward_split_hook(fn):
    """Install (or, with None, remove) the hook; returns the previous one."""
    global _split_hook
    old, _split_hook = _split_hook, fn
    return old


# ---- gradient accumulation IN PLACE for steps that render several views of one model (include/vegs_rast.h:
# VR_FLAG_ACCUMULATE_GRADS).  PyTorch adds the gradients of a leaf's second, third ... use out of place: per view the op
# writes (56 + 12 K) bytes per Gaussian of dense gradients and autograd then reads them and the running sums and writes the
# new sums -- 0.22 ms per view at 2 M Gaussians.  With accumulate_grads(True) the backward of a view whose op inputs are
# LEAVES that already hold a `.grad` (i.e. from the step's second view on; the first one takes the plain path and its dense
# arrays BECOME the `.grad`s) adds its rows straight into those `.grad` tensors -- only the rows the view renders -- and
# returns None for these inputs.  Same numbers as autograd's own accumulation, in the same order (bit-equal in the
# deterministic mode); what differs is what autograd SEES: no gradient arrives at the leaf's AccumulateGrad node for those
# views, so hooks on the leaves (DDP-style post-accumulate hooks) do not fire for them.  Off by default; means2D (a
# per-view tensor) and the SH factor are returned as always.  Views whose backwards run on different streams
# (vegs_amd.views.ViewStreams) are ordered by an event between consecutive accumulations.
_accumulate = False
_acc_last = {}          # device -> (event recorded after the last accumulating backward, the stream it ran on)


def accumulate_grads(enabled):
    """In-place accumulation of the op's input gradients into existing leaf `.grad`s (see above); returns the previous
    setting."""
    global _accumulate
    old, _accumulate = _accumulate, bool(enabled)
    return old


def _acc_target(t):
    """The `.grad` tensor the backward may add into for op input `t`, or None."""
    if t is None or not t.requires_grad or not t.is_leaf:
        return None
    g = t.grad
    if g is None or g.dtype != torch.float32 or g.shape != t.shape or not g.is_contiguous() or g.device != t.device:
        return None
    return g


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, sh_rest=None, sh_color_grad=None, sh_tail=None):
        lib = _capi.load()
        rs = raster_settings
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise ValueError("means3D must have dimensions (num_points, 3)")
        if not means3D.is_cuda:
            raise ValueError("means3D must be a GPU tensor (the rasterizer has no CPU path)")
        device = means3D.device
        P = means3D.shape[0]
        means3D = _prep(means3D, "means3D", device) if P > 0 else means3D.contiguous()
        sh = _prep(sh, "shs", device)
        colors_precomp = _prep(colors_precomp, "colors_precomp", device)
        opacities = _prep(opacities, "opacities", device)
        scales = _prep(scales, "scales", device)
        rotations = _prep(rotations, "rotations", device)
        cov3Ds_precomp = _prep(cov3Ds_precomp, "cov3D_precomp", device)
        sh_rest = _prep(sh_rest, "shs[1] (features_rest)", device)
        sh_tail = _prep(sh_tail, "shs[2] (SH tail)", device)
        # SH tail: the rows of the last Gaussians live in a second whole tensor (the instances behind the static model)
        P0 = P
        ctx.tail_is_whole = False
        if sh_tail is not None and sh is None:       # nothing in front of the tail (an empty head was dropped by _prep):
            sh, sh_rest, sh_tail = sh_tail, None, None   # it is simply the whole tensor; its gradient goes back as the tail's
            ctx.tail_is_whole = True
        if sh_tail is not None:                      # (an empty tail was dropped by _prep as well)
            if sh.dim() != 3 or sh_tail.dim() != 3 or sh_tail.shape[2] != 3 or sh.shape[0] + sh_tail.shape[0] != P:
                raise ValueError("an SH tail needs shs [P0,.,3] and the tail [P - P0, M, 3]")
            P0 = sh.shape[0]
            M_all = sh.shape[1] + (sh_rest.shape[1] if sh_rest is not None else 0)
            if sh_tail.shape[1] != M_all or (3 * M_all) % 4 != 0 or M_all > 16:
                raise ValueError("the SH tail must hold the same number of coefficients (a multiple of 4, at most 16)")
        if sh_rest is not None:
            # split SH storage: shs = (features_dc [P,1,3], features_rest [P,M-1,3]) as the model keeps them
            if sh is None or sh.dim() != 3 or sh.shape[1] != 1 or sh_rest.dim() != 3 or sh_rest.shape[0] != P0 \
                    or sh_rest.shape[2] != 3:
                raise ValueError("split shs must be (features_dc [P,1,3], features_rest [P,M-1,3])")
        for t, name, shape in ((sh, "shs", (P0, None, 3)), (colors_precomp, "colors_precomp", (P, 3)),
                               (opacities, "opacities", None), (scales, "scales", (P, 3)),
                               (rotations, "rotations", (P, 4)), (cov3Ds_precomp, "cov3D_precomp", (P, 6))):
            if t is None:
                continue
            if shape is None:
                if t.numel() != P:
                    raise ValueError(f"{name} must hold one value per Gaussian")
            elif t.dim() != len(shape) or any(s is not None and s != d for s, d in zip(shape, t.shape)):
                raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {shape}")
        H, W = int(rs.image_height), int(rs.image_width)
        keep = []
        flag_bits = _flags
        with torch.cuda.device(device):
            st = _settings_struct(rs, device, keep, flag_bits)
            inp = _inputs_struct(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, sh_rest, sh_tail)
            # one [12,H,W] block: colour(3) depth(1) quat(4) scale(3) alpha(1) -- sliced into the 5 outputs
            img = torch.empty((12, H, W), dtype=torch.float32, device=device)
            color, depth, cov_quat, cov_scale, alpha = img[0:3], img[3:4], img[4:8], img[8:11], img[11:12]
            radii = torch.empty((P,), dtype=torch.int32, device=device)
            out = _capi.VrOutputs(color.data_ptr(), depth.data_ptr(), cov_quat.data_ptr(), cov_scale.data_ptr(),
                                  alpha.data_ptr(), _capi.ptr(radii))
            arena = _capi.Arena(device)
            saved = _capi.VrSaved()
            hint_key = (device.index, H, W, P >> 16)
            hint = _LAST_R.get(hint_key, 0)
            saved.binning_capacity = int(hint * 1.125) + 65536 if hint > 0 else 0
            stream = torch.cuda.current_stream(device).cuda_stream
            need_key, need_t = None, None
            # (ctx.needs_input_grad says which inputs require grad, also under no_grad: the caller's grad mode decides)
            hinted = _use_hints == "always" or (_use_hints and not (_will_differentiate and any(ctx.needs_input_grad)))
            if hinted and P > 0 and isinstance(rs.viewmatrix, torch.Tensor) and isinstance(rs.projmatrix, torch.Tensor):
                need_key = (device.index, H, W, rs.viewmatrix.data_ptr(), rs.projmatrix.data_ptr())
                if need_key not in _NEEDED:
                    # FIRST sighting: only remember the key.  Cameras that are rendered once and thrown away (the
                    # reference's augmented views are fresh tensors every iteration, train.py:177 /
                    # scene/cameras.py:126-227; eval and video cameras) never cost an allocation or a fill launch.
                    if len(_NEEDED) >= _NEEDED_MAX:
                        _NEEDED.pop(next(iter(_NEEDED)))
                    _NEEDED[need_key] = None
                else:
                    need_t = _NEEDED[need_key]
                    if need_t is None:           # second sighting: "no idea" -- this forward fills in what it needed
                        need_t = torch.full((((W + 15) // 16) * ((H + 15) // 16),), 0x3FFFFFFF, dtype=torch.int32,
                                            device=device)
                        _NEEDED[need_key] = need_t
                    saved.needed_hint = need_t.data_ptr()
            cpu_args = _cpu_args_copy((means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)) \
                if rs.debug else None
            cb = arena.callback()
            rc = lib.vr_forward(C.byref(st), C.byref(inp), C.byref(out), cb, None, stream, C.byref(saved))
            del cb
            arena.release_scratch()
            if rc != 0:
                if rs.debug:
                    torch.save((cpu_args, tuple(rs)), "snapshot_fw.dump")
                    print("rasterizer forward failed; inputs written to snapshot_fw.dump")
                if arena.error is not None:
                    raise arena.error
                _capi.check(rc)
        ctx.raster_settings = rs
        ctx.flag_bits = flag_bits
        ctx.num_rendered = int(saved.num_rendered)
        ctx.num_visible = int(saved.num_visible)
        ctx.binning_capacity = int(saved.binning_capacity)
        ctx.ticket = int(saved.ticket)
        _LAST_R[hint_key] = max(_LAST_R.get(hint_key, 0), ctx.num_rendered)
        ctx.buffers = (arena.kept[_capi.VR_BUF_GEOM], arena.kept[_capi.VR_BUF_BINNING], arena.kept[_capi.VR_BUF_IMAGE])
        ctx.has = (sh is not None, colors_precomp is not None, scales is not None, cov3Ds_precomp is not None)
        if sh_color_grad is not None and (sh is None or tuple(sh_color_grad.shape) != (P, 3)):
            raise ValueError("sh_color_grad needs shs and must be a [P,3] tensor")
        ctx.sh_factored = sh_color_grad is not None
        ctx.save_for_backward(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, radii, sh_rest,
                              sh_tail)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)   # unused outputs arrive as None -> NULL, no zero tensors
        return color, depth, cov_quat, cov_scale, alpha, radii

    @staticmethod
    def backward(ctx, g_color, g_depth, g_quat, g_scale, g_alpha, _g_radii):
        lib = _capi.load()
        rs = ctx.raster_settings
        means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, radii, sh_rest, sh_tail = ctx.saved_tensors
        geom, binning, image = ctx.buffers
        device = means3D.device
        P = means3D.shape[0]
        keep = []

        def g(t):
            return None if t is None else _dev_f32(t, device)
        g_color, g_depth, g_quat, g_scale, g_alpha = g(g_color), g(g_depth), g(g_quat), g(g_scale), g(g_alpha)
        with torch.cuda.device(device):
            # factored SH gradient: only the clamp-masked dL/d(colour) [P,3] is produced (it lands on the caller's
            # `sh_color_grad` tensor); shs / features_rest receive no gradient from this op
            factored = ctx.sh_factored
            # accumulate mode (accumulate_grads): ALL the gradient arrays this call would allocate must have a leaf `.grad`
            # to add into -- one flag for the call; otherwise (a step's first view, non-leaf inputs) the plain path
            wanted = [t for t in (means3D, opacities, colors_precomp, scales, rotations, cov3Ds_precomp) if t is not None]
            if not factored:
                wanted += [t for t in (sh, sh_rest, sh_tail) if t is not None]
            targets = {id(t): _acc_target(t) for t in wanted} if (_accumulate and P > 0) else {}
            acc = bool(targets) and all(g is not None for g in targets.values())
            flag_bits = ctx.flag_bits | (FLAG_ACCUMULATE_GRADS if acc else 0)

            def out(t):
                return None if t is None else (targets[id(t)] if acc else torch.empty_like(t))
            st = _settings_struct(rs, device, keep, flag_bits)
            inp = _inputs_struct(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, sh_rest, sh_tail)
            d_means3D = out(means3D)
            d_means2D = torch.empty((P, 3), dtype=torch.float32, device=device)
            d_opac = out(opacities)
            d_sink = torch.empty((P, 3), dtype=torch.float32, device=device) if factored else None
            d_sh = out(sh) if not factored else None
            d_sh_rest = out(sh_rest) if not factored else None
            d_sh_tail = out(sh_tail) if not factored else None
            d_col = out(colors_precomp)
            d_scales = out(scales)
            d_rot = out(rotations)
            d_cov = out(cov3Ds_precomp)
            if acc:      # order this accumulation behind the previous one if that ran on another stream
                cur = torch.cuda.current_stream(device)
                last = _acc_last.get(device)
                if last is not None and last[1] != cur:
                    cur.wait_event(last[0])
            gout = _capi.VrOutGrads(_capi.ptr(g_color), _capi.ptr(g_depth), _capi.ptr(g_quat), _capi.ptr(g_scale),
                                    _capi.ptr(g_alpha))
            gin = _capi.VrInGrads(_capi.ptr(d_means3D), _capi.ptr(d_means2D), _capi.ptr(d_sh), _capi.ptr(d_col),
                                  _capi.ptr(d_opac), _capi.ptr(d_scales), _capi.ptr(d_rot), _capi.ptr(d_cov),
                                  _capi.ptr(d_sh_rest), _capi.ptr(d_sink), _capi.ptr(d_sh_tail))
            saved = _capi.VrSaved(geom.data_ptr(), binning.data_ptr(), image.data_ptr(), ctx.num_rendered,
                                  ctx.num_visible, ctx.binning_capacity, None, ctx.ticket)
            arena = _capi.Arena(device)
            stream = torch.cuda.current_stream(device).cuda_stream
            cpu_args = _cpu_args_copy((means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                       radii, g_color, g_depth, g_quat, g_scale, g_alpha)) if rs.debug else None
            cb = arena.callback()
            hook = _split_hook if factored else None
            if hook is None:
                rc = lib.vr_backward(C.byref(st), C.byref(inp), _capi.ptr(radii), C.byref(saved), C.byref(gout),
                                     C.byref(gin), cb, None, stream)
            else:
                state = C.c_void_p()
                rc = lib.vr_backward_render(C.byref(st), C.byref(inp), _capi.ptr(radii), C.byref(saved), C.byref(gout),
                                            C.byref(gin), cb, None, stream, C.byref(state))
                if rc == 0:
                    hook(d_sink)
                    rc = lib.vr_backward_preprocess(C.byref(st), C.byref(inp), _capi.ptr(radii), C.byref(saved),
                                                    C.byref(gin), state, stream)
            del cb
            arena.release_scratch()
            if rc != 0:
                if rs.debug:
                    torch.save((cpu_args, tuple(rs)), "snapshot_bw.dump")
                    print("rasterizer backward failed; inputs written to snapshot_bw.dump")
                if arena.error is not None:
                    raise arena.error
                _capi.check(rc)
            if _accumulate and P > 0:
                # the next accumulating backward (possibly on another stream) must come after this call's writes -- also
                # after a PLAIN call's: its dense arrays become the `.grad`s the next view adds into
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(device))
                _acc_last[device] = (ev, torch.cuda.current_stream(device))
        # input order: means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings
        if acc:          # added in place: nothing for autograd to accumulate
            return None, d_means2D, None, None, None, None, None, None, None, None, d_sink, None
        if ctx.tail_is_whole:        # the tail stood in for the whole tensor (empty head): its gradient goes back as the tail's
            d_sh, d_sh_tail = None, d_sh
        return d_means3D, d_means2D, d_sh, d_col, d_opac, d_scales, d_rot, d_cov, None, d_sh_rest, d_sink, d_sh_tail


_will_differentiate = True      # grad mode of the CALLER of the op (inside Function.forward it is always off)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, sh_rest=None, sh_color_grad=None, sh_tail=None):
    global _will_differentiate
    _will_differentiate = torch.is_grad_enabled()
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, sh_rest, sh_color_grad, sh_tail)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool[P]: view-space depth > 0.2 (reference utils/norminit_utils.py:179 uses it as a mask)."""
        lib = _capi.load()
        rs = self.raster_settings
        with torch.no_grad():
            if not positions.is_cuda:
                raise ValueError("positions must be a GPU tensor (the rasterizer has no CPU path)")
            device = positions.device
            pos = positions.detach().to(torch.float32).contiguous()
            P = pos.shape[0]
            view = _dev_f32(rs.viewmatrix, device)
            proj = _dev_f32(rs.projmatrix, device)
            present = torch.empty((P,), dtype=torch.uint8, device=device)
            with torch.cuda.device(device):
                rc = lib.vr_mark_visible(_capi.ptr(pos), P, view.data_ptr(), proj.data_ptr(), _capi.ptr(present),
                                         torch.cuda.current_stream(device).cuda_stream)
            _capi.check(rc)
            return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, sh_color_grad=None):
        """The reference's eight keyword arguments (gaussian_renderer/__init__.py:86-94), plus one extension:
        `sh_color_grad`, a [P,3] float32 tensor with requires_grad (its contents are never read: `torch.empty` will do).  When given, the backward deposits on it the
        clamp-masked dL/d(colour) -- the 3-float factor of the rank-1 SH gradient dL/dshs[i,k,c] = basis_k(dir_i) *
        factor[i,c] -- and `shs` itself receives NO gradient (vegs_amd.optim.sh_grad_from_factors / Adam.step_sh_factored
        rebuild or consume it; vegs_amd.dist exchanges 3 instead of 48 floats per Gaussian and view)."""
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("exactly one of shs and colors_precomp must be given")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("exactly one of (scales, rotations) and cov3D_precomp must be given")
        # Extension: shs may be the PAIR (features_dc [P,1,3], features_rest [P,M-1,3]) the reference's model stores
        # (scene/gaussian_model.py:112-116 concatenates them on every call); the kernels then read the two tensors
        # in place and return their gradients separately -- no torch.cat, no slicing copies.
        # A THIRD element is the SH TAIL: the whole [P - P0, M, 3] tensor of the Gaussians behind the first P0 (the
        # dynamic instances that render_all / render_dyn put behind the static model, merge_kwargs :182-186) -- the static
        # model's tensors are then read where they are instead of being concatenated with a few thousand instance rows;
        # (whole [P0,M,3], None, tail) is accepted as well.
        sh_rest, sh_tail = None, None
        if isinstance(shs, (tuple, list)):
            if len(shs) == 3:
                shs, sh_rest, sh_tail = shs
            elif len(shs) == 2:
                shs, sh_rest = shs
            else:
                raise Exception("split shs must be (features_dc, features_rest) or (features_dc, features_rest, tail)")
            if sh_rest is not None and sh_rest.shape[1] == 0:
                sh_rest = None
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, rs, sh_rest, sh_color_grad, sh_tail)
