"""ctypes binding of libvegsrast.so (C ABI declared in include/vegs_rast.h).

There is no fallback: if the HIP library is missing or does not export the ABI, importing the
product path raises.  PyTorch is plumbing here (device memory, current stream); the signatures
crossing the boundary are plain pointers and sizes.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VEGS_LIB") or os.path.join(_HERE, "_lib", "libvegsrast.so")   # (VEGS_LIB: reproducer builds, vegs_amd/build.py --variant)
ABI_VERSION = 9

# VrSettings.flags (include/vegs_rast.h, VrFlags)
FLAG_SCALE_MODIFIED, FLAG_DEPTH_NORMALIZED, FLAG_EXTRA_NO_ALPHA_GRAD, FLAG_FILL_EMPTY, FLAG_DETERMINISTIC = 1, 2, 4, 8, 256
FLAG_SCAN_BINNING = 512
FLAG_ROUNDS_OFF, FLAG_ROUNDS_ON = 1024, 2048
FLAG_RAW_PARAMS = 4096
FLAG_FAST_EXP = 8192
FLAG_VERIFY_BINNING = 16384
FLAG_FULL_TILE_LISTS = 32768
FLAG_ACCUMULATE_GRADS = 65536

VR_BUF_GEOM, VR_BUF_BINNING, VR_BUF_IMAGE, VR_BUF_SCRATCH, VR_BUF_BACKWARD = 0, 1, 2, 3, 4

# every symbol include/vegs_rast.h declares
EXPORTS = ["vr_abi_version", "vr_last_error", "vr_forward", "vr_backward", "vr_backward_render", "vr_backward_preprocess", "vr_mark_visible", "vr_get_counters",
           "vr_count_fragments", "vr_count_blended", "vr_count_flushes", "vr_export_needed", "vr_debug_export_binning", "vr_debug_set_guard", "vr_debug_raise_guard", "vr_debug_rebinned", "vr_profile_level", "vr_profile_collect",
           "vr_knn3_mean_dist2", "vr_photometric_forward", "vr_photometric_backward",
           "vr_normal_guidance_forward", "vr_normal_guidance_backward", "vr_training_loss_forward", "vr_training_loss_backward", "vr_adam_step", "vr_densify_stats",
           "vr_densify_plan_words", "vr_densify_plan", "vr_densify_apply", "vr_reset_opacity",
           "vr_sh_grad_from_factors", "vr_sh_adam_step",
           "vr_instances_forward", "vr_instances_backward", "vr_activations_forward", "vr_activations_backward",
           "vr_boxmodel_forward", "vr_boxmodel_backward", "vr_boxmodel_regularizer_grad",
           "vr_xgmi_create", "vr_xgmi_detach", "vr_xgmi_destroy", "vr_xgmi_layout", "vr_xgmi_window", "vr_xgmi_handle", "vr_xgmi_attach",
           "vr_xgmi_allreduce", "vr_xgmi_allgather_begin", "vr_xgmi_allgather_wait", "vr_xgmi_check", "vr_xgmi_failed", "vr_xgmi_set_wait_bound"]
STAGES = ["preprocess", "compact", "depth_sort", "emit", "tile_sort", "ranges", "render_fwd", "bwd_zero",
          "render_bwd", "preprocess_bwd", "k_seg_bwd"]


class VrSettings(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32), ("tanfovx", C.c_float),
                ("tanfovy", C.c_float), ("scale_modifier", C.c_float), ("sh_degree", C.c_int32),
                ("prefiltered", C.c_int32), ("debug", C.c_int32), ("bg", C.c_void_p), ("viewmatrix", C.c_void_p),
                ("projmatrix", C.c_void_p), ("campos", C.c_void_p), ("flags", C.c_uint32)]


class VrInputs(C.Structure):
    _fields_ = [("P", C.c_int32), ("M", C.c_int32), ("means3D", C.c_void_p), ("shs", C.c_void_p),
                ("colors_precomp", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p),
                ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("shs_rest", C.c_void_p),
                ("shs_tail", C.c_void_p), ("tail_start", C.c_int64)]


class VrOutputs(C.Structure):
    _fields_ = [("color", C.c_void_p), ("depth", C.c_void_p), ("cov_quat", C.c_void_p), ("cov_scale", C.c_void_p),
                ("alpha", C.c_void_p), ("radii", C.c_void_p)]


class VrSaved(C.Structure):
    _fields_ = [("geom", C.c_void_p), ("binning", C.c_void_p), ("image", C.c_void_p), ("num_rendered", C.c_int64),
                ("num_visible", C.c_int64), ("binning_capacity", C.c_int64), ("needed_hint", C.c_void_p),
                ("ticket", C.c_uint64)]


class VrOutGrads(C.Structure):
    _fields_ = [("dL_dcolor", C.c_void_p), ("dL_ddepth", C.c_void_p), ("dL_dcov_quat", C.c_void_p),
                ("dL_dcov_scale", C.c_void_p), ("dL_dalpha", C.c_void_p)]


class VrInGrads(C.Structure):
    _fields_ = [("dL_dmeans3D", C.c_void_p), ("dL_dmeans2D", C.c_void_p), ("dL_dshs", C.c_void_p),
                ("dL_dcolors_precomp", C.c_void_p), ("dL_dopacities", C.c_void_p), ("dL_dscales", C.c_void_p),
                ("dL_drotations", C.c_void_p), ("dL_dcov3D_precomp", C.c_void_p), ("dL_dshs_rest", C.c_void_p),
                ("dL_dcolors_sh", C.c_void_p), ("dL_dshs_tail", C.c_void_p)]


class VrAdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_double), ("step", C.c_int64), ("eps", C.c_double)]


class VrDensifySettings(C.Structure):
    _fields_ = [("max_grad", C.c_double), ("min_opacity", C.c_double), ("extent", C.c_double),
                ("percent_dense", C.c_double), ("prune_big", C.c_int32)]


class VrDensifyTensor(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("m_src", C.c_void_p), ("m_dst", C.c_void_p),
                ("v_src", C.c_void_p), ("v_dst", C.c_void_p), ("width", C.c_int32), ("role", C.c_int32)]


class VrShAdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("lr", C.c_double),
                ("step", C.c_int64)]


class VrInstance(C.Structure):
    _fields_ = [("means", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p), ("box2world", C.c_void_p),
                ("n", C.c_int64), ("offset", C.c_int64)]


class VrInstanceGrads(C.Structure):
    _fields_ = [("dL_dmeans", C.c_void_p), ("dL_dscales", C.c_void_p), ("dL_drotations", C.c_void_p),
                ("dL_dbox2world", C.c_void_p)]


class VrBoxModel(C.Structure):
    _fields_ = [("box2world", C.c_void_p), ("delta_r", C.c_void_p), ("delta_s", C.c_void_p), ("delta_t", C.c_void_p)]


class VrBoxModelGrads(C.Structure):
    _fields_ = [("d_delta_r", C.c_void_p), ("d_delta_s", C.c_void_p), ("d_delta_t", C.c_void_p)]


class VrXgmiLayout(C.Structure):
    _fields_ = [("recv_offset", C.c_int64), ("result_offset", C.c_int64), ("gather_offset", C.c_int64 * 2),
                ("gather_slot", C.c_int64), ("total_floats", C.c_int64)]


class VrXgmiSegment(C.Structure):
    _fields_ = [("src", C.c_void_p), ("n", C.c_int64)]


class VrCounters(C.Structure):
    _fields_ = [("P", C.c_int64), ("num_visible", C.c_int64), ("num_rendered", C.c_int64), ("num_tiles", C.c_int64),
                ("num_pixels", C.c_int64)]


VrAllocFn = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, C.c_size_t)

_lib = None


def load():
    """Load libvegsrast.so; raise (never fall back) if it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python -m vegs_amd.build` "
                          "(there is no CPU or PyTorch fallback for the rasterizer)")
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise ImportError(f"{LIB_PATH} does not export {name}")
    lib.vr_abi_version.restype = C.c_int
    lib.vr_last_error.restype = C.c_char_p
    lib.vr_forward.restype = C.c_int
    lib.vr_forward.argtypes = [C.POINTER(VrSettings), C.POINTER(VrInputs), C.POINTER(VrOutputs), VrAllocFn,
                               C.c_void_p, C.c_void_p, C.POINTER(VrSaved)]
    lib.vr_backward.restype = C.c_int
    lib.vr_backward.argtypes = [C.POINTER(VrSettings), C.POINTER(VrInputs), C.c_void_p, C.POINTER(VrSaved),
                                C.POINTER(VrOutGrads), C.POINTER(VrInGrads), VrAllocFn, C.c_void_p, C.c_void_p]
    lib.vr_backward_render.restype = C.c_int
    lib.vr_backward_render.argtypes = [C.POINTER(VrSettings), C.POINTER(VrInputs), C.c_void_p, C.POINTER(VrSaved),
                                       C.POINTER(VrOutGrads), C.POINTER(VrInGrads), VrAllocFn, C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_void_p)]
    lib.vr_backward_preprocess.restype = C.c_int
    lib.vr_backward_preprocess.argtypes = [C.POINTER(VrSettings), C.POINTER(VrInputs), C.c_void_p, C.POINTER(VrSaved),
                                           C.POINTER(VrInGrads), C.c_void_p, C.c_void_p]
    lib.vr_mark_visible.restype = C.c_int
    lib.vr_mark_visible.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vr_get_counters.restype = None
    lib.vr_get_counters.argtypes = [C.POINTER(VrCounters)]
    lib.vr_count_fragments.restype = C.c_int
    lib.vr_count_fragments.argtypes = [C.POINTER(VrSaved), C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_int64)]
    lib.vr_count_blended.restype = C.c_int
    lib.vr_count_blended.argtypes = [C.POINTER(VrSaved), C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_int64)]
    lib.vr_count_flushes.restype = C.c_int
    lib.vr_count_flushes.argtypes = [C.POINTER(VrSaved), C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_int64)]
    lib.vr_export_needed.restype = C.c_int
    lib.vr_export_needed.argtypes = [C.POINTER(VrSaved), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.vr_debug_export_binning.restype = C.c_int
    lib.vr_debug_export_binning.argtypes = [C.POINTER(VrSaved), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                            C.c_void_p]
    lib.vr_debug_set_guard.restype = C.c_int
    lib.vr_debug_set_guard.argtypes = [C.c_uint32, C.c_void_p]
    lib.vr_debug_raise_guard.restype = C.c_int
    lib.vr_debug_raise_guard.argtypes = [C.c_int]
    lib.vr_debug_rebinned.restype = C.c_int
    lib.vr_debug_rebinned.argtypes = []
    lib.vr_knn3_mean_dist2.restype = C.c_int
    lib.vr_knn3_mean_dist2.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, VrAllocFn, C.c_void_p, C.c_void_p]
    vp, i32 = C.c_void_p, C.c_int32
    lib.vr_photometric_forward.restype = C.c_int
    lib.vr_photometric_forward.argtypes = [vp, vp, i32, i32, i32, vp, vp, VrAllocFn, vp, vp]
    lib.vr_photometric_backward.restype = C.c_int
    lib.vr_photometric_backward.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.vr_normal_guidance_forward.restype = C.c_int
    lib.vr_normal_guidance_forward.argtypes = [vp, vp, vp, C.POINTER(C.c_float), i32, i32, vp, VrAllocFn, vp, vp]
    lib.vr_normal_guidance_backward.restype = C.c_int
    lib.vr_normal_guidance_backward.argtypes = [vp, vp, vp, C.POINTER(C.c_float), i32, i32, vp, vp, vp, vp]
    lib.vr_training_loss_forward.restype = C.c_int
    lib.vr_training_loss_forward.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, C.POINTER(C.c_float), C.c_float, C.c_float, i32,
                                             vp, vp, vp, VrAllocFn, vp, vp]
    lib.vr_training_loss_backward.restype = C.c_int
    lib.vr_training_loss_backward.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, C.POINTER(C.c_float), C.c_float, C.c_float,
                                              i32, vp, vp, vp, vp, vp]
    lib.vr_adam_step.restype = C.c_int
    lib.vr_adam_step.argtypes = [C.POINTER(VrAdamTensor), i32, C.c_double, C.c_double, C.c_double, vp]
    lib.vr_densify_stats.restype = C.c_int
    lib.vr_densify_stats.argtypes = [vp, vp, i32, vp, vp, vp, vp]
    lib.vr_densify_plan_words.restype = C.c_int64
    lib.vr_densify_plan_words.argtypes = [i32]
    lib.vr_densify_plan.restype = C.c_int
    lib.vr_densify_plan.argtypes = [vp, vp, vp, vp, i32, C.POINTER(VrDensifySettings), vp, vp, vp]
    lib.vr_densify_apply.restype = C.c_int
    lib.vr_densify_apply.argtypes = [vp, i32, i32, C.POINTER(VrDensifyTensor), i32, vp, vp, vp, vp]
    lib.vr_reset_opacity.restype = C.c_int
    lib.vr_reset_opacity.argtypes = [vp, vp, vp, C.c_int64, C.c_float, vp]
    lib.vr_sh_grad_from_factors.restype = C.c_int
    lib.vr_sh_grad_from_factors.argtypes = [vp, i32, vp, vp, i32, C.c_int64, i32, i32, C.c_float, vp, vp, vp]
    lib.vr_sh_adam_step.restype = C.c_int
    lib.vr_sh_adam_step.argtypes = [vp, i32, vp, vp, i32, C.c_int64, i32, i32, C.c_float, C.POINTER(VrShAdamTensor),
                                    C.POINTER(VrShAdamTensor), C.c_double, C.c_double, C.c_double, vp]
    lib.vr_instances_forward.restype = C.c_int
    lib.vr_instances_forward.argtypes = [C.POINTER(VrInstance), i32, vp, vp, vp, vp]
    lib.vr_instances_backward.restype = C.c_int
    lib.vr_instances_backward.argtypes = [C.POINTER(VrInstance), C.POINTER(VrInstanceGrads), i32, vp, vp, vp, VrAllocFn, vp, vp]
    lib.vr_activations_forward.restype = C.c_int
    lib.vr_activations_forward.argtypes = [vp, vp, vp, C.c_int64, vp, vp, vp, vp]
    lib.vr_activations_backward.restype = C.c_int
    lib.vr_activations_backward.argtypes = [vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp]
    lib.vr_boxmodel_forward.restype = C.c_int
    lib.vr_boxmodel_forward.argtypes = [C.POINTER(VrBoxModel), i32, vp, vp]
    lib.vr_boxmodel_backward.restype = C.c_int
    lib.vr_boxmodel_backward.argtypes = [C.POINTER(VrBoxModel), C.POINTER(VrBoxModelGrads), i32, vp, i32, vp]
    lib.vr_boxmodel_regularizer_grad.restype = C.c_int
    lib.vr_boxmodel_regularizer_grad.argtypes = [C.POINTER(VrBoxModel), C.POINTER(VrBoxModelGrads), i32, C.c_float, vp]
    lib.vr_xgmi_create.restype = C.c_int
    lib.vr_xgmi_create.argtypes = [i32, i32, C.c_int64, C.c_int64, C.POINTER(vp)]
    lib.vr_xgmi_destroy.restype = C.c_int
    lib.vr_xgmi_destroy.argtypes = [vp]
    lib.vr_xgmi_detach.restype = C.c_int
    lib.vr_xgmi_detach.argtypes = [vp]
    lib.vr_xgmi_layout.restype = C.c_int
    lib.vr_xgmi_layout.argtypes = [vp, C.POINTER(VrXgmiLayout)]
    lib.vr_xgmi_window.restype = vp
    lib.vr_xgmi_window.argtypes = [vp]
    lib.vr_xgmi_handle.restype = C.c_int
    lib.vr_xgmi_handle.argtypes = [vp, vp]
    lib.vr_xgmi_attach.restype = C.c_int
    lib.vr_xgmi_attach.argtypes = [vp, vp]
    lib.vr_xgmi_allreduce.restype = C.c_int
    lib.vr_xgmi_allreduce.argtypes = [vp, C.POINTER(VrXgmiSegment), i32, C.c_float, C.POINTER(C.c_int64), vp]
    lib.vr_xgmi_allgather_begin.restype = C.c_int
    lib.vr_xgmi_allgather_begin.argtypes = [vp, C.POINTER(VrXgmiSegment), i32, i32, C.POINTER(C.c_int64), vp]
    lib.vr_xgmi_allgather_wait.restype = C.c_int
    lib.vr_xgmi_allgather_wait.argtypes = [vp, i32, vp]
    lib.vr_xgmi_check.restype = C.c_int
    lib.vr_xgmi_check.argtypes = [vp, vp]
    lib.vr_xgmi_failed.restype = C.c_int
    lib.vr_xgmi_failed.argtypes = [vp]
    lib.vr_xgmi_set_wait_bound.restype = C.c_int
    lib.vr_xgmi_set_wait_bound.argtypes = [vp, C.c_double]
    lib.vr_profile_level.restype = C.c_int
    lib.vr_profile_level.argtypes = [C.c_int]
    lib.vr_profile_collect.restype = C.c_int
    lib.vr_profile_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    if lib.vr_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.vr_abi_version()} != {ABI_VERSION}")
    _lib = lib
    return lib


class VegsRastError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise VegsRastError(f"libvegsrast error {rc}: {load().vr_last_error().decode()}")


def ptr(t):
    """Device pointer of a tensor, or NULL for None / empty tensors."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


class Arena:
    """Allocator handed to the library: byte tensors from torch's caching allocator.

    GEOM / BINNING / IMAGE tensors are kept (they are saved for backward); SCRATCH tensors
    are dropped when the arena is released, i.e. returned to the caching allocator, which is
    safe because all work was enqueued on the same stream.
    """

    def __init__(self, device):
        self.device = device
        self.kept = {}
        self.scratch = []
        self.error = None

    def callback(self):
        """A fresh C callback bound to this arena.  Deliberately NOT stored on the arena: arena ->
        callback -> bound method -> arena would be a reference cycle that keeps hundreds of MB of
        buffers alive until the cyclic GC happens to run (and sends the caching allocator back to
        hipMalloc in the meantime)."""
        return VrAllocFn(self._alloc)

    def _alloc(self, _user, kind, nbytes):
        try:
            t = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=self.device)
            if kind == VR_BUF_SCRATCH:
                self.scratch.append(t)
            else:
                self.kept[kind] = t
            return t.data_ptr()
        except BaseException as e:  # never let an exception cross the C boundary
            self.error = e
            return 0

    def release_scratch(self):
        self.scratch.clear()


def profile_level(level):
    return load().vr_profile_level(int(level))


def profile_collect():
    """dict stage -> (total ms, launches) accumulated since the last collect."""
    n = len(STAGES)
    ms = (C.c_double * n)()
    cnt = (C.c_int64 * n)()
    check(load().vr_profile_collect(ms, cnt))
    return {STAGES[i]: (ms[i], cnt[i]) for i in range(n)}


def counters():
    c = VrCounters()
    load().vr_get_counters(C.byref(c))
    return dict(P=c.P, V=c.num_visible, R=c.num_rendered, T=c.num_tiles, N=c.num_pixels)


def saved_of(grad_fn):
    """VrSaved of the forward behind `grad_fn` (the op's autograd node)."""
    geom, binning, image = grad_fn.buffers
    return VrSaved(geom.data_ptr(), binning.data_ptr(), image.data_ptr(), grad_fn.num_rendered, grad_fn.num_visible,
                   grad_fn.binning_capacity, None, getattr(grad_fn, "ticket", 0))


def count_blended(grad_fn, H, W, device):
    """B = (pixel, splat) pairs actually blended by the forward behind `grad_fn` (alpha >= 1/255, before the stop)."""
    saved = saved_of(grad_fn)
    out = C.c_int64(0)
    with torch.cuda.device(device):
        check(load().vr_count_blended(C.byref(saved), H, W, torch.cuda.current_stream(device).cuda_stream, C.byref(out)))
    return out.value


def count_flushes(grad_fn, H, W, device):
    """(list entry, region) flushes of the render backward behind `grad_fn`: x 17 = its global fp32 atomics."""
    saved = saved_of(grad_fn)
    out = C.c_int64(0)
    with torch.cuda.device(device):
        check(load().vr_count_flushes(C.byref(saved), H, W, torch.cuda.current_stream(device).cuda_stream, C.byref(out)))
    return out.value


def count_fragments(grad_fn, H, W, device):
    """F = sum of n_contrib of the forward behind `grad_fn` (the op's autograd node)."""
    geom, binning, image = grad_fn.buffers
    saved = saved_of(grad_fn)
    out = C.c_int64(0)
    with torch.cuda.device(device):
        check(load().vr_count_fragments(C.byref(saved), H, W, torch.cuda.current_stream(device).cuda_stream,
                                        C.byref(out)))
    return out.value
