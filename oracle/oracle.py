"""ctypes front-end of the CPU oracle (oracle/vr_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of vr_oracle.c ("parity unpinned").  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under vegs_amd/ or diff_gaussian_rasterization/ does.

The functions take/return numpy arrays and mirror the operator boundary used by the
reference at gaussian_renderer/__init__.py:38-53,86-94 (settings + the 8 op kwargs ->
color, depth, cov_quat, cov_scale, alpha, radii).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvr_oracle.so")
NCH = 11


class OrCam(C.Structure):
    _fields_ = [
        ("H", C.c_int), ("W", C.c_int),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("bg", C.c_float * 3),
        ("scale_modifier", C.c_float),
        ("view", C.c_float * 16),
        ("proj", C.c_float * 16),
        ("campos", C.c_float * 3),
        ("sh_degree", C.c_int),
        ("M", C.c_int),
        ("flags", C.c_uint),
    ]


def build(force=False):
    """Compile libvr_oracle.so with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "vr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.or_count_rendered.restype = C.c_long
    return _lib


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_cam(H, W, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix, campos, sh_degree, M, flags=0):
    """flags: VrFlags bits 0-3 of include/vegs_rast.h (the fork assumptions of SURVEY.md A.8 as switches) and bit 15
    (VR_FLAG_FULL_TILE_LISTS: the reference's full tile rectangles instead of the tight lists)."""
    cam = OrCam()
    cam.flags = int(flags) & (0xF | 32768)
    cam.H, cam.W = int(H), int(W)
    cam.tanfovx, cam.tanfovy = float(tanfovx), float(tanfovy)
    cam.bg[:] = [float(x) for x in np.asarray(bg, dtype=np.float32).reshape(3)]
    cam.scale_modifier = float(scale_modifier)
    cam.view[:] = [float(x) for x in np.asarray(viewmatrix, dtype=np.float32).reshape(16)]
    cam.proj[:] = [float(x) for x in np.asarray(projmatrix, dtype=np.float32).reshape(16)]
    cam.campos[:] = [float(x) for x in np.asarray(campos, dtype=np.float32).reshape(3)]
    cam.sh_degree = int(sh_degree)
    cam.M = int(M)
    return cam


def cov3d(scales, scale_modifier, rotations):
    """[n,6] upper-triangular covariance R diag(mod*s)^2 R^T exactly as or_preprocess computes it."""
    scales, rotations = _f32(scales), _f32(rotations)
    out = np.zeros((scales.shape[0], 6), np.float32)
    lib().or_cov3d(scales.shape[0], _p(scales), C.c_float(scale_modifier), _p(rotations), _p(out))
    return out


def mark_visible(cam, means3D):
    means3D = _f32(means3D)
    P = means3D.shape[0]
    out = np.zeros(P, dtype=np.uint8)
    lib().or_mark_visible(C.byref(cam), P, _p(means3D), _p(out))
    return out.astype(bool)


def forward(cam, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp):
    """Full forward.  Returns (outputs dict, state dict for backward)."""
    L = lib()
    means3D, shs, colors_precomp = _f32(means3D), _f32(shs), _f32(colors_precomp)
    opacities, scales, rotations, cov3D_precomp = _f32(opacities), _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    P = means3D.shape[0]
    H, W = cam.H, cam.W
    N = H * W
    gx, gy = (W + 15) // 16, (H + 15) // 16
    st = dict(
        depth=np.zeros(P, np.float32), xy=np.zeros((P, 2), np.float32), cov3D=np.zeros((P, 6), np.float32),
        conic_op=np.zeros((P, 4), np.float32), rgb=np.zeros((P, 3), np.float32),
        clamped=np.zeros((P, 3), np.uint8), radii=np.zeros(P, np.int32), rect=np.zeros((P, 4), np.int32),
        tiles_touched=np.zeros(P, np.uint32), tile_mask=np.zeros(P, np.uint64),
    )
    L.or_preprocess(C.byref(cam), P, _p(means3D), _p(shs), _p(colors_precomp), _p(opacities), _p(scales),
                    _p(rotations), _p(cov3D_precomp), _p(st["depth"]), _p(st["xy"]), _p(st["cov3D"]),
                    _p(st["conic_op"]), _p(st["rgb"]), _p(st["clamped"]), _p(st["radii"]), _p(st["rect"]),
                    _p(st["tiles_touched"]), _p(st["tile_mask"]))
    R = int(L.or_count_rendered(P, _p(st["tiles_touched"])))
    keys = np.zeros(max(R, 1), np.uint64)
    point_list = np.zeros(max(R, 1), np.uint32)
    ranges = np.zeros((gx * gy, 2), np.int32)
    L.or_binning(C.byref(cam), P, _p(st["depth"]), _p(st["rect"]), _p(st["tiles_touched"]), _p(st["tile_mask"]), C.c_long(R),
                 _p(keys), _p(point_list), _p(ranges))
    out = dict(
        color=np.zeros((3, H, W), np.float32), depth=np.zeros((1, H, W), np.float32),
        cov_quat=np.zeros((4, H, W), np.float32), cov_scale=np.zeros((3, H, W), np.float32),
        alpha=np.zeros((1, H, W), np.float32), radii=st["radii"],
    )
    final_T = np.zeros(N, np.float32)
    n_contrib = np.zeros(N, np.uint32)
    L.or_render_fwd(C.byref(cam), _p(ranges), _p(point_list), _p(st["xy"]), _p(st["conic_op"]), _p(st["rgb"]),
                    _p(st["depth"]), _p(rotations), _p(scales), _p(out["color"]), _p(out["depth"]),
                    _p(out["cov_quat"]), _p(out["cov_scale"]), _p(out["alpha"]), _p(final_T), _p(n_contrib))
    st.update(out_depth=out["depth"].copy(), R=R, keys=keys[:R], point_list=point_list[:R], ranges=ranges, final_T=final_T, n_contrib=n_contrib,
              inputs=dict(means3D=means3D, shs=shs, colors_precomp=colors_precomp, opacities=opacities,
                          scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp))
    return out, st


def backward(cam, st, dL_dcolor=None, dL_ddepth=None, dL_dquat=None, dL_dscale=None, dL_dalpha=None, abs_sums=False):
    """Backward of forward().  Returns dense input gradients (dict of numpy float32).
    abs_sums: also return, under "_per_gaussian"["abs"], the [P,17] sums of the absolute per-fragment terms."""
    L = lib()
    inp = st["inputs"]
    P = inp["means3D"].shape[0]
    g_mean2D = np.zeros((P, 2), np.float64)
    g_conic = np.zeros((P, 3), np.float64)
    g_opacity = np.zeros(P, np.float64)
    g_attr = np.zeros((P, NCH), np.float64)
    g_abs = np.zeros((P, 17), np.float64) if abs_sums else None
    dL_dcolor, dL_ddepth, dL_dquat = _f32(dL_dcolor), _f32(dL_ddepth), _f32(dL_dquat)
    dL_dscale, dL_dalpha = _f32(dL_dscale), _f32(dL_dalpha)
    pl = st["point_list"] if st["R"] > 0 else np.zeros(1, np.uint32)
    L.or_render_bwd(C.byref(cam), P, _p(st["ranges"]), _p(pl), _p(st["xy"]), _p(st["conic_op"]), _p(st["rgb"]),
                    _p(st["depth"]), _p(inp["rotations"]), _p(inp["scales"]), _p(st["final_T"]), _p(st["n_contrib"]),
                    _p(st["out_depth"]), _p(dL_dcolor), _p(dL_ddepth), _p(dL_dquat), _p(dL_dscale), _p(dL_dalpha),
                    _p(g_mean2D), _p(g_conic), _p(g_opacity), _p(g_attr), _p(g_abs))
    grads = preprocess_backward(cam, st, g_mean2D, g_conic, g_opacity, g_attr)
    grads["_per_gaussian"] = dict(mean2D=g_mean2D, conic=g_conic, opacity=g_opacity, attr=g_attr, abs=g_abs)
    return grads


def preprocess_backward(cam, st, g_mean2D, g_conic, g_opacity, g_attr, perturb=None):
    """Second half of backward(): the per-Gaussian sums of the render backward (float64) -> dense input gradients.
    perturb = (eps, seed, abs [P,17]): add eps * abs * N(0,1) to every sum first, abs = the sums of the absolute
    per-fragment terms (backward(abs_sums=True)) -- the model of what an fp32 summation of those terms carries.  The
    tests use it to MEASURE which rows are ill-conditioned with respect to the sums they start from (conic -> cov2D ->
    cov3D divides by det^2 of the 2D covariance; for edge-on 1e-5-thin discs such a perturbation moves the result by
    far more than 1e-3): those rows cannot be held to a tight tolerance by ANY fp32 implementation, this one included."""
    L = lib()
    inp = st["inputs"]
    P = inp["means3D"].shape[0]
    if perturb is not None:
        eps, seed, ab = perturb
        rng = np.random.default_rng(seed)
        g_mean2D = g_mean2D + eps * ab[:, 0:2] * rng.standard_normal(g_mean2D.shape)
        g_conic = g_conic + eps * ab[:, 2:5] * rng.standard_normal(g_conic.shape)
        g_opacity = g_opacity + eps * ab[:, 5] * rng.standard_normal(g_opacity.shape)
        g_attr = g_attr + eps * ab[:, 6:17] * rng.standard_normal(g_attr.shape)
    M = cam.M
    has_sh = inp["shs"] is not None
    has_sr = inp["scales"] is not None
    grads = dict(
        means3D=np.zeros((P, 3), np.float32),
        means2D=np.concatenate([g_mean2D.astype(np.float32), np.zeros((P, 1), np.float32)], axis=1),
        shs=np.zeros((P, M, 3), np.float32) if has_sh else None,
        colors_precomp=None if has_sh else np.zeros((P, 3), np.float32),
        opacities=g_opacity.astype(np.float32).reshape(P, 1),
        scales=np.zeros((P, 3), np.float32) if has_sr else None,
        rotations=np.zeros((P, 4), np.float32) if has_sr else None,
        cov3D_precomp=None if has_sr else np.zeros((P, 6), np.float32),
    )
    gm, gc, ga = g_mean2D.astype(np.float32), g_conic.astype(np.float32), g_attr.astype(np.float32)
    L.or_preprocess_bwd(C.byref(cam), P, _p(inp["means3D"]), _p(inp["shs"]), _p(inp["colors_precomp"]),
                        _p(inp["scales"]), _p(inp["rotations"]), _p(inp["cov3D_precomp"]), _p(st["radii"]),
                        _p(st["cov3D"]), _p(st["clamped"]), _p(gm), _p(gc), _p(ga),
                        _p(grads["means3D"]), _p(grads["shs"]), _p(grads["colors_precomp"]), _p(grads["scales"]),
                        _p(grads["rotations"]), _p(grads["cov3D_precomp"]))
    return grads
