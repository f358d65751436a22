/*
 * vegs_rast.h -- C ABI of libvegsrast.so: MI355X (gfx950) differentiable Gaussian-splatting
 * rasterizer, the drop-in for the native half of VEGS' `diff_gaussian_rasterization`
 * extension (un-vendored submodule, reference .gitmodules:7-9).
 *
 * What each entry point replaces (the reference binds these through the extension's
 * pybind module `_C`; its callers are cited):
 *   vr_forward       <- _C.rasterize_gaussians          called by GaussianRasterizer.forward,
 *                       reference gaussian_renderer/__init__.py:86-94, :230, :303
 *   vr_backward      <- _C.rasterize_gaussians_backward triggered by loss.backward(),
 *                       reference train.py:196
 *   vr_mark_visible  <- _C.mark_visible                 called by GaussianRasterizer.markVisible,
 *                       reference utils/norminit_utils.py:55, :179
 * Settings mirror the 12 fields of GaussianRasterizationSettings as filled at reference
 * gaussian_renderer/__init__.py:38-51.
 *
 * Conventions: plain C structs of DEVICE pointers (fp32, contiguous) plus scalars; no
 * exceptions cross the ABI; every function returns 0 on success or a negative VrStatus and
 * vr_last_error() describes the failure.  All work is enqueued on `stream` (a hipStream_t
 * passed as void*) of the CURRENT device.  vr_forward blocks the host once (to learn the
 * number of tile-list entries) exactly as the reference op does.  Memory is owned by the
 * caller: the library asks for buffers through VrAllocFn and never frees them.
 */
#ifndef VEGS_RAST_H
#define VEGS_RAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VR_ABI_VERSION 9

typedef enum VrStatus {
    VR_OK = 0,
    VR_ERR_INVALID_ARGUMENT = -1, /* shape / NULL / either-or violations (same rules as the reference shim) */
    VR_ERR_ALLOC = -2,            /* allocator callback returned NULL */
    VR_ERR_HIP = -3,              /* a HIP runtime call or kernel failed */
    VR_ERR_NO_DEVICE = -4
} VrStatus;

/* which buffer the library is asking for */
typedef enum VrBufferKind {
    VR_BUF_GEOM = 0,    /* per-Gaussian state, kept for backward  (O(P)) */
    VR_BUF_BINNING = 1, /* sorted tile lists, kept for backward   (O(R)) */
    VR_BUF_IMAGE = 2,   /* per-pixel state, kept for backward     (O(H*W)) */
    VR_BUF_SCRATCH = 3, /* transient; may be released when the call returns (several requests per call) */
    VR_BUF_BACKWARD = 4 /* per-Gaussian sums between vr_backward_render and vr_backward_preprocess (64 B per Gaussian) */
} VrBufferKind;

/* Must return a device pointer to at least `bytes` bytes, 256-byte aligned, valid on `stream`
 * (NULL = failure).  GEOM/BINNING/IMAGE must stay alive until the matching vr_backward. */
typedef void* (*VrAllocFn)(void* user, int kind, size_t bytes);

typedef struct VrSettings {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t sh_degree;   /* active degree 0..3 */
    int32_t prefiltered; /* accepted for API parity; unused (as upstream) */
    int32_t debug;       /* !=0: synchronise and check after every kernel */
    const float* bg;         /* device [3] */
    const float* viewmatrix; /* device [16], row-major 4x4, row-vector convention (scene/cameras.py:76) */
    const float* projmatrix; /* device [16], full_proj_transform (scene/cameras.py:87) */
    const float* campos;     /* device [3] */
    uint32_t flags;          /* OR of VrFlags; 0 = the documented defaults (ABI v4) */
} VrSettings;

/* VrSettings.flags.
 *
 * Bits 0-7: the FORK ASSUMPTIONS as switches.  The rasterizer VEGS pins (emjay73/diff_gaussian_rasterization_with_depth,
 * reference .gitmodules:7-9) is not vendored, so what exactly it blends into its extra outputs is inferred from the
 * call sites only (assumptions A-1..A-6 and the open questions of A.8).  Each open question
 * that would change training behaviour is a flag here, implemented in the kernels, the CPU oracle and the float64
 * autograd restatement alike; 0 selects the assumption SURVEY.md states.  A maintainer who can read the fork flips
 * the bit instead of rewriting a kernel.
 * Bit 8: execution mode of the backward pass (no effect on the forward).  Bit 9: execution mode of the forward
 * binning (no effect on any result). */
typedef enum VrFlags {
    /* A-3 / A.8(1): cov_scale blends scale_modifier * scales (the row the covariance is built from) instead of the
     * raw `scales` input row.  Identical in training, where the modifier is 1.0 (gaussian_renderer/__init__.py:263). */
    VR_FLAG_SCALE_MODIFIED = 1u << 0,
    /* A-1 / A.8(3): depth = sum(w z) / sum(w) with sum(w) = alpha = 1 - T_final (0 where nothing contributes)
     * instead of the un-normalised sum(w z). */
    VR_FLAG_DEPTH_NORMALIZED = 1u << 1,
    /* A.5 / A.8(2): the extra channels (depth, cov_quat, cov_scale) send gradients to their per-Gaussian attributes
     * only; nothing flows from them through alpha (opacity, conic, 2D mean).  Default: they flow through both. */
    VR_FLAG_EXTRA_NO_ALPHA_GRAD = 1u << 2,
    /* A-5 / A.8(4): cov_quat += T_final * (1,0,0,0) -- an identity "background" rotation, so that a pixel nothing
     * covers holds (1,0,0,0) instead of exact zeros (the reference's quaternion_to_matrix divides by |q|^2,
     * utils/graphics_utils.py:217, and returns NaN for zeros). */
    VR_FLAG_FILL_EMPTY = 1u << 3,
    /* Backward without floating-point atomics: every (tile-list entry, 8x8 pixel region) writes its partial sums to
     * its own slot and a second kernel adds the slots of each Gaussian in list order.  Gradients are then
     * bit-reproducible from run to run (and independent of scheduling); costs 272 bytes per list entry of scratch. */
    VR_FLAG_DETERMINISTIC = 1u << 8,
    /* Forward binning with the scan-based radix passes (histogram, device-wide scan, scatter: four launches per pass)
     * instead of the single-launch passes in which a workgroup waits -- bounded -- for sums posted by the workgroups
     * before it.  Same lists bit for bit; ~25 us per view slower at the headline size.  Also what inputs beyond 16.7 M
     * visible Gaussians or list entries use. */
    VR_FLAG_SCAN_BINNING = 1u << 9,
    /* (bits 10 and 11: tuning overrides of the forward's segment rounds, include/vegs_rast_debug.h -- not part of the stable set) */
    /* opacities / scales / rotations are the model's RAW parameters (scene/gaussian_model.py:_opacity, _scaling,
     * _rotation): the library applies the reference's activations itself -- sigmoid, exp, F.normalize (eps 1e-12),
     * scene/gaussian_model.py:37-45, the arithmetic of vr_activations_forward -- inside the preprocess kernel, and
     * dL_dopacities / dL_dscales / dL_drotations come back with respect to the RAW values.  Saves the two activation
     * launches and their 64 bytes per Gaussian each way in a training iteration.  Needs scales and rotations (not
     * cov3D_precomp).  With an SH tail (VrInputs.shs_tail) only the rows in front of tail_start are raw: the rows of the
     * box instances behind the static model arrive activated and transformed (gaussian_renderer/__init__.py:121-186). */
    VR_FLAG_RAW_PARAMS = 1u << 12,
    /* ABI v8.  The Gaussian's 2^x in the compositing kernels by the hardware's transcendental instruction (v_exp_f32,
     * ~1 ulp) instead of the bit-exact polynomial the CPU checker restates -- in the forward AND in the backward of the
     * view (the same instruction both ways: they agree on which fragments reach alpha >= 1/255).  Tile lists, radii and
     * ranges are unaffected (bit-exact); images move by ~1e-6, except that a fragment whose alpha sits within an ulp of
     * 1/255 (about one in 10^7) may be classified the other way, which moves its pixel by up to ~1/255 of a channel --
     * as any two correct implementations of exp() differ.  The bit-exact mode stays the default and the test mode. */
    VR_FLAG_FAST_EXP = 1u << 13,
    /* ABI v8.  The host waits for the forward's binning guard word before it queues the render stage and, if a bounded
     * look-back wait of the single-launch radix passes gave up, bins the view once more with the wait-free passes of
     * VR_FLAG_SCAN_BINNING (same lists): the view is rendered correctly instead of being failed at its backward.  Costs
     * the forward its run-ahead over the binning (about one launch latency of GPU idle per view); the default leaves a
     * tripped view empty and reports it (vr_backward: VR_ERR_HIP). */
    VR_FLAG_VERIFY_BINNING = 1u << 14,
    /* ABI v8.  TILE LISTS.  The reference emits one list entry per tile of a Gaussian's rectangle, and the rectangle
     * comes from the 3-sigma radius of the LARGER axis: on a street scene a third of those (Gaussian, tile) pairs cannot
     * reach alpha >= 1/255 at any pixel centre of the tile -- for every pixel the blend rule skips them.  By default the
     * library leaves such pairs out (rectangles of up to 64 tiles are tested tile by tile, larger ones in cells of k x k
     * tiles, with a conservative ellipse-vs-rectangle test written in IEEE basic operations, which the CPU checker restates
     * bit for bit): images, radii and gradients are what the full rectangles give -- radii exactly, images to rounding
     * (~5e-7: the sums are grouped by 256-entry list segments, which start at other entries; a pixel whose transmittance
     * sits within an ulp of the 1e-4 stop test may take one fragment of weight < 1e-4 more or fewer) -- while the lists the
     * sorts and the compositing kernels work on are a third shorter (more than half with large splats).  num_rendered,
     * n_contrib and vr_count_fragments then refer to the shorter lists.  This flag restores the reference's full
     * rectangles (for comparisons with the fork's internal buffers, or with BASELINE.md's definition of a fragment). */
    VR_FLAG_FULL_TILE_LISTS = 1u << 15,
    /* ABI v9.  Execution mode of the BACKWARD (ignored by vr_forward): every array of VrInGrads except dL_dmeans2D and the SH
     * factor dL_dcolors_sh -- which are per-view quantities and are overwritten as always -- RECEIVES this view's gradient
     * instead of being overwritten: grad[i] += g_view[i] (fp32, one add per element) for the rows with radii > 0; rows of
     * culled Gaussians are neither read nor written.  The caller owns the arrays across the views of a step (and their
     * clearing, or a first view without the flag).  A step that renders N views of one model otherwise pays per view the
     * dense write of (56 + 12 K) bytes per Gaussian AND the framework's accumulation of the same arrays (read, read,
     * write): 0.22 ms per view at 2 M Gaussians.  In the deterministic mode the result is bit-equal to adding the views'
     * dense gradients in call order.  Calls that add into the same arrays must be ordered by the caller (one stream, or
     * events between the streams). */
    VR_FLAG_ACCUMULATE_GRADS = 1u << 16
} VrFlags;

/* The op's tensor arguments (reference gaussian_renderer/__init__.py:86-94). Exactly one of
 * shs / colors_precomp and exactly one of (scales, rotations) / cov3D_precomp is non-NULL. */
typedef struct VrInputs {
    int32_t P;                   /* number of Gaussians */
    int32_t M;                   /* SH coefficients stored per Gaussian IN TOTAL (shs.shape[1], or 1 + shs_rest.shape[1]
                                    with split storage); 0 if shs == NULL */
    const float* means3D;        /* [P,3] */
    const float* shs;            /* [P,M,3] or NULL; with split storage: [P,1,3], the DC coefficient */
    const float* colors_precomp; /* [P,3] or NULL */
    const float* opacities;      /* [P,1] */
    const float* scales;         /* [P,3] or NULL */
    const float* rotations;      /* [P,4] (w,x,y,z) or NULL */
    const float* cov3D_precomp;  /* [P,6] or NULL */
    const float* shs_rest;       /* optional, split SH storage: [P,M-1,3], the remaining coefficients.  The reference's
                                    model keeps _features_dc / _features_rest apart and concatenates them on every
                                    call (get_features, scene/gaussian_model.py:112-116); passing the two tensors as
                                    they are saves that 2 x 384 MB copy per view at 2 M Gaussians.  NULL = shs is whole. */
    const float* shs_tail;       /* optional (ABI v6), SH TAIL: the SH rows of Gaussians tail_start .. P-1 live in this
                                    second, whole [P - tail_start, M, 3] tensor; shs (/ shs_rest) then hold only the
                                    first tail_start rows.  render_all / render_dyn put the dynamic instances' Gaussians
                                    behind the static model's (merge_kwargs, gaussian_renderer/__init__.py:182-186): with
                                    a tail the static model's SH tensors are read where they are instead of being
                                    concatenated with a few thousand instance rows (2 x 0.96 GB of copies per view at
                                    5 M Gaussians).  Needs M*3 % 4 == 0 and 16-byte aligned SH arrays.  NULL = no tail. */
    int64_t tail_start;          /* first Gaussian whose SH row is in shs_tail (0 <= tail_start <= P); ignored without */
} VrInputs;

typedef struct VrOutputs {
    float* color;     /* [3,H,W] */
    float* depth;     /* [1,H,W] */
    float* cov_quat;  /* [4,H,W] */
    float* cov_scale; /* [3,H,W] */
    float* alpha;     /* [1,H,W] */
    int32_t* radii;   /* [P] */
} VrOutputs;

/* Filled by vr_forward; pass it unchanged to vr_backward.
 * On ENTRY to vr_forward, binning_capacity may hold a hint: the expected number of tile-list entries
 * (e.g. the previous call's num_rendered plus some headroom; 0 = no hint).  With a hint the library
 * requests its R-sized buffers BEFORE the host synchronisation, so the device does not idle while the
 * host allocates; if the hint turns out too small the buffers are simply requested again. */
typedef struct VrSaved {
    void* geom;
    void* binning;
    void* image;
    int64_t num_rendered;     /* R: tile-list entries */
    int64_t num_visible;      /* V: Gaussians with at least one tile-list entry (= radii > 0 with VR_FLAG_FULL_TILE_LISTS) */
    int64_t binning_capacity; /* entries the binning buffer was laid out for (>= R) */
    uint32_t* needed_hint;    /* IN/OUT, optional (NULL = none): device array [tiles], tiles = ceil(W/16)*ceil(H/16) in
                                 row-major tile order, owned by the caller and kept PER CAMERA.  On entry: the number of
                                 256-entry list segments each tile needed in the previous forward of this camera at this
                                 image size (fill a new array with 0x3FFFFFFF = "no idea").  With a hint h for a tile
                                 the forward computes only the tile's first  h + 2 + h/8  segments up front (the hint
                                 plus a margin of two segments and 12 %; the array is read ONCE per call, by the first
                                 render kernel) and redoes the rest on the spot wherever that turns out too small --
                                 results never depend on the hint, bit-exact either way; only the time does (about half
                                 of the segments of a KITTI-shaped view are never needed).  The forward overwrites the
                                 array, asynchronously on `stream`, with what THIS forward needed: exactly the values
                                 vr_export_needed returns, i.e. per tile the number of leading segments in which at
                                 least one pixel was still alive. */
    uint64_t ticket;          /* OUT (ABI v5): identifies this forward's slot in the library's pinned guard ring.  The
                                 single-launch binning passes bound every wait for another workgroup's posted sum; the
                                 last binning kernel posts whether one ran out, and vr_backward / vr_count_* /
                                 vr_export_needed / vr_debug_export_binning called with this VrSaved return VR_ERR_HIP
                                 for exactly this view (0 = nothing to check).  Pass it on unchanged.
                                 ABI v9: every forward in flight has its OWN device guard word (views on different
                                 streams cannot fail each other).  vr_backward asks for the slot without blocking; not
                                 posted yet = it queues its kernels (a failed view's tile ranges are empty: they compute
                                 zeros) and then WAITS for the slot before it returns (round 6: the forward's last binning
                                 kernel lies in front of everything the call queued, so healthy runs lose nothing, and a
                                 failed view always fails its OWN backward).  A view's binning fails when a bounded wait
                                 runs out or when the depth sort's output is not a permutation of the visible ids (an
                                 always-on checksum comparison); a view that never gets a backward is reported by the
                                 thread's next vr_forward. */
} VrSaved;

/* Incoming gradients, one per differentiable output (NULL = zero). */
typedef struct VrOutGrads {
    const float* dL_dcolor;     /* [3,H,W] */
    const float* dL_ddepth;     /* [1,H,W] */
    const float* dL_dcov_quat;  /* [4,H,W] */
    const float* dL_dcov_scale; /* [3,H,W] */
    const float* dL_dalpha;     /* [1,H,W] */
} VrOutGrads;

/* Dense gradients w.r.t. the inputs; the library fully overwrites every non-NULL array
 * (zeros for culled Gaussians) -- or, with VR_FLAG_ACCUMULATE_GRADS, adds into them (see VrFlags).  dL_dmeans2D is [P,3] with z == 0 and x,y = d loss / d NDC
 * (the quantity scene/gaussian_model.py:411-413 norms).  Arrays whose input was NULL are NULL. */
typedef struct VrInGrads {
    float* dL_dmeans3D;        /* [P,3] */
    float* dL_dmeans2D;        /* [P,3] */
    float* dL_dshs;            /* [P,M,3] or NULL; with split storage [P,1,3] */
    float* dL_dcolors_precomp; /* [P,3] or NULL */
    float* dL_dopacities;      /* [P,1] */
    float* dL_dscales;         /* [P,3] or NULL */
    float* dL_drotations;      /* [P,4] or NULL */
    float* dL_dcov3D_precomp;  /* [P,6] or NULL */
    float* dL_dshs_rest;       /* [P,M-1,3], required when VrInputs.shs_rest is given ([tail_start,M-1,3] with a tail) */
    float* dL_dcolors_sh;      /* optional (SH mode, ABI v4): FACTORED SH gradient.  dL/dshs of one view is a rank-1
                                  product per Gaussian, dL_dshs[i][k][c] = basis_k(dir_i) * g[i][c] with g = dL/d(colour)
                                  zeroed where the colour was clamped.  When this [P,3] array is given the library
                                  writes g into it (zeros for culled Gaussians) and does NOT write dL_dshs /
                                  dL_dshs_rest (they may be NULL): 12 instead of 192 bytes per Gaussian leave the
                                  kernel, and a multi-GPU job exchanges 3 instead of 48 floats per Gaussian and view
                                  (vegs_optim.h: vr_sh_grad_from_factors / vr_sh_adam_step rebuild or consume it). */
    float* dL_dshs_tail;       /* [P - tail_start, M, 3], required when VrInputs.shs_tail is given (unless the factored
                                  gradient is requested); dL_dshs / dL_dshs_rest then have tail_start rows */
} VrInGrads;

/* Work counters of the most recent vr_forward on this thread (roofline accounting). */
typedef struct VrCounters {
    int64_t P;           /* Gaussians */
    int64_t num_visible; /* V */
    int64_t num_rendered;/* R */
    int64_t num_tiles;   /* T */
    int64_t num_pixels;  /* N */
} VrCounters;

int vr_abi_version(void);
const char* vr_last_error(void);

int vr_forward(const VrSettings* settings, const VrInputs* in, const VrOutputs* out,
               VrAllocFn alloc, void* alloc_user, void* stream, VrSaved* saved);

int vr_backward(const VrSettings* settings, const VrInputs* in, const int32_t* radii,
                const VrSaved* saved, const VrOutGrads* gout, const VrInGrads* gin,
                VrAllocFn alloc, void* alloc_user, void* stream);

/* The backward in TWO calls (ABI v5), for callers that have something to do in between -- a multi-GPU job starts the
 * exchange of the SH factors (VrInGrads.dL_dcolors_sh) while the second half still runs:
 *   vr_backward_render      zeroing, render backward (the per-pixel -> per-Gaussian sums, dL_dmeans2D) and, with the
 *                           factored SH gradient, dL_dcolors_sh -- COMPLETE when this call's work is (stream order);
 *                           *state receives a VR_BUF_BACKWARD buffer to hand to the second call
 *   vr_backward_preprocess  every other gradient (means3D, opacities, scales, rotations, cov3D, dense shs / colours)
 * Same arguments and results as vr_backward, which is the two in a row. */
int vr_backward_render(const VrSettings* settings, const VrInputs* in, const int32_t* radii, const VrSaved* saved,
                       const VrOutGrads* gout, const VrInGrads* gin, VrAllocFn alloc, void* alloc_user, void* stream,
                       void** state);
int vr_backward_preprocess(const VrSettings* settings, const VrInputs* in, const int32_t* radii, const VrSaved* saved,
                           const VrInGrads* gin, void* state, void* stream);

/* present[i] = view-space z of xyz[i] > 0.2 (no screen-bounds test). */
int vr_mark_visible(const float* xyz, int32_t P, const float* viewmatrix, const float* projmatrix,
                    uint8_t* present, void* stream);

/* Replacement for the reference's second native dependency, simple_knn._C.distCUDA2 (imported at
 * scene/gaussian_model.py:21, called at :140 and :517): out[i] = mean of the squared distances from
 * points[i] to its three nearest other points (exact).  points [N,3], out [N]; scratch via `alloc`. */
int vr_knn3_mean_dist2(const float* points, int32_t N, float* out, VrAllocFn alloc, void* alloc_user, void* stream);

void vr_get_counters(VrCounters* out);

/* Copies the per-tile number of needed list segments of the forward whose state is `saved` into `out`
 * (device, uint32 [tiles], tiles = ceil(W/16) * ceil(H/16)); asynchronous on `stream`.  Feed it back as
 * VrSaved.needed_hint of the next forward of the same camera. */
int vr_export_needed(const VrSaved* saved, int32_t image_height, int32_t image_width, uint32_t* out, void* stream);

/* Benchmark bookkeeping (vr_count_*), stage timers (vr_profile_*), test hooks (vr_debug_*) and the tuning overrides of the
 * segment rounds are declared in include/vegs_rast_debug.h: EXPERIMENTAL, no binding of a product should use them. */

#ifdef __cplusplus
}
#endif
#endif /* VEGS_RAST_H */
