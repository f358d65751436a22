/* vegs_instances.h -- C ABI of the fused instance transform + concatenation that feeds the rasterizer when
 * dynamic box instances are in frame.  Same library and conventions as
 * vegs_rast.h.
 *
 * Reference interface replaced: the box2world branch of prepare_rasterization and merge_kwargs for
 * means3D / scales / rotations (gaussian_renderer/__init__.py:122-126, :140-153, :182-186), as render_all /
 * render_dyn call them once per instance (:199-230, :274-303):
 *     means'     = homogeneous(box2world @ [x;1])
 *     S, Rb      = decompose_T_to_RS(box2world)                    utils/graphics_utils.py:49-53
 *     rotations' = matrix_to_quaternion(Rb @ quaternion_to_matrix(rotations))   utils/graphics_utils.py:140-248
 *     scales'    = scales * S
 *     out        = torch.cat over the static model and all instances
 */
#ifndef VEGS_INSTANCES_H
#define VEGS_INSTANCES_H

#include "vegs_rast.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct VrInstance {
    const float* means;     /* [n,3] */
    const float* scales;    /* [n,3] */
    const float* rotations; /* [n,4] (w,x,y,z) */
    const float* box2world; /* device [16], row-major 4x4; NULL = identity (the static model: plain copy) */
    int64_t n;
    int64_t offset;         /* first row of this instance in the concatenated outputs */
} VrInstance;

/* Gradient destinations of one instance (device; all four required when the instance has a box2world). */
typedef struct VrInstanceGrads {
    float* dL_dmeans;     /* [n,3] */
    float* dL_dscales;    /* [n,3] */
    float* dL_drotations; /* [n,4] */
    float* dL_dbox2world; /* [16] */
} VrInstanceGrads;

/* `inst` is a HOST array.  Writes rows [offset, offset+n) of the three concatenated outputs for every instance;
 * one kernel launch per 16 instances. */
int vr_instances_forward(const VrInstance* inst, int32_t count, float* out_means, float* out_scales,
                         float* out_rotations, void* stream);

/* g_* are the gradients of the concatenated outputs.  Instances with box2world == NULL are skipped (their
 * gradients are the corresponding rows of g_*).  Deterministic: per-block partial sums of the box2world
 * gradient are reduced in a fixed order. */
int vr_instances_backward(const VrInstance* inst, const VrInstanceGrads* grads, int32_t count, const float* g_means,
                          const float* g_scales, const float* g_rotations, VrAllocFn alloc, void* alloc_user,
                          void* stream);

/* ---- the model's activations, the first thing prepare_rasterization reads (gaussian_renderer/__init__.py:128-137 ->
 * scene/gaussian_model.py:98-120 with the functions of :37-45):
 *     opacity   = torch.sigmoid(_opacity)                 [P,1]
 *     scales    = torch.exp(_scaling)                     [P,3]
 *     rotations = torch.nn.functional.normalize(_rotation) [P,4]   x / max(|x|_2, 1e-12) per row
 * one launch forward, one backward (ATen: ~15 launches over all Gaussians per iteration).  Rotation arrays must be
 * 16-byte aligned. */
int vr_activations_forward(const float* raw_opacity, const float* raw_scaling, const float* raw_rotation, int64_t P,
                           float* opacity, float* scales, float* rotations, void* stream);

/* Gradients w.r.t. the raw parameters from the gradients of the activated ones.  `opacity` and `scales` are the
 * forward's OUTPUTS, `raw_rotation` its input.  A NULL g_* means a zero gradient; a NULL dL_draw_* is not computed. */
int vr_activations_backward(const float* opacity, const float* scales, const float* raw_rotation, int64_t P,
                            const float* g_opacity, const float* g_scales, const float* g_rotations,
                            float* dL_draw_opacity, float* dL_draw_scaling, float* dL_draw_rotation, void* stream);

#ifdef __cplusplus
}
#endif
#endif
