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


/* ---- BoxModel (model/boxmodel.py:6-49): per-instance learnable pose correction, optimised at train.py:270-274.
 *     D        = [[diag(delta_s) @ quaternion_to_matrix(delta_r), delta_t], [0 0 0 1]]        d_box2world, :30-38
 *     adjusted = box2world @ D                                                                  adjustbox2world, :40-42
 * All instances of a frame in ONE launch each way (one thread per instance; the reference: ~15 ATen launches per instance
 * forward, ~30 backward).  `boxes` / `grads` are HOST arrays of device pointers. */
typedef struct VrBoxModel {
    const float* box2world; /* [16] the annotated pose (obj_box2world, model/boxmodel.py:16-21); unused by the regularizer */
    const float* delta_r;   /* [4] (w,x,y,z), NOT normalised (quaternion_to_matrix divides by |q|^2) */
    const float* delta_s;   /* [3] */
    const float* delta_t;   /* [3] */
} VrBoxModel;

typedef struct VrBoxModelGrads {
    float* d_delta_r; /* [4] */
    float* d_delta_s; /* [3] */
    float* d_delta_t; /* [3] */
} VrBoxModelGrads;

/* adjusted [count,16] (row-major 4x4 each) */
int vr_boxmodel_forward(const VrBoxModel* boxes, int32_t count, float* adjusted, void* stream);

/* g_adjusted [count,16] = dL/d(adjusted); the three gradients of every instance are OVERWRITTEN.  nan_guard != 0 applies
 * train.py:199-205: if dL/d(delta_r) or dL/d(delta_s) holds a NaN, all three gradients of that instance become zeros. */
int vr_boxmodel_backward(const VrBoxModel* boxes, const VrBoxModelGrads* grads, int32_t count, const float* g_adjusted,
                         int32_t nan_guard, void* stream);

/* Gradient of BoxModel.regularize's loss (model/boxmodel.py:44-46)
 *     lambda_reg * (|delta_r - (1,0,0,0)| + |delta_s - 1| + |delta_t|)
 * as autograd returns it (x * lambda / |x|; zeros where the norm is 0), written over the three gradients. */
int vr_boxmodel_regularizer_grad(const VrBoxModel* boxes, const VrBoxModelGrads* grads, int32_t count, float lambda_reg,
                                 void* stream);

#ifdef __cplusplus
}
#endif
#endif
