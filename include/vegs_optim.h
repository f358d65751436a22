/* vegs_optim.h -- C ABI of the fused per-Gaussian update that follows loss.backward() in a VEGS iteration
 * (SURVEY.md section 8f, row N2).  Same library and conventions as vegs_rast.h.
 *
 * Reference interfaces replaced:
 *   torch.optim.Adam(l, lr=0.0, eps=1e-15).step()   scene/gaussian_model.py:159-168 (groups xyz, f_dc, f_rest,
 *                                                   opacity, scaling, rotation), stepped at train.py:319
 *   add_densification_stats + max_radii2D update    scene/gaussian_model.py:411-413, train.py:299-300
 *   densify_and_prune (clone + split + prune, with   scene/gaussian_model.py:278-403, called at train.py:312
 *     the optimizer-state surgery), reset_opacity    scene/gaussian_model.py:215-218, called at train.py:315
 */
#ifndef VEGS_OPTIM_H
#define VEGS_OPTIM_H

#include "vegs_rast.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One optimised tensor (flat fp32 arrays of n elements, all on the device). */
typedef struct VrAdamTensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    double lr;      /* the group's learning rate at this step */
    int64_t step;   /* this tensor's step count AFTER the increment (>= 1) */
    double eps;     /* this tensor's eps; negative = the call's `eps` (lets optimizers with different eps -- the models'
                       1e-15, scene/gaussian_model.py:168, and the BoxModels' default 1e-8, model/boxmodel.py:13 -- share a launch) */
} VrAdamTensor;

/* Adam (no weight decay, no amsgrad), the arithmetic of torch.optim.Adam's default path:
 *   m += (g - m) * (1 - beta1);  v = v * beta2 + (1 - beta2) * g * g;
 *   p -= (lr / (1 - beta1^step)) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * `tensors` is a HOST array; all tensors are updated by one kernel launch per 64 tensors (the static model's six, every
 * in-frame instance model's six and every BoxModel's three of a frame with dynamic objects: train.py:254-275 as ONE
 * launch up to 64 tensors). */
int vr_adam_step(const VrAdamTensor* tensors, int32_t count, double beta1, double beta2, double eps, void* stream);

/* For every Gaussian with radii > 0:  xyz_gradient_accum += ||means2D_grad[:2]||;  denom += 1;
 * max_radii2D = max(max_radii2D, radii).  means2D_grad [P,3], radii int32 [P], the three statistics fp32 [P]. */
int vr_densify_stats(const float* means2D_grad, const int32_t* radii, int32_t P, float* xyz_gradient_accum,
                     float* denom, float* max_radii2D, void* stream);

/* ---- densification (scene/gaussian_model.py:384-403 densify_and_prune).  The reference's sequence of concatenations and
 * mask gathers amounts to one gather: every output row copies one input row.  In output order:
 *   A  the originals that are neither split nor pruned          parameters and Adam moments carried along
 *   B  one clone of every original with |g| >= max_grad whose largest scale <= percent_dense * extent      moments 0
 *   C1, C2  the two samples of every original with g >= max_grad and a larger scale (copy-major):
 *           xyz = R(rotation) (noise * exp(scaling)) + xyz,  scaling = log(exp(scaling) / 1.6),  moments 0
 * g = xyz_gradient_accum / denom with NaN -> 0.  Pruned (all rows, after the above): sigmoid(opacity) < min_opacity and,
 * if prune_big, largest scale > 0.1 * extent.  (prune_big = the reference's `max_screen_size` being truthy; its
 * screen-size test max_radii2D > max_screen_size can never fire, because densification_postfix, :349-351, has reset
 * max_radii2D to zeros by then.)  The caller zeroes xyz_gradient_accum / denom / max_radii2D at the new length. */
typedef struct VrDensifySettings {
    double max_grad;       /* densify_grad_threshold (train.py:305-312) */
    double min_opacity;    /* 0.005 at train.py:312 */
    double extent;         /* scene.cameras_extent */
    double percent_dense;  /* training_args.percent_dense (scene/gaussian_model.py:155) */
    int32_t prune_big;     /* 0: prune by opacity only; 1: also by world size; 2: the reference's prune=False (:394,:397) --
                              clone and split only, no Gaussian is pruned (the split originals are still replaced) */
} VrDensifySettings;

/* int32 words of the `plan` buffer for P Gaussians */
int64_t vr_densify_plan_words(int32_t P);

/* Classify and lay out.  opacity [P], scaling [P,3] raw parameters; the statistics [P].  plan: device buffer of
 * vr_densify_plan_words(P) int32.  counts: device int32[5] = { rows out, A, B, C, S } with S = number of split originals =
 * rows per copy of the caller's unit-normal draw `noise` [2 S, 3] (torch.normal's, :367). */
int vr_densify_plan(const float* opacity, const float* scaling, const float* xyz_gradient_accum, const float* denom,
                    int32_t P, const VrDensifySettings* settings, int32_t* plan, int32_t* counts, void* stream);

/* One model tensor [P, width] -> [rows out, width], with its Adam moments (all four pointers or none).
 * role: 0 copied columns, 1 the position tensor, 2 the scaling tensor (the computed columns of the split samples). */
typedef struct VrDensifyTensor {
    const float* src;   float* dst;
    const float* m_src; float* m_dst;   /* exp_avg */
    const float* v_src; float* v_dst;   /* exp_avg_sq */
    int32_t width;
    int32_t role;
} VrDensifyTensor;

/* Gather: n_out = counts[0], n_split = counts[4] (the caller has read them back to size the outputs).  scaling [P,3],
 * rotation [P,4] (16-byte aligned) = the INPUT model's raw parameters, noise [2 n_split, 3].  `tensors` is a HOST array;
 * one launch per tensor moves the parameter and both moments. */
int vr_densify_apply(const int32_t* plan, int32_t n_out, int32_t n_split, const VrDensifyTensor* tensors, int32_t count,
                     const float* scaling, const float* rotation, const float* noise, void* stream);

/* reset_opacity (scene/gaussian_model.py:215-218): opacity = logit(min(sigmoid(opacity), cap)) in place, cap = 0.01 in the
 * reference; the moments (both or none) are zeroed as replace_tensor_to_optimizer does. */
int vr_reset_opacity(float* opacity, float* exp_avg, float* exp_avg_sq, int64_t P, float cap, void* stream);

/* ---- factored SH gradients (VrInGrads.dL_dcolors_sh of vegs_rast.h).
 * Dense gradient from the factors of n_views views (what a view-sharded job all-gathers: 3 floats per Gaussian and
 * view instead of all-reducing 48):
 *   dL_dshs[i][k][c] = scale * sum_v basis_k(normalize(means3D[i] - campos[v])) * factors[v][i][c]   k < (deg+1)^2,
 * zeros for the inactive coefficients.  campos [n_views,3], factors [n_views,P,3], all on the device;
 * factor_view_stride = floats between the blocks of consecutive views (0 = packed, 3 P; larger when the P rows are the
 * head of longer per-view blocks, e.g. the static model's rows of an all-gathered [n_views, P + instance rows, 3]).  Output either
 * whole (dL_dshs [P,M,3], dL_dshs_rest NULL) or split as the model stores it (dL_dshs [P,1,3] + dL_dshs_rest [P,M-1,3]). */
int vr_sh_grad_from_factors(const float* means3D, int32_t P, const float* campos, const float* factors, int32_t n_views,
                            int64_t factor_view_stride, int32_t sh_degree, int32_t M, float scale, float* dL_dshs,
                            float* dL_dshs_rest, void* stream);

/* One SH tensor of the optimizer (same meaning as VrAdamTensor, the gradient being implicit). */
typedef struct VrShAdamTensor {
    float* param;
    float* exp_avg;
    float* exp_avg_sq;
    double lr;
    int64_t step;
} VrShAdamTensor;

/* Adam step of the SH tensors straight from the factors: the gradient of vr_sh_grad_from_factors is built per Gaussian
 * in registers and consumed at once -- the [P,M,3] gradient (384 MB at 2 M Gaussians) is never written or read.
 * `dc` = f_dc [P,1,3] and `rest` = f_rest [P,M-1,3] (scene/gaussian_model.py:159-166), or dc = the whole [P,M,3] tensor
 * and rest = NULL.  Arithmetic = vr_adam_step's. */
int vr_sh_adam_step(const float* means3D, int32_t P, const float* campos, const float* factors, int32_t n_views,
                    int64_t factor_view_stride, int32_t sh_degree, int32_t M, float scale, const VrShAdamTensor* dc,
                    const VrShAdamTensor* rest, double beta1, double beta2, double eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif
