/* vegs_optim.h -- C ABI of the fused per-Gaussian update that follows loss.backward() in a VEGS iteration
 * (SURVEY.md section 8f, row N2).  Same library and conventions as vegs_rast.h.
 *
 * Reference interfaces replaced:
 *   torch.optim.Adam(l, lr=0.0, eps=1e-15).step()   scene/gaussian_model.py:159-168 (groups xyz, f_dc, f_rest,
 *                                                   opacity, scaling, rotation), stepped at train.py:319
 *   add_densification_stats + max_radii2D update    scene/gaussian_model.py:411-413, train.py:299-300
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
} VrAdamTensor;

/* Adam (no weight decay, no amsgrad), the arithmetic of torch.optim.Adam's default path:
 *   m += (g - m) * (1 - beta1);  v = v * beta2 + (1 - beta2) * g * g;
 *   p -= (lr / (1 - beta1^step)) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * `tensors` is a HOST array; all tensors are updated by one kernel launch per 8 tensors. */
int vr_adam_step(const VrAdamTensor* tensors, int32_t count, double beta1, double beta2, double eps, void* stream);

/* For every Gaussian with radii > 0:  xyz_gradient_accum += ||means2D_grad[:2]||;  denom += 1;
 * max_radii2D = max(max_radii2D, radii).  means2D_grad [P,3], radii int32 [P], the three statistics fp32 [P]. */
int vr_densify_stats(const float* means2D_grad, const int32_t* radii, int32_t P, float* xyz_gradient_accum,
                     float* denom, float* max_radii2D, void* stream);

/* ---- factored SH gradients (VrInGrads.dL_dcolors_sh of vegs_rast.h).
 * Dense gradient from the factors of n_views views (what a view-sharded job all-gathers: 3 floats per Gaussian and
 * view instead of all-reducing 48):
 *   dL_dshs[i][k][c] = scale * sum_v basis_k(normalize(means3D[i] - campos[v])) * factors[v][i][c]   k < (deg+1)^2,
 * zeros for the inactive coefficients.  campos [n_views,3], factors [n_views,P,3], all on the device.  Output either
 * whole (dL_dshs [P,M,3], dL_dshs_rest NULL) or split as the model stores it (dL_dshs [P,1,3] + dL_dshs_rest [P,M-1,3]). */
int vr_sh_grad_from_factors(const float* means3D, int32_t P, const float* campos, const float* factors, int32_t n_views,
                            int32_t sh_degree, int32_t M, float scale, float* dL_dshs, float* dL_dshs_rest, void* stream);

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
                    int32_t sh_degree, int32_t M, float scale, const VrShAdamTensor* dc, const VrShAdamTensor* rest,
                    double beta1, double beta2, double eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif
