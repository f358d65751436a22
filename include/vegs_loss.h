/* vegs_loss.h -- C ABI of the fused per-pixel losses that follow the rasterizer in a VEGS training
 * iteration.  Same library (libvegsrast.so), same conventions as
 * vegs_rast.h: plain device pointers and sizes, 0 / negative error code, vr_last_error() for the text,
 * scratch through the caller's VrAllocFn, kernels on the caller's stream, no host synchronisation
 * (scalar results and upstream scalar gradients live in DEVICE memory).
 *
 * Reference interfaces replaced:
 *   l1_loss(network_output, gt)                    utils/loss_utils.py:18-22   (call: train.py:162)
 *   ssim(img1, img2)  window 11, size_average      utils/loss_utils.py:39-79   (call: train.py:164)
 *   loss_normal_guidance(cam, cov_quat, cov_scale) loss/normal_guidance.py:3-22 (call: train.py:167)
 */
#ifndef VEGS_LOSS_H
#define VEGS_LOSS_H

#include "vegs_rast.h"

#ifdef __cplusplus
extern "C" {
#endif

/* image, gt: [C,H,W] fp32.  sums[0] = mean |image-gt|, sums[1] = mean SSIM map (device, 2 floats).
 * dmaps: [3,C,H,W] fp32 workspace kept for vr_photometric_backward (the SSIM map's partial derivatives
 * w.r.t. the three window moments of `image`), or NULL when no gradient is wanted. */
int vr_photometric_forward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, float* sums,
                           float* dmaps, VrAllocFn alloc, void* alloc_user, void* stream);

/* dL_dimage [C,H,W] = g_l1 * d(l1 mean)/dimage + g_ssim * d(ssim mean)/dimage; g_l1 / g_ssim are device
 * scalars (NULL = 0), e.g. (1-lambda) and -lambda times the upstream gradient for train.py:164. */
int vr_photometric_backward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, const float* dmaps,
                            const float* g_l1, const float* g_ssim, float* dL_dimage, void* stream);

/* cov_quat [4,H,W], cov_scale [3,H,W] (rasterizer outputs), normal [3,H,W] (camera-frame normal map),
 * R_cam2world: 9 HOST floats, row-major (viewpoint_cam.R).  loss: device scalar. */
int vr_normal_guidance_forward(const float* cov_quat, const float* cov_scale, const float* normal,
                               const float* R_cam2world, int32_t H, int32_t W, float* loss, VrAllocFn alloc,
                               void* alloc_user, void* stream);

/* dL_dquat [4,H,W], dL_dscale [3,H,W] = g * d loss / d(...); g: device scalar (required). */
int vr_normal_guidance_backward(const float* cov_quat, const float* cov_scale, const float* normal,
                                const float* R_cam2world, int32_t H, int32_t W, const float* g, float* dL_dquat,
                                float* dL_dscale, void* stream);

/* The whole loss block of an iteration (train.py:162-168) in one call each way:
 *   Ll1 = l1_loss(image, gt);  loss = (1 - lambda_dssim) * Ll1 + lambda_dssim * (1 - ssim(image, gt));
 *   loss += lambda_dnormal * loss_normal_guidance(cam, cov_quat, cov_scale)
 * Three launches forward (the two losses' partial sums, one combining reduction), two backward, no scalar glue kernels
 * in between.  loss: device scalar; aux: device float[3] = { Ll1, mean SSIM, Lng } (train.py logs Ll1); dmaps as for
 * vr_photometric_forward (NULL when no gradient is wanted).  guard_empty: pixels no Gaussian covers (cov_quat == 0, where
 * the reference's quaternion_to_matrix returns NaN, utils/graphics_utils.py:217) are evaluated with q = (1,1,1,1) and get
 * no quaternion gradient -- what `torch.where(|q|^2 > 0, q, 1)` in front of the loss does. */
int vr_training_loss_forward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, const float* cov_quat,
                             const float* cov_scale, const float* normal, const float* R_cam2world, float lambda_dssim,
                             float lambda_dnormal, int32_t guard_empty, float* loss, float* aux, float* dmaps,
                             VrAllocFn alloc, void* alloc_user, void* stream);

/* g: device scalar dL/dloss.  dL_dimage [C,H,W], dL_dquat [4,H,W], dL_dscale [3,H,W]. */
int vr_training_loss_backward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, const float* dmaps,
                              const float* cov_quat, const float* cov_scale, const float* normal, const float* R_cam2world,
                              float lambda_dssim, float lambda_dnormal, int32_t guard_empty, const float* g,
                              float* dL_dimage, float* dL_dquat, float* dL_dscale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
