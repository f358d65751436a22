// preprocess.hip -- per-Gaussian forward: near-plane cull, projection, cov3D -> EWA cov2D -> conic,
// integer radius + tile rectangle, SH colour.  One lane per Gaussian; visible Gaussians write an
// 80-byte Splat record that the render kernels gather with dwordx4 loads.
//
// Replaces the native "preprocess" stage behind GaussianRasterizer.forward (reference call site
// gaussian_renderer/__init__.py:86-94).  In-repo definitions of the same math: SH colour
// utils/sh_utils.py:57-112 and the +0.5 / clamp at gaussian_renderer/__init__.py:79-80; cov3D
// utils/general_utils.py:97-129; camera matrices scene/cameras.py:76-88.
#include "vr_host.h"

namespace vr {

__global__ void __launch_bounds__(256)
k_preprocess(Camera cam, int P, const float* __restrict__ means3D, const float* __restrict__ shs,
             const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
             const float* __restrict__ scales, const float* __restrict__ rotations,
             const float* __restrict__ cov3D_precomp, Splat* __restrict__ rec, int* __restrict__ radii,
             uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ depth_key)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    // culled unless proven visible
    int rad_out = 0;
    uint32_t tiles_out = 0;
    uint32_t key_out = 0;

    float px3 = means3D[3 * i], py3 = means3D[3 * i + 1], pz3 = means3D[3 * i + 2];
    float t0, t1, t2;
    xform43(cam.view, px3, py3, pz3, t0, t1, t2);
    if (t2 > NEAR_Z) {
        float h0, h1, h2;
        xform43(cam.proj, px3, py3, pz3, h0, h1, h2);
        float hw = xform_w(cam.proj, px3, py3, pz3);
        float pw = 1.0f / (hw + 0.0000001f);
        float ndcx = h0 * pw, ndcy = h1 * pw;
        float c6[6];
        float q[4] = {0.f, 0.f, 0.f, 0.f}, sc[3] = {0.f, 0.f, 0.f};
        if (cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * (size_t)i + k];
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) sc[k] = scales[3 * (size_t)i + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = rotations[4 * (size_t)i + k];
            cov3d_from_scale_rot(sc, cam.mod, q, c6);
        }
        Cov2D cv;
        cov2d(cam, cam.view, t0, t1, t2, c6, cv);
        float det = cv.a * cv.c - cv.b * cv.b;
        if (det != 0.0f) {
            float det_inv = 1.0f / det;
            float mid = 0.5f * (cv.a + cv.c);
            float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
            float lam = fmaxf(mid + sq, mid - sq);
            int rad = (int)ceilf(3.0f * sqrtf(lam));
            float px = ndc2pix(ndcx, cam.W), py = ndc2pix(ndcy, cam.H);
            int x0, y0, x1, y1;
            tile_rect(px, py, rad, cam.gx, cam.gy, x0, y0, x1, y1);
            int ntiles = (x1 - x0) * (y1 - y0);
            if (ntiles != 0) {
                float rgb[3];
                uint32_t clampbits = 0;
                if (colors_precomp) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) rgb[c] = colors_precomp[3 * (size_t)i + c];
                } else {
                    float dx = px3 - cam.campos[0], dy = py3 - cam.campos[1], dz = pz3 - cam.campos[2];
                    float len = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
                    dx = dx / len; dy = dy / len; dz = dz / len;
                    float b[16];
                    sh_basis(cam.deg, dx, dy, dz, b);
                    int K = (cam.deg + 1) * (cam.deg + 1);
                    const float* sh = shs + (size_t)i * cam.M * 3;
                    float acc0 = b[0] * sh[0], acc1 = b[0] * sh[1], acc2 = b[0] * sh[2];
                    for (int k = 1; k < K; ++k) {
                        acc0 = fmaf(b[k], sh[3 * k + 0], acc0);
                        acc1 = fmaf(b[k], sh[3 * k + 1], acc1);
                        acc2 = fmaf(b[k], sh[3 * k + 2], acc2);
                    }
                    acc0 += 0.5f; acc1 += 0.5f; acc2 += 0.5f;
                    clampbits = (acc0 < 0.0f ? 1u : 0u) | (acc1 < 0.0f ? 2u : 0u) | (acc2 < 0.0f ? 4u : 0u);
                    rgb[0] = fmaxf(acc0, 0.0f); rgb[1] = fmaxf(acc1, 0.0f); rgb[2] = fmaxf(acc2, 0.0f);
                }
                Splat s;
                const float opac = opacities[i];
                s.x = px; s.y = py; s.conA = cv.c * det_inv; s.conB = -cv.b * det_inv;
                s.conC = cv.a * det_inv; s.opacity = opac; s.thr = splat_thr(opac); s.depth = t2;
                s.r = rgb[0]; s.g = rgb[1]; s.b = rgb[2]; s.qw = q[0];
                s.qx = q[1]; s.qy = q[2]; s.qz = q[3]; s.s0 = sc[0];
                s.s1 = sc[1]; s.s2 = sc[2]; s.clamped = clampbits; s.pad0 = 0;
                float4* dst = reinterpret_cast<float4*>(rec + i);
                const float4* src = reinterpret_cast<const float4*>(&s);
#pragma unroll
                for (int k = 0; k < 5; ++k) dst[k] = src[k];
                rad_out = rad;
                tiles_out = (uint32_t)ntiles;
                key_out = __float_as_uint(t2);
            }
        }
    }
    radii[i] = rad_out;
    tiles_touched[i] = tiles_out;
    depth_key[i] = key_out;
}

__global__ void __launch_bounds__(256)
k_mark_visible(const float* __restrict__ xyz, int P, const float* __restrict__ view, uint8_t* __restrict__ present)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float t0, t1, t2;
    xform43(view, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], t0, t1, t2);
    present[i] = t2 > NEAR_Z ? 1 : 0;
}

int launch_preprocess(const Camera& cam, int P, const float* means3D, const float* shs, const float* colors_precomp,
                      const float* opacities, const float* scales, const float* rotations,
                      const float* cov3D_precomp, Splat* rec, int* radii, uint32_t* tiles_touched,
                      uint32_t* depth_key, hipStream_t s, bool debug)
{
    if (P == 0) return 0;
    hipLaunchKernelGGL(k_preprocess, dim3(cdiv(P, 256)), dim3(256), 0, s, cam, P, means3D, shs, colors_precomp,
                       opacities, scales, rotations, cov3D_precomp, rec, radii, tiles_touched, depth_key);
    VR_KERNEL_CHECK("preprocess", s, debug);
    return 0;
}

int launch_mark_visible(const float* xyz, int P, const float* view, uint8_t* present, hipStream_t s)
{
    if (P == 0) return 0;
    hipLaunchKernelGGL(k_mark_visible, dim3(cdiv(P, 256)), dim3(256), 0, s, xyz, P, view, present);
    VR_KERNEL_CHECK("mark_visible", s, false);
    return 0;
}

}  // namespace vr
