// optim.hip -- the per-Gaussian update after loss.backward(): Adam over the six parameter groups
// (reference scene/gaussian_model.py:159-168, stepped at train.py:319) and the densification statistics
// (scene/gaussian_model.py:411-413, train.py:299-300).
//
// torch.optim.Adam's default path walks every group with ~10 foreach launches and several temporaries
// (118 M elements at 2 M Gaussians x 59 floats: each temporary is another 472 MB round trip).  Here ALL
// tensors are updated by ONE launch: a block->tensor table in the kernel arguments, float4 streaming of
// (param, grad, exp_avg, exp_avg_sq), 28 B per element -- the algorithmic minimum (read 4, write 3 floats).
// Pure HBM streaming; bound = 8 TB/s.
#include <math.h>

#include "../../include/vegs_optim.h"
#include "vr_host.h"

namespace vr {

constexpr int ADAM_MAX_T = 8;            // tensors per launch
constexpr int ADAM_EPB = 256 * 4 * 4;    // elements per block: 256 threads x float4 x 4

struct AdamSeg {
    float* p;
    const float* g;
    float* m;
    float* v;
    long n;
    float step_size;       // lr / (1 - beta1^step)
    float inv_bc2_sqrt;    // 1 / sqrt(1 - beta2^step)
    int block0;            // first block of this tensor
};
struct AdamArgs {
    AdamSeg seg[ADAM_MAX_T];
    int count;
    float one_minus_b1, b2, one_minus_b2, eps;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamSeg& s, const AdamArgs& a)
{
    m = m + (g - m) * a.one_minus_b1;                      // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.b2 + (a.one_minus_b2 * g) * g;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) * s.inv_bc2_sqrt + a.eps; // (sqrt(v) / bias_correction2_sqrt).add_(eps)
    p = p - s.step_size * (m / denom);                     // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ void __launch_bounds__(256) k_adam(AdamArgs a)
{
    int t = 0;
#pragma unroll
    for (int i = 1; i < ADAM_MAX_T; ++i)
        if (i < a.count && (int)blockIdx.x >= a.seg[i].block0) t = i;
    const AdamSeg s = a.seg[t];
    const long base = (long)(blockIdx.x - s.block0) * ADAM_EPB;
    const bool vec = ((((uintptr_t)s.p | (uintptr_t)s.g | (uintptr_t)s.m | (uintptr_t)s.v) & 15) == 0);
    if (vec && base + ADAM_EPB <= s.n) {
        float4 p[4], g[4], m[4], v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long o = base / 4 + j * 256 + threadIdx.x;
            p[j] = reinterpret_cast<const float4*>(s.p)[o];
            g[j] = nt_load4(reinterpret_cast<const float4*>(s.g) + o);
            m[j] = reinterpret_cast<const float4*>(s.m)[o];
            v[j] = reinterpret_cast<const float4*>(s.v)[o];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            adam_elem(p[j].x, g[j].x, m[j].x, v[j].x, s, a);
            adam_elem(p[j].y, g[j].y, m[j].y, v[j].y, s, a);
            adam_elem(p[j].z, g[j].z, m[j].z, v[j].z, s, a);
            adam_elem(p[j].w, g[j].w, m[j].w, v[j].w, s, a);
            const long o = base / 4 + j * 256 + threadIdx.x;
            reinterpret_cast<float4*>(s.p)[o] = p[j];
            reinterpret_cast<float4*>(s.m)[o] = m[j];
            reinterpret_cast<float4*>(s.v)[o] = v[j];
        }
    } else {
        for (long o = base + threadIdx.x; o < base + ADAM_EPB && o < s.n; o += 256) {
            float p = s.p[o], m = s.m[o], v = s.v[o];
            adam_elem(p, s.g[o], m, v, s, a);
            s.p[o] = p; s.m[o] = m; s.v[o] = v;
        }
    }
}

__global__ void __launch_bounds__(256)
k_densify_stats(const float* __restrict__ g2d, const int32_t* __restrict__ radii, int P, float* __restrict__ accum,
                float* __restrict__ denom, float* __restrict__ max_radii)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r <= 0) return;
    const float gx = g2d[3 * (size_t)i], gy = g2d[3 * (size_t)i + 1];
    accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
    max_radii[i] = fmaxf(max_radii[i], (float)r);
}

}  // namespace vr

using namespace vr;

extern "C" int vr_adam_step(const VrAdamTensor* tensors, int32_t count, double beta1, double beta2, double eps, void* stream)
{
    if (count < 0 || (count > 0 && !tensors)) { set_error("adam: bad tensor list"); return VR_ERR_INVALID_ARGUMENT; }
    for (int i = 0; i < count; ++i) {
        const VrAdamTensor& t = tensors[i];
        if (t.n < 0 || t.step < 1 || (t.n > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq)))
            { set_error("adam: tensor with NULL array, negative size or step < 1"); return VR_ERR_INVALID_ARGUMENT; }
    }
    // `first` advances by the entries CONSUMED (empty tensors are skipped without taking a slot), not by
    // ADAM_MAX_T: otherwise a batch that skipped an empty tensor would be followed by one that repeats its tail
    for (int first = 0, next = 0; first < count; first = next) {
        AdamArgs a;
        a.count = 0;
        a.one_minus_b1 = (float)(1.0 - beta1);
        a.b2 = (float)beta2;
        a.one_minus_b2 = (float)(1.0 - beta2);
        a.eps = (float)eps;
        int blocks = 0;
        for (next = first; next < count && a.count < ADAM_MAX_T; ++next) {
            const VrAdamTensor& t = tensors[next];
            if (t.n == 0) continue;
            AdamSeg& s = a.seg[a.count++];
            s.p = t.param; s.g = t.grad; s.m = t.exp_avg; s.v = t.exp_avg_sq; s.n = (long)t.n;
            // the scalars torch computes in Python doubles (torch/optim/adam.py, non-capturable path)
            const double bc1 = 1.0 - pow(beta1, (double)t.step), bc2 = 1.0 - pow(beta2, (double)t.step);
            s.step_size = (float)(t.lr / bc1);
            s.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
            s.block0 = blocks;
            blocks += cdiv((long)t.n, ADAM_EPB);
        }
        for (int i = a.count; i < ADAM_MAX_T; ++i) a.seg[i] = AdamSeg{nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0.f, 0x7fffffff};
        if (blocks == 0) continue;
        hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
        if (hipGetLastError() != hipSuccess) { set_error("adam: kernel launch failed"); return VR_ERR_HIP; }
    }
    return VR_OK;
}

extern "C" int vr_densify_stats(const float* means2D_grad, const int32_t* radii, int32_t P, float* xyz_gradient_accum,
                                float* denom, float* max_radii2D, void* stream)
{
    if (P < 0 || (P > 0 && (!means2D_grad || !radii || !xyz_gradient_accum || !denom || !max_radii2D)))
        { set_error("densify_stats: bad arguments"); return VR_ERR_INVALID_ARGUMENT; }
    if (P == 0) return VR_OK;
    hipLaunchKernelGGL(k_densify_stats, dim3(cdiv(P, 256)), dim3(256), 0, (hipStream_t)stream, means2D_grad, radii, P,
                       xyz_gradient_accum, denom, max_radii2D);
    if (hipGetLastError() != hipSuccess) { set_error("densify_stats: kernel launch failed"); return VR_ERR_HIP; }
    return VR_OK;
}
