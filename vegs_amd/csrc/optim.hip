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

constexpr int ADAM_MAX_T = 64;           // tensors per launch (the table is a kernel argument: 64 x 56 B < the 4 KB limit)
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
    float eps;             // per tensor: the model's groups use 1e-15, the BoxModels' optimizers torch's default 1e-8
};
struct AdamArgs {
    AdamSeg seg[ADAM_MAX_T];
    int count;
    float one_minus_b1, b2, one_minus_b2;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamSeg& s, const AdamArgs& a)
{
    m = m + (g - m) * a.one_minus_b1;                      // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.b2 + (a.one_minus_b2 * g) * g;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) * s.inv_bc2_sqrt + s.eps; // (sqrt(v) / bias_correction2_sqrt).add_(eps)
    p = p - s.step_size * (m / denom);                     // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ void __launch_bounds__(256) k_adam(AdamArgs a)
{
    int t = 0, hi = a.count - 1;          // the last tensor whose first block is <= blockIdx.x (scalar binary search)
    while (t < hi) {
        const int mid = (t + hi + 1) >> 1;
        if ((int)blockIdx.x >= a.seg[mid].block0) t = mid; else hi = mid - 1;
    }
    const AdamSeg s = a.seg[t];
    const long base = (long)(blockIdx.x - s.block0) * ADAM_EPB;
    const bool vec = ((((uintptr_t)s.p | (uintptr_t)s.g | (uintptr_t)s.m | (uintptr_t)s.v) & 15) == 0);
    if (vec && base + ADAM_EPB <= s.n) {
        float4 p[4], g[4], m[4], v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long o = base / 4 + j * 256 + threadIdx.x;
            p[j] = nt_load4(reinterpret_cast<const float4*>(s.p) + o);
            g[j] = nt_load4(reinterpret_cast<const float4*>(s.g) + o);
            m[j] = nt_load4(reinterpret_cast<const float4*>(s.m) + o);
            v[j] = nt_load4(reinterpret_cast<const float4*>(s.v) + o);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            adam_elem(p[j].x, g[j].x, m[j].x, v[j].x, s, a);
            adam_elem(p[j].y, g[j].y, m[j].y, v[j].y, s, a);
            adam_elem(p[j].z, g[j].z, m[j].z, v[j].z, s, a);
            adam_elem(p[j].w, g[j].w, m[j].w, v[j].w, s, a);
            const long o = base / 4 + j * 256 + threadIdx.x;
            nt_store4(p[j], reinterpret_cast<float4*>(s.p) + o);
            nt_store4(m[j], reinterpret_cast<float4*>(s.m) + o);
            nt_store4(v[j], reinterpret_cast<float4*>(s.v) + o);
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

// ---- factored SH gradients (VrInGrads.dL_dcolors_sh).  One lane per Gaussian rebuilds its dense row
//   g[k][c] = scale * sum_v basis_k(dir(means[i], campos[v])) * factors[v][i][c]
// in registers (n_views x (direction + 16-term basis + 3K fma)), parks it in LDS, and the wave then walks the 64 rows --
// contiguous in memory -- linearly: either storing them (vr_sh_grad_from_factors) or applying Adam to param / exp_avg /
// exp_avg_sq in place (vr_sh_adam_step), so the 192-byte gradient row never exists in HBM.  Rows of the whole
// [P,M,3] tensor are parked at stride 3M + 1 (odd for even M: bank-conflict free); with split storage the rest rows use
// stride 3(M-1) + 1 and the DC row is handled by its own lane.
constexpr int SHF_ROW_MAX = 48;
struct ShAdamSeg {
    float* p; float* m; float* v;          // Adam mode
    float* out;                            // store mode
    float step_size, inv_bc2_sqrt;
};
struct ShFactorArgs {
    const float* means3D; const float* campos; const float* factors;
    long view_stride;                      // floats between the factor blocks of consecutive views (3 P when packed)
    int P, n_views, deg, M;
    float scale;
    ShAdamSeg dc, rest;                    // rest.p / rest.out == nullptr: `dc` is the whole [P,M,3] tensor
    float one_minus_b1, b2, one_minus_b2, eps;
};

template <bool ADAM>
__device__ __forceinline__ void sh_consume(const ShFactorArgs& a, const ShAdamSeg& s, size_t e, float g)
{
    if (ADAM) {
        float p = s.p[e], m = s.m[e], v = s.v[e];
        m = m + (g - m) * a.one_minus_b1;
        v = v * a.b2 + (a.one_minus_b2 * g) * g;
        const float denom = sqrtf(v) * s.inv_bc2_sqrt + a.eps;
        p = p - s.step_size * (m / denom);
        s.p[e] = p; s.m[e] = m; s.v[e] = v;
    } else {
        s.out[e] = g;
    }
}

// The wave's 64 parked rows (row length `rowlen`, LDS stride `stride`) against `total` = rows x rowlen contiguous
// elements starting at element `base`: four elements per lane and trip (dwordx4 on param / exp_avg / exp_avg_sq: a
// quarter of the memory instructions of the one-dword walk, 0.47 -> 0.4x ms at 2 M Gaussians), a dword tail for a last
// partial wave.  `base` is a multiple of 64 x rowlen, so the quads are 16-byte aligned whenever the arrays are.
template <bool ADAM>
__device__ __forceinline__ void sh_walk(const ShFactorArgs& a, const ShAdamSeg& s, size_t base, int total, int rowlen,
                                        int stride, const float* __restrict__ lrows, int lane)
{
    const uintptr_t align = ADAM ? ((uintptr_t)s.p | (uintptr_t)s.m | (uintptr_t)s.v) : (uintptr_t)s.out;
    int done = 0;
    if ((align & 15u) == 0u) {
        const int quads = total >> 2;
        for (int q = lane; q < quads; q += 64) {
            float g[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = 4 * q + c, r = e / rowlen;
                g[c] = lrows[r * stride + (e - r * rowlen)];
            }
            if (ADAM) {
                float4* pp = reinterpret_cast<float4*>(s.p + base) + q;
                float4* pm = reinterpret_cast<float4*>(s.m + base) + q;
                float4* pv = reinterpret_cast<float4*>(s.v + base) + q;
                const float4 p4 = nt_load4(pp), m4 = nt_load4(pm), v4 = nt_load4(pv);   // streamed once per step
                float p[4] = {p4.x, p4.y, p4.z, p4.w}, m[4] = {m4.x, m4.y, m4.z, m4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    m[c] = m[c] + (g[c] - m[c]) * a.one_minus_b1;
                    v[c] = v[c] * a.b2 + (a.one_minus_b2 * g[c]) * g[c];
                    const float denom = sqrtf(v[c]) * s.inv_bc2_sqrt + a.eps;
                    p[c] = p[c] - s.step_size * (m[c] / denom);
                }
                nt_store4(make_float4(p[0], p[1], p[2], p[3]), pp);
                nt_store4(make_float4(m[0], m[1], m[2], m[3]), pm);
                nt_store4(make_float4(v[0], v[1], v[2], v[3]), pv);
            } else {
                reinterpret_cast<float4*>(s.out + base)[q] = make_float4(g[0], g[1], g[2], g[3]);
            }
        }
        done = quads << 2;
    }
    for (int e = done + lane; e < total; e += 64) {
        const int r = e / rowlen;
        sh_consume<ADAM>(a, s, base + e, lrows[r * stride + (e - r * rowlen)]);
    }
}

template <bool ADAM>
__global__ void __launch_bounds__(256) k_sh_factors(ShFactorArgs a)
{
    __shared__ float rows[4][64 * (SHF_ROW_MAX + 1)];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wave_first = (blockIdx.x * 4 + w) * 64;
    if (wave_first >= a.P) return;
    const int i = wave_first + lane;
    const int rows_here = min(64, a.P - wave_first);
    const bool split = ADAM ? a.rest.p != nullptr : a.rest.out != nullptr;
    const int K = (a.deg + 1) * (a.deg + 1);
    float g[SHF_ROW_MAX];
#pragma unroll
    for (int q = 0; q < SHF_ROW_MAX; ++q) g[q] = 0.0f;
    if (i < a.P) {
        const float px = a.means3D[3 * (size_t)i], py = a.means3D[3 * (size_t)i + 1], pz = a.means3D[3 * (size_t)i + 2];
        for (int v = 0; v < a.n_views; ++v) {
            const float* f = a.factors + (size_t)v * a.view_stride + (size_t)i * 3;
            const float f0 = f[0], f1 = f[1], f2v = f[2];
            if (f0 == 0.0f && f1 == 0.0f && f2v == 0.0f) continue;      // not visible in this view (or fully clamped)
            const float d0 = px - a.campos[3 * v], d1 = py - a.campos[3 * v + 1], d2 = pz - a.campos[3 * v + 2];
            const float il = 1.0f / sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
            float bas[16];
            sh_basis(a.deg, d0 * il, d1 * il, d2 * il, bas);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k < K) {
                    g[3 * k + 0] = fmaf(bas[k], f0, g[3 * k + 0]);
                    g[3 * k + 1] = fmaf(bas[k], f1, g[3 * k + 1]);
                    g[3 * k + 2] = fmaf(bas[k], f2v, g[3 * k + 2]);
                }
            }
        }
    }
    const int row = 3 * a.M;
    if (split) {
        if (i < a.P) {
#pragma unroll
            for (int c = 0; c < 3; ++c) sh_consume<ADAM>(a, a.dc, 3 * (size_t)i + c, a.scale * g[c]);
        }
        const int rowr = row - 3, stride = rowr + 1;
        float* my = rows[w] + lane * stride;
#pragma unroll
        for (int q = 0; q < SHF_ROW_MAX - 3; ++q)
            if (q < rowr) my[q] = a.scale * g[3 + q];
        __builtin_amdgcn_wave_barrier();
        sh_walk<ADAM>(a, a.rest, (size_t)wave_first * rowr, rows_here * rowr, rowr, stride, rows[w], lane);
    } else {
        const int stride = row + 1;
        float* my = rows[w] + lane * stride;
#pragma unroll
        for (int q = 0; q < SHF_ROW_MAX; ++q)
            if (q < row) my[q] = a.scale * g[q];
        __builtin_amdgcn_wave_barrier();
        sh_walk<ADAM>(a, a.dc, (size_t)wave_first * row, rows_here * row, row, stride, rows[w], lane);
    }
}

static int sh_factor_args(ShFactorArgs& a, const float* means3D, int32_t P, const float* campos, const float* factors,
                          int32_t n_views, int64_t view_stride, int32_t sh_degree, int32_t M, float scale)
{
    if (view_stride != 0 && view_stride < 3 * (int64_t)P)
        { set_error("sh factors: factor_view_stride must be 0 (packed) or >= 3 P"); return VR_ERR_INVALID_ARGUMENT; }
    if (P < 0 || n_views < 1 || sh_degree < 0 || sh_degree > 3 || M < (sh_degree + 1) * (sh_degree + 1) || M > 16)
        { set_error("sh factors: need P >= 0, n_views >= 1, sh_degree 0..3 and (sh_degree+1)^2 <= M <= 16"); return VR_ERR_INVALID_ARGUMENT; }
    if (P > 0 && (!means3D || !campos || !factors)) { set_error("sh factors: means3D, campos and factors are required"); return VR_ERR_INVALID_ARGUMENT; }
    a.means3D = means3D; a.campos = campos; a.factors = factors;
    a.P = P; a.n_views = n_views; a.deg = sh_degree; a.M = M; a.scale = scale;
    a.view_stride = view_stride ? (long)view_stride : 3L * P;
    a.dc = ShAdamSeg{nullptr, nullptr, nullptr, nullptr, 0.f, 0.f};
    a.rest = a.dc;
    a.one_minus_b1 = a.b2 = a.one_minus_b2 = a.eps = 0.f;
    return VR_OK;
}

}  // namespace vr

using namespace vr;

extern "C" int vr_sh_grad_from_factors(const float* means3D, int32_t P, const float* campos, const float* factors,
                                       int32_t n_views, int64_t factor_view_stride, int32_t sh_degree, int32_t M, float scale,
                                       float* dL_dshs, float* dL_dshs_rest, void* stream)
{
    ShFactorArgs a;
    int rc = sh_factor_args(a, means3D, P, campos, factors, n_views, factor_view_stride, sh_degree, M, scale);
    if (rc) return rc;
    if (P == 0) return VR_OK;
    if (!dL_dshs || (dL_dshs_rest && M < 2)) { set_error("sh factors: dL_dshs is required (and M >= 2 for split storage)"); return VR_ERR_INVALID_ARGUMENT; }
    a.dc.out = dL_dshs;
    a.rest.out = dL_dshs_rest;
    hipLaunchKernelGGL(k_sh_factors<false>, dim3(cdiv(P, 256)), dim3(256), 0, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) { set_error("sh factors: kernel launch failed"); return VR_ERR_HIP; }
    return VR_OK;
}

extern "C" int vr_sh_adam_step(const float* means3D, int32_t P, const float* campos, const float* factors, int32_t n_views,
                               int64_t factor_view_stride, int32_t sh_degree, int32_t M, float scale, const VrShAdamTensor* dc,
                               const VrShAdamTensor* rest, double beta1, double beta2, double eps, void* stream)
{
    ShFactorArgs a;
    int rc = sh_factor_args(a, means3D, P, campos, factors, n_views, factor_view_stride, sh_degree, M, scale);
    if (rc) return rc;
    if (P == 0) return VR_OK;
    if (!dc || !dc->param || !dc->exp_avg || !dc->exp_avg_sq || dc->step < 1 ||
        (rest && (!rest->param || !rest->exp_avg || !rest->exp_avg_sq || rest->step < 1 || M < 2)))
        { set_error("sh adam: tensor with NULL array or step < 1"); return VR_ERR_INVALID_ARGUMENT; }
    auto fill = [&](ShAdamSeg& s, const VrShAdamTensor& t) {
        s.p = t.param; s.m = t.exp_avg; s.v = t.exp_avg_sq; s.out = nullptr;
        const double bc1 = 1.0 - pow(beta1, (double)t.step), bc2 = 1.0 - pow(beta2, (double)t.step);
        s.step_size = (float)(t.lr / bc1);
        s.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    };
    fill(a.dc, *dc);
    if (rest) fill(a.rest, *rest);
    a.one_minus_b1 = (float)(1.0 - beta1);
    a.b2 = (float)beta2;
    a.one_minus_b2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    hipLaunchKernelGGL(k_sh_factors<true>, dim3(cdiv(P, 256)), dim3(256), 0, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) { set_error("sh adam: kernel launch failed"); return VR_ERR_HIP; }
    return VR_OK;
}


extern "C" int vr_adam_step(const VrAdamTensor* tensors, int32_t count, double beta1, double beta2, double eps, void* stream)
{
    if (count < 0 || (count > 0 && !tensors)) { set_error("adam: bad tensor list"); return VR_ERR_INVALID_ARGUMENT; }
    for (int i = 0; i < count; ++i) {
        const VrAdamTensor& t = tensors[i];
        if (t.n < 0 || t.step < 1 || (t.n > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq)))
            { set_error("adam: tensor with NULL array, negative size or step < 1"); return VR_ERR_INVALID_ARGUMENT; }
        if (t.eps < 0.0 && !(eps >= 0.0)) { set_error("adam: tensor %d has no eps of its own and the call gives none", i); return VR_ERR_INVALID_ARGUMENT; }
    }
    // `first` advances by the entries CONSUMED (empty tensors are skipped without taking a slot), not by
    // ADAM_MAX_T: otherwise a batch that skipped an empty tensor would be followed by one that repeats its tail
    for (int first = 0, next = 0; first < count; first = next) {
        AdamArgs a;
        a.count = 0;
        a.one_minus_b1 = (float)(1.0 - beta1);
        a.b2 = (float)beta2;
        a.one_minus_b2 = (float)(1.0 - beta2);
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
            s.eps = (float)(t.eps >= 0.0 ? t.eps : eps);
            blocks += cdiv((long)t.n, ADAM_EPB);
        }
        for (int i = a.count; i < ADAM_MAX_T; ++i) a.seg[i] = AdamSeg{nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0.f, 0x7fffffff, 0.f};
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
