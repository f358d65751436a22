// densify.hip -- the densification step of the Gaussian model ("densification bookkeeping"):
// scene/gaussian_model.py:384-403 densify_and_prune = densify_and_clone (:375-382) + densify_and_split (:353-373) +
// prune_points (:296-309), with the optimizer-state surgery of cat_tensors_to_optimizer (:311-331) and _prune_optimizer
// (:278-294), and reset_opacity (:215-218).
//
// The reference does it as a sequence of whole-model passes: two torch.cat per tensor and Adam moment (clone, split) and
// two boolean-mask gathers (prune the split originals, prune by opacity / size), over 6 parameters x 3 arrays -- ~70
// ATen launches and four read+write passes over 3 x 236 bytes per Gaussian.  What that sequence amounts to is ONE
// gather: every output row is a copy of one input row (kept original / clone / one of the two split samples), in a
// fixed order.  Here:
//   vr_densify_plan   classifies every Gaussian (one thread each; 3 small launches: count, scan, map) and writes, for
//                     every OUTPUT row, its source row, its kind and -- for split samples -- the row of the random draw
//   vr_densify_apply  one gather launch per tensor: parameter + both Adam moments in the same pass, the two computed
//                     columns of the split samples (position, scale) evaluated on the fly
// i.e. one read of what survives and one write of the result: HBM-bound streaming, 2 x 708 bytes per output Gaussian.
// Row order and arithmetic as the reference's sequence leaves them (pinned by the outputs of the reference's own
// methods: tests/golden/ref_densify.npz).
#include "../../include/vegs_optim.h"
#include "vr_host.h"

namespace vr {

constexpr int DN_THREADS = 256;
constexpr int DN_ITEMS = 4;                       // consecutive Gaussians per thread
constexpr int DN_BLOCK = DN_THREADS * DN_ITEMS;   // Gaussians per workgroup
constexpr int KIND_KEEP = 0, KIND_CLONE = 1, KIND_SPLIT0 = 2, KIND_SPLIT1 = 3;

struct DensifyRule {
    float max_grad, min_opacity, dense_thr, world_thr;
    int prune_big;
};

// bit 0: kept original (A), bit 1: clone (B), bit 2: split sample pair kept (C), bit 3: split original (owns a draw)
__device__ __forceinline__ uint32_t classify(const DensifyRule& r, int i, const float* __restrict__ opacity,
                                             const float* __restrict__ scaling, const float* __restrict__ accum,
                                             const float* __restrict__ denom)
{
    float g = accum[i] / denom[i];
    if (g != g) g = 0.0f;                                                   // never seen: 0 / 0 (:391-392)
    const float s0 = expf(scaling[3 * (size_t)i]), s1 = expf(scaling[3 * (size_t)i + 1]), s2 = expf(scaling[3 * (size_t)i + 2]);
    const float smax = fmaxf(s0, fmaxf(s1, s2));
    const bool big = smax > r.dense_thr;
    const bool clone = fabsf(g) >= r.max_grad && !big;                      // :377-380
    const bool split = g >= r.max_grad && big;                              // :356-362
    const float op = 1.0f / (1.0f + expf(-opacity[i]));
    bool prune_old = op < r.min_opacity, prune_new = prune_old;             // :396-402 (the screen-size test cannot fire:
    if (r.prune_big == 2) prune_old = prune_new = false;                    //  max_radii2D was reset by :349-351); prune=False (:394, :397)
    if (r.prune_big == 1) {
        prune_old = prune_old || smax > r.world_thr;
        const float n0 = expf(logf(s0 / 1.6f)), n1 = expf(logf(s1 / 1.6f)), n2 = expf(logf(s2 / 1.6f));
        prune_new = prune_new || fmaxf(n0, fmaxf(n1, n2)) > r.world_thr;
    }
    return (uint32_t)(!split && !prune_old) | (uint32_t)(clone && !prune_old) << 1 | (uint32_t)(split && !prune_new) << 2 |
           (uint32_t)split << 3;
}

// four 16-bit counters in one 64-bit word (a workgroup holds 1024 Gaussians)
__device__ __forceinline__ unsigned long long counters_of(uint32_t bits)
{
    return (unsigned long long)(bits & 1u) | (unsigned long long)((bits >> 1) & 1u) << 16 |
           (unsigned long long)((bits >> 2) & 1u) << 32 | (unsigned long long)((bits >> 3) & 1u) << 48;
}

// exclusive scan of `mine` over the workgroup (thread order); total in `total`
__device__ __forceinline__ unsigned long long block_exclusive(unsigned long long mine, unsigned long long& total)
{
    __shared__ unsigned long long wsum[DN_THREADS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    unsigned long long off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < DN_THREADS / 64; ++k) {
        if (k < w) off += wsum[k];
        tot += wsum[k];
    }
    total = tot;
    return off + incl - mine;
}

__global__ void __launch_bounds__(DN_THREADS)
k_densify_count(DensifyRule r, int P, const float* __restrict__ opacity, const float* __restrict__ scaling,
                const float* __restrict__ accum, const float* __restrict__ denom, uint32_t* __restrict__ block_sums)
{
    unsigned long long mine = 0;
    const int base = blockIdx.x * DN_BLOCK + threadIdx.x * DN_ITEMS;
#pragma unroll
    for (int k = 0; k < DN_ITEMS; ++k)
        if (base + k < P) mine += counters_of(classify(r, base + k, opacity, scaling, accum, denom));
    unsigned long long total;
    block_exclusive(mine, total);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) block_sums[4 * (size_t)blockIdx.x + c] = (uint32_t)(total >> (16 * c)) & 0xFFFFu;
    }
}

// exclusive scan of the per-workgroup counts (4 interleaved sequences), single workgroup; totals -> counts[0..4]:
// {rows out, A, B, C, split originals}
__global__ void __launch_bounds__(1024)
k_densify_scan(int nblocks, uint32_t* __restrict__ block_sums, int32_t* __restrict__ counts)
{
    __shared__ uint32_t wsum[16][4];
    __shared__ uint32_t carry[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < 4) carry[threadIdx.x] = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int b = base + threadIdx.x;
        uint32_t v[4], incl[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) incl[c] = v[c] = b < nblocks ? block_sums[4 * (size_t)b + c] : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t o = __shfl_up(incl[c], d, 64);
                if (lane >= d) incl[c] += o;
            }
        if (lane == 63)
#pragma unroll
            for (int c = 0; c < 4; ++c) wsum[w][c] = incl[c];
        __syncthreads();
        uint32_t off[4], tot[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            off[c] = carry[c];
            tot[c] = 0;
            for (int k = 0; k < 16; ++k) {
                if (k < w) off[c] += wsum[k][c];
                tot[c] += wsum[k][c];
            }
        }
        if (b < nblocks)
#pragma unroll
            for (int c = 0; c < 4; ++c) block_sums[4 * (size_t)b + c] = off[c] + incl[c] - v[c];
        __syncthreads();
        if (threadIdx.x < 4) carry[threadIdx.x] += tot[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counts[0] = (int32_t)(carry[0] + carry[1] + 2u * carry[2]);
        counts[1] = (int32_t)carry[0];
        counts[2] = (int32_t)carry[1];
        counts[3] = (int32_t)carry[2];
        counts[4] = (int32_t)carry[3];
    }
}

// plan entry of output row r: { source row | kind << 30, row of the unit-normal draw (split samples) }
__device__ __forceinline__ int2 plan_entry(int i, int kind, int draw)
{
    return make_int2((int)((uint32_t)i | (uint32_t)kind << 30), draw);
}

__global__ void __launch_bounds__(DN_THREADS)
k_densify_map(DensifyRule r, int P, const float* __restrict__ opacity, const float* __restrict__ scaling,
              const float* __restrict__ accum, const float* __restrict__ denom, const uint32_t* __restrict__ block_offs,
              const int32_t* __restrict__ counts, int2* __restrict__ plan)
{
    uint32_t bits[DN_ITEMS];
    unsigned long long mine = 0;
    const int base = blockIdx.x * DN_BLOCK + threadIdx.x * DN_ITEMS;
#pragma unroll
    for (int k = 0; k < DN_ITEMS; ++k) {
        bits[k] = base + k < P ? classify(r, base + k, opacity, scaling, accum, denom) : 0u;
        mine += counters_of(bits[k]);
    }
    unsigned long long total;
    unsigned long long off = block_exclusive(mine, total);
    const uint32_t nA = (uint32_t)counts[1], nB = (uint32_t)counts[2], nC = (uint32_t)counts[3], S = (uint32_t)counts[4];
    uint32_t a = block_offs[4 * (size_t)blockIdx.x] + (uint32_t)(off & 0xFFFFu);
    uint32_t b = block_offs[4 * (size_t)blockIdx.x + 1] + (uint32_t)((off >> 16) & 0xFFFFu);
    uint32_t c = block_offs[4 * (size_t)blockIdx.x + 2] + (uint32_t)((off >> 32) & 0xFFFFu);
    uint32_t s = block_offs[4 * (size_t)blockIdx.x + 3] + (uint32_t)((off >> 48) & 0xFFFFu);
#pragma unroll
    for (int k = 0; k < DN_ITEMS; ++k) {
        const int i = base + k;
        if (bits[k] & 1u) plan[a++] = plan_entry(i, KIND_KEEP, -1);
        if (bits[k] & 2u) plan[nA + b++] = plan_entry(i, KIND_CLONE, -1);
        if (bits[k] & 4u) {                                               // copy-major, like `repeat(N, 1)` (:364-373)
            plan[nA + nB + c] = plan_entry(i, KIND_SPLIT0, (int)s);
            plan[nA + nB + nC + c] = plan_entry(i, KIND_SPLIT1, (int)(S + s));
            ++c;
        }
        if (bits[k] & 8u) ++s;
    }
}

enum { ROLE_COPY = 0, ROLE_XYZ = 1, ROLE_SCALING = 2 };

// Output element (row, column) of one tensor and of its two Adam moments.  W = row width (0: runtime `width`).
template <int W>
__global__ void __launch_bounds__(256)
k_densify_rows(const int2* __restrict__ plan, long n_elems, int width, int role, const float* __restrict__ src,
               float* __restrict__ dst, const float* __restrict__ m_src, float* __restrict__ m_dst,
               const float* __restrict__ v_src, float* __restrict__ v_dst, const float* __restrict__ scaling,
               const float* __restrict__ rotation, const float* __restrict__ noise)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_elems) return;
    const int w = W ? W : width;
    const int row = (int)(e / w), col = (int)(e - (long)row * w);
    const int2 pe = plan[row];
    const size_t i = (size_t)(pe.x & 0x3FFFFFFF);
    const int kind = (int)((uint32_t)pe.x >> 30);
    float val = src[i * w + col];
    if (kind >= KIND_SPLIT0 && role != ROLE_COPY) {
        if (role == ROLE_SCALING) {
            val = logf(expf(val) / 1.6f);                                   // log(get_scaling / (0.8 N)), N = 2 (:369)
        } else {
            // xyz = R(q) (noise * exp(scaling)) + xyz; R = utils/general_utils.py:97-118 on the RAW quaternion (:367-368)
            const float4 q4 = reinterpret_cast<const float4*>(rotation)[i];
            const float norm = sqrtf(q4.x * q4.x + q4.y * q4.y + q4.z * q4.z + q4.w * q4.w);
            const float qr = q4.x / norm, qx = q4.y / norm, qy = q4.z / norm, qz = q4.w / norm;
            float R0, R1, R2;
            if (col == 0) { R0 = 1.0f - 2.0f * (qy * qy + qz * qz); R1 = 2.0f * (qx * qy - qr * qz); R2 = 2.0f * (qx * qz + qr * qy); }
            else if (col == 1) { R0 = 2.0f * (qx * qy + qr * qz); R1 = 1.0f - 2.0f * (qx * qx + qz * qz); R2 = 2.0f * (qy * qz - qr * qx); }
            else { R0 = 2.0f * (qx * qz - qr * qy); R1 = 2.0f * (qy * qz + qr * qx); R2 = 1.0f - 2.0f * (qx * qx + qy * qy); }
            const float* nz = noise + 3 * (size_t)pe.y;
            const float* sc = scaling + 3 * i;
            const float t0 = nz[0] * expf(sc[0]), t1 = nz[1] * expf(sc[1]), t2 = nz[2] * expf(sc[2]);
            val = (R0 * t0 + R1 * t1 + R2 * t2) + val;
        }
    }
    dst[e] = val;
    if (m_dst) {
        const bool keep = kind == KIND_KEEP;                                // new rows start with zero moments (:318-319)
        m_dst[e] = keep ? m_src[i * w + col] : 0.0f;
        v_dst[e] = keep ? v_src[i * w + col] : 0.0f;
    }
}

__global__ void __launch_bounds__(256)
k_reset_opacity(float* __restrict__ opacity, float* __restrict__ m, float* __restrict__ v, long P, float cap)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float x = fminf(1.0f / (1.0f + expf(-opacity[i])), cap);         // :216, utils/general_utils.py:21-22
    opacity[i] = logf(x / (1.0f - x));
    if (m) { m[i] = 0.0f; v[i] = 0.0f; }                                     // replace_tensor_to_optimizer (:263-276)
}

static int nblocks_of(int P) { return cdiv(P, DN_BLOCK); }

static bool make_rule(const VrDensifySettings* s, DensifyRule& r)
{
    if (!s || !(s->extent >= 0.0) || !(s->percent_dense >= 0.0) || s->max_grad != s->max_grad || s->min_opacity != s->min_opacity)
        return false;
    r.max_grad = (float)s->max_grad;
    r.min_opacity = (float)s->min_opacity;
    r.dense_thr = (float)(s->percent_dense * s->extent);     // Python-float products, compared in float32
    r.world_thr = (float)(0.1 * s->extent);
    r.prune_big = s->prune_big == 2 ? 2 : (s->prune_big != 0);
    return true;
}

}  // namespace vr

using namespace vr;

extern "C" int64_t vr_densify_plan_words(int32_t P)
{
    if (P < 0) return -1;
    return 4 * (int64_t)P + 4 * (int64_t)nblocks_of(P) + 4;   // int2 per output row (<= 2 P) + the workgroup counts
}

extern "C" int vr_densify_plan(const float* opacity, const float* scaling, const float* xyz_gradient_accum,
                               const float* denom, int32_t P, const VrDensifySettings* settings, int32_t* plan,
                               int32_t* counts, void* stream)
{
    DensifyRule r;
    if (P < 0 || !counts || !make_rule(settings, r) ||
        (P > 0 && (!opacity || !scaling || !xyz_gradient_accum || !denom || !plan))) {
        set_error("densify_plan: bad arguments");
        return VR_ERR_INVALID_ARGUMENT;
    }
    if (P >= (1 << 30)) { set_error("densify_plan: at most 2^30 - 1 Gaussians"); return VR_ERR_INVALID_ARGUMENT; }
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) { VR_HIP(hipMemsetAsync(counts, 0, 5 * sizeof(int32_t), s)); return VR_OK; }
    const int nb = nblocks_of(P);
    uint32_t* sums = reinterpret_cast<uint32_t*>(plan) + 4 * (size_t)P;
    hipLaunchKernelGGL(k_densify_count, dim3(nb), dim3(DN_THREADS), 0, s, r, P, opacity, scaling, xyz_gradient_accum, denom, sums);
    hipLaunchKernelGGL(k_densify_scan, dim3(1), dim3(1024), 0, s, nb, sums, counts);
    hipLaunchKernelGGL(k_densify_map, dim3(nb), dim3(DN_THREADS), 0, s, r, P, opacity, scaling, xyz_gradient_accum, denom,
                       (const uint32_t*)sums, (const int32_t*)counts, reinterpret_cast<int2*>(plan));
    if (hipGetLastError() != hipSuccess) { set_error("densify_plan: kernel launch failed"); return VR_ERR_HIP; }
    return VR_OK;
}

extern "C" int vr_densify_apply(const int32_t* plan, int32_t n_out, int32_t n_split, const VrDensifyTensor* tensors,
                                int32_t count, const float* scaling, const float* rotation, const float* noise, void* stream)
{
    if (n_out < 0 || n_split < 0 || count < 0 || (count > 0 && !tensors) || (n_out > 0 && !plan)) {
        set_error("densify_apply: bad arguments");
        return VR_ERR_INVALID_ARGUMENT;
    }
    for (int t = 0; t < count; ++t) {
        const VrDensifyTensor& d = tensors[t];
        const bool moments = d.m_src || d.m_dst || d.v_src || d.v_dst;
        if (d.width <= 0 || d.role < ROLE_COPY || d.role > ROLE_SCALING || (n_out > 0 && (!d.src || !d.dst)) ||
            (moments && n_out > 0 && !(d.m_src && d.m_dst && d.v_src && d.v_dst))) {
            set_error("densify_apply: tensor %d: NULL array, bad width or role, or an incomplete set of moments", t);
            return VR_ERR_INVALID_ARGUMENT;
        }
        if ((d.role == ROLE_XYZ || d.role == ROLE_SCALING) && d.width != 3) {
            set_error("densify_apply: tensor %d: the position and scaling tensors have 3 columns", t);
            return VR_ERR_INVALID_ARGUMENT;
        }
        if (d.role == ROLE_XYZ && n_out > 0 && (!scaling || !rotation || (n_split > 0 && !noise))) {
            set_error("densify_apply: the position tensor needs scaling, rotation and the draw");
            return VR_ERR_INVALID_ARGUMENT;
        }
        if (d.role == ROLE_XYZ && ((uintptr_t)rotation & 15)) {
            set_error("densify_apply: rotation must be 16-byte aligned");
            return VR_ERR_INVALID_ARGUMENT;
        }
    }
    if (n_out == 0) return VR_OK;
    hipStream_t s = (hipStream_t)stream;
    const int2* p2 = reinterpret_cast<const int2*>(plan);
    for (int t = 0; t < count; ++t) {
        const VrDensifyTensor& d = tensors[t];
        const long n = (long)n_out * d.width;
        const dim3 grid((unsigned)((n + 255) / 256));
#define VR_ROWS(W)                                                                                                      \
    hipLaunchKernelGGL(k_densify_rows<W>, grid, dim3(256), 0, s, p2, n, d.width, d.role, d.src, d.dst, d.m_src, d.m_dst, \
                       d.v_src, d.v_dst, scaling, rotation, noise)
        switch (d.width) {
            case 1: VR_ROWS(1); break;
            case 3: VR_ROWS(3); break;
            case 4: VR_ROWS(4); break;
            case 9: VR_ROWS(9); break;
            case 24: VR_ROWS(24); break;
            case 45: VR_ROWS(45); break;
            default: VR_ROWS(0); break;
        }
#undef VR_ROWS
    }
    if (hipGetLastError() != hipSuccess) { set_error("densify_apply: kernel launch failed"); return VR_ERR_HIP; }
    return VR_OK;
}

extern "C" int vr_reset_opacity(float* opacity, float* exp_avg, float* exp_avg_sq, int64_t P, float cap, void* stream)
{
    if (P < 0 || (P > 0 && !opacity) || ((exp_avg != nullptr) != (exp_avg_sq != nullptr)) || !(cap > 0.0f && cap < 1.0f)) {
        set_error("reset_opacity: bad arguments");
        return VR_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return VR_OK;
    hipLaunchKernelGGL(k_reset_opacity, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, opacity, exp_avg,
                       exp_avg_sq, (long)P, cap);
    if (hipGetLastError() != hipSuccess) { set_error("reset_opacity: kernel launch failed"); return VR_ERR_HIP; }
    return VR_OK;
}
