// knn.hip -- mean squared distance to the 3 nearest neighbours of every point.
//
// Replaces `simple_knn._C.distCUDA2` (second un-vendored CUDA submodule of the reference,
// .gitmodules:1-3), which scene/gaussian_model.py:21 imports at module load and calls at :140 and :517
// to initialise the Gaussian scales -- without it VEGS cannot start on ROCm.
// Contract: points[N,3] fp32 -> out[N] fp32, out[i] = mean of the squared distances from point i to its
// three nearest OTHER points (exact, not approximate); missing neighbours (N < 4) count as FLT_MAX.
//
// Method (exact k-NN with spatial pruning): Morton-order the points (30-bit codes over the bounding
// box, sorted with the library's wave-ballot radix sort), cut the sorted sequence into boxes of 512
// points with their AABBs, and let every point (one lane each; a wave = 64 Morton-neighbours) visit only
// the boxes whose AABB is closer than its current third-best distance.  The box loop is wave-uniform:
// a box is scanned when ANY lane needs it and its points are then read as wave-uniform broadcasts.
#include <float.h>

#include "vr_host.h"

namespace vr {

constexpr int KNN_BOX = 512;

__device__ __forceinline__ uint32_t float_order_key(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float float_from_order_key(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// minmax[0..2] = min x,y,z ; minmax[3..5] = max x,y,z (as order-preserving uint keys)
__global__ void __launch_bounds__(256) k_knn_bbox(const float* __restrict__ pts, int N, uint32_t* __restrict__ minmax)
{
    __shared__ uint32_t red[6][4];
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const uint32_t k = float_order_key(pts[3 * (size_t)i + a]);
            lo[a] = min(lo[a], k);
            hi[a] = max(hi[a], k);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            lo[a] = min(lo[a], (uint32_t)__shfl_xor((int)lo[a], d, 64));
            hi[a] = max(hi[a], (uint32_t)__shfl_xor((int)hi[a], d, 64));
        }
        if ((threadIdx.x & 63) == 0) { red[a][threadIdx.x >> 6] = lo[a]; red[3 + a][threadIdx.x >> 6] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        atomicMin(&minmax[a], min(min(red[a][0], red[a][1]), min(red[a][2], red[a][3])));
        atomicMax(&minmax[3 + a], max(max(red[3 + a][0], red[3 + a][1]), max(red[3 + a][2], red[3 + a][3])));
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x)
{
    x &= 0x3FFu;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ void __launch_bounds__(256)
k_knn_morton(const float* __restrict__ pts, int N, const uint32_t* __restrict__ minmax, uint32_t* __restrict__ keys,
             uint32_t* __restrict__ vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    uint32_t code = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = float_from_order_key(minmax[a]), hi = float_from_order_key(minmax[3 + a]);
        const float ext = hi - lo;
        const float t = ext > 0.0f ? (pts[3 * (size_t)i + a] - lo) / ext : 0.0f;
        const uint32_t q = (uint32_t)fminf(fmaxf(t * 1023.0f, 0.0f), 1023.0f);
        code |= spread10(q) << a;
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

// sorted[i] = (x, y, z, original index as bits)
__global__ void __launch_bounds__(256)
k_knn_gather(const float* __restrict__ pts, const uint32_t* __restrict__ order, int N, float4* __restrict__ sorted)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t id = order[i];
    sorted[i] = make_float4(pts[3 * (size_t)id], pts[3 * (size_t)id + 1], pts[3 * (size_t)id + 2], __uint_as_float(id));
}

__global__ void __launch_bounds__(256)
k_knn_boxes(const float4* __restrict__ sorted, int N, float4* __restrict__ bmin, float4* __restrict__ bmax)
{
    __shared__ float red[6][4];
    const int b = blockIdx.x;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int j = b * KNN_BOX + threadIdx.x; j < min(N, (b + 1) * KNN_BOX); j += 256) {
        const float4 p = sorted[j];
        lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64));
        }
        if ((threadIdx.x & 63) == 0) { red[a][threadIdx.x >> 6] = lo[a]; red[3 + a][threadIdx.x >> 6] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        bmin[b] = make_float4(fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3])),
                              fminf(fminf(red[1][0], red[1][1]), fminf(red[1][2], red[1][3])),
                              fminf(fminf(red[2][0], red[2][1]), fminf(red[2][2], red[2][3])), 0.f);
        bmax[b] = make_float4(fmaxf(fmaxf(red[3][0], red[3][1]), fmaxf(red[3][2], red[3][3])),
                              fmaxf(fmaxf(red[4][0], red[4][1]), fmaxf(red[4][2], red[4][3])),
                              fmaxf(fmaxf(red[5][0], red[5][1]), fmaxf(red[5][2], red[5][3])), 0.f);
    }
}

__device__ __forceinline__ void knn_insert(float d, float& b0, float& b1, float& b2)
{
    // keep the three smallest, b0 <= b1 <= b2
    const float n2 = fminf(b2, fmaxf(b1, d));
    const float n1 = fminf(b1, fmaxf(b0, d));
    const float n0 = fminf(b0, d);
    b0 = n0; b1 = n1; b2 = n2;
}

__global__ void __launch_bounds__(256)
k_knn_search(const float4* __restrict__ sorted, int N, const float4* __restrict__ bmin, const float4* __restrict__ bmax,
             int nbox, float* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < N;
    const float4 p = live ? sorted[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    // The +-3 neighbours in Morton order give a first, usually tight, UPPER BOUND on the third-nearest
    // distance.  It is only used for pruning: the neighbours themselves are met again in the box scans,
    // so the running best-three starts empty (no double counting).
    float bound = FLT_MAX;
    if (live) {
        float t0 = FLT_MAX, t1 = FLT_MAX, t2 = FLT_MAX;
#pragma unroll
        for (int d = -3; d <= 3; ++d) {
            const int j = i + d;
            if (d == 0 || j < 0 || j >= N) continue;
            const float4 q = sorted[j];
            const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
            knn_insert(fmaf(dz, dz, fmaf(dy, dy, dx * dx)), t0, t1, t2);
        }
        bound = t2;
    }
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    for (int b = 0; b < nbox; ++b) {
        const float4 lo = bmin[b], hi = bmax[b];
        const float dx = fmaxf(fmaxf(lo.x - p.x, p.x - hi.x), 0.0f);
        const float dy = fmaxf(fmaxf(lo.y - p.y, p.y - hi.y), 0.0f);
        const float dz = fmaxf(fmaxf(lo.z - p.z, p.z - hi.z), 0.0f);
        const float dbox = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        // a box farther than min(current third best, bound) cannot hold one of the three nearest
        if (__ballot(live && dbox <= fminf(b2, bound)) == 0ull) continue;
        const int j1 = min(N, (b + 1) * KNN_BOX);
        for (int j = b * KNN_BOX; j < j1; ++j) {
            const float4 q = sorted[j];                        // wave-uniform address
            const float ex = p.x - q.x, ey = p.y - q.y, ez = p.z - q.z;
            const float d = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
            if (j != i) knn_insert(d, b0, b1, b2);
        }
    }
    if (live) out[__float_as_uint(p.w)] = (b0 + b1 + b2) / 3.0f;
}

size_t knn3_scratch_bytes(int N)
{
    const size_t n = (size_t)(N > 0 ? N : 1);
    const size_t nbox = (n + KNN_BOX - 1) / KNN_BOX;
    return 256 + 4 * align_up(n * 4, 256) + align_up(n * 16, 256) + 2 * align_up(nbox * 16, 256) +
           sort_pairs_scratch_bytes((long)n);
}

int launch_knn3(const float* points, int N, float* out, void* scratch, hipStream_t s, bool debug)
{
    if (N <= 0) return 0;
    const size_t n = (size_t)N, arr = align_up(n * 4, 256);
    const int nbox = (N + KNN_BOX - 1) / KNN_BOX;
    char* base = (char*)scratch;
    uint32_t* minmax = (uint32_t*)base;
    uint32_t* k0 = (uint32_t*)(base + 256);
    uint32_t* v0 = (uint32_t*)(base + 256 + arr);
    uint32_t* k1 = (uint32_t*)(base + 256 + 2 * arr);
    uint32_t* v1 = (uint32_t*)(base + 256 + 3 * arr);
    float4* sorted = (float4*)(base + 256 + 4 * arr);
    float4* bmin = (float4*)((char*)sorted + align_up(n * 16, 256));
    float4* bmax = (float4*)((char*)bmin + align_up((size_t)nbox * 16, 256));
    void* sort_scr = (char*)bmax + align_up((size_t)nbox * 16, 256);

    VR_HIP(hipMemsetD32Async((hipDeviceptr_t)minmax, (int)0xFFFFFFFF, 3, s));
    VR_HIP(hipMemsetD32Async((hipDeviceptr_t)(minmax + 3), 0, 3, s));
    int grid = cdiv(N, 256);
    hipLaunchKernelGGL(k_knn_bbox, dim3(grid > 512 ? 512 : grid), dim3(256), 0, s, points, N, minmax);
    VR_KERNEL_CHECK("knn_bbox", s, debug);
    hipLaunchKernelGGL(k_knn_morton, dim3(grid), dim3(256), 0, s, points, N, (const uint32_t*)minmax, k0, v0);
    VR_KERNEL_CHECK("knn_morton", s, debug);
    int where = 0;
    int rc = launch_sort_pairs(k0, v0, k1, v1, N, 0u, 30, sort_scr, s, debug, &where);
    if (rc) return rc;
    const uint32_t* order = where ? v1 : v0;
    hipLaunchKernelGGL(k_knn_gather, dim3(grid), dim3(256), 0, s, points, order, N, sorted);
    VR_KERNEL_CHECK("knn_gather", s, debug);
    hipLaunchKernelGGL(k_knn_boxes, dim3(nbox), dim3(256), 0, s, (const float4*)sorted, N, bmin, bmax);
    VR_KERNEL_CHECK("knn_boxes", s, debug);
    hipLaunchKernelGGL(k_knn_search, dim3(grid), dim3(256), 0, s, (const float4*)sorted, N, (const float4*)bmin,
                       (const float4*)bmax, nbox, out);
    VR_KERNEL_CHECK("knn_search", s, debug);
    return 0;
}

}  // namespace vr
