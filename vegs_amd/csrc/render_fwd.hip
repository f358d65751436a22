// render_fwd.hip -- tile-based front-to-back alpha compositing of 11 channels
// (rgb, depth, rotation quaternion, scale) + alpha, final transmittance and contributor count.
//
// Replaces the native render stage behind GaussianRasterizer.forward (reference call site
// gaussian_renderer/__init__.py:86-94; the 6-tuple it returns is consumed at :109-119 and by
// loss/normal_guidance.py:3-22).  Semantics: SURVEY.md A.4 with fork assumptions A-1..A-5.
//
// Mapping: one 256-thread workgroup per 16x16 tile = 4 wave64, each wave owning a 16x4 pixel strip,
// one lane per pixel (the reduction target of the forward pass is the pixel, so the pixel owns the
// lane and accumulates in registers).  Splat records are gathered by id with five dwordx4 loads per
// lane and staged in LDS in batches of 256; the inner loop reads them back as wave-uniform
// broadcasts.  The gather of batch k+1 is issued before batch k is blended (register staging) so
// HBM/L2 latency hides behind the blend loop.  Waves whose 64 pixels are all saturated skip the
// blend loop (wave-uniform branch) but keep staging for the others.
#include "vr_host.h"

namespace vr {

constexpr int BATCH = 256;

__global__ void __launch_bounds__(256)
k_render_fwd(Camera cam, const int2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
             const Splat* __restrict__ rec, float* __restrict__ out_color, float* __restrict__ out_depth,
             float* __restrict__ out_quat, float* __restrict__ out_scale, float* __restrict__ out_alpha,
             float* __restrict__ final_T, uint32_t* __restrict__ n_contrib)
{
    __shared__ float4 lds[5][BATCH];
    const int ntiles = cam.gx * cam.gy;
    const int tile = xcd_tile(blockIdx.x, ntiles);
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int px = tx * TILE + (lane & 15), py = ty * TILE + w * 4 + (lane >> 4);
    const bool inside = px < cam.W && py < cam.H;
    const float pxf = (float)px, pyf = (float)py;
    const int2 range = ranges[tile];

    float T = 1.0f;
    float C[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) C[k] = 0.0f;
    uint32_t contributor = 0, last = 0;
    bool done = !inside;

    // register-staged prefetch of the first batch
    float4 st[5];
    int j = range.x + threadIdx.x;
    if (j < range.y) {
        const float4* src = reinterpret_cast<const float4*>(rec + point_list[j]);
#pragma unroll
        for (int k = 0; k < 5; ++k) st[k] = src[k];
    }
    for (int base = range.x; base < range.y; base += BATCH) {
        // all pixels of the tile saturated -> stop (the barrier also protects LDS reuse)
        if (__syncthreads_and(done)) break;
        if (base + (int)threadIdx.x < range.y) {
#pragma unroll
            for (int k = 0; k < 5; ++k) lds[k][threadIdx.x] = st[k];
        }
        __syncthreads();
        // issue the gather of the next batch now; it lands while this batch is blended
        j = base + BATCH + threadIdx.x;
        if (j < range.y) {
            const float4* src = reinterpret_cast<const float4*>(rec + point_list[j]);
#pragma unroll
            for (int k = 0; k < 5; ++k) st[k] = src[k];
        }
        const int n = min(BATCH, range.y - base);
        if (__ballot(!done) != 0ull) {
            for (int k = 0; k < n; ++k) {
                if (done) continue;
                ++contributor;
                const float4 a = lds[0][k];  // x y A B
                const float4 b = lds[1][k];  // C opacity depth r
                float dx, dy;
                const float power = splat_power(a.x, a.y, a.z, a.w, b.x, pxf, pyf, dx, dy);
                if (power > 0.0f) continue;
                const float alpha = fminf(ALPHA_MAX, b.y * vr_exp(power));
                if (alpha < ALPHA_MIN) continue;
                const float test_T = T * (1.0f - alpha);
                if (test_T < T_EPS) { done = true; continue; }
                const float wgt = alpha * T;
                const float4 c = lds[2][k];  // g b qw qx
                const float4 d = lds[3][k];  // qy qz s0 s1
                const float s2 = lds[4][k].x;
                C[0] = fmaf(b.w, wgt, C[0]);
                C[1] = fmaf(c.x, wgt, C[1]);
                C[2] = fmaf(c.y, wgt, C[2]);
                C[3] = fmaf(b.z, wgt, C[3]);
                C[4] = fmaf(c.z, wgt, C[4]);
                C[5] = fmaf(c.w, wgt, C[5]);
                C[6] = fmaf(d.x, wgt, C[6]);
                C[7] = fmaf(d.y, wgt, C[7]);
                C[8] = fmaf(d.z, wgt, C[8]);
                C[9] = fmaf(d.w, wgt, C[9]);
                C[10] = fmaf(s2, wgt, C[10]);
                T = test_T;
                last = contributor;
            }
        }
    }
    if (inside) {
        const size_t N = (size_t)cam.H * cam.W;
        const size_t pix = (size_t)py * cam.W + px;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = fmaf(T, cam.bg[0], C[0]);
        out_color[N + pix] = fmaf(T, cam.bg[1], C[1]);
        out_color[2 * N + pix] = fmaf(T, cam.bg[2], C[2]);
        out_depth[pix] = C[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) out_quat[k * N + pix] = C[4 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) out_scale[k * N + pix] = C[8 + k];
        out_alpha[pix] = 1.0f - T;
    }
}

__global__ void __launch_bounds__(256)
k_count_fragments(const uint32_t* __restrict__ n_contrib, long N, unsigned long long* __restrict__ out)
{
    unsigned long long acc = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256) acc += n_contrib[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

int launch_render_fwd(const Camera& cam, const int2* ranges, const uint32_t* point_list, const Splat* rec,
                      float* out_color, float* out_depth, float* out_quat, float* out_scale, float* out_alpha,
                      float* final_T, uint32_t* n_contrib, hipStream_t s, bool debug)
{
    int ntiles = cam.gx * cam.gy;
    if (ntiles == 0) return 0;
    hipLaunchKernelGGL(k_render_fwd, dim3(ntiles), dim3(256), 0, s, cam, ranges, point_list, rec, out_color,
                       out_depth, out_quat, out_scale, out_alpha, final_T, n_contrib);
    VR_KERNEL_CHECK("render_fwd", s, debug);
    return 0;
}

int launch_count_fragments(const uint32_t* n_contrib, long N, unsigned long long* out_dev, hipStream_t s)
{
    VR_HIP(hipMemsetAsync(out_dev, 0, sizeof(unsigned long long), s));
    if (N > 0) {
        int nb = cdiv(N, 256);
        if (nb > 1024) nb = 1024;
        hipLaunchKernelGGL(k_count_fragments, dim3(nb), dim3(256), 0, s, n_contrib, N, out_dev);
        VR_KERNEL_CHECK("count_fragments", s, false);
    }
    return 0;
}

}  // namespace vr
