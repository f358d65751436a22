// losses.hip -- the per-pixel losses that sit right after the rasterizer in a VEGS training iteration
// (SURVEY.md section 8f, row N1), fused so that they produce dL/d(rasterizer outputs) directly:
//
//   photometric:      Ll1 = l1_loss(image, gt)                         reference utils/loss_utils.py:18-22
//                     ssim(image, gt) (11x11 Gaussian window, sigma 1.5, zero padding, mean)  :30-79
//                     combined by train.py:162-164 as (1-l)*Ll1 + l*(1-ssim)
//   normal guidance:  loss/normal_guidance.py:3-22 with quaternion_to_matrix (utils/graphics_utils.py:204-248)
//                     and cam_normal_to_world_normal (:362-368)
//
// The reference runs these as ~40 ATen launches (five depthwise conv2d forward + their backward, elementwise
// maps, permute/reshape copies of [n_pix,3,3] matrices).  Here: SSIM forward = ONE kernel (tile + halo in LDS,
// separable 11+11 taps on the five window moments, SSIM map, its three partial-derivative maps, block partial
// sums); backward = ONE kernel (separable window over the three derivative maps + the L1 sign term).  The
// normal-guidance loss is one streaming kernel per direction.  All HBM-bound streaming: photometric forward
// reads 2 and writes 3 image-sized planes per channel, backward reads 5 and writes 1.
// Loss sums: per-block partials in double, reduced by a second tiny kernel in a fixed order (deterministic).
#include "vr_host.h"

namespace vr {

// Tile: TW x TH = 16 x 54 output pixels per 256-thread workgroup, input tile 26 x 64 (halo 5).  Both separable passes are
// REGISTER-BLOCKED: a thread produces four adjacent outputs from 14 consecutive inputs it reads once (the 16 x 16 tile
// with one output per thread read 11 inputs per output and was bound by LDS bandwidth: 95 LDS reads per pixel, now 27).
// The horizontal pass has exactly 64 rows x 4 quarters = 256 work items; the vertical one 16 columns x 14 row groups.
// Per output the taps are accumulated in the same order as before (k = 0 .. 10), so the maps are bit-identical.
constexpr int SSIM_R = 5, SSIM_W = 11, TW = 16, TH = 54, TIN_W = TW + 2 * SSIM_R, TIN_H = TH + 2 * SSIM_R;   // 26 x 64
constexpr int TX_STRIDE = TIN_W + 1, HZ_STRIDE = 20, BLK = 4, RUN = BLK + SSIM_W - 1;                        // 14 inputs per run
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;
static_assert(TIN_H * (TW / BLK) == 256, "one horizontal work item per thread");

struct SsimWin { float g[SSIM_W]; };

__device__ __forceinline__ double block_sum_double(double v, double* red)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double s = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return s;
}

__global__ void __launch_bounds__(256)
k_photo_fwd(const float* __restrict__ x, const float* __restrict__ y, int H, int W, SsimWin win,
            float* __restrict__ dmaps, size_t plane_all, double* __restrict__ partial)
{
    __shared__ float tx[TIN_H][TX_STRIDE], ty[TIN_H][TX_STRIDE];
    __shared__ float hz[5][TIN_H][HZ_STRIDE];
    __shared__ double red[4];
    const int c = blockIdx.z, bx = blockIdx.x * TW, by = blockIdx.y * TH;
    const size_t cbase = (size_t)c * H * W;
    for (int i = threadIdx.x; i < TIN_H * TIN_W; i += 256) {
        const int r = i / TIN_W, q = i - r * TIN_W;
        const int gy = by + r - SSIM_R, gx = bx + q - SSIM_R;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        tx[r][q] = in ? x[cbase + (size_t)gy * W + gx] : 0.0f;
        ty[r][q] = in ? y[cbase + (size_t)gy * W + gx] : 0.0f;
    }
    __syncthreads();
    {   // horizontal pass on the five moments x, y, x^2, y^2, xy: row r, outputs 4 qt .. 4 qt + 3
        const int r = threadIdx.x >> 2, q0 = (threadIdx.x & 3) * BLK;
        float a[RUN], b[RUN];
#pragma unroll
        for (int j = 0; j < RUN; ++j) { a[j] = tx[r][q0 + j]; b[j] = ty[r][q0 + j]; }
#pragma unroll
        for (int o = 0; o < BLK; ++o) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
            for (int k = 0; k < SSIM_W; ++k) {
                const float g = win.g[k];
                const float ga = g * a[o + k], gb = g * b[o + k];
                s0 += ga; s1 += gb;
                s2 = fmaf(ga, a[o + k], s2); s3 = fmaf(gb, b[o + k], s3); s4 = fmaf(ga, b[o + k], s4);
            }
            hz[0][r][q0 + o] = s0; hz[1][r][q0 + o] = s1; hz[2][r][q0 + o] = s2; hz[3][r][q0 + o] = s3; hz[4][r][q0 + o] = s4;
        }
    }
    __syncthreads();
    // vertical pass: column lx, output rows 4 grp .. 4 grp + 3
    const int lx = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int gx = bx + lx;
    double l1 = 0.0, ss = 0.0;
    if (grp * BLK < TH && gx < W) {
        float col[5][RUN];
#pragma unroll
        for (int m = 0; m < 5; ++m)
#pragma unroll
            for (int j = 0; j < RUN; ++j) col[m][j] = hz[m][min(grp * BLK + j, TIN_H - 1)][lx];   // (the last group's spare rows)
#pragma unroll
        for (int o = 0; o < BLK; ++o) {
            const int ly = grp * BLK + o, gy = by + ly;
            if (ly >= TH || gy >= H) continue;
            float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
            for (int k = 0; k < SSIM_W; ++k) {
                const float g = win.g[k];
                mu1 = fmaf(g, col[0][o + k], mu1);
                mu2 = fmaf(g, col[1][o + k], mu2);
                e11 = fmaf(g, col[2][o + k], e11);
                e22 = fmaf(g, col[3][o + k], e22);
                e12 = fmaf(g, col[4][o + k], e12);
            }
            const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = e11 - mu1s, s2 = e22 - mu2s, s12 = e12 - mu12;
            const float A1 = 2.0f * mu12 + SSIM_C1, A2 = 2.0f * s12 + SSIM_C2;
            const float B1 = mu1s + mu2s + SSIM_C1, B2 = s1 + s2 + SSIM_C2;
            const float iB1 = 1.0f / B1, iB2 = 1.0f / B2;
            const float S = A1 * A2 * iB1 * iB2;
            ss += (double)S;
            l1 += (double)fabsf(tx[ly + SSIM_R][lx + SSIM_R] - ty[ly + SSIM_R][lx + SSIM_R]);
            if (dmaps) {
                const size_t oo = cbase + (size_t)gy * W + gx;
                // dS/dmu1 (total: also through sigma1^2 = E11 - mu1^2 and sigma12 = E12 - mu1 mu2), dS/dE11, dS/dE12
                dmaps[oo] = 2.0f * mu2 * (A2 - A1) * iB1 * iB2 - 2.0f * mu1 * S * (iB1 - iB2);
                dmaps[plane_all + oo] = -S * iB2;
                dmaps[2 * plane_all + oo] = 2.0f * A1 * iB1 * iB2;
            }
        }
    }
    const double bl1 = block_sum_double(l1, red);
    const double bss = block_sum_double(ss, red);
    if (threadIdx.x == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * b] = bl1;
        partial[2 * b + 1] = bss;
    }
}

// out[j] = scale * sum_b partial[b*stride + j], j < stride <= 4; one block, fixed order
__global__ void __launch_bounds__(256)
k_reduce_partials(const double* __restrict__ partial, long nblocks, int stride, double scale, float* __restrict__ out)
{
    __shared__ double red[4];
    for (int j = 0; j < stride; ++j) {
        double v = 0.0;
        for (long b = threadIdx.x; b < nblocks; b += 256) v += partial[b * stride + j];
        const double s = block_sum_double(v, red);
        if (threadIdx.x == 0) out[j] = (float)(s * scale);
    }
}

__global__ void __launch_bounds__(256)
k_photo_bwd(const float* __restrict__ x, const float* __restrict__ y, int H, int W, SsimWin win,
            const float* __restrict__ dmaps, size_t plane_all, const float* __restrict__ g_l1,
            const float* __restrict__ g_ssim, float w_l1, float w_ssim, float inv_n, float* __restrict__ dx)
{
    __shared__ float t[3][TIN_H][TX_STRIDE];
    __shared__ float hz[3][TIN_H][HZ_STRIDE];
    const int c = blockIdx.z, bx = blockIdx.x * TW, by = blockIdx.y * TH;
    const size_t cbase = (size_t)c * H * W;
    // (upstream scalars from device memory, times the host-side weights of the combined training loss)
    const float wl1 = ((g_l1 ? *g_l1 : 0.0f) * w_l1) * inv_n, wss = ((g_ssim ? *g_ssim : 0.0f) * w_ssim) * inv_n;
    for (int i = threadIdx.x; i < TIN_H * TIN_W; i += 256) {
        const int r = i / TIN_W, q = i - r * TIN_W;
        const int gy = by + r - SSIM_R, gx = bx + q - SSIM_R;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t o = cbase + (size_t)gy * W + gx;
#pragma unroll
        for (int m = 0; m < 3; ++m) t[m][r][q] = in ? dmaps[m * plane_all + o] : 0.0f;
    }
    __syncthreads();
    {
        const int r = threadIdx.x >> 2, q0 = (threadIdx.x & 3) * BLK;
        float v[3][RUN];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int j = 0; j < RUN; ++j) v[m][j] = t[m][r][q0 + j];
#pragma unroll
        for (int o = 0; o < BLK; ++o) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < SSIM_W; ++k) {
                const float g = win.g[k];
                s0 = fmaf(g, v[0][o + k], s0); s1 = fmaf(g, v[1][o + k], s1); s2 = fmaf(g, v[2][o + k], s2);
            }
            hz[0][r][q0 + o] = s0; hz[1][r][q0 + o] = s1; hz[2][r][q0 + o] = s2;
        }
    }
    __syncthreads();
    const int lx = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int gx = bx + lx;
    if (grp * BLK >= TH || gx >= W) return;
    float col[3][RUN];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int j = 0; j < RUN; ++j) col[m][j] = hz[m][min(grp * BLK + j, TIN_H - 1)][lx];
#pragma unroll
    for (int o = 0; o < BLK; ++o) {
        const int ly = grp * BLK + o, gy = by + ly;
        if (ly >= TH || gy >= H) continue;
        float cmu = 0.f, c11 = 0.f, c12 = 0.f;
#pragma unroll
        for (int k = 0; k < SSIM_W; ++k) {
            const float g = win.g[k];
            cmu = fmaf(g, col[0][o + k], cmu); c11 = fmaf(g, col[1][o + k], c11); c12 = fmaf(g, col[2][o + k], c12);
        }
        const size_t oo = cbase + (size_t)gy * W + gx;
        const float xv = x[oo], yv = y[oo];
        const float d = xv - yv;
        const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
        dx[oo] = fmaf(wss, cmu + 2.0f * xv * c11 + yv * c12, wl1 * sgn);
    }
}

// ---------------------------------------------------------------- normal guidance
struct Mat3 { float m[9]; };

struct NgPixel {
    float c[3];      // column j of R(q) . n_world
    float nw[3];
    float q[4], s[3], two_s;
    bool empty;      // guard: no Gaussian covers the pixel (q == 0), evaluated with q = (1,1,1,1) and no gradient to q
};

__device__ __forceinline__ void ng_load(const float* __restrict__ cov_quat, const float* __restrict__ cov_scale,
                                        const float* __restrict__ normal, const Mat3& Rc, size_t p, size_t N, NgPixel& o,
                                        bool guard = false)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) o.q[k] = cov_quat[k * N + p];
    o.empty = guard && o.q[0] * o.q[0] + o.q[1] * o.q[1] + o.q[2] * o.q[2] + o.q[3] * o.q[3] <= 0.0f;
    if (o.empty) { o.q[0] = 1.0f; o.q[1] = 1.0f; o.q[2] = 1.0f; o.q[3] = 1.0f; }
#pragma unroll
    for (int k = 0; k < 3; ++k) o.s[k] = cov_scale[k * N + p];
    const float n0 = normal[p], n1 = normal[N + p], n2 = normal[2 * N + p];
#pragma unroll
    for (int i = 0; i < 3; ++i) o.nw[i] = Rc.m[3 * i] * n0 + Rc.m[3 * i + 1] * n1 + Rc.m[3 * i + 2] * n2;
    const float r = o.q[0], i = o.q[1], j = o.q[2], k = o.q[3];
    o.two_s = 2.0f / (r * r + i * i + j * j + k * k);     // |q| = 0 (uncovered pixel) -> inf -> NaN, as in the reference
    const float t = o.two_s;
    const float R00 = 1.0f - t * (j * j + k * k), R01 = t * (i * j - k * r), R02 = t * (i * k + j * r);
    const float R10 = t * (i * j + k * r), R11 = 1.0f - t * (i * i + k * k), R12 = t * (j * k - i * r);
    const float R20 = t * (i * k - j * r), R21 = t * (j * k + i * r), R22 = 1.0f - t * (i * i + j * j);
    o.c[0] = R00 * o.nw[0] + R10 * o.nw[1] + R20 * o.nw[2];
    o.c[1] = R01 * o.nw[0] + R11 * o.nw[1] + R21 * o.nw[2];
    o.c[2] = R02 * o.nw[0] + R12 * o.nw[1] + R22 * o.nw[2];
}

__global__ void __launch_bounds__(256)
k_ng_fwd(const float* __restrict__ cov_quat, const float* __restrict__ cov_scale, const float* __restrict__ normal,
         Mat3 Rc, size_t N, double* __restrict__ partial, bool guard)
{
    __shared__ double red[4];
    double t1 = 0.0, t2 = 0.0;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < N; p += (size_t)gridDim.x * 256) {
        NgPixel px;
        ng_load(cov_quat, cov_scale, normal, Rc, p, N, px, guard);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            t1 += (double)fabsf(px.c[j]);
            t2 += (double)fabsf(px.c[j] * px.s[j]);
        }
    }
    const double b1 = block_sum_double(t1, red);
    const double b2 = block_sum_double(t2, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = 0.8 * b1 + 0.2 * b2;
}

__device__ __forceinline__ float sgnf(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : (v == 0.0f ? 0.0f : v)); }

__global__ void __launch_bounds__(256)
k_ng_bwd(const float* __restrict__ cov_quat, const float* __restrict__ cov_scale, const float* __restrict__ normal,
         Mat3 Rc, size_t N, const float* __restrict__ gup, float weight, float inv_3n, float* __restrict__ dquat,
         float* __restrict__ dscale, bool guard)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    NgPixel px;
    ng_load(cov_quat, cov_scale, normal, Rc, p, N, px, guard);
    const float gl = (*gup * weight) * inv_3n;
    const float k1 = 0.8f * gl, k2 = 0.2f * gl;
    float sg[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        dscale[j * N + p] = k2 * sgnf(px.c[j] * px.s[j]) * px.c[j];
        sg[j] = k1 * sgnf(px.c[j]);
    }
    // G[i][j] = dL/dR[i][j] = k1 sign(c_j) n_i ; chain through R = I + two_s * M(q)
    float G[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) G[i][j] = sg[j] * px.nw[i];
    const float r = px.q[0], i = px.q[1], j = px.q[2], k = px.q[3], t = px.two_s;
    const float GM = G[0][0] * -(j * j + k * k) + G[0][1] * (i * j - k * r) + G[0][2] * (i * k + j * r) +
                     G[1][0] * (i * j + k * r) + G[1][1] * -(i * i + k * k) + G[1][2] * (j * k - i * r) +
                     G[2][0] * (i * k - j * r) + G[2][1] * (j * k + i * r) + G[2][2] * -(i * i + j * j);
    const float dr = -k * G[0][1] + j * G[0][2] + k * G[1][0] - i * G[1][2] - j * G[2][0] + i * G[2][1];
    const float di = j * (G[0][1] + G[1][0]) + k * (G[0][2] + G[2][0]) - 2.0f * i * (G[1][1] + G[2][2]) + r * (G[2][1] - G[1][2]);
    const float dj = -2.0f * j * (G[0][0] + G[2][2]) + i * (G[0][1] + G[1][0]) + r * (G[0][2] - G[2][0]) + k * (G[1][2] + G[2][1]);
    const float dk = -2.0f * k * (G[0][0] + G[1][1]) + r * (G[1][0] - G[0][1]) + i * (G[0][2] + G[2][0]) + j * (G[1][2] + G[2][1]);
    const float tt = t * t * GM;
    dquat[p] = px.empty ? 0.0f : t * dr - tt * r;
    dquat[N + p] = px.empty ? 0.0f : t * di - tt * i;
    dquat[2 * N + p] = px.empty ? 0.0f : t * dj - tt * j;
    dquat[3 * N + p] = px.empty ? 0.0f : t * dk - tt * k;
}

// The combined loss of train.py:162-168 from the two kernels' partial sums, one block, fixed order:
//   aux = { Ll1, mean SSIM, Lng };  loss = (1 - lambda) Ll1 + lambda (1 - ssim) + lambda_n Lng
__global__ void __launch_bounds__(256)
k_loss_terms(const double* __restrict__ photo_partial, long photo_blocks, double photo_scale,
             const double* __restrict__ ng_partial, long ng_blocks, double ng_scale, float lambda_dssim, float lambda_dnormal,
             float* __restrict__ loss, float* __restrict__ aux)
{
    __shared__ double red[4];
    double v0 = 0.0, v1 = 0.0, v2 = 0.0;
    for (long b = threadIdx.x; b < photo_blocks; b += 256) { v0 += photo_partial[2 * b]; v1 += photo_partial[2 * b + 1]; }
    for (long b = threadIdx.x; b < ng_blocks; b += 256) v2 += ng_partial[b];
    const double s0 = block_sum_double(v0, red), s1 = block_sum_double(v1, red), s2 = block_sum_double(v2, red);
    if (threadIdx.x == 0) {
        const float l1 = (float)(s0 * photo_scale), ss = (float)(s1 * photo_scale), ng = (float)(s2 * ng_scale);
        aux[0] = l1; aux[1] = ss; aux[2] = ng;
        // the reference's float32 sequence: (1 - l) * Ll1 + l * (1 - ssim), then += l_n * Lng
        *loss = ((1.0f - lambda_dssim) * l1 + lambda_dssim * (1.0f - ss)) + lambda_dnormal * ng;
    }
}

// ---------------------------------------------------------------- host side
static SsimWin make_window()
{
    // utils/loss_utils.py:30-32: float32 tensor of exp(-(x-5)^2 / (2*1.5^2)) divided by its float32 sum
    SsimWin w;
    float sum = 0.0f;
    for (int x = 0; x < SSIM_W; ++x) {
        w.g[x] = (float)exp(-(double)((x - SSIM_R) * (x - SSIM_R)) / (2.0 * 1.5 * 1.5));
        sum += w.g[x];
    }
    for (int x = 0; x < SSIM_W; ++x) w.g[x] /= sum;
    return w;
}

static dim3 photo_grid(int C, int H, int W) { return dim3(cdiv(W, TW), cdiv(H, TH), C); }

size_t photometric_scratch_bytes(int C, int H, int W)
{
    const dim3 g = photo_grid(C, H, W);
    return align_up((size_t)g.x * g.y * g.z * 2 * sizeof(double), 256);
}

int launch_photometric_fwd(const float* image, const float* gt, int C, int H, int W, float* sums, float* dmaps,
                           void* scratch, hipStream_t s, bool debug)
{
    const dim3 g = photo_grid(C, H, W);
    const size_t n = (size_t)C * H * W;
    double* partial = (double*)scratch;
    hipLaunchKernelGGL(k_photo_fwd, g, dim3(256), 0, s, image, gt, H, W, make_window(), dmaps, n, partial);
    VR_KERNEL_CHECK("photo_fwd", s, debug);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, s, (const double*)partial, (long)g.x * g.y * g.z, 2,
                       1.0 / (double)n, sums);
    VR_KERNEL_CHECK("photo_reduce", s, debug);
    return 0;
}

int launch_photometric_bwd(const float* image, const float* gt, int C, int H, int W, const float* dmaps,
                           const float* g_l1, const float* g_ssim, float* dL_dimage, hipStream_t s, bool debug)
{
    const size_t n = (size_t)C * H * W;
    hipLaunchKernelGGL(k_photo_bwd, photo_grid(C, H, W), dim3(256), 0, s, image, gt, H, W, make_window(), dmaps, n, g_l1,
                       g_ssim, 1.0f, 1.0f, (float)(1.0 / (double)n), dL_dimage);
    VR_KERNEL_CHECK("photo_bwd", s, debug);
    return 0;
}

static int ng_blocks(size_t N) { return (int)std::min<size_t>(cdiv((long)N, 256), 2048); }

size_t normal_guidance_scratch_bytes(int H, int W) { return align_up((size_t)ng_blocks((size_t)H * W) * sizeof(double), 256); }

int launch_normal_guidance_fwd(const float* cov_quat, const float* cov_scale, const float* normal, const float* R9, int H,
                               int W, float* loss, void* scratch, hipStream_t s, bool debug)
{
    Mat3 Rc;
    for (int i = 0; i < 9; ++i) Rc.m[i] = R9[i];
    const size_t N = (size_t)H * W;
    const int nb = ng_blocks(N);
    hipLaunchKernelGGL(k_ng_fwd, dim3(nb), dim3(256), 0, s, cov_quat, cov_scale, normal, Rc, N, (double*)scratch, false);
    VR_KERNEL_CHECK("ng_fwd", s, debug);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, s, (const double*)scratch, (long)nb, 1,
                       1.0 / (3.0 * (double)N), loss);
    VR_KERNEL_CHECK("ng_reduce", s, debug);
    return 0;
}

int launch_normal_guidance_bwd(const float* cov_quat, const float* cov_scale, const float* normal, const float* R9, int H,
                               int W, const float* g, float* dL_dquat, float* dL_dscale, hipStream_t s, bool debug)
{
    Mat3 Rc;
    for (int i = 0; i < 9; ++i) Rc.m[i] = R9[i];
    const size_t N = (size_t)H * W;
    hipLaunchKernelGGL(k_ng_bwd, dim3(cdiv((long)N, 256)), dim3(256), 0, s, cov_quat, cov_scale, normal, Rc, N, g, 1.0f,
                       (float)(1.0 / (3.0 * (double)N)), dL_dquat, dL_dscale, false);
    VR_KERNEL_CHECK("ng_bwd", s, debug);
    return 0;
}

size_t training_loss_scratch_bytes(int C, int H, int W)
{
    return photometric_scratch_bytes(C, H, W) + normal_guidance_scratch_bytes(H, W);
}

// train.py:162-168 in three launches: both losses' partial sums, then k_loss_terms
int launch_training_loss_fwd(const float* image, const float* gt, int C, int H, int W, const float* cov_quat,
                             const float* cov_scale, const float* normal, const float* R9, float lambda_dssim,
                             float lambda_dnormal, bool guard, float* loss, float* aux, float* dmaps, void* scratch,
                             hipStream_t s, bool debug)
{
    const dim3 g = photo_grid(C, H, W);
    const size_t n = (size_t)C * H * W, N = (size_t)H * W;
    double* pp = (double*)scratch;
    double* np = (double*)((char*)scratch + photometric_scratch_bytes(C, H, W));
    hipLaunchKernelGGL(k_photo_fwd, g, dim3(256), 0, s, image, gt, H, W, make_window(), dmaps, n, pp);
    VR_KERNEL_CHECK("photo_fwd", s, debug);
    Mat3 Rc;
    for (int i = 0; i < 9; ++i) Rc.m[i] = R9[i];
    const int nb = ng_blocks(N);
    hipLaunchKernelGGL(k_ng_fwd, dim3(nb), dim3(256), 0, s, cov_quat, cov_scale, normal, Rc, N, np, guard);
    VR_KERNEL_CHECK("ng_fwd", s, debug);
    hipLaunchKernelGGL(k_loss_terms, dim3(1), dim3(256), 0, s, (const double*)pp, (long)g.x * g.y * g.z, 1.0 / (double)n,
                       (const double*)np, (long)nb, 1.0 / (3.0 * (double)N), lambda_dssim, lambda_dnormal, loss, aux);
    VR_KERNEL_CHECK("loss_terms", s, debug);
    return 0;
}

int launch_training_loss_bwd(const float* image, const float* gt, int C, int H, int W, const float* dmaps,
                             const float* cov_quat, const float* cov_scale, const float* normal, const float* R9,
                             float lambda_dssim, float lambda_dnormal, bool guard, const float* g, float* dL_dimage,
                             float* dL_dquat, float* dL_dscale, hipStream_t s, bool debug)
{
    const size_t n = (size_t)C * H * W, N = (size_t)H * W;
    hipLaunchKernelGGL(k_photo_bwd, photo_grid(C, H, W), dim3(256), 0, s, image, gt, H, W, make_window(), dmaps, n, g, g,
                       1.0f - lambda_dssim, -lambda_dssim, (float)(1.0 / (double)n), dL_dimage);
    VR_KERNEL_CHECK("photo_bwd", s, debug);
    Mat3 Rc;
    for (int i = 0; i < 9; ++i) Rc.m[i] = R9[i];
    hipLaunchKernelGGL(k_ng_bwd, dim3(cdiv((long)N, 256)), dim3(256), 0, s, cov_quat, cov_scale, normal, Rc, N, g,
                       lambda_dnormal, (float)(1.0 / (3.0 * (double)N)), dL_dquat, dL_dscale, guard);
    VR_KERNEL_CHECK("ng_bwd", s, debug);
    return 0;
}

}  // namespace vr
