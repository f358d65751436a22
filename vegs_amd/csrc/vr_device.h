// vr_device.h -- shared device-side definitions for the gfx950 rasterizer kernels.
//
// The fp32 operation order of every function here is part of the behavioural spec (see
// DESIGN.md "numerics"): translation units are compiled with -ffp-contract=off and every
// fused multiply-add is an explicit fmaf(), so that radii, tile rectangles, sort keys and
// forward images are bit-reproducible against the CPU oracle used by the tests.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vr {

constexpr int TILE = 16;          // pixels per tile edge (16x16 = 256 threads = 4 waves of 64)
constexpr int NCH = 11;           // blended channels: rgb(3) depth(1) quat(4) scale(3)
constexpr float NEAR_Z = 0.2f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float T_EPS = 0.0001f;

// Per-Gaussian record written by preprocess for visible Gaussians and gathered by the render
// kernels: 5 x 16 B so a record is fetched with five dwordx4 loads.
struct __attribute__((aligned(16))) Splat {
    float x, y, conA, conB;          // q0: pixel centre, conic A,B
    float conC, opacity, thr, depth; // q1: conic C, opacity, power threshold (see splat_thr), view depth
    float r, g, b, qw;               // q2: colour, quaternion w       (input rotation row, A-2)
    float qx, qy, qz, s0;            // q3: quaternion x,y,z, scale 0  (input scale row, A-3)
    float s1, s2;                    // q4: scale 1,2
    uint32_t clamped;                //     bit c set: colour channel c was clamped at 0
    uint32_t pad0;
};
static_assert(sizeof(Splat) == 80, "Splat must be 80 bytes");

// VrSettings.flags (include/vegs_rast.h, VrFlags; api.hip asserts that the values agree)
constexpr uint32_t FLAG_SCALE_MODIFIED = 1u << 0;       // cov_scale blends scale_modifier * scales
constexpr uint32_t FLAG_DEPTH_NORMALIZED = 1u << 1;     // depth = sum(w z) / (1 - T_final)
constexpr uint32_t FLAG_EXTRA_NO_ALPHA_GRAD = 1u << 2;  // depth/quat/scale channels: no gradient through alpha
constexpr uint32_t FLAG_FILL_EMPTY = 1u << 3;           // cov_quat += T_final * (1,0,0,0)
constexpr uint32_t FLAG_DETERMINISTIC = 1u << 8;        // backward without atomics
constexpr uint32_t FLAG_SCAN_BINNING = 1u << 9;         // binning with the scan-based (multi-launch) radix passes
constexpr uint32_t FLAG_ROUNDS_OFF = 1u << 10;          // forward: all list segments at once
constexpr uint32_t FLAG_ROUNDS_ON = 1u << 11;           // forward: segment rounds whatever the list density
constexpr uint32_t FLAG_RAW_PARAMS = 1u << 12;          // opacities / scales / rotations are raw parameters (activated here)
constexpr uint32_t FLAG_FAST_EXP = 1u << 13;            // the compositing's 2^x by the hardware's v_exp_f32 (forward AND backward)
constexpr uint32_t FLAG_VERIFY_BINNING = 1u << 14;      // forward: wait for the binning's guard word; a tripped view is re-binned without waits
constexpr uint32_t FLAG_FULL_TILE_LISTS = 1u << 15;     // tile lists hold the reference's full rectangles (no tile test)
constexpr uint32_t FLAG_ACCUMULATE_GRADS = 1u << 16;    // backward ADDS the view's gradients into the caller's arrays (visible rows only)

// The model's activations (scene/gaussian_model.py:37-45), shared by vr_activations_* and the VR_FLAG_RAW_PARAMS path
constexpr float NORMALIZE_EPS = 1e-12f;   // F.normalize's default eps
__device__ __forceinline__ float act_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ void act_normalize(const float q[4], float y[4])
{
    const float n = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), NORMALIZE_EPS);
    y[0] = q[0] / n; y[1] = q[1] / n; y[2] = q[2] / n; y[3] = q[3] / n;
}
// y = x / n, n = max(|x|, eps): dx = (g - y <y, g>) / n where the norm is not clamped, g / eps where it is
__device__ __forceinline__ void act_normalize_bwd(const float q[4], const float g[4], float d[4])
{
    const float len = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (len > NORMALIZE_EPS) {
        const float inv = 1.0f / len;
        const float y0 = q[0] * inv, y1 = q[1] * inv, y2 = q[2] * inv, y3 = q[3] * inv;
        const float dot = y0 * g[0] + y1 * g[1] + y2 * g[2] + y3 * g[3];
        d[0] = (g[0] - y0 * dot) * inv; d[1] = (g[1] - y1 * dot) * inv; d[2] = (g[2] - y2 * dot) * inv; d[3] = (g[3] - y3 * dot) * inv;
    } else {
        d[0] = g[0] / NORMALIZE_EPS; d[1] = g[1] / NORMALIZE_EPS; d[2] = g[2] / NORMALIZE_EPS; d[3] = g[3] / NORMALIZE_EPS;
    }
}

struct Camera {
    int H, W, gx, gy;
    float tanfovx, tanfovy, fx, fy;
    float mod;
    int deg, M;
    uint32_t flags;
    const float* view;
    const float* proj;
    const float* campos;
    const float* bg;
};

// non-temporal 16-byte accesses for data that is streamed exactly once (SH rows, dL/dSH rows)
typedef float vr_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p)
{
    const vr_f4 v = __builtin_nontemporal_load(reinterpret_cast<const vr_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float4 v, float4* p)
{
    vr_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<vr_f4*>(p));
}

// 2^x, x <= 0, from IEEE basic operations only: bit-identical on host and device (the CPU checker restates it operation for operation).
// x = n + f, n = rint(x), |f| <= 1/2 (exact subtraction); 2^f by a degree-5 polynomial fitted on [-1/2, 1/2] with
// p(0) = 1 (relative error 1.6e-7 in fp32); scaled by 2^n.  The compositing evaluates exp(power) as 2^(power log2 e)
// with log2 e folded into the conic once per splat (splat_k2 / splat_power2): 9 instructions per evaluation where
// exp(x) by range reduction in natural units took 12 (round 3; the loops are VALU-bound).
constexpr float LOG2E = 1.44269504088896341f;
constexpr float EXP2_C1 = 0.6931470036506653f, EXP2_C2 = 0.24022243916988373f, EXP2_C3 = 0.05550731346011162f,
                EXP2_C4 = 0.009671415202319622f, EXP2_C5 = 0.0013264892622828484f;
// Unspecified below x = -126 (all the compositing kernels ever USE: they only keep the value for power2 >= thr2 > -9).
__device__ __forceinline__ float vr_exp2_unclamped(float x)
{
    const float n = rintf(x);
    const float f = x - n;
    float p = EXP2_C5;
    p = fmaf(p, f, EXP2_C4);
    p = fmaf(p, f, EXP2_C3);
    p = fmaf(p, f, EXP2_C2);
    p = fmaf(p, f, EXP2_C1);
    p = fmaf(p, f, 1.0f);
    return ldexpf(p, (int)n);
}
__device__ __forceinline__ float vr_exp2(float x) { return x < -126.0f ? 0.0f : vr_exp2_unclamped(x); }
// The compositing kernels' 2^x: the bit-exact polynomial above (FAST = false: what the CPU checker restates), or ONE
// transcendental instruction (VR_FLAG_FAST_EXP: v_exp_f32, ~1 ulp, flushes below 2^-126 -- never used there).  The same
// choice in k_seg_alpha, k_seg_blend and k_seg_bwd of a view: forward and backward then agree on every fragment's alpha.
template <bool FAST>
__device__ __forceinline__ float exp2_sel(float x) { return FAST ? __builtin_amdgcn_exp2f(x) : vr_exp2_unclamped(x); }

// ---- wave-cooperative linear copies between a contiguous global block and LDS (n floats, one wave).  16-byte
// vector path when the global address is aligned (all of a lane's loads in flight before the first store), plus a
// scalar tail; scalar path otherwise.  Used for the split SH storage, whose LDS row stride is the memory's own.
// `rows` / `row_len`: when row_len > 0 the block is `rows`-masked rows of row_len floats and only float4s that touch a
// row whose bit is set are fetched (the others are never read back by their lanes).
template <int MAXV>
__device__ __forceinline__ void wave_copy_to_lds(const float* __restrict__ src, float* __restrict__ lds, int n, int lane,
                                                 unsigned long long rows = ~0ull, int row_len = 0)
{
    if ((reinterpret_cast<size_t>(src) & 15) == 0) {
        const int nv = n >> 2;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(lds);
        float4 tmp[MAXV];
        bool want[MAXV];
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int v = lane + 64 * j;
            want[j] = v < nv;
            if (row_len > 0 && want[j]) {
                const int r0 = (4 * v) / row_len, r1 = (4 * v + 3) / row_len;
                want[j] = (((rows >> r0) | (rows >> (r1 < 64 ? r1 : 63))) & 1ull) != 0ull;
            }
            if (want[j]) tmp[j] = nt_load4(&s4[v]);
        }
#pragma unroll
        for (int j = 0; j < MAXV; ++j)
            if (want[j]) d4[lane + 64 * j] = tmp[j];
        if (lane < (n & 3)) lds[(nv << 2) + lane] = src[(nv << 2) + lane];
    } else {
        for (int e = lane; e < n; e += 64) lds[e] = src[e];
    }
}
template <int MAXV>
__device__ __forceinline__ void wave_copy_from_lds(float* __restrict__ dst, const float* __restrict__ lds, int n, int lane,
                                                   bool zeros)
{
    if ((reinterpret_cast<size_t>(dst) & 15) == 0) {
        const int nv = n >> 2;
        float4* d4 = reinterpret_cast<float4*>(dst);
        const float4* s4 = reinterpret_cast<const float4*>(lds);
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            if (lane + 64 * j < nv) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!zeros) v = s4[lane + 64 * j];
                nt_store4(v, &d4[lane + 64 * j]);
            }
        }
        if (lane < (n & 3)) dst[(nv << 2) + lane] = zeros ? 0.0f : lds[(nv << 2) + lane];
    } else {
        for (int e = lane; e < n; e += 64) dst[e] = zeros ? 0.0f : lds[e];
    }
}

// ---- packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 process two floats per lane per issue):
// two-wide helpers for the compositing arithmetic of the backward pass
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 f2_splat(float v) { f2 r = {v, v}; return r; }
__device__ __forceinline__ f2 f2_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void xform43(const float* __restrict__ m, float px, float py, float pz, float& ox,
                                        float& oy, float& oz)
{
    ox = fmaf(m[8], pz, fmaf(m[4], py, fmaf(m[0], px, m[12])));
    oy = fmaf(m[9], pz, fmaf(m[5], py, fmaf(m[1], px, m[13])));
    oz = fmaf(m[10], pz, fmaf(m[6], py, fmaf(m[2], px, m[14])));
}
__device__ __forceinline__ float xform_w(const float* __restrict__ m, float px, float py, float pz)
{
    return fmaf(m[11], pz, fmaf(m[7], py, fmaf(m[3], px, m[15])));
}

__device__ __forceinline__ float ndc2pix(float v, int S) { return ((v + 1.0f) * (float)S - 1.0f) * 0.5f; }

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// tile rectangle [x0,x1) x [y0,y1) touched by a splat of integer radius rad centred at (px,py)
__device__ __forceinline__ void tile_rect(float px, float py, int rad, int gx, int gy, int& x0, int& y0, int& x1,
                                          int& y1)
{
    x0 = clampi((int)((px - (float)rad) / (float)TILE), 0, gx);
    y0 = clampi((int)((py - (float)rad) / (float)TILE), 0, gy);
    x1 = clampi((int)((px + (float)rad + (float)(TILE - 1)) / (float)TILE), 0, gx);
    y1 = clampi((int)((py + (float)rad + (float)(TILE - 1)) / (float)TILE), 0, gy);
}

// Conservative lower bound on the exponent below which alpha = opacity*exp(power) is certainly
// < 1/255 (1 % slack in the exponent >> the error of vr_exp): lets a whole wave skip a splat without
// evaluating exp.  Never decides anything by itself -- the exact alpha test still follows.
__device__ __forceinline__ float splat_thr(float opacity) { return -__logf(255.0f * opacity) - 0.01f; }

// Strip relevance.  The reference's tile rectangle comes from the 3-sigma radius of the LARGER axis, so
// most tile-list entries never reach alpha >= 1/255 inside a given 8x8 pixel region (VEGS discs are
// thin ellipses).  Does the footprint ellipse  A dx^2 + 2B dx dy + C dy^2 <= k,  k = -2 thr, intersect
// the strip's rectangle of pixel centres?  Minimum of the convex quadratic over the rectangle: 0 if the
// centre is inside, otherwise on one of the four edges (1-D clamped minimum).  Conservative by
// construction (thr already carries slack; the comparison adds a margin): a strip is only skipped when
// the splat provably contributes nothing to it, so skipping never changes a result.
__device__ __forceinline__ float quad_edge_min(float a, float inv_a, float b, float c, float fixed, float lo, float hi)
{
    // min over t in [lo,hi] of  a t^2 + 2 b t fixed + c fixed^2   (inv_a ~ 1/a: an inexact minimiser only
    // raises the value by a*delta^2, far below the safety margin of the caller)
    const float t = fminf(hi, fmaxf(lo, -b * fixed * inv_a));
    return fmaf(fmaf(a, t, 2.0f * b * fixed), t, c * fixed * fixed);
}
// Pixel regions of a 16x16 tile: wave w of a segment workgroup owns region w, lane l its pixel
// (REGION_W x REGION_H pixels, REGION_W * REGION_H = 64).
constexpr int REGION_W = 8, REGION_H = 8;
constexpr int REGIONS_X = TILE / REGION_W;   // regions per tile row
__device__ __forceinline__ int region_x(int w, int lane) { return (w % REGIONS_X) * REGION_W + (lane % REGION_W); }
__device__ __forceinline__ int region_y(int w, int lane) { return (w / REGIONS_X) * REGION_H + (lane / REGION_W); }

// Does the footprint ellipse reach the rectangle of pixel centres [xl, xh] x [yl, yh] (coordinates relative to the
// splat centre)?  lim = k * 1.001 + 0.001 with k = -2 thr; A, C > 0 and A C - B^2 > 0 are the caller's business.
__device__ __forceinline__ bool rect_relevant(float A, float inv_A, float B, float C, float inv_C, float lim, float xl,
                                              float xh, float yl, float yh)
{
    const bool in = xl <= 0.0f && xh >= 0.0f && yl <= 0.0f && yh >= 0.0f;
    float q = quad_edge_min(A, inv_A, B, C, yl, xl, xh);
    q = fminf(q, quad_edge_min(A, inv_A, B, C, yh, xl, xh));
    q = fminf(q, quad_edge_min(C, inv_C, B, A, xl, yl, yh));
    q = fminf(q, quad_edge_min(C, inv_C, B, A, xh, yl, yh));
    return in || q <= lim;
}

// the same with the two edges FACING the centre only (the minimum of a convex quadratic over a rectangle that does not hold
// its centre lies on them): half the work of rect_relevant, equally conservative
__device__ __forceinline__ bool rect_relevant_facing(float A, float inv_A, float B, float C, float inv_C, float lim, float xl,
                                                     float xh, float yl, float yh)
{
    const bool in_x = xl <= 0.0f && xh >= 0.0f, in_y = yl <= 0.0f && yh >= 0.0f;
    const float fy = yl > 0.0f ? yl : yh, fx = xl > 0.0f ? xl : xh;
    const float qy = quad_edge_min(A, inv_A, B, C, fy, xl, xh);
    const float qx = quad_edge_min(C, inv_C, B, A, fx, yl, yh);
    const float q = in_y ? qx : (in_x ? qy : fminf(qx, qy));
    return (in_x && in_y) || q <= lim;
}

// relevance of one splat for the four regions of a tile (bit s of the result = region s)
__device__ __forceinline__ uint32_t strips_relevant_exact(float sx, float sy, float A, float B, float C, float thr,
                                                          float x0, float y0)
{
    const float k = -2.0f * thr;
    if (!(k > 0.0f)) return 0u;
    // The edge minimisation assumes a positive-definite conic.  det = a*c - b*b of a huge, thin splat can cancel to
    // <= 0 in fp32 (and an invalid cov3D_precomp can be indefinite outright); then "closest point on the edge" is a
    // maximum, not a minimum.  Such splats are simply relevant everywhere: the exact per-pixel rule decides.
    if (!(A > 0.0f) || !(C > 0.0f) || !(A * C - B * B > 0.0f)) return 0xFu;
    const float lim = k * 1.001f + 0.001f;
    const float inv_A = __builtin_amdgcn_rcpf(A), inv_C = __builtin_amdgcn_rcpf(C);
    uint32_t bits = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float xl = x0 + (float)((s % REGIONS_X) * REGION_W) - sx, xh = xl + (float)(REGION_W - 1);
        const float yl = y0 + (float)((s / REGIONS_X) * REGION_H) - sy, yh = yl + (float)(REGION_H - 1);
        bits |= rect_relevant(A, inv_A, B, C, inv_C, lim, xl, xh, yl, yh) ? (1u << s) : 0u;
    }
    return bits;
}

// ---- TIGHT TILE LISTS (round 4).  The reference's tile rectangle comes from the 3-sigma radius of the LARGER axis: on a
// street scene a third of the (Gaussian, tile) pairs it emits cannot reach alpha >= 1/255 at ANY pixel centre of the tile
// (thin discs, faint splats; profiles/experiments/README.md).  Such a pair is a no-op for every pixel -- the per-pixel rule
// skips it -- so leaving it out of the tile list changes no radius and images / gradients by rounding only (the sums are
// grouped by 256-entry list segments, which start at other entries: ~5e-7, or one fragment of weight < 1e-4 in a pixel whose
// transmittance sits within an ulp of the 1e-4 stop test); it shortens the lists the
// sorts, the compositing kernels and their per-segment buffers are sized by.  The preprocess kernel therefore tests every
// tile of a rectangle of up to 64 tiles -- every cell of k x k tiles of a larger one, tile_cells -- (the same
// conservative ellipse-vs-rectangle test as the strip masks, over the tiles' 16 x 16 pixel centres) and hands the binning a
// 64-bit mask with the rectangle.  The test is part of the
// LIST DEFINITION -- the CPU checker builds the same lists -- so it is written with IEEE basic operations only (no rcp,
// no hardware log): bit-identical on host and device.  VR_FLAG_FULL_TILE_LISTS restores the reference's full rectangles.
constexpr int TIGHT_MAX_TILES = 64;
constexpr uint32_t TILE_COUNT_MASK = 0x7FFFFFFFu;  // tile_count[i]: list entries of Gaussian i; bit 31: its rectangle has more than 64 tiles
constexpr uint32_t DEPTH_KEY_NONE = 0xFFFFFFFFu;   // depth key of a Gaussian without list entries (real keys: bits of a positive float)
constexpr float VR_LN2 = 0.693147180559945309f;
// ln(v), v > 0: v = m 2^e with m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh(s), s = (m - 1) / (m + 1), four odd terms
// (|s| <= 0.172: truncation < 4e-8).  The caller adds 1 % slack; what matters is that host and device agree bit for bit.
__host__ __device__ inline float vr_ln_repro(float v)
{
    uint32_t b;
    __builtin_memcpy(&b, &v, 4);
    int e = (int)(b >> 23) - 127;
    uint32_t mb = (b & 0x007FFFFFu) | 0x3F800000u;
    float m;
    __builtin_memcpy(&m, &mb, 4);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float s2 = s * s;
    float p = fmaf(s2, 0.142857149f, 0.2f);
    p = fmaf(s2, p, 0.333333343f);
    p = fmaf(s2, p, 1.0f);
    return fmaf((float)e, VR_LN2, (2.0f * s) * p);
}
// Per-splat constants of the tile test.  mode: 0 = reaches no tile at all (opacity below 1/255), 1 = every tile counts (not
// a proper ellipse in fp32: the per-pixel rule decides), 2 = test with lim / inv_A / inv_C.
struct TileTest { int mode; float lim, inv_A, inv_C; };
__host__ __device__ inline TileTest tile_test_setup(float A, float B, float C, float opacity)
{
    TileTest t;
    const float k = 2.0f * (vr_ln_repro(255.0f * opacity) + 0.01f);     // = -2 (ln(1 / (255 opacity)) - 0.01)
    t.lim = k * 1.001f + 0.001f;
    t.inv_A = 1.0f / A;
    t.inv_C = 1.0f / C;
    const float det = A * C - B * B;
    t.mode = !(k > 0.0f) ? 0 : ((!(A > 0.0f) || !(C > 0.0f) || !(det > 0.0f)) ? 1 : 2);
    return t;
}
__host__ __device__ inline float tile_edge_min(float a, float inv_a, float b, float c, float fixed, float lo, float hi)
{
    const float t = fminf(hi, fmaxf(lo, -b * fixed * inv_a));
    return fmaf(fmaf(a, t, 2.0f * b * fixed), t, c * fixed * fixed);
}
// can the splat reach alpha >= 1/255 at a pixel centre of the block of ntx x nty tiles whose first tile is (tx, ty)?
// (conservative: true when in doubt).  Minimum of the convex quadratic over the block's rectangle of pixel centres: 0 when
// the centre lies inside; otherwise on the boundary FACING the centre -- the horizontal edge on the centre's side when it
// lies above / below, the vertical one when it lies left / right, the smaller of the two when both (two edge minimisations,
// not four).
__host__ __device__ inline bool tile_reachable(const TileTest& t, float sx, float sy, float A, float B, float C, int tx, int ty,
                                               int ntx = 1, int nty = 1)
{
    if (t.mode != 2) return t.mode == 1;
    const float xl = (float)(tx * TILE) - sx, xh = xl + (float)(ntx * TILE - 1);
    const float yl = (float)(ty * TILE) - sy, yh = yl + (float)(nty * TILE - 1);
    const bool in_x = xl <= 0.0f && xh >= 0.0f, in_y = yl <= 0.0f && yh >= 0.0f;
    const float fy = yl > 0.0f ? yl : yh, fx = xl > 0.0f ? xl : xh;
    const float qy = tile_edge_min(A, t.inv_A, B, C, fy, xl, xh);       // along the facing horizontal edge
    const float qx = tile_edge_min(C, t.inv_C, B, A, fx, yl, yh);       // along the facing vertical edge
    const float q = in_y ? qx : (in_x ? qy : fminf(qx, qy));
    return (in_x && in_y) || q <= t.lim;
}
// Rectangles of more than 64 tiles are tested in CELLS of k x k tiles, k the smallest size for which the rectangle has at
// most 32 cells (mask bit j = cell j, row-major over cw x ch = ceil(w / k) x ceil(h / k) cells; the last column / row of
// cells may be narrower): a cell none of whose pixel centres can be reached drops all its tiles.  Their mask needs one
// word; the other one holds the number of kept tiles (up to 64 tiles it is the mask's population count), so that nobody has
// to recount.  Without a division: cw = ceil(w / k) is the c with (c - 1) k < w <= c k, and it only shrinks as k grows.
constexpr int TIGHT_BIG_CELLS = 32;
__host__ __device__ inline void tile_cells(int w, int h, int& k, int& cw, int& ch)
{
    k = 1; cw = w; ch = h;
    if (w * h <= TIGHT_MAX_TILES) return;
    while (cw * ch > TIGHT_BIG_CELLS) {
        ++k;
        while ((cw - 1) * k >= w) --cw;
        while ((ch - 1) * k >= h) --ch;
    }
}

// The Gaussian exponent at a pixel in units of log2 e:  power2 = log2(e) (-1/2 (A dx^2 + C dy^2) - B dx dy)
//   = ((kA dx) dx + (kC dy) dy) + (kB dx) dy,   kA = (-1/2 log2 e) A, kB = -(log2 e) B, kC = (-1/2 log2 e) C
// (splat_k2: rounded once per splat, where the record is staged; thr2 = thr log2 e is the pre-filter threshold in the
// same units).  alpha = min(0.99, opacity * 2^power2).  Same operation order in the CPU checker.
constexpr float K_HALF = -0.5f * LOG2E;
__device__ __forceinline__ void splat_k2(float A, float B, float C, float thr, float& kA, float& kB, float& kC, float& thr2)
{
    kA = K_HALF * A;
    kB = -LOG2E * B;
    kC = K_HALF * C;
    thr2 = thr * LOG2E;
}
__device__ __forceinline__ float splat_power2(float sx, float sy, float kA, float kB, float kC, float pxf, float pyf,
                                              float& dx, float& dy)
{
    dx = sx - pxf;
    dy = sy - pyf;
    const float q = fmaf(kC * dy, dy, (kA * dx) * dx);
    return fmaf(kB * dx, dy, q);
}

// the same for one splat against two pixels
__device__ __forceinline__ f2 splat_power2_x2(float sx, float sy, float kA, float kB, float kC, f2 pxf, f2 pyf, f2& dx, f2& dy)
{
    dx = f2_splat(sx) - pxf;
    dy = f2_splat(sy) - pyf;
    const f2 q = f2_fma(f2_splat(kC) * dy, dy, (f2_splat(kA) * dx) * dx);
    return f2_fma(f2_splat(kB) * dx, dy, q);
}

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

// Real SH basis up to degree 3 at unit direction (x,y,z) (polynomials: reference utils/sh_utils.py:74-100).
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* b)
{
    b[0] = SH_C0;
    if (deg < 1) return;
    b[1] = -SH_C1 * y;
    b[2] = SH_C1 * z;
    b[3] = -SH_C1 * x;
    if (deg < 2) return;
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = SH_C2[0] * xy;
    b[5] = SH_C2[1] * yz;
    b[6] = SH_C2[2] * (2.0f * zz - xx - yy);
    b[7] = SH_C2[3] * xz;
    b[8] = SH_C2[4] * (xx - yy);
    if (deg < 3) return;
    b[9] = SH_C3[0] * y * (3.0f * xx - yy);
    b[10] = SH_C3[1] * xy * z;
    b[11] = SH_C3[2] * y * (4.0f * zz - xx - yy);
    b[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    b[13] = SH_C3[4] * x * (4.0f * zz - xx - yy);
    b[14] = SH_C3[5] * z * (xx - yy);
    b[15] = SH_C3[6] * x * (xx - 3.0f * yy);
}

__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, float* bx, float* by, float* bz)
{
#pragma unroll
    for (int k = 0; k < 16; ++k) { bx[k] = 0.f; by[k] = 0.f; bz[k] = 0.f; }
    if (deg < 1) return;
    by[1] = -SH_C1; bz[2] = SH_C1; bx[3] = -SH_C1;
    if (deg < 2) return;
    float xx = x * x, yy = y * y, zz = z * z;
    bx[4] = SH_C2[0] * y; by[4] = SH_C2[0] * x;
    by[5] = SH_C2[1] * z; bz[5] = SH_C2[1] * y;
    bx[6] = SH_C2[2] * -2.0f * x; by[6] = SH_C2[2] * -2.0f * y; bz[6] = SH_C2[2] * 4.0f * z;
    bx[7] = SH_C2[3] * z; bz[7] = SH_C2[3] * x;
    bx[8] = SH_C2[4] * 2.0f * x; by[8] = SH_C2[4] * -2.0f * y;
    if (deg < 3) return;
    bx[9] = SH_C3[0] * 6.0f * x * y; by[9] = SH_C3[0] * (3.0f * xx - 3.0f * yy);
    bx[10] = SH_C3[1] * y * z; by[10] = SH_C3[1] * x * z; bz[10] = SH_C3[1] * x * y;
    bx[11] = SH_C3[2] * -2.0f * x * y; by[11] = SH_C3[2] * (4.0f * zz - xx - 3.0f * yy); bz[11] = SH_C3[2] * 8.0f * y * z;
    bx[12] = SH_C3[3] * -6.0f * x * z; by[12] = SH_C3[3] * -6.0f * y * z; bz[12] = SH_C3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
    bx[13] = SH_C3[4] * (4.0f * zz - 3.0f * xx - yy); by[13] = SH_C3[4] * -2.0f * x * y; bz[13] = SH_C3[4] * 8.0f * x * z;
    bx[14] = SH_C3[5] * 2.0f * x * z; by[14] = SH_C3[5] * -2.0f * y * z; bz[14] = SH_C3[5] * (xx - yy);
    bx[15] = SH_C3[6] * (3.0f * xx - 3.0f * yy); by[15] = SH_C3[6] * -6.0f * x * y;
}

// D[3*c + a] = d colour_c / d direction_a = sum_k grad_a basis_k(dir) * sh[k][c]  (before normalisation of the
// direction and before the clamp).  Computed in the FORWARD, next to the colour itself, and kept (9 floats per
// visible Gaussian): the backward then needs neither the 192-byte SH row nor the basis gradients again.
// One basis function and its gradient at a time (index K known at compile time): b = basis_K(x, y, z) and (bx, by, bz) its
// partial derivatives, with the SAME expressions as sh_basis / sh_basis_grad above (bit-identical values) -- for callers
// that consume the coefficients in order and cannot afford the 64 registers the whole tables take (k_preprocess, HALF path).
// xx, yy, zz, xy, yz, xz: the products sh_basis forms (x * x ...).
template <int K>
__device__ __forceinline__ void sh_term(float x, float y, float z, float xx, float yy, float zz, float xy, float yz, float xz,
                                        float& b, float& bx, float& by, float& bz)
{
    bx = 0.f; by = 0.f; bz = 0.f;
    if (K == 0) { b = SH_C0; }
    else if (K == 1) { b = -SH_C1 * y; by = -SH_C1; }
    else if (K == 2) { b = SH_C1 * z; bz = SH_C1; }
    else if (K == 3) { b = -SH_C1 * x; bx = -SH_C1; }
    else if (K == 4) { b = SH_C2[0] * xy; bx = SH_C2[0] * y; by = SH_C2[0] * x; }
    else if (K == 5) { b = SH_C2[1] * yz; by = SH_C2[1] * z; bz = SH_C2[1] * y; }
    else if (K == 6) { b = SH_C2[2] * (2.0f * zz - xx - yy); bx = SH_C2[2] * -2.0f * x; by = SH_C2[2] * -2.0f * y; bz = SH_C2[2] * 4.0f * z; }
    else if (K == 7) { b = SH_C2[3] * xz; bx = SH_C2[3] * z; bz = SH_C2[3] * x; }
    else if (K == 8) { b = SH_C2[4] * (xx - yy); bx = SH_C2[4] * 2.0f * x; by = SH_C2[4] * -2.0f * y; }
    else if (K == 9) { b = SH_C3[0] * y * (3.0f * xx - yy); bx = SH_C3[0] * 6.0f * x * y; by = SH_C3[0] * (3.0f * xx - 3.0f * yy); }
    else if (K == 10) { b = SH_C3[1] * xy * z; bx = SH_C3[1] * y * z; by = SH_C3[1] * x * z; bz = SH_C3[1] * x * y; }
    else if (K == 11) { b = SH_C3[2] * y * (4.0f * zz - xx - yy); bx = SH_C3[2] * -2.0f * x * y; by = SH_C3[2] * (4.0f * zz - xx - 3.0f * yy); bz = SH_C3[2] * 8.0f * y * z; }
    else if (K == 12) { b = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy); bx = SH_C3[3] * -6.0f * x * z; by = SH_C3[3] * -6.0f * y * z; bz = SH_C3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy); }
    else if (K == 13) { b = SH_C3[4] * x * (4.0f * zz - xx - yy); bx = SH_C3[4] * (4.0f * zz - 3.0f * xx - yy); by = SH_C3[4] * -2.0f * x * y; bz = SH_C3[4] * 8.0f * x * z; }
    else if (K == 14) { b = SH_C3[5] * z * (xx - yy); bx = SH_C3[5] * 2.0f * x * z; by = SH_C3[5] * -2.0f * y * z; bz = SH_C3[5] * (xx - yy); }
    else { b = SH_C3[6] * x * (xx - 3.0f * yy); bx = SH_C3[6] * (3.0f * xx - 3.0f * yy); by = SH_C3[6] * -6.0f * x * y; }
}
// acc / D of sh_dot / sh_ddir9 advanced by coefficient K (sv = the coefficient's three channels): the same operations in
// the same order as those two loops, including the fmas whose basis gradient is an exact zero
template <int K>
__device__ __forceinline__ void sh_accumulate(float x, float y, float z, float xx, float yy, float zz, float xy, float yz,
                                              float xz, const float* sv, float* acc, float* D)
{
    float b, bx, by, bz;
    sh_term<K>(x, y, z, xx, yy, zz, xy, yz, xz, b, bx, by, bz);
    if (K == 0) { acc[0] = b * sv[0]; acc[1] = b * sv[1]; acc[2] = b * sv[2]; }
    else { acc[0] = fmaf(b, sv[0], acc[0]); acc[1] = fmaf(b, sv[1], acc[1]); acc[2] = fmaf(b, sv[2], acc[2]); }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        D[3 * c + 0] = fmaf(bx, sv[c], D[3 * c + 0]);
        D[3 * c + 1] = fmaf(by, sv[c], D[3 * c + 1]);
        D[3 * c + 2] = fmaf(bz, sv[c], D[3 * c + 2]);
    }
}
__device__ __forceinline__ void sh_ddir9(const float* bx, const float* by, const float* bz, int K, const float* sh, float* D)
{
#pragma unroll
    for (int q = 0; q < 9; ++q) D[q] = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k < K) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float sv = sh[3 * k + c];
                D[3 * c + 0] = fmaf(bx[k], sv, D[3 * c + 0]);
                D[3 * c + 1] = fmaf(by[k], sv, D[3 * c + 1]);
                D[3 * c + 2] = fmaf(bz[k], sv, D[3 * c + 2]);
            }
        }
    }
}


// Rotation matrix of a (not re-normalised) quaternion (w,x,y,z), row-major.
__device__ __forceinline__ void quat_to_R(float r, float x, float y, float z, float* R)
{
    R[0] = 1.0f - 2.0f * (y * y + z * z); R[1] = 2.0f * (x * y - r * z); R[2] = 2.0f * (x * z + r * y);
    R[3] = 2.0f * (x * y + r * z); R[4] = 1.0f - 2.0f * (x * x + z * z); R[5] = 2.0f * (y * z - r * x);
    R[6] = 2.0f * (x * z - r * y); R[7] = 2.0f * (y * z + r * x); R[8] = 1.0f - 2.0f * (x * x + y * y);
}

// Sigma = R diag(mod*s)^2 R^T as 6 upper-triangular floats (xx xy xz yy yz zz).
__device__ __forceinline__ void cov3d_from_scale_rot(const float* s, float mod, const float* q, float* c6)
{
    float R[9], L[9];
    quat_to_R(q[0], q[1], q[2], q[3], R);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) L[3 * i + k] = R[3 * i + k] * (mod * s[k]);
    c6[0] = fmaf(L[2], L[2], fmaf(L[1], L[1], L[0] * L[0]));
    c6[1] = fmaf(L[2], L[5], fmaf(L[1], L[4], L[0] * L[3]));
    c6[2] = fmaf(L[2], L[8], fmaf(L[1], L[7], L[0] * L[6]));
    c6[3] = fmaf(L[5], L[5], fmaf(L[4], L[4], L[3] * L[3]));
    c6[4] = fmaf(L[5], L[8], fmaf(L[4], L[7], L[3] * L[6]));
    c6[5] = fmaf(L[8], L[8], fmaf(L[7], L[7], L[6] * L[6]));
}

// EWA splat projection: rows m0,m1 of J*W (2x3), clamped view-space point and the 2D covariance.
struct Cov2D {
    float m0[3], m1[3];
    float tx, ty, tz;
    bool clampx, clampy;
    float a, b, c;
};

__device__ __forceinline__ void cov2d(const Camera& cam, const float* __restrict__ v, float t0, float t1, float t2,
                                      const float* c6, Cov2D& o)
{
    float limx = 1.3f * cam.tanfovx, limy = 1.3f * cam.tanfovy;
    float tz = t2;
    float txtz = t0 / tz, tytz = t1 / tz;
    o.clampx = (txtz < -limx) || (txtz > limx);
    o.clampy = (tytz < -limy) || (tytz > limy);
    float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    o.tx = tx; o.ty = ty; o.tz = tz;
    float j00 = cam.fx / tz, j02 = -(cam.fx * tx) / (tz * tz);
    float j11 = cam.fy / tz, j12 = -(cam.fy * ty) / (tz * tz);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o.m0[i] = fmaf(j02, v[4 * i + 2], j00 * v[4 * i + 0]);
        o.m1[i] = fmaf(j12, v[4 * i + 2], j11 * v[4 * i + 1]);
    }
    const float* m0 = o.m0; const float* m1 = o.m1;
    float u0 = fmaf(c6[2], m0[2], fmaf(c6[1], m0[1], c6[0] * m0[0]));
    float u1 = fmaf(c6[4], m0[2], fmaf(c6[3], m0[1], c6[1] * m0[0]));
    float u2 = fmaf(c6[5], m0[2], fmaf(c6[4], m0[1], c6[2] * m0[0]));
    float w0 = fmaf(c6[2], m1[2], fmaf(c6[1], m1[1], c6[0] * m1[0]));
    float w1 = fmaf(c6[4], m1[2], fmaf(c6[3], m1[1], c6[1] * m1[0]));
    float w2 = fmaf(c6[5], m1[2], fmaf(c6[4], m1[1], c6[2] * m1[0]));
    o.a = fmaf(m0[2], u2, fmaf(m0[1], u1, m0[0] * u0)) + 0.3f;
    o.b = fmaf(m1[2], u2, fmaf(m1[1], u1, m1[0] * u0));
    o.c = fmaf(m1[2], w2, fmaf(m1[1], w1, m1[0] * w0)) + 0.3f;
}

// XCD-aware block -> tile map: the dispatcher places block b on XCD b % 8 (speed only, never
// correctness); give each XCD a contiguous run of tiles so neighbouring tiles, which share
// splats, hit the same L2.  Bijective for any tile count.
__device__ __forceinline__ int xcd_tile(int b, int ntiles)
{
    constexpr int NX = 8;
    int q = ntiles / NX, r = ntiles % NX;
    int xcd = b % NX, k = b / NX;
    int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

}  // namespace vr
