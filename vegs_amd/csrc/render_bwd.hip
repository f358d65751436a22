// render_bwd.hip -- backward of the tile compositor.
//
// Replaces the native render-backward stage that loss.backward() reaches through the op's
// autograd.Function (reference train.py:196); semantics SURVEY.md A.5 + fork channels: for every
// fragment (pixel p, splat s) with weight w = alpha*T,
//   dL/dattr_s[k] += w * g_p[k]                                    (11 blended channels)
//   dL/dalpha     = T*u - (behind + T_final*bgterm) / (1 - alpha),  u = <attr_s, g_p>,
//                   behind = sum_{j behind s} w_j u_j
// and dL/dalpha fans out to opacity, conic and the 2D mean.
//
// MI355X design.  (1) SEGMENTED like the forward: the unit of work is (tile, 256-entry segment, 8x8 pixel region),
// so the 10^5-entry vanishing-point tiles spread over the whole chip.  The two compositing carries are made
// available per segment: the transmittance at the segment end comes from the forward's boundary buffer (Tbuf,
// kept in the binning buffer), and the "behind" sum comes from a cheap pixel-parallel pass (k_seg_u: U = sum of
// w*u per segment, straight from the forward's segment-local channel sums) followed by a per-tile suffix sum
// (k_seg_suffix).
// (2) SPLAT-parallel inside a segment with wave64 DPP scans (not the pixel-parallel + per-fragment-atomic scheme
// of CUDA rasterizers): the reduction target of the backward pass is the splat, so the splat owns the lane.  A
// wave (= one 64-thread workgroup) holds 64 entries relevant to its region in registers (record + 17 gradient
// accumulators) and walks the region's pixels two at a time; the pixel quantities are wave-uniform (LDS
// broadcasts).  The two recurrences become wave scans in DPP (row_shr / row_bcast):
//   T in front of splat s = T_after_chunk / prod_{j>=s}(1-alpha_j)   (inclusive scan-product)
//   behind(s)             = carry + sum_{j>s} w_j u_j                (inclusive scan-sum)
// Lanes hold a chunk back-to-front (lane l <-> entry 63-l) so both are PREFIX scans over lanes.
// Per-splat gradients stay in registers until the chunk is finished, are transposed through LDS and leave as one
// coalesced set of global atomics per (region, splat) -- ~250x fewer atomics than one per fragment.
#include "../../include/vegs_rast_debug.h"
#include "vr_host.h"
#include "vr_segment.h"

namespace vr {

constexpr int NACC = 17;  // conic(3) opacity(1) attr(11) mean2D(2)
#ifndef VR_BWD_PREFETCH
#define VR_BWD_PREFETCH 1        // (0: A/B build that gathers a chunk's records when it starts, --variant nopref)
#endif
#ifndef VR_BWD_PACK_TAILS
#define VR_BWD_PACK_TAILS 1      // (0: the A/B build without row-packed tail chunks, python -m vegs_amd.build --variant nopack)
#endif

// wave64 inclusive prefix scans in DPP: row_shr 1,2,4,8 inside each 16-lane row, then row_bcast 15 / 31.
// VOP2-DPP semantics do the masking for free: a lane whose DPP source is out of range (or whose row is
// masked off) is simply not written, i.e. keeps its value -- one instruction per scan step.  A DPP read needs
// two wait states after the VALU write of the same VGPR; two independent scans are interleaved, so the other
// chain's instruction fills one of them.
#define VR_DPP2(op, ctl)                                                        \
    op " %0, %0, %0 " ctl "\n\t" op " %1, %1, %1 " ctl "\n\ts_nop 0\n\t"
__device__ __forceinline__ void wave_prefix_mul_x2(float& a, float& b)
{
    asm volatile("s_nop 1\n\t"
                 VR_DPP2("v_mul_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")
                 VR_DPP2("v_mul_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
                 VR_DPP2("v_mul_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf")
                 VR_DPP2("v_mul_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
                 VR_DPP2("v_mul_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
                 VR_DPP2("v_mul_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 0"
                 : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void wave_prefix_add_x2(float& a, float& b)
{
    asm volatile("s_nop 1\n\t"
                 VR_DPP2("v_add_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")
                 VR_DPP2("v_add_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
                 VR_DPP2("v_add_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf")
                 VR_DPP2("v_add_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
                 VR_DPP2("v_add_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
                 VR_DPP2("v_add_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 0"
                 : "+v"(a), "+v"(b));
}

// the scan result of the lane BELOW (lane 0: zero) -- an inclusive prefix turned into the exclusive one without a subtraction
__device__ __forceinline__ void wave_shift_up_x2(float a, float b, float& xa, float& xb)
{
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_mov_b32_dpp %1, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 0"
                 : "=&v"(xa), "=&v"(xb)
                 : "v"(a), "v"(b));
}

// ---- the same scans restricted to GROUPS of 64 / ROWS lanes (ROWS = 2: the wave's halves, ROWS = 4: its 16-lane DPP rows), for
// the row-packed tail chunks of k_seg_bwd (below): the full scan minus its last one / two steps
template <int ROWS>
__device__ __forceinline__ void group_prefix_mul_x2(float& a, float& b)
{
    if (ROWS == 1) { wave_prefix_mul_x2(a, b); return; }
    if (ROWS == 2)
        asm volatile("s_nop 1\n\t"
                     VR_DPP2("v_mul_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")
                     VR_DPP2("v_mul_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
                     VR_DPP2("v_mul_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf")
                     VR_DPP2("v_mul_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
                     VR_DPP2("v_mul_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
                     "s_nop 0"
                     : "+v"(a), "+v"(b));
    else
        asm volatile("s_nop 1\n\t"
                     VR_DPP2("v_mul_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")
                     VR_DPP2("v_mul_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
                     VR_DPP2("v_mul_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf")
                     VR_DPP2("v_mul_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
                     "s_nop 0"
                     : "+v"(a), "+v"(b));
}
template <int ROWS>
__device__ __forceinline__ void group_prefix_add_x2(float& a, float& b)
{
    if (ROWS == 1) { wave_prefix_add_x2(a, b); return; }
    if (ROWS == 2)
        asm volatile("s_nop 1\n\t"
                     VR_DPP2("v_add_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")
                     VR_DPP2("v_add_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
                     VR_DPP2("v_add_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf")
                     VR_DPP2("v_add_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
                     VR_DPP2("v_add_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
                     "s_nop 0"
                     : "+v"(a), "+v"(b));
    else
        asm volatile("s_nop 1\n\t"
                     VR_DPP2("v_add_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")
                     VR_DPP2("v_add_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
                     VR_DPP2("v_add_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf")
                     VR_DPP2("v_add_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
                     "s_nop 0"
                     : "+v"(a), "+v"(b));
}
// the scan result of the lane below INSIDE the group (the group's first lane: zero)
template <int ROWS>
__device__ __forceinline__ void group_shift_up_x2(float a, float b, float& xa, float& xb, int lane)
{
    if (ROWS == 4) {
        asm volatile("s_nop 1\n\t"
                     "v_mov_b32_dpp %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                     "v_mov_b32_dpp %1, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                     "s_nop 0"
                     : "=&v"(xa), "=&v"(xb)
                     : "v"(a), "v"(b));
        return;
    }
    wave_shift_up_x2(a, b, xa, xb);
    if (ROWS == 2 && lane == 32) { xa = 0.0f; xb = 0.0f; }
}

// per-pixel upstream gradients of the 11 blended channels
struct PixGrad {
    float g[NCH];
    float galpha;
};
// With VR_FLAG_DEPTH_NORMALIZED the depth image is D / A (D = sum w z, A = 1 - T_final): the kernels keep
// working on the un-normalised sums, so the incoming gradient is carried back through the quotient here:
// dL/dD = g / A and (unless the extra channels are cut off from alpha) dL/dA -= g D / A^2.
__device__ __forceinline__ void load_pixgrad(bool inside, size_t pix, size_t N, const float* dL_dcolor,
                                             const float* dL_ddepth, const float* dL_dquat, const float* dL_dscale,
                                             const float* dL_dalpha, uint32_t flags, const float* final_T,
                                             const float* dsum, PixGrad& o)
{
#pragma unroll
    for (int k = 0; k < NCH; ++k) o.g[k] = 0.0f;
    o.galpha = 0.0f;
    if (!inside) return;
    if (dL_dcolor) { o.g[0] = dL_dcolor[pix]; o.g[1] = dL_dcolor[N + pix]; o.g[2] = dL_dcolor[2 * N + pix]; }
    if (dL_ddepth) o.g[3] = dL_ddepth[pix];
    if (dL_dquat) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o.g[4 + k] = dL_dquat[k * N + pix];
    }
    if (dL_dscale) {
#pragma unroll
        for (int k = 0; k < 3; ++k) o.g[8 + k] = dL_dscale[k * N + pix];
    }
    if (dL_dalpha) o.galpha = dL_dalpha[pix];
    if ((flags & FLAG_DEPTH_NORMALIZED) && dL_ddepth) {
        const float A = 1.0f - final_T[pix];
        const float inv = A > 0.0f ? 1.0f / A : 0.0f;
        const float g3 = o.g[3];
        o.g[3] = g3 * inv;
        if (!(flags & FLAG_EXTRA_NO_ALPHA_GRAD)) o.galpha -= (g3 * dsum[pix]) * (inv * inv);
    }
}
// channels whose upstream gradient also flows through alpha (u = <attr, g> over these): all 11, or colour only
__device__ __forceinline__ int alpha_grad_channels(uint32_t flags) { return (flags & FLAG_EXTRA_NO_ALPHA_GRAD) ? 3 : NCH; }

// ---- A': U[seg][pix] = sum over the segment's applied entries of w*u = <dL/dout(pix), segment-local
// channel sums> -- the sums the forward already produced (`part`), so no second pass over the splats.
// Fully parallel over (segment, pixel); the sequential part (B') then only touches one float per segment.
constexpr int NPART_B = 13;
__global__ void __launch_bounds__(256)
k_seg_u(Camera cam, const int2* __restrict__ ranges, const uint32_t* __restrict__ seg_off, uint32_t cap,
        const uint32_t* __restrict__ seg_needed, const float* __restrict__ part,
        const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor,
        const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dquat, const float* __restrict__ dL_dscale,
        const float* __restrict__ final_T, const float* __restrict__ dsum, float* __restrict__ Ubuf,
        float* __restrict__ zero_a, size_t n_a, float* __restrict__ zero_b, size_t n_b)
{
    // The backward's accumulators (gacc [P][16] and dL/dmeans2D [P][3], contiguous quads) are cleared HERE, by every
    // workgroup of the launch before it looks at its segment: this kernel waits for gathers most of its life, the
    // stores ride along (two fill launches of 25 us per view are gone).  k_seg_bwd, the first to add to them, is the
    // next launch but one.
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        float* const z = which ? zero_b : zero_a;
        const size_t n = which ? n_b : n_a;
        if (!z) continue;
        const size_t head = (reinterpret_cast<size_t>(z) & 15) ? n : 0;         // unaligned array: all of it by dwords
        const size_t nq = (n - head) >> 2;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (size_t)gridDim.x * 256)
            nt_store4(make_float4(0.f, 0.f, 0.f, 0.f), reinterpret_cast<float4*>(z) + i);
        for (size_t i = (nq << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) z[i] = 0.0f;
    }
    SegCtx c;
    const int ntiles = cam.gx * cam.gy;
    if (blockIdx.x >= seg_off[seg_counts_offset(ntiles, cap) + SEG_LIST_NEEDED]) return;    // beyond the needed list (vr_segment.h)
    if (!seg_setup_at(cam, ranges, seg_off, seg_off[seg_list_offset(ntiles, cap, SEG_LIST_NEEDED) + blockIdx.x],
                      threadIdx.x >> 6, c)) return;
    // U of a tile's FIRST segment is never used: it would only enter the "behind" sums of earlier segments, and there
    // are none (k_seg_suffix overwrites the slot with the suffix before it adds the slot's old content to a running sum
    // nobody reads).  2 k of the ~9.5 k needed segments of the headline view.
    if (c.flag == 0u || c.sl == 0) return;
    const size_t N = (size_t)cam.H * cam.W;
    PixGrad pg;
    load_pixgrad(c.inside, c.pix, N, dL_dcolor, dL_ddepth, dL_dquat, dL_dscale, nullptr, cam.flags, final_T, dsum, pg);
    const int nu = alpha_grad_channels(cam.flags);
    const int nc = c.inside ? (int)n_contrib[c.pix] : 0;
    float U = 0.0f;
    if (c.sl * SEG < nc) {   // the pixel applied entries of this segment (otherwise `part` is not defined for it)
        const float* src = part + (size_t)c.seg * (NPART_B * SEG) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (k == 3 && dL_ddepth == nullptr) continue;      // (no gradient on the depth image: fma(x, 0, U) = U, the plane is not read)
            if (k < nu) U = fmaf(src[k * SEG], pg.g[k], U);
        }
    }
    Ubuf[(size_t)c.seg * SEG + threadIdx.x] = U;
}

// ---- B': per tile, in place: Ubuf[seg][pix] <- sum of U over the LATER segments of the tile
__global__ void __launch_bounds__(256)
k_seg_suffix(int ntiles, uint32_t cap, const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ seg_needed,
             float* __restrict__ Ubuf)
{
    // heavy tiles first (vr_segment.h): the first `ntiles` workgroups take the forward's heavy list from its end, the others
    // their own tile unless it is on that list
    int tile;
    if ((int)blockIdx.x < ntiles) {
        const uint32_t hcount = seg_off[seg_counts_offset(ntiles, cap) + SEG_COUNT_HEAVY];
        if (blockIdx.x >= hcount) return;
        tile = (int)seg_off[seg_actoff_offset(ntiles, cap) + (hcount - 1u - blockIdx.x)];
    } else {
        tile = (int)blockIdx.x - ntiles;
        if (seg_needed[tile] >= HEAVY_TILE) return;
    }
    const uint32_t s0 = seg_off[tile];
    const int needed = (int)seg_needed[tile];
    constexpr int CU = 24;
    float run = 0.0f;
    for (int s = needed - 1; s >= 0; s -= CU) {
        float v[CU];
#pragma unroll
        for (int j = 0; j < CU; ++j) v[j] = Ubuf[(size_t)(s0 + max(s - j, 0)) * SEG + threadIdx.x];
#pragma unroll
        for (int j = 0; j < CU; ++j) {
            if (s - j < 0) continue;
            Ubuf[(size_t)(s0 + s - j) * SEG + threadIdx.x] = run;
            run += v[j];
        }
    }
}

// ---- C': gradients of one (tile, segment, 8x8 region = "strip").  ONE WAVE PER WORKGROUP: the four strips of a segment have
// different numbers of relevant entries and live pixels; as four waves of one workgroup the fast ones idled at
// the final barrier while still holding their occupancy slot (SQ_WAIT_ANY was 48 % of the wave cycles).  As
// independent 64-thread workgroups they are scheduled -- and leave -- individually.  An entry is relevant to
// 1.05 strips on average, so combining the strips in LDS before the global flush bought almost nothing.
// NO_EXTRA / DET are compile-time (VR_FLAG_EXTRA_NO_ALPHA_GRAD, VR_FLAG_DETERMINISTIC): the common instantiation carries
// neither their branches nor their registers (round 2 added them as run-time tests: 360 -> 373 us).
// 1 / x: v_rcp_f32 (1 ulp) in the production instantiations, the correctly rounded division in the deterministic one (the
// mode the full-size tests compare with the checker: its reciprocals are then the checker's)
template <bool EXACT>
__device__ __forceinline__ float bwd_rcp(float x) { return EXACT ? 1.0f / x : __builtin_amdgcn_rcpf(x); }

// The pixel loop of one chunk.  ROWS = 1: the wave holds 64 entries and walks the region's 32 pixel pairs one per trip.
// ROWS = 4 / 2 (round 6, the ROW-PACKED TAIL): a region's relevant entries are cut into chunks of 64 from the front, so the last
// chunk holds 1 ... 64 of them -- and a chunk of <= 16 (<= 32) entries used to cost the same 32 trips with three quarters
// (half) of the lanes empty.  Such a chunk is now held once per 16-lane DPP row (once per half), and every row (half) walks
// ANOTHER pixel pair of the trip: 8 (16) trips, the scans stop at the row (half) boundary, the pixel state arrives as one LDS
// broadcast per group instead of per wave, and the groups' partial sums are added at the flush.  On the headline view 29 % of
// the chunks are such tails (profiles/tools/pairstats.py: tail_stats): accepted trips 820 k -> 725 k.
typedef float f4n __attribute__((ext_vector_type(4)));          // (a native vector: loads through an LDS-qualified pointer)
typedef __attribute__((address_space(3))) f4n lds_f4;
template <int ROWS, bool NO_EXTRA, bool DET, bool FAST, bool NO_DEPTH>
__device__ __forceinline__ void bwd_chunk_trips(const int lane, const int e, const int chunk_lo, const int v_nc, const float sx,
                                                const float sy, const float kA, const float kB, const float kC, const float thr2,
                                                const float op, const float cA, const float cB, const float cC,
                                                const float (&at)[NCH], f2 (&acc)[NACC], float4 (*pix)[32], const int2* ncp)
{
    constexpr int SPAN = 64 / ROWS;
    constexpr bool no_extra = NO_EXTRA;
    const int grp = ROWS == 1 ? 0 : lane / SPAN;
    for (int t = 0; t < 32 / ROWS; ++t) {
        int pp, nc0, nc1;
        if (ROWS == 1) {
            pp = t;
            nc0 = __builtin_amdgcn_readlane(v_nc, 2 * t);
            nc1 = __builtin_amdgcn_readlane(v_nc, 2 * t + 1);
            if (max(nc0, nc1) <= chunk_lo) continue;  // neither pixel has a contributor in this chunk (wave-uniform)
        } else {        // every group its own pair: the pixels' last contributors come from LDS, per lane
            pp = t * ROWS + grp;
            const int2 n = ncp[pp];
            nc0 = n.x;
            nc1 = n.y;
            if (__builtin_amdgcn_ballot_w64(max(nc0, nc1) > chunk_lo) == 0ull) continue;
        }
        // the pair's LDS row address in ONE vector register for all nine accesses of the trip (the compiler re-materialised it
        // from a scalar three times per trip)
        uint32_t paddr = (uint32_t)(uintptr_t)(lds_f4*)&pix[0][pp];
        asm volatile("" : "+v"(paddr));
        lds_f4* const prow = (lds_f4*)(uintptr_t)paddr;        // prow[32 * slot] = pix[slot][pp]
        const f4n r0 = prow[0];
        const f2 pxf = {r0.x, r0.y}, pyf = {r0.z, r0.w};
        f2 dx, dy;
        const f2 power = splat_power2_x2(sx, sy, kA, kB, kC, pxf, pyf, dx, dy);    // in units of log2 e, as the forward
        // thr2 <= power <= 0 as ONE comparison: the median of (power, thr2, 0) is power itself.  (A splat too faint
        // to reach 1/255 anywhere has thr2 > 0 and passes this at power == 0 exactly; the alpha test below drops it.)
        const bool pre0 = (e < nc0) & (__builtin_amdgcn_fmed3f(power.x, thr2, 0.0f) == power.x);
        const bool pre1 = (e < nc1) & (__builtin_amdgcn_fmed3f(power.y, thr2, 0.0f) == power.y);
        if (__builtin_amdgcn_ballot_w64(pre0 | pre1) == 0ull) continue;  // no splat of the chunk reaches the pair: carries unchanged
        // The forward's own 2^x (bit for bit: same operations), not the hardware's: WHICH fragments contributed is
        // the forward's decision (alpha >= 1/255), and v_exp_f32 agrees with it only to ~2 ulp -- rare to matter,
        // but a faint splat has ALL its fragments at the threshold, and one fragment of twenty classified the other
        // way moved its gradient by 4.5 % (the C-harness test caught it).  Evaluated for both pixels in packed
        // arithmetic; +14 us per view over two v_exp_f32 -- and so did the two cheaper-looking alternatives (a band
        // test around the threshold with the exact function inline or out of line in the rare branch).
        f2 G;
        if (FAST) {      // VR_FLAG_FAST_EXP: the instruction the forward of this view used
            G.x = __builtin_amdgcn_exp2f(power.x);
            G.y = __builtin_amdgcn_exp2f(power.y);
        } else {
            const f2 n = {rintf(power.x), rintf(power.y)};
            const f2 f = power - n;
            f2 p = f2_splat(EXP2_C5);
            p = f2_fma(p, f, f2_splat(EXP2_C4));
            p = f2_fma(p, f, f2_splat(EXP2_C3));
            p = f2_fma(p, f, f2_splat(EXP2_C2));
            p = f2_fma(p, f, f2_splat(EXP2_C1));
            p = f2_fma(p, f, f2_splat(1.0f));
            G.x = ldexpf(p.x, (int)n.x);
            G.y = ldexpf(p.y, (int)n.y);
        }
        const f2 alpha = {fminf(ALPHA_MAX, op * G.x), fminf(ALPHA_MAX, op * G.y)};
        const bool contrib0 = pre0 && !(alpha.x < ALPHA_MIN), contrib1 = pre1 && !(alpha.y < ALPHA_MIN);
        const f2 a_eff = {contrib0 ? alpha.x : 0.0f, contrib1 ? alpha.y : 0.0f};
        G.x = contrib0 ? G.x : 0.0f;              // (exp of a positive exponent may be inf: keep it out of 0*inf)
        G.y = contrib1 ? G.y : 0.0f;
        const f4n r1 = prow[32], r2 = prow[64], r3 = prow[96], r4 = prow[128],
                  r5 = prow[160], r6 = prow[192], r7 = prow[224];
        const f2 bgterm = {r1.x, r1.y}, Tc = {r7.x, r7.y}, Sc = {r7.z, r7.w};
        const f2 g[NCH] = {{r1.z, r1.w}, {r2.x, r2.y}, {r2.z, r2.w}, {r3.x, r3.y}, {r3.z, r3.w}, {r4.x, r4.y},
                           {r4.z, r4.w}, {r5.x, r5.y}, {r5.z, r5.w}, {r6.x, r6.y}, {r6.z, r6.w}};
        const f2 om = f2_splat(1.0f) - a_eff;
        // 1 / (1 - alpha) for dL/dalpha below, taken HERE: the scan then runs in place on `om`'s registers (two moves per trip
        // fewer; a trip that reaches this point almost always has a contributing lane)
        // (finite also for a lane that does not contribute: its alpha is 0, so 1 - alpha = 1, and the prefix
        // product of at most 64 factors >= 0.01 it divides by cannot reach zero before Tc itself has)
        f2 inv_om = {bwd_rcp<DET>(om.x), bwd_rcp<DET>(om.y)};
        {   // (pinned in front of the scan: sunk into the block that uses it, the reciprocals keep `om` alive across the scan)
            float ia = inv_om.x, ib = inv_om.y;
            asm volatile("" : "+v"(ia), "+v"(ib));
            inv_om.x = ia; inv_om.y = ib;
        }
        float pa = om.x, pb = om.y;
        group_prefix_mul_x2<ROWS>(pa, pb);                               // prod over entries >= mine
        const f2 Tl = Tc * (f2){bwd_rcp<DET>(pa), bwd_rcp<DET>(pb)};   // T in front of my splat
        const f2 wgt = a_eff * Tl;
        // <attr, g> in two independent chains (a dependent v_pk_fma_f32 costs an extra wait state)
        f2 u0 = f2_splat(at[0]) * g[0], u1 = f2_splat(at[1]) * g[1];
        f2 u;
        if (!no_extra) {
#pragma unroll
            for (int k = 2; k + 1 < NCH; k += 2) {
                u0 = f2_fma(f2_splat(at[k]), g[k], u0);
                if (!(NO_DEPTH && k + 1 == 3)) u1 = f2_fma(f2_splat(at[k + 1]), g[k + 1], u1);
            }
            u = f2_fma(f2_splat(at[NCH - 1]), g[NCH - 1], u0) + u1;
        } else {
            u = f2_fma(f2_splat(at[2]), g[2], u0) + u1;     // colour channels only (wave-uniform branch)
        }
        const f2 wu = wgt * u;
        float sa = wu.x, sb = wu.y;
        group_prefix_add_x2<ROWS>(sa, sb);                               // sum over entries >= mine
        const f2 psum = {sa, sb};
        // "behind" = the carry + the EXCLUSIVE prefix, taken from the lane below (round 6).  It used to be psum - wu: for a
        // near-opaque splat in front of faint ones (opacity 1: alpha clamped at 0.99) that subtraction leaves the small
        // sum behind it with the absolute rounding error of the large one, and dL/dalpha then divides by 1 - alpha = 0.01
        // -- fuzz seed 11136: an opacity gradient off by 4.7 % where the checker's sequential sum and the float64
        // restatement agree to 1e-4.  Same two issue slots as the subtraction.
        float ea, eb;
        group_shift_up_x2<ROWS>(sa, sb, ea, eb, lane);
        const f2 behind = Sc + (f2){ea, eb};
        // carries for the next (nearer) chunk: values at the chunk's first entry = the group's last lane
        if ((lane & (SPAN - 1)) == SPAN - 1) {
            const f2 Sn = Sc + psum;
            prow[224] = (f4n){Tl.x, Tl.y, Sn.x, Sn.y};
        }
        if (contrib0 || contrib1) {
            // Per-fragment work kept to what depends on the pixel.  With a = dL/dG * G (zero for a lane that does
            // not contribute: G and alpha are masked above, everything below is finite):
            //   conic:   dA += -1/2 a dx^2   dB += -a dx dy   dC += -1/2 a dy^2
            //   mean2D:  dx += -a (A dx + B dy)   dy += -a (C dy + B dx)
            // the factors -1/2 and -1 belong to the SPLAT, i.e. to the lane: sum(a dx dx), sum(a dx dy),
            // sum(a dy dy), sum(a (A dx + B dy)), sum(a (C dy + B dx)) are accumulated here (12 packed operations
            // instead of 20) and the signs applied once per entry at the flush.  (The conic cannot be pulled out of
            // the mean2D sums as well: A sum(a dx) + B sum(a dy) cancels AFTER the sums were rounded, and for edge-on
            // discs that lost two digits -- the C-harness test caught rows off by 2 %.)
            const f2 dLda = f2_fma(Tl, u, -(behind + bgterm) * inv_om);
            const f2 a = (f2_splat(op) * dLda) * G;
            const f2 adx = a * dx, ady = a * dy;
            acc[0] = f2_fma(adx, dx, acc[0]);
            acc[1] = f2_fma(adx, dy, acc[1]);
            acc[2] = f2_fma(ady, dy, acc[2]);
            acc[3] = f2_fma(G, dLda, acc[3]);
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (!(NO_DEPTH && k == 3)) acc[4 + k] = f2_fma(wgt, g[k], acc[4 + k]);
            acc[15] = f2_fma(a, f2_fma(f2_splat(cA), dx, f2_splat(cB) * dy), acc[15]);
            acc[16] = f2_fma(a, f2_fma(f2_splat(cC), dy, f2_splat(cB) * dx), acc[16]);
        }
    }
}

// NO_DEPTH (round 6): no upstream gradient on the depth image (dL_ddepth == NULL: VEGS' losses reach colour, cov_quat and
// cov_scale only, train.py:152-168) -- its channel's terms, fma(depth_s, 0, u) in the <attr, g> chain and fma(w, 0, acc) in the
// accumulators, are the identity for every finite depth and are left out: bit-identical sums, two packed instructions fewer
// per trip.
template <bool NO_EXTRA, bool DET, bool FAST, bool NO_DEPTH = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_seg_bwd(Camera cam, const int2* __restrict__ ranges, const uint32_t* __restrict__ seg_off, uint32_t cap,
          const uint32_t* __restrict__ seg_needed, const uint32_t* __restrict__ point_list,
          const Splat* __restrict__ rec, const float* __restrict__ Tbuf, const float* __restrict__ Ubuf,
          const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
          const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dquat,
          const float* __restrict__ dL_dscale, const float* __restrict__ dL_dalpha, float* __restrict__ gacc,
          float* __restrict__ gmean2D, const unsigned long long* __restrict__ segmask, const float* __restrict__ dsum,
          float* __restrict__ gpart)
{
    __shared__ float stage[64 * NACC];            // one chunk's per-entry sums, entry-major, for the coalesced flush
    __shared__ uint32_t rel_gid[SEG];             // the strip's relevant entries, ascending: Gaussian id ...
    __shared__ unsigned short rel_j[SEG];         // ... and entry index inside the segment
    __shared__ float4 pix[8][32];                 // per PIXEL PAIR [slot][pair]: coords, bg term, 11 upstream grads, carries
    __shared__ int2 ncp[32];                      // per pixel pair: the two pixels' n_contrib (row-packed tail chunks read it per lane)
    SegCtx c;
    const int w = (int)(blockIdx.x & 3u);
    const int ntiles = cam.gx * cam.gy;
    if ((blockIdx.x >> 2) >= seg_off[seg_counts_offset(ntiles, cap) + SEG_LIST_NEEDED]) return;    // beyond the needed list (vr_segment.h)
    if (!seg_setup_at(cam, ranges, seg_off, seg_off[seg_list_offset(ntiles, cap, SEG_LIST_NEEDED) + (blockIdx.x >> 2)], w, c)) return;
    const int lane = threadIdx.x;
    const int pixslot = w * 64 + lane;            // this pixel's slot in the [segment][256] buffers

    // ---- ONE batch of independent global loads right after the segment descriptor: relevance masks, the segment's
    // list entries (all four 64-entry parts: which ones are relevant is only known once the masks are back) and the
    // whole pixel state.  Issued in dependency order (masks -> list -> pixel state) these were three memory round
    // trips in the life of a short workgroup that has only ~3 others on its SIMD to hide them behind.
    const unsigned long long* masks = segmask + (size_t)c.seg * 16 + w * 4;
    const unsigned long long mraw0 = masks[0], mraw1 = masks[1], mraw2 = masks[2], mraw3 = masks[3];
    uint32_t gid_q[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) gid_q[q] = q * 64 + lane < c.count ? point_list[c.first + q * 64 + lane] : 0u;
    const size_t N = (size_t)cam.H * cam.W;
    const float v_pxf = (float)c.px, v_pyf = (float)c.py;
    PixGrad pg;
    load_pixgrad(c.inside, c.pix, N, dL_dcolor, dL_ddepth, dL_dquat, dL_dscale, dL_dalpha, cam.flags, final_T, dsum, pg);
    float v_Tf = 1.0f;
    int v_nc = 0;
    if (c.inside) { v_Tf = final_T[c.pix]; v_nc = (int)n_contrib[c.pix]; }
    // carries at the END of this segment: transmittance behind its last entry (the next segment's boundary value
    // when the pixel is still alive there), and the w*u sum of everything behind the segment
    const float Tnext = c.flag == 2u ? Tbuf[(size_t)(c.seg + 1) * SEG + pixslot] : -1.0f;
    float v_Scar = Ubuf[(size_t)c.seg * SEG + pixslot];

    // ---- the entries relevant to this strip, compacted in list order
    const unsigned long long m0 = uniform64(mraw0), m1 = uniform64(mraw1), m2 = uniform64(mraw2), m3 = uniform64(mraw3);
    const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
    const int nrel = n0 + n1 + n2 + n3;
    if (nrel == 0) return;
    {
        const unsigned long long lt = (1ull << lane) - 1ull;
        const unsigned long long mm[4] = {m0, m1, m2, m3};
        const int before[4] = {0, n0, n0 + n1, n0 + n1 + n2};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if ((mm[q] >> lane) & 1ull) {
                const int pos = before[q] + __popcll(mm[q] & lt);
                rel_j[pos] = (unsigned short)(q * 64 + lane);
                rel_gid[pos] = gid_q[q];
            }
        }
    }

    // ---- pixel state, lane = pixel of this wave's 8x8 region
    // "background" terms weighted by the final transmittance: bg colour, alpha = 1 - T and, with VR_FLAG_FILL_EMPTY,
    // the identity rotation added to cov_quat (an extra channel: cut off together with the others)
    constexpr bool no_extra = NO_EXTRA;
    const float v_fill = ((cam.flags & FLAG_FILL_EMPTY) && !no_extra) ? pg.g[4] : 0.0f;
    const float v_bgterm =
        v_Tf * ((fmaf(cam.bg[2], pg.g[2], fmaf(cam.bg[1], pg.g[1], cam.bg[0] * pg.g[0])) + v_fill) - pg.galpha);
    float v_Tcar = v_Tf;
    if (!(Tnext < 0.0f)) v_Tcar = Tnext;   // pixel still alive at the next segment
    const int seg_lo = c.sl * SEG;       // first list entry (tile-relative) of this segment
    // Per-pixel record in LDS, interleaved by PIXEL PAIR (a = even pixel, b = odd pixel of the strip): the
    // pixel loop reads it back as wave-uniform broadcasts (LDS pipe, not 16 v_readlane on the VALU pipe) and
    // gets every quantity as a ready-made (a,b) register pair for packed fp32 math.  float4 slots:
    //   0 {px_a px_b py_a py_b}  1 {bg_a bg_b g0_a g0_b}  2 {g1 g2}  3 {g3 g4}  4 {g5 g6}  5 {g7 g8}
    //   6 {g9 g10}  7 {Tcar_a Tcar_b Scar_a Scar_b} -- the two carries, updated in place by lane 63.
    // SLOT-MAJOR, pix[slot][pair], and written as whole float4s by the EVEN lanes (the odd pixel's values come over DPP):
    // eight ds_write_b128 at 32 consecutive addresses.  Round 4 had every lane store its 16 scalars into pixrec[pair][.] +
    // (lane & 1): a lane stride of 32 floats per pair put 32 lanes on each of two banks -- 16 stores of ~32 cycles per
    // workgroup, and that, not the loop's broadcast reads, was the kernel's SQ_LDS_BANK_CONFLICT count (0.71 of its
    // SQ_ACTIVE_INST_LDS; profiles/tools/ubench/lds_pixrec.hip measures both patterns).
    {
        float mine[16];
        mine[0] = v_pxf; mine[1] = v_pyf; mine[2] = v_bgterm;
#pragma unroll
        for (int k = 0; k < NCH; ++k) mine[3 + k] = pg.g[k];
        mine[14] = v_Tcar; mine[15] = v_Scar;
        float other[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) other[k] = __shfl_xor(mine[k], 1, 64);
        const int nc_other = __shfl_xor(v_nc, 1, 64);
        if ((lane & 1) == 0) {
#pragma unroll
            for (int sl = 0; sl < 8; ++sl)
                pix[sl][lane >> 1] = make_float4(mine[2 * sl], other[2 * sl], mine[2 * sl + 1], other[2 * sl + 1]);
            ncp[lane >> 1] = make_int2(v_nc, nc_other);
        }
    }
    __syncthreads();

    int mx = v_nc;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d, 64));
    const int wave_maxc = mx;
    const int nchunks = (nrel + 63) >> 6;

    // The chunk's records are gathered one chunk AHEAD (round 6): the next (nearer) chunk is always a full one (lane l <-> entry
    // 63 - l), its 80-byte records are requested right after this chunk's sums are staged and travel under the flush.
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = make_float4(0.f, 0.f, 1.0f, 0.f), q2 = q0, q3 = q0, q4 = q0;     // (an empty lane: thr = 1)
    auto gather = [&](const int r_, const bool has_) {
        asm volatile("" ::: "memory");        // (not hoisted above the pixel loop: twenty more live registers there spill)
        if (has_) {
            const float4* src = reinterpret_cast<const float4*>(rec + rel_gid[r_]);
            q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3]; q4 = src[4];
        }
    };
    // (the last chunk walked redefines the registers too -- with anything: a value carried around the loop would have to
    // survive the pixel loop, twenty registers the kernel does not have)
    auto next_records = [&](const int ch_) {
        if (ch_ > 0) gather((ch_ - 1) * 64 + (63 - lane), true);
        else q0 = q1 = q2 = q3 = q4 = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    if (VR_BWD_PREFETCH) {
        const int nch0 = (nrel + 63) >> 6, n_last = nrel - (nch0 - 1) * 64;
        const int span0 = !VR_BWD_PACK_TAILS ? 64 : (n_last <= 16 ? 16 : n_last <= 32 ? 32 : 64);
        const int r0 = (nch0 - 1) * 64 + (span0 - 1 - (lane & (span0 - 1)));
        gather(r0, r0 < nrel);
    }
    for (int ch = nchunks - 1; ch >= 0; --ch) {
        // ---- lane l owns the strip's relevant entry number ch*64 + (63-l): back-to-front over lanes.  The LAST chunk (the first
        // one walked) may hold few entries: <= 16 of them are held once per 16-lane row, <= 32 once per half (bwd_chunk_trips)
        const int span = !VR_BWD_PACK_TAILS ? 64 : (nrel - ch * 64 <= 16 ? 16 : nrel - ch * 64 <= 32 ? 32 : 64);      // (wave-uniform)
        const int r = ch * 64 + (span - 1 - (lane & (span - 1)));
        const bool has = r < nrel;
        const int ej = has ? (int)rel_j[r] : 0;        // entry index inside the segment
        // tile-relative list index; a lane without an entry sits behind every pixel's last contributor, so the
        // per-pixel test "e < n_contrib" also covers "the lane has an entry" (one compare instead of two masked blocks)
        const int e = has ? seg_lo + ej : 0x7FFFFFFF;
        float sx = 0.f, sy = 0.f, cA = 0.f, cB = 0.f, cC = 0.f, op = 0.f, thr = 1.0f;
        float at[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) at[k] = 0.0f;
        if (!VR_BWD_PREFETCH) gather(r, has);
        if (has) {
            sx = q0.x; sy = q0.y; cA = q0.z; cB = q0.w; cC = q1.x; op = q1.y; thr = q1.z;
            at[0] = q2.x; at[1] = q2.y; at[2] = q2.z; at[3] = q1.w;
            at[4] = q2.w; at[5] = q3.x; at[6] = q3.y; at[7] = q3.z;
            at[8] = q3.w; at[9] = q4.x; at[10] = q4.y;
        }
        // the conic and the threshold in log2 units, as the forward stages them (an empty lane: thr2 > 0, nothing passes)
        float kA, kB, kC, thr2;
        splat_k2(cA, cB, cC, thr, kA, kB, kC, thr2);
        // accumulators: .x sums over the even pixels, .y over the odd ones (added at the end of the chunk)
        f2 acc[NACC];
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = f2_splat(0.0f);
        // nearest list index held by this chunk (its first relevant entry): pixels whose last
        // contributor lies in front of it have nothing to do here
        const int chunk_lo = __builtin_amdgcn_readfirstlane(seg_lo + (int)rel_j[ch * 64]);       // (wave-uniform: a scalar compare per trip)

        if (chunk_lo < wave_maxc) {
#define VR_TRIPS(R) bwd_chunk_trips<R, NO_EXTRA, DET, FAST, NO_DEPTH>(lane, e, chunk_lo, v_nc, sx, sy, kA, kB, kC, thr2, op, cA, cB, cC, at, acc, pix, ncp)
            if (span == 16) VR_TRIPS(4);
            else if (span == 32) VR_TRIPS(2);
            else VR_TRIPS(1);
#undef VR_TRIPS
            // ---- per-entry sums of the chunk -> LDS (entry-major) -> one coalesced set of global atomics per entry
            if (has) {
                float o[NACC];
#pragma unroll
                for (int k = 0; k < NACC; ++k) o[k] = acc[k].x + acc[k].y;
                // the per-splat signs / factors of the conic and mean2D sums (see the pixel loop)
                o[0] = -0.5f * o[0];
                o[1] = -o[1];
                o[2] = -0.5f * o[2];
                o[15] = -o[15];
                o[16] = -o[16];
#pragma unroll
                for (int k = 0; k < NACC; ++k) stage[lane * NACC + k] = o[k];
            }
            if (VR_BWD_PREFETCH) next_records(ch);
            __syncthreads();
            for (int v = lane; v < span * NACC; v += 64) {
                const int l = v / NACC, k = v - l * NACC;
                const int rr = ch * 64 + (span - 1 - l);
                if (rr >= nrel) continue;
                float sum = stage[v];
                for (int gq = span; gq < 64; gq += span) sum += stage[v + gq * NACC];      // (row-packed chunk: the groups' partial sums)
                if (sum == 0.0f) continue;
                const float val = k < 15 ? sum : sum * (k == 15 ? 0.5f * (float)cam.W : 0.5f * (float)cam.H);
                if (DET) {   // deterministic mode: this (list entry, region)'s own slot, summed later in list order
                    gpart[((size_t)(c.first + (int)rel_j[rr]) * 4 + w) * NACC + k] = val;
                    continue;
                }
                const uint32_t gid = rel_gid[rr];
                if (k < 15) atomicAdd(&gacc[(size_t)gid * 16 + k], val);
                else atomicAdd(&gmean2D[(size_t)gid * 3 + (k - 15)], val);
            }
            __syncthreads();
        } else if (VR_BWD_PREFETCH) {
            next_records(ch);
        }
    }
}

// ---- deterministic mode (VR_FLAG_DETERMINISTIC): k_seg_bwd wrote one slot of NACC sums per (list entry, region);
// the list entries of every Gaussian are found by a stable sort of (Gaussian id, list index) pairs and their
// slots are added in list order, regions 0..3, by ONE thread per Gaussian -- a fixed order, so the per-Gaussian sums
// are bit-reproducible.  A test/debug mode: 272 bytes per list entry of scratch and a sort on top of the normal pass.
__global__ void __launch_bounds__(256)
k_det_pairs(const uint32_t* __restrict__ point_list, long R, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j < R) { keys[j] = point_list[j]; vals[j] = (uint32_t)j; }
}

__global__ void __launch_bounds__(256)
k_det_reduce(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, long R, const float* __restrict__ gpart,
             float* __restrict__ gacc, float* __restrict__ gmean2D)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    const uint32_t gid = keys[i];
    if (i > 0 && keys[i - 1] == gid) return;          // not the first list entry of its Gaussian
    // DOUBLE accumulators (round 4): a test mode, its speed is irrelevant -- what the comparison with the checker's double
    // sums should see is the kernels' per-fragment arithmetic, not the rounding of a second fp32 summation over a
    // Gaussian's (up to tens of thousands of) list entries
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
    for (long t = i; t < R && keys[t] == gid; ++t) {
        const float* src = gpart + (size_t)vals[t] * 4 * NACC;
        for (int q = 0; q < 4 * NACC; q += NACC) {
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] += (double)src[q + k];
        }
    }
#pragma unroll
    for (int k = 0; k < 15; ++k) gacc[(size_t)gid * 16 + k] = (float)acc[k];
    gmean2D[(size_t)gid * 3 + 0] = (float)acc[15];
    gmean2D[(size_t)gid * 3 + 1] = (float)acc[16];
}

size_t render_bwd_det_bytes(long R, int P)
{
    (void)P;
    const size_t r = (size_t)(R > 0 ? R : 1);
    return align_up(r * 4 * NACC * sizeof(float), 256) + 4 * align_up(r * 4, 256) + sort_pairs_scratch_bytes(R) + 256;
}

size_t render_bwd_scratch_bytes(long R, int ntiles)
{
    return align_up(seg_capacity(R, ntiles) * SEG * sizeof(float), 256);
}

int launch_render_bwd(const Camera& cam, long R, const int2* ranges, const uint32_t* point_list, const Splat* rec,
                      const uint32_t* seg_off, const uint32_t* seg_needed, const float* Tbuf, const float* part,
                      const unsigned long long* segmask, void* scratch, const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                      const float* dL_ddepth, const float* dL_dquat, const float* dL_dscale,
                      const float* dL_dalpha, float* gacc, float* gmean2D, const float* dsum, void* det_scratch, int P,
                      bool zero_accumulators, hipStream_t s, bool debug)
{
    const int ntiles = cam.gx * cam.gy;
    if (ntiles == 0 || R == 0) return 0;
    const unsigned nseg = (unsigned)seg_capacity(R, ntiles);
    float* Ubuf = (float*)scratch;
    float* gpart = nullptr;
    if (det_scratch) {
        gpart = (float*)det_scratch;
        VR_HIP(hipMemsetAsync(gpart, 0, (size_t)R * 4 * NACC * sizeof(float), s));
    }
    hipLaunchKernelGGL(k_seg_u, dim3(nseg), dim3(256), 0, s, cam, ranges, seg_off, (uint32_t)nseg, seg_needed, part, n_contrib,
                       dL_dcolor, dL_ddepth, dL_dquat, dL_dscale, final_T, dsum, Ubuf,
                       zero_accumulators ? gacc : nullptr, (size_t)P * 16, zero_accumulators ? gmean2D : nullptr, (size_t)P * 3);
    VR_KERNEL_CHECK("seg_u", s, debug);
    hipLaunchKernelGGL(k_seg_suffix, dim3(2 * ntiles), dim3(256), 0, s, ntiles, (uint32_t)nseg, seg_off, seg_needed, Ubuf);
    VR_KERNEL_CHECK("seg_suffix", s, debug);
    prof_begin(VR_STAGE_K_SEG_BWD, s);
#define VR_BWD3(NOX, DETM, FST, NODEP)                                                                                 \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_seg_bwd<NOX, DETM, FST, NODEP>), dim3(nseg * 4), dim3(64), 0, s, cam, ranges, seg_off, \
                       (uint32_t)nseg, seg_needed, point_list, rec, Tbuf, (const float*)Ubuf, final_T, n_contrib, dL_dcolor, dL_ddepth, \
                       dL_dquat, dL_dscale, dL_dalpha, gacc, gmean2D, segmask, dsum, gpart)
    // (NO_DEPTH: the production instantiations only -- not the all-channels-off colour-only one, which never sees channel 3, nor
    // the deterministic test mode)
#define VR_BWD2(NOX, DETM, FST) do { if (!(NOX) && !(DETM) && dL_ddepth == nullptr) VR_BWD3(NOX, DETM, FST, true); else VR_BWD3(NOX, DETM, FST, false); } while (0)
#define VR_BWD(NOX, DETM) do { if (cam.flags & FLAG_FAST_EXP) VR_BWD2(NOX, DETM, true); else VR_BWD2(NOX, DETM, false); } while (0)
    {
        const bool nox = (cam.flags & FLAG_EXTRA_NO_ALPHA_GRAD) != 0u;
        if (gpart) { if (nox) VR_BWD(true, true); else VR_BWD(false, true); }
        else { if (nox) VR_BWD(true, false); else VR_BWD(false, false); }
    }
#undef VR_BWD
#undef VR_BWD2
#undef VR_BWD3
    prof_end(VR_STAGE_K_SEG_BWD, s);
    VR_KERNEL_CHECK("seg_bwd", s, debug);
    if (gpart) {
        const size_t r4 = align_up((size_t)R * 4, 256);
        char* base = (char*)det_scratch + align_up((size_t)R * 4 * NACC * sizeof(float), 256);
        uint32_t *k0 = (uint32_t*)base, *v0 = (uint32_t*)(base + r4), *k1 = (uint32_t*)(base + 2 * r4),
                 *v1 = (uint32_t*)(base + 3 * r4);
        void* sort_scr = base + 4 * r4;
        hipLaunchKernelGGL(k_det_pairs, dim3(cdiv(R, 256)), dim3(256), 0, s, point_list, R, k0, v0);
        int bits = 1, where = 0;
        while ((1L << bits) < (long)P) ++bits;
        int rc = launch_sort_pairs(k0, v0, k1, v1, R, 0u, bits, sort_scr, s, debug, &where);
        if (rc) return rc;
        hipLaunchKernelGGL(k_det_reduce, dim3(cdiv(R, 256)), dim3(256), 0, s, (const uint32_t*)(where ? k1 : k0),
                           (const uint32_t*)(where ? v1 : v0), R, (const float*)gpart, gacc, gmean2D);
        VR_KERNEL_CHECK("det_reduce", s, debug);
    }
    return 0;
}

}  // namespace vr
