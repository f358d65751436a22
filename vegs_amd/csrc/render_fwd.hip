// render_fwd.hip -- tile-based front-to-back alpha compositing of 11 channels
// (rgb, depth, rotation quaternion, scale) + alpha, final transmittance and contributor count.
//
// Replaces the native render stage behind GaussianRasterizer.forward (reference call site
// gaussian_renderer/__init__.py:86-94; the 6-tuple it returns is consumed at :109-119 and by
// loss/normal_guidance.py:3-22).  Semantics: SURVEY.md A.4 with fork assumptions A-1..A-5.
//
// MI355X design -- SEGMENTED compositing (a chunked scan over each tile's list), not one
// sequential loop per tile.  KITTI-shaped scenes pile tens of thousands of far splats onto the
// vanishing-point tiles (lists of 10^5 entries while the mean is 10^3); a loop per tile leaves
// 255 CUs idle behind the longest tile.  The list of every tile is cut into segments of 256
// entries and the unit of work is (tile, segment), ~R/256 + T workgroups of equal size:
//   A  k_seg_alpha   every (tile, segment): per pixel, P = product of (1 - alpha) over the segment
//   B  k_seg_scan    per tile: Tb[s+1] = Tb[s] * P[s] per pixel (the only sequential step: one
//                    multiply per segment); a pixel is finished in the first segment with
//                    Tb*P < 1e-4 (fp32 products by factors <= 1 are monotone, so this is exactly
//                    "the stop test fires inside this segment"); segments behind that are skipped
//   C  k_seg_blend   every needed (tile, segment): blend the segment from its known Tb into
//                    segment-local sums
//   D  k_seg_combine per tile: add the segment-local sums in order, write the images
// The arithmetic (local product / local sums per segment, added in order) is the one the spec
// fixes (oracle: or_render_fwd), so the images stay bit-identical to the sequential oracle.
// One 256-thread workgroup = one 16x16 tile = 4 wave64, lane = pixel (one 8x8 region = "strip" per wave); splat
// records are gathered with dwordx4 loads into LDS and read back as wave-uniform broadcasts.
#include <stdlib.h>

#include "vr_host.h"
#include "vr_segment.h"

namespace vr {

constexpr int NPART = 13;      // planes of `part` per segment: 11 channel sums, the local product, the last contributor
__device__ __forceinline__ uint32_t hinted_limit(uint32_t h) { return h + 2u + (h >> 3); }
// Segment rounds without a hint ("auto").  Re-tuned in round 4 on the tight tile lists (profiles/tools/ab/rounds.sh: the street
// scene with its discs x 1 / 1.5 / 2 / 3 / 5 = 5.0 / 5.9 / 7.0 / 10 / 20 list segments per tile; render forward, ms):
//   x1    all at once 0.358   rounds 6 + 48: 0.400, 6 + 64: 0.401, 12 + 96: 0.383   (deep vanishing-point tiles need the third
//                                                                                  round's whole rest: a tail)
//   x1.5  all at once 0.370   6 + 48: 0.344   6 + 64: 0.320
//   x2    all at once 0.398   6 + 48: 0.319   6 + 64: 0.317   4 + 32: 0.309
//   x3    all at once 0.488   6 + 24: 0.347   3 + 8: 0.292    2 + 8: 0.272    2 + 16: 0.273
//   x5    all at once 0.842   6 + 24: 0.336   3 + 8: 0.242    2 + 8: 0.219    2 + 4: 0.216
// Large, opaque discs saturate a tile within one or two segments.  List density alone does not predict the outcome in
// between: the x1 view on the reference's FULL rectangles (7.4 segments per tile, a third of them no-ops) loses 0.06 ms with
// rounds, the x2 view (7.0) wins 0.08 -- so the rounds are used from 8.5 segments per tile on (x3 and denser: 2 + 8), with
// the list length of a VR_FLAG_FULL_TILE_LISTS forward counted at two thirds; VR_FLAG_ROUNDS_ON below that: 6 + 64.  The host
// knows the density when it launches.
constexpr uint32_t AUTO_FIRST_SPARSE = 6u, AUTO_FIRST_DENSE = 2u;     // segments of every tile computed in round 0
constexpr uint32_t AUTO_SECOND_SPARSE = 64u, AUTO_SECOND_DENSE = 8u;  // ... and at least this many more in round 1
constexpr uint32_t AUTO_DENSE_X2 = 17u;                               // threshold in HALF segments per tile (8.5)
constexpr uint32_t AUTO_FIRST = AUTO_FIRST_SPARSE;                    // (grid bound of round 0: the larger of the two)

// seg_off[t] = first global segment id of tile t; seg_off[T] = total.  Single workgroup.  (k_seg_tiles then
// fills seg_tile[s] = tile of segment s, 0xFFFFFFFF beyond the total: the launch grids cover `cap` segments.)
// Also the ONE place where the caller's needed-segment hint is read: limit[t] = number of leading segments of tile t
// that k_seg_alpha computes up front (all of them without a hint).  The snapshot lives in this call's own binning
// buffer (the seg_needed array, which k_seg_scan overwrites with its result), so k_seg_alpha and k_seg_scan agree on
// it even if the caller's array changes underneath them (another stream rendering the same camera).
constexpr int SEGOFF_THREADS = 1024;
__global__ void __launch_bounds__(SEGOFF_THREADS)
k_seg_offsets(const int2* __restrict__ ranges, int ntiles, uint32_t* __restrict__ seg_off,
              const uint32_t* __restrict__ hint, uint32_t* __restrict__ limit, uint32_t cap, uint32_t auto_first, int first_fused)
{
    constexpr int NW = SEGOFF_THREADS / 64;
    __shared__ uint32_t wsum[NW], wsum_a[NW];
    __shared__ uint32_t carry_s, carry_a;
    uint32_t* const counts = seg_off + seg_counts_offset(ntiles, cap);
    uint32_t* const act_off = seg_off + seg_actoff_offset(ntiles, cap);
    if (threadIdx.x == 0) { carry_s = 0; carry_a = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int base = 0; base < ntiles; base += SEGOFF_THREADS) {
        const int t = base + threadIdx.x;
        uint32_t n = 0, a = 0;
        if (t < ntiles) {
            const int2 r = ranges[t];
            n = (uint32_t)((r.y - r.x + SEG - 1) / SEG);
            // no hint: the first AUTO_FIRST segments of every tile up front, the rest in the catch-up rounds if any pixel
            // is still alive behind them (k_seg_scan) -- most tiles never need more
            a = hint ? min(n, hinted_limit(min(hint[t], 0x3FFFFFFFu))) : min(n, auto_first);
            limit[t] = a;
            if (first_fused) a -= a > 0u ? 1u : 0u;      // the tile's FIRST segment is not on the list: k_seg_first takes it (below)
        }
        uint32_t incl = n, incl_a = a;      // two scans: all segments (global ids), round-0 segments (list positions)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d, 64), oa = __shfl_up(incl_a, d, 64);
            if (lane >= d) { incl += o; incl_a += oa; }
        }
        if (lane == 63) { wsum[w] = incl; wsum_a[w] = incl_a; }
        __syncthreads();
        uint32_t woff = 0, woff_a = 0;
        for (int k = 0; k < w; ++k) { woff += wsum[k]; woff_a += wsum_a[k]; }
        const uint32_t carry = carry_s, ca = carry_a;
        if (t < ntiles) { seg_off[t] = carry + woff + incl - n; act_off[t] = ca + woff_a + incl_a - a; }
        __syncthreads();
        if (threadIdx.x == SEGOFF_THREADS - 1) { carry_s = carry + woff + incl; carry_a = ca + woff_a + incl_a; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        seg_off[ntiles] = carry_s;
        act_off[ntiles] = carry_a;
        counts[0] = carry_a;          // entries of the round-0 list; the catch-up lists start empty
        counts[1] = 0;
        counts[2] = 0;
        counts[SEG_LIST_NEEDED] = 0;
        counts[SEG_COUNT_HEAVY] = 0;
        for (int q = 0; q < SEG_QUEUES; ++q) seg_off[seg_qcount_offset(ntiles, cap, q)] = 0;
    }
}

// segment table entries (vr_segment.h): one thread per segment of the launch grid
__global__ void __launch_bounds__(256)
k_seg_tiles(int ntiles, const int2* __restrict__ ranges, uint32_t* __restrict__ seg_off,
            const uint32_t* __restrict__ limit, uint32_t cap, int first_fused)
{
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= cap) return;
    int4* __restrict__ seg_info = reinterpret_cast<int4*>(seg_off + seg_tile_offset(ntiles));
    if (b >= seg_off[ntiles]) { seg_info[b] = make_int4(-1, 0, 0, 0); return; }
    int lo = 0, hi = ntiles;       // largest t with seg_off[t] <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg_off[mid] <= b) lo = mid; else hi = mid - 1;
    }
    const int sl = (int)(b - seg_off[lo]);
    const int2 r = ranges[lo];
    const int first = r.x + sl * SEG;
    // flag 3 = behind the round-0 prefix of its tile: not in the round-0 list, k_seg_scan decides (and rewrites the flag)
    const bool up_front = (uint32_t)sl < limit[lo];
    seg_info[b] = make_int4(lo, first, min(SEG, r.y - first), sl | (up_front ? 0 : (int)(3u << 30)));
    if (up_front && sl >= first_fused) seg_off[seg_list_offset(ntiles, cap, 0) + seg_off[seg_actoff_offset(ntiles, cap) + lo] + sl - first_fused] = b;
}

// k_seg_offsets + k_seg_tiles in ONE launch (round 5): every workgroup of the table kernel repeats the two prefix sums over
// the tiles in LDS -- a view has ~2 k tiles: 8 per thread, ~2 us, against a launch of its own for a single workgroup (6 us)
// plus the dispatch gap -- and then fills its 256 segment slots from the LDS copies; workgroup 0 also writes the global
// arrays the later kernels read.  Same values in the same places as the two kernels (which remain: frames of more than
// SEGTAB_MAX_TILES tiles, 3.1 Mpixel).  Render forward -6 us.
constexpr int SEGTAB_MAX_TILES = 3072;

// ---- CHAIN MODE (round 5; a forward without rounds).  Half of a street view's list segments are DEAD -- behind the point
// where every pixel of their tile has stopped -- and nearly all of those lie in the few dozen deepest (vanishing-point)
// tiles: up to ~950 segments of which at most ~200 are needed.  Which ones is the RESULT of the chain k_seg_scan walks
// after k_seg_alpha.  In chain mode the chains of the HEAVY tiles (more than CHAIN_PREFIX segments) are walked INSIDE the
// k_seg_alpha launch, by one WALKER workgroup per heavy tile that follows the tile's segment products as they are
// published (a flag per segment) and, when the chain ends, publishes the number of needed segments (`dead`): a workgroup
// whose segment lies behind it returns at once.  It only pays if the walkers know before the dead segments are dispatched,
// so the launch's work is ordered for them:
//   [ the heavy tiles' first segments (fused) | A: their levels 1 .. CHAIN_PREFIX-1 | the walkers |
//     the other tiles' first segments | B: their other segments   (the bulk: ~40 us during which the walkers get through A) |
//     C: the heavy tiles' levels >= CHAIN_PREFIX, LEVEL-major (all tiles' level k before any tile's level k + 1) ]
// Nothing waits but the walkers (bounded; a walker that runs out of patience leaves its tile to k_seg_scan), the same
// segment products feed the same chain arithmetic, and a skipped segment is one no pixel needs: results bit for bit.
constexpr int CHAIN_PREFIX = 32;               // (16 and 48 measured the same)
constexpr int CHAIN_MAX_HEAVY = 128;           // more heavy tiles than this: the view runs without walkers (classic order)
constexpr int SEG_COUNT_CHAIN_HEAVY = 8;       // counts[]: heavy tiles of this view (0 = no chain mode)
constexpr int SEG_COUNT_CHAIN_A = 9;           //           list entries of region A
constexpr int SEG_COUNT_CHAIN_B = 10;          //           ... of region B
constexpr uint32_t CHAIN_OPEN = 0xFFFFFFFFu;   // dead[tile]: the chain has not ended yet
// chain mode's arrays live in the lists of the catch-up rounds it excludes: flags [cap] | dead [ntiles], heavy tile ids
__host__ __device__ inline size_t chain_flags_offset(int ntiles, size_t cap) { return seg_list_offset(ntiles, cap, 1); }
__host__ __device__ inline size_t chain_dead_offset(int ntiles, size_t cap) { return seg_list_offset(ntiles, cap, 2); }
// (the tiles in launch order: the heavy ones, ascending, then the others: [ntiles])
__host__ __device__ inline size_t chain_heavy_offset(int ntiles, size_t cap) { return seg_list_offset(ntiles, cap, 2) + (size_t)seg_tile_offset(ntiles); }

__global__ void __launch_bounds__(256)
k_seg_table(const int2* __restrict__ ranges, int ntiles, uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ hint,
            uint32_t* __restrict__ limit, uint32_t cap, uint32_t auto_first, int first_fused, int chain)
{
    __shared__ uint32_t so[SEGTAB_MAX_TILES + 1], ao[SEGTAB_MAX_TILES + 1], lim[SEGTAB_MAX_TILES];
    __shared__ uint32_t ha[SEGTAB_MAX_TILES + 1];      // chain mode: list entries of the HEAVY tiles before tile t
    __shared__ uint32_t hv_t[CHAIN_MAX_HEAVY], hv_n[CHAIN_MAX_HEAVY];
    __shared__ uint32_t wsum[4], wsum_a[4], wsum_h[4], wsum_ha[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int per = (ntiles + 255) / 256;                    // consecutive tiles per thread
    const int t0 = min(ntiles, (int)threadIdx.x * per), t1 = min(ntiles, t0 + per);
    uint32_t sum_n = 0, sum_a = 0, sum_h = 0, sum_ha = 0;
    for (int t = t0; t < t1; ++t) {
        const int2 r = ranges[t];
        const uint32_t n = (uint32_t)((r.y - r.x + SEG - 1) / SEG);
        uint32_t a = hint ? min(n, hinted_limit(min(hint[t], 0x3FFFFFFFu))) : min(n, auto_first);
        lim[t] = a;
        if (first_fused) a -= a > 0u ? 1u : 0u;
        so[t] = sum_n;                                       // thread-local exclusive prefixes, rebased below
        ao[t] = sum_a;
        ha[t] = sum_ha;
        sum_n += n;
        sum_a += a;
        if (chain && n > (uint32_t)CHAIN_PREFIX) { sum_h += 1u; sum_ha += a; }
    }
    uint32_t incl = sum_n, incl_a = sum_a, incl_h = sum_h, incl_ha = sum_ha;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64), oa = __shfl_up(incl_a, d, 64);
        const uint32_t oh = __shfl_up(incl_h, d, 64), oha = __shfl_up(incl_ha, d, 64);
        if (lane >= d) { incl += o; incl_a += oa; incl_h += oh; incl_ha += oha; }
    }
    if (lane == 63) { wsum[w] = incl; wsum_a[w] = incl_a; wsum_h[w] = incl_h; wsum_ha[w] = incl_ha; }
    __syncthreads();
    uint32_t base = incl - sum_n, base_a = incl_a - sum_a, base_h = incl_h - sum_h, base_ha = incl_ha - sum_ha;
    for (int k = 0; k < w; ++k) { base += wsum[k]; base_a += wsum_a[k]; base_h += wsum_h[k]; base_ha += wsum_ha[k]; }
    const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3], total_a = wsum_a[0] + wsum_a[1] + wsum_a[2] + wsum_a[3];
    const uint32_t nheavy_all = wsum_h[0] + wsum_h[1] + wsum_h[2] + wsum_h[3];
    const uint32_t total_ha = wsum_ha[0] + wsum_ha[1] + wsum_ha[2] + wsum_ha[3];
    // chain mode for THIS view: some heavy tile, and no more of them than there are walkers
    const uint32_t nheavy = (chain && nheavy_all <= (uint32_t)CHAIN_MAX_HEAVY) ? nheavy_all : 0u;
    {
        uint32_t h = base_h;
        for (int t = t0; t < t1; ++t) {
            const uint32_t n = (t + 1 < t1 ? so[t + 1] : sum_n) - so[t];
            so[t] += base; ao[t] += base_a; ha[t] += base_ha;
            if (nheavy && n > (uint32_t)CHAIN_PREFIX) {
                hv_t[h] = (uint32_t)t; hv_n[h] = n;
                if (blockIdx.x == 0) seg_off[chain_heavy_offset(ntiles, cap) + h] = (uint32_t)t;
                ++h;
            } else if (nheavy && blockIdx.x == 0) {
                seg_off[chain_heavy_offset(ntiles, cap) + nheavy + ((uint32_t)t - h)] = (uint32_t)t;     // (h = heavy tiles before t)
            }
        }
    }
    if (threadIdx.x == 0) { so[ntiles] = total; ao[ntiles] = total_a; ha[ntiles] = total_ha; }
    __syncthreads();
    // region sizes of the chain order (every heavy tile has all of the levels 1 .. CHAIN_PREFIX-1)
    const uint32_t size_a = nheavy * (uint32_t)(CHAIN_PREFIX - 1);
    const uint32_t size_b = total_a - total_ha;
    if (blockIdx.x == 0) {                                   // the global copies (k_seg_offsets' outputs)
        uint32_t* const counts = seg_off + seg_counts_offset(ntiles, cap);
        uint32_t* const act_off = seg_off + seg_actoff_offset(ntiles, cap);
        for (int t = threadIdx.x; t <= ntiles; t += 256) {
            seg_off[t] = so[t];
            act_off[t] = ao[t];
            if (t < ntiles) limit[t] = lim[t];
            if (chain && t < ntiles) seg_off[chain_dead_offset(ntiles, cap) + t] = CHAIN_OPEN;
        }
        if (threadIdx.x == 0) {
            counts[0] = total_a;
            counts[1] = 0;
            counts[2] = 0;
            counts[SEG_LIST_NEEDED] = 0;
            counts[SEG_COUNT_HEAVY] = 0;
            counts[SEG_COUNT_CHAIN_HEAVY] = nheavy;
            counts[SEG_COUNT_CHAIN_A] = size_a;
            counts[SEG_COUNT_CHAIN_B] = size_b;
            for (int q = 0; q < SEG_QUEUES; ++q) seg_off[seg_qcount_offset(ntiles, cap, q)] = 0;
        }
    }
    // ---- k_seg_tiles' part: one segment slot per thread
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= cap) return;
    if (chain) seg_off[chain_flags_offset(ntiles, cap) + b] = 0u;       // "product published" flag of the slot
    int4* __restrict__ seg_info = reinterpret_cast<int4*>(seg_off + seg_tile_offset(ntiles));
    if (b >= total) { seg_info[b] = make_int4(-1, 0, 0, 0); return; }
    int lo = 0, hi = ntiles;       // largest t with so[t] <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (so[mid] <= b) lo = mid; else hi = mid - 1;
    }
    const int sl = (int)(b - so[lo]);
    const int2 r = ranges[lo];
    const int first = r.x + sl * SEG;
    const bool up_front = (uint32_t)sl < lim[lo];
    seg_info[b] = make_int4(lo, first, min(SEG, r.y - first), sl | (up_front ? 0 : (int)(3u << 30)));
    if (!(up_front && sl >= first_fused)) return;
    uint32_t pos = ao[lo] + (uint32_t)(sl - first_fused);                  // classic: tile-major
    if (nheavy) {
        const uint32_t n_lo = so[lo + 1] - so[lo];
        if (n_lo <= (uint32_t)CHAIN_PREFIX) {
            pos = size_a + (ao[lo] - ha[lo]) + (uint32_t)(sl - first_fused);               // B
        } else {
            int hl = 0, hh = (int)nheavy - 1;      // the tile's index among the heavy ones (ascending tile ids)
            while (hl < hh) {
                const int mid = (hl + hh) >> 1;
                if (hv_t[mid] < (uint32_t)lo) hl = mid + 1; else hh = mid;
            }
            if (sl < CHAIN_PREFIX) {
                pos = (uint32_t)hl * (uint32_t)(CHAIN_PREFIX - 1) + (uint32_t)(sl - 1);    // A
            } else {
                uint32_t before = 0;               // C, level-major: the deeper levels of all heavy tiles below `sl`, then the
                for (uint32_t h = 0; h < nheavy; ++h) {                                    // tiles before this one on level `sl`
                    before += min(hv_n[h], (uint32_t)sl) - (uint32_t)CHAIN_PREFIX;
                    before += (h < (uint32_t)hl && hv_n[h] > (uint32_t)sl) ? 1u : 0u;
                }
                pos = size_a + size_b + before;
            }
        }
    }
    seg_off[seg_list_offset(ntiles, cap, 0) + pos] = b;
}

// ---- NEEDED-SEGMENT HINT.  Half of the segments lie behind the point where every pixel of their tile has stopped
// (a few vanishing-point tiles have up to ~950 segments and need at most ~200): k_seg_alpha computes their products
// for nothing, because which segments are needed is only known after the chain.  Training revisits every camera
// hundreds of times while the Gaussians move slowly, so the caller may hand back, as a HINT, the per-tile number of
// needed segments of its previous forward of the same camera (VrSaved.needed_hint; vr_export_needed).  k_seg_alpha
// then only computes the segments sl < hinted_limit(hint[tile]); k_seg_scan walks that prefix as before and, if any
// pixel of the tile is STILL alive at its end (the hint was too small: a stale or foreign hint, a scene that
// changed), the tile is finished by a second, equally parallel round over its remaining segments (k_seg_scan below) --
// so the result never depends on the hint, only the time does (bit-exact either way;
// tests/test_gpu_parity.py::test_needed_hint_*).  Margin: 2 segments + 12 %: 4.6 k instead of 2.1 k of the 9.5 k dead
// segments of the headline view are still computed, and a model drifting by 2 cm / +-0.3 opacity logits / +-5 % scales
// misses in 3-5 of 2064 tiles (profiles/tools/staleness.py).

// ---- chain mode (the segment table's comment): device-scope accesses of what workgroups of ONE launch hand each other.  The
// 8 XCD L2s are not coherent with each other: payload AND flag go through device-scope (sc1) stores / loads, which are
// performed at the level all XCDs share -- so no cache maintenance is needed, only ORDER: every wave waits for its own
// stores (s_waitcnt vmcnt(0): a written-through store is acknowledged from that level) before the workgroup's barrier, the
// flag is stored behind the barrier; the reader uses the payload only after it has seen the flag.  (The fences that say the
// same in the memory model -- release: buffer_wbl2, acquire: buffer_inv -- write back / drop the WHOLE L2 of the XCD, with
// every other workgroup's records and lists in it: 7 k of them per launch took k_seg_alpha from 170 to 364 us.)
__device__ __forceinline__ void chain_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float chain_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t chain_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every thread of the workgroup has stored its pixel's product: raise the segment's flag (all 256 threads call this)
__device__ __forceinline__ void chain_publish(uint32_t* __restrict__ flags, uint32_t seg)
{
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): this wave's device-scope stores have been performed
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&flags[seg], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Product of (1 - alpha) over one segment for the calling thread's pixel; also builds the segment's strip-relevance
// masks and stores them.  Called by all 256 threads of a workgroup (contains block barriers).
template <bool FAST>
__device__ __forceinline__ float seg_alpha_body(const SegCtx& c, const uint32_t* __restrict__ point_list,
                                                const Splat* __restrict__ rec, float4 (*lds)[SEG],
                                                unsigned long long* masks, unsigned long long* __restrict__ segmask, void* qspace)
{
    {
        const bool have = (int)threadIdx.x < c.count;
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
        if (have) {
            const float4* src = reinterpret_cast<const float4*>(rec + point_list[c.first + threadIdx.x]);
            q0 = src[0];
            q1 = src[1];
            // staged for the loop with the conic and the threshold in log2 units (splat_k2): x y kA kB | kC opacity thr2 depth
            float4 s0 = q0, s1 = q1;
            splat_k2(q0.z, q0.w, q1.x, q1.z, s0.z, s0.w, s1.x, s1.z);
            lds[0][threadIdx.x] = s0;
            lds[1][threadIdx.x] = s1;
        }
        seg_build_masks(c, have, q0, q1, masks);
    }
    __syncthreads();
    if (threadIdx.x < 16) segmask[(size_t)c.seg * 16 + threadIdx.x] = masks[threadIdx.x];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float pxf = (float)c.px, pyf = (float)c.py;
    float p = 1.0f;
    // ---- QUARTER-WAVE LISTS (round 4).  An entry relevant to the 8 x 8 strip reaches 1.9 of its four 4 x 4 quadrants on
    // average: every quadrant (16 lanes) walks its OWN list of the entries that can reach it, the wave makes as many trips as
    // the longest of the four lists -- 36 % fewer than with one list per strip (profiles/tools/pairstats.py), for three more
    // instructions per trip and ~250 per wave to build the lists (the strip's entries compacted first, then four
    // rectangle tests per entry, two facing edges each).  An entry left out of a quadrant's list has alpha < 1/255 at every
    // one of its pixels: a no-op there, the products are bit for bit what they were.  Four entries per trip: their LDS
    // reads are issued before any of them is used; the entries are still applied strictly in list order.
    // (render forward 0.360 -> 0.340 ms.  Not in seg_first_body -- 5 KB more of LDS would cost it a workgroup per CU, and the
    // near splats of a first segment reach most quadrants anyway -- nor in k_seg_blend, where per-chunk lists bought nothing.)
    // List stride 260 bytes, not 256 (round 6): the four quadrants read their lists' dword t / 4 in ONE ds_read_b32, and lists
    // 256 bytes apart put the four addresses on one bank (a 4-way conflict per four entries: part of the kernel's
    // SQ_LDS_BANK_CONFLICT count, profiles/experiments/README.md); 65 dwords apart they fall on four consecutive banks.
    constexpr int QL_STRIDE = 260;
    uint8_t* const ql = reinterpret_cast<uint8_t*>(qspace) + w * (4 * QL_STRIDE);     // 4 lists of up to 256 entry indices
    uint8_t* const rel = reinterpret_cast<uint8_t*>(qspace) + 4 * (4 * QL_STRIDE) + w * 256; // the strip's relevant entries, compacted
    int n0 = 0, n1 = 0, n2 = 0, n3 = 0;
    {
        const unsigned long long lt = (1ull << lane) - 1ull;
        int nrel = 0;
#pragma unroll
        for (int part = 0; part < 4; ++part) {
            const unsigned long long m = uniform64(masks[w * 4 + part]);
            if ((m >> lane) & 1ull) rel[nrel + __popcll(m & lt)] = (uint8_t)(part * 64 + lane);
            nrel += __popcll(m);
        }
        const float sx0 = c.x0 + (float)((w % REGIONS_X) * REGION_W), sy0 = c.y0 + (float)((w / REGIONS_X) * REGION_H);
        for (int base = 0; base < nrel; base += 64) {
            const bool have = base + lane < nrel;
            const int j = have ? (int)rel[base + lane] : 0;
            const float4 a = lds[0][j], b = lds[1][j];        // x y kA kB | kC opacity thr2 depth
            // -power2 = A' dx^2 + 2 B' dx dy + C' dy^2 with A' = -kA, B' = -kB / 2, C' = -kC; the entry reaches a quadrant iff
            // the minimum over the quadrant's pixel centres is <= -thr2 (with the strip test's slack)
            const float A = -a.z, B = -0.5f * a.w, Cc = -b.x;
            const float lim = (-b.z) * 1.001f + 0.001f;
            const bool odd = !(A > 0.0f) || !(Cc > 0.0f) || !(A * Cc - B * B > 0.0f);
            const float inv_A = __builtin_amdgcn_rcpf(A), inv_C = __builtin_amdgcn_rcpf(Cc);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xl = sx0 + (float)((q & 1) * 4) - a.x, xh = xl + 3.0f;
                const float yl = sy0 + (float)((q >> 1) * 4) - a.y, yh = yl + 3.0f;
                const bool r = have && (odd || rect_relevant_facing(A, inv_A, B, Cc, inv_C, lim, xl, xh, yl, yh));
                const unsigned long long bm = __ballot(r);
                int& n = q == 0 ? n0 : (q == 1 ? n1 : (q == 2 ? n2 : n3));
                if (r) ql[q * QL_STRIDE + n + __popcll(bm & lt)] = (uint8_t)j;
                n += __popcll(bm);
            }
        }
    }
    {
        const int myq = ((lane >> 5) & 1) * 2 + ((lane >> 2) & 1);
        const int n_mine = myq == 0 ? n0 : (myq == 1 ? n1 : (myq == 2 ? n2 : n3));
        const int trips = max(max(n0, n1), max(n2, n3));
        const uint8_t* const myl = ql + myq * QL_STRIDE;
        auto apply = [&](const float4 a, const float4 b, const bool act) {
            float dx, dy;
            const float power = splat_power2(a.x, a.y, a.z, a.w, b.x, pxf, pyf, dx, dy);
            const bool pre = act && !(power > 0.0f) && power >= b.z;
            const float alpha = fminf(ALPHA_MAX, b.y * exp2_sel<FAST>(power));
            const bool valid = pre && !(alpha < ALPHA_MIN);
            p = valid ? p * (1.0f - alpha) : p;
        };
        for (int t = 0; t < trips; t += 4) {
            const uint32_t idx4 = *reinterpret_cast<const uint32_t*>(myl + t);
            float4 av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = (int)((idx4 >> (8 * u)) & 255u);
                av[u] = lds[0][k];
                bv[u] = lds[1][k];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (t + u < trips) apply(av[u], bv[u], t + u < n_mine);
        }
    }
    return p;
}

// the forward's per-pixel outputs (what k_seg_combine writes; a tile with ONE list segment is finished by seg_first_body)
struct FwdOut {
    float *color, *depth, *quat, *scale, *alpha, *final_T;
    uint32_t* n_contrib;
    float* dsum;
};

// ---- A + C for a tile's FIRST segment in one pass.  Its boundary transmittance is 1 by definition, so nothing has to
// wait for the chain: the workgroup builds the relevance masks and the local product like k_seg_alpha AND blends like
// k_seg_blend (same expressions, same order: `part`, last contributor and the stop test are bit for bit what the two
// kernels produced) -- alpha is evaluated once instead of twice for what are the most expensive segments of a view (every
// pixel alive), their records are gathered once, and k_seg_blend's list loses a fifth of its entries.  For a pixel whose
// stop test fires inside the segment the stored product is the one that fired it (< 1e-4: k_seg_scan takes the same
// decision as with the product over the whole segment, fp32 products by factors <= 1 being monotone).
// The first `ntiles` workgroups of k_seg_alpha<0>'s launch run this (one launch: the heavy first-segment workgroups start
// first and the plain ones fill in behind them; as a launch of its own the 2064 workgroups -- a single resident set --
// took 56 us, most of it waiting for the longest of them).
template <bool FAST>
__device__ __forceinline__ void seg_first_body(const Camera& cam, const int tile, const int2* __restrict__ ranges,
                                               const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ point_list,
                                               const Splat* __restrict__ rec, float* __restrict__ Pbuf,
                                               unsigned long long* __restrict__ segmask, float* __restrict__ part,
                                               float4 (*lds)[SEG], float2* lds_s, unsigned long long* masks, const FwdOut& o,
                                               uint32_t* __restrict__ chain_flags)
{
    const uint32_t seg0 = seg_off[tile];
    if (seg_off[tile + 1] == seg0) return;                     // empty tile
    const bool single = seg_off[tile + 1] - seg0 == 1u;        // (half of a street view's tiles: finished here, below)
    // chain mode, heavy tile: the product is published for the tile's walker (chain_flags: this view's flags, or null)
    const bool publish = chain_flags != nullptr && seg_off[tile + 1] - seg0 > (uint32_t)CHAIN_PREFIX;
    SegCtx c;
    if (!seg_setup_at(cam, ranges, seg_off, seg0, threadIdx.x >> 6, c)) return;
    {
        const bool have = (int)threadIdx.x < c.count;
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
        if (have) {
            const float4* src = reinterpret_cast<const float4*>(rec + point_list[c.first + threadIdx.x]);
            q0 = src[0];
            q1 = src[1];
            float4 s0 = q0, s1 = q1;       // x y kA kB | kC opacity thr2 depth (splat_k2: log2 units)
            splat_k2(q0.z, q0.w, q1.x, q1.z, s0.z, s0.w, s1.x, s1.z);
            lds[0][threadIdx.x] = s0;
            lds[1][threadIdx.x] = s1;
            lds[2][threadIdx.x] = src[2];  // r g b qw
            lds[3][threadIdx.x] = src[3];  // qx qy qz s0
            const float4 q4 = src[4];      // s1 s2 (clamp bits, pad)
            lds_s[threadIdx.x] = make_float2(q4.x, q4.y);
        }
        seg_build_masks(c, have, q0, q1, masks);
    }
    __syncthreads();
    if (threadIdx.x < 16) segmask[(size_t)c.seg * 16 + threadIdx.x] = masks[threadIdx.x];
    const int w = threadIdx.x >> 6;
    const float pyf = (float)c.py;
    constexpr float GATED = 1e30f;            // (k_seg_blend: a finished pixel is moved out of reach)
    constexpr float Tb = 1.0f;
    float gx = c.inside ? (float)c.px : GATED;
    float p = 1.0f, pstop = 1.0f;
    f2 Cp[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) Cp[k] = f2_splat(0.0f);
    float Cd = 0.0f;
    int lastk = -1;
    auto blend_one = [&](const int i, const float4 a, const float4 b) {
        float dx, dy;
        const float power = splat_power2(a.x, a.y, a.z, a.w, b.x, gx, pyf, dx, dy);
        const bool pre = !(power > 0.0f) && power >= b.z;
        if (__builtin_amdgcn_ballot_w64(pre) == 0ull) return;
        const float alpha = fminf(ALPHA_MAX, b.y * exp2_sel<FAST>(power));
        const bool valid = pre && !(alpha < ALPHA_MIN);
        const float pn = p * (1.0f - alpha);
        const bool stop = valid && (Tb * pn < T_EPS);
        const bool apply = valid && !stop;
        gx = stop ? GATED : gx;
        pstop = stop ? pn : pstop;
        // (no "nobody applies it" exit here any more -- round 6: the ballot of `valid && !stop` is lowered through a vector
        // register, v_cndmask + v_cmp + a branch on EVERY entry that reaches this point, to skip a block that the rare entry
        // nobody applies walks with wgt = 0: exact no-ops)
        const float wgt = apply ? alpha * (Tb * p) : 0.0f;
        const float4 cc = lds[2][i];
        const float4 d = lds[3][i];
        const float2 e2 = lds_s[i];
        const f2 w2 = f2_splat(wgt);
        Cp[0] = f2_fma((f2){cc.x, cc.y}, w2, Cp[0]);
        Cp[1] = f2_fma((f2){cc.z, cc.w}, w2, Cp[1]);
        Cp[2] = f2_fma((f2){d.x, d.y}, w2, Cp[2]);
        Cp[3] = f2_fma((f2){d.z, d.w}, w2, Cp[3]);
        Cp[4] = f2_fma((f2){e2.x, e2.y}, w2, Cp[4]);
        Cd = fmaf(b.w, wgt, Cd);
        p = apply ? pn : p;
        lastk = apply ? i : lastk;
    };
    for (int part_i = 0; part_i < 4; ++part_i) {
        unsigned long long m = uniform64(masks[w * 4 + part_i]);
        while (m) {
            constexpr int NB = 4;
            int kk[NB];
            float4 av[NB], bv[NB];
            int nb = 0;
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                if (m) {
                    kk[t] = part_i * 64 + __builtin_ctzll(m);
                    m &= m - 1;
                    nb = t + 1;
                } else {
                    kk[t] = kk[0];
                }
                av[t] = lds[0][kk[t]];
                bv[t] = lds[1][kk[t]];
            }
#pragma unroll
            for (int t = 0; t < NB; ++t)
                if (t < nb) blend_one(kk[t], av[t], bv[t]);
        }
        if (__builtin_amdgcn_ballot_w64(gx != GATED) == 0ull) break;     // every pixel of the strip is finished
    }
    const bool stopped = c.inside && gx == GATED;
    if (publish) chain_store(&Pbuf[(size_t)c.seg * SEG + threadIdx.x], stopped ? pstop : p);
    else Pbuf[(size_t)c.seg * SEG + threadIdx.x] = stopped ? pstop : p;
    if (c.inside) {
        const float Cs[NCH] = {Cp[0].x, Cp[0].y, Cp[1].x, Cd, Cp[1].y, Cp[2].x, Cp[2].y, Cp[3].x, Cp[3].y, Cp[4].x, Cp[4].y};
        const uint32_t last = lastk >= 0 ? (uint32_t)(lastk + 1) : 0u;       // (sl = 0: tile-relative index + 1)
        if (single) {
            // The tile's ONLY segment: its sums are the pixel's sums -- the images are written here, with k_seg_combine's
            // operations in k_seg_combine's order (0 + sum, Tb * product with Tb = 1), and the tile's 13 `part` planes
            // are neither written nor read back (k_seg_combine skips the tile).
            const size_t N = (size_t)cam.H * cam.W, pix = c.pix;
            const float T = Tb * p;
            float C[NCH];
#pragma unroll
            for (int k = 0; k < NCH; ++k) C[k] = 0.0f + Cs[k];
            o.final_T[pix] = T;
            o.n_contrib[pix] = last;
            o.color[pix] = fmaf(T, cam.bg[0], C[0]);
            o.color[N + pix] = fmaf(T, cam.bg[1], C[1]);
            o.color[2 * N + pix] = fmaf(T, cam.bg[2], C[2]);
            o.alpha[pix] = 1.0f - T;
            float depth_out = C[3];
            if (cam.flags & FLAG_DEPTH_NORMALIZED) {
                const float A = 1.0f - T;
                o.dsum[pix] = C[3];
                depth_out = A > 0.0f ? C[3] / A : 0.0f;
            }
            o.depth[pix] = depth_out;
            if (cam.flags & FLAG_FILL_EMPTY) C[4] += T;
#pragma unroll
            for (int k = 0; k < 4; ++k) o.quat[k * N + pix] = C[4 + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) o.scale[k * N + pix] = C[8 + k];
        } else {
            float* dst = part + (size_t)c.seg * (NPART * SEG) + threadIdx.x;
#pragma unroll
            for (int k = 0; k < NCH; ++k) dst[k * SEG] = Cs[k];
            dst[11 * SEG] = p;
            dst[12 * SEG] = __uint_as_float(last | (stopped ? 0x80000000u : 0u));
        }
    }
    if (publish) chain_publish(chain_flags, c.seg);
}

// Window of a short tile's list that catch-up round ROUND (1, 2) of a hinted forward covers, given where the previous
// round stopped (`lo`, from seg_needed) and the tile's segment count: round 1 looks half as far again (+ 8 segments) --
// after an epoch of training the heavy tiles' counts have moved by up to +-50 % (profiles/tools/epoch_drift.py: 55 short
// tiles per view, 429 segments beyond their limits, out of ~10 k needed) --, round 2 takes whatever is left.
constexpr uint32_t TILE_SHORT = 0x80000000u;
__device__ __forceinline__ uint32_t catchup_end(int round, uint32_t lo, uint32_t nseg, uint32_t second)
{
    return round == 1 ? min(nseg, lo + max(8u + (lo >> 1), second)) : nseg;
}

// The tile's chain has ended: `needed` segments are needed by some pixel, the calling wave's pixels by `mine` of them.
// (k_seg_scan's last step, also taken by the walkers of chain mode; all 256 threads; wbase: one word of LDS.)
__device__ __forceinline__ void seg_tile_finish(const Camera& cam, uint32_t* __restrict__ seg_off, uint32_t cap, int tile,
                                                uint32_t s0, uint32_t s1, uint32_t mine, uint32_t needed, float* __restrict__ Tbuf,
                                                uint32_t* __restrict__ seg_needed, uint32_t* __restrict__ hint, uint32_t* wbase,
                                                const bool enqueue = true)
{
    for (uint32_t s = s0 + mine; s < s0 + needed; ++s) Tbuf[(size_t)s * SEG + threadIdx.x] = -1.0f;
    // the tile's needed segments go onto the list of its dispatch queue, in the order in which tiles finish (vr_segment.h)
    // (enqueue = false, a walker of chain mode: k_seg_merge puts the walked tiles' segments BEHIND the queues' -- where the
    // heavy tiles end up when k_seg_scan walks them, being the last to finish: the order k_seg_blend / k_seg_bwd want)
    const int ntiles_all = cam.gx * cam.gy, queue = (int)(blockIdx.x % SEG_QUEUES);
    if (threadIdx.x == 0) {
        seg_needed[tile] = needed;
        if (hint) hint[tile] = needed;     // in/out: what this forward needed is the hint of the camera's next visit
        *wbase = enqueue ? atomicAdd(&seg_off[seg_qcount_offset(ntiles_all, cap, queue)], needed) : 0u;
        if (needed >= HEAVY_TILE) {      // long chains start first in the per-tile kernels that follow (vr_segment.h)
            const uint32_t hp = atomicAdd(&seg_off[seg_counts_offset(ntiles_all, cap) + SEG_COUNT_HEAVY], 1u);
            seg_off[seg_actoff_offset(ntiles_all, cap) + hp] = (uint32_t)tile;     // (act_off's space: free after k_seg_tiles)
        }
    }
    __syncthreads();
    if (enqueue) {
        uint32_t* const ql = seg_off + seg_qlist_offset(ntiles_all, cap, queue) + *wbase;
        for (uint32_t k = threadIdx.x; k < needed; k += 256) ql[k] = s0 + k;
    }
    // per-segment flag for the segment kernels (vr_segment.h): 0 not needed / 1 needed, last / 2 needed, next too
    int4* seg_info = reinterpret_cast<int4*>(seg_off + seg_tile_offset(ntiles_all));
    for (uint32_t k = threadIdx.x; k < s1 - s0; k += 256) {
        const uint32_t flag = k >= needed ? 0u : (k + 1 < needed ? 2u : 1u);
        seg_info[s0 + k].w = (int)(k | (flag << 30));
    }
}

// ---- chain mode: the WALKER of one heavy tile (one workgroup of k_seg_alpha<0>'s launch; the segment table's comment).
// k_seg_scan<0>'s walk -- the same multiplications in the same order, the same Tbuf rows -- over products that are still
// being computed: each wave polls the flags of the next WALK segments and takes as many as are published in a row.  When
// the chain ends it publishes the tile's needed-segment count first (`dead`: workgroups of segments behind it return at
// once), then finishes the tile like k_seg_scan.  The wait is bounded: after CHAIN_MAX_POLLS polls without a new segment
// (a producer that is not scheduled: another process holding the GPU) the walker leaves and k_seg_scan walks the tile
// behind the launch, as without chain mode (nothing has been skipped then: `dead` was never set).
constexpr int CHAIN_MAX_POLLS = 20000;         // x (~1 us sleep + a device-scope load): tens of milliseconds
__device__ __forceinline__ void seg_chain_walk(const Camera& cam, int tile, uint32_t* __restrict__ seg_off, uint32_t cap,
                                               const float* __restrict__ Pbuf, float* __restrict__ Tbuf,
                                               uint32_t* __restrict__ seg_needed, uint32_t* wsh, const int patience)
{
    constexpr int WALK = 16;
    uint32_t* const wneed = wsh, * const wgave = wsh + 4, * const wbase = wsh + 8;
    const int ntiles = cam.gx * cam.gy;
    const uint32_t* const flags = seg_off + chain_flags_offset(ntiles, cap);
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int px = tx * TILE + region_x(w, lane), py = ty * TILE + region_y(w, lane);
    const uint32_t s0 = seg_off[tile], s1 = seg_off[tile + 1];
    bool alive = px < cam.W && py < cam.H;
    float Tb = 1.0f;
    uint32_t mine = 0;
    bool wave_alive = __ballot(alive) != 0ull, gave_up = false;
    uint32_t s = s0;
    while (s < s1 && wave_alive) {
        int r = 0;
        for (int polls = 0;; ++polls) {
            uint32_t f = 1u;
            if (lane < WALK && s + lane < s1) f = chain_load(&flags[s + lane]);
            const unsigned long long missing = __ballot(f == 0u);
            r = missing ? (int)__builtin_ctzll(missing) : WALK;
            if (r > 0) break;
            if (polls >= patience) { gave_up = true; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (gave_up) break;
        r = min(min(r, WALK), (int)(s1 - s));
        __atomic_signal_fence(__ATOMIC_SEQ_CST);       // (compiler only: the products are loaded after the flags were seen)
        float Pv[WALK];
#pragma unroll
        for (int k = 0; k < WALK; ++k) Pv[k] = k < r ? chain_load(&Pbuf[(size_t)(s + k) * SEG + threadIdx.x]) : 1.0f;
#pragma unroll
        for (int k = 0; k < WALK; ++k) {
            if (k < r && wave_alive) {
                mine = s + k - s0 + 1;
                Tbuf[(size_t)(s + k) * SEG + threadIdx.x] = alive ? Tb : -1.0f;
                const float Tn = Tb * Pv[k];
                if (alive && Tn < T_EPS) alive = false;  // the stop test fires inside this segment
                else if (alive) Tb = Tn;
                wave_alive = __ballot(alive) != 0ull;
            }
        }
        s += (uint32_t)r;
    }
    if (lane == 0) { wneed[w] = mine; wgave[w] = gave_up ? 1u : 0u; }
    __syncthreads();
    if ((wgave[0] | wgave[1] | wgave[2] | wgave[3]) != 0u) return;
    const uint32_t needed = max(max(wneed[0], wneed[1]), max(wneed[2], wneed[3]));
    if (threadIdx.x == 0)
        __hip_atomic_store(seg_off + chain_dead_offset(ntiles, cap) + tile, needed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    seg_tile_finish(cam, seg_off, cap, tile, s0, s1, mine, needed, Tbuf, seg_needed, nullptr, wbase, false);
}

// ---- A: per (tile, segment, pixel) product of (1 - alpha).  ROUND 0 = the segments inside the hinted prefix of their
// tile (all of them without a hint); ROUND 1, 2 = the catch-up rounds for the short tiles of a hinted forward (see
// k_seg_scan): the segments still flagged 3 that fall into the round's window.
template <int ROUND, bool FAST, bool CHAIN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))      // (chain mode: 102 SGPRs without it, 7 waves)
k_seg_alpha(Camera cam, const int2* __restrict__ ranges, uint32_t* __restrict__ seg_off, uint32_t cap,
            const uint32_t* __restrict__ point_list, const Splat* __restrict__ rec, float* __restrict__ Pbuf,
            unsigned long long* __restrict__ segmask, float* __restrict__ part, int first_fused, FwdOut fwd_out,
            float* __restrict__ Tbuf, uint32_t* __restrict__ seg_needed, int chain_patience)
{
    // (18.5 KB in round 0: eight workgroups per CU, as with the 8 KB of the plain path alone)
    __shared__ float4 lds[ROUND == 0 ? 4 : 2][SEG];
    __shared__ float2 lds_s[ROUND == 0 ? SEG : 1];
    __shared__ unsigned long long masks[16];
    __shared__ uint32_t qextra[ROUND == 0 ? 1 : 1296];                  // (quarter-wave lists: round 0 borrows the fused path's planes)
    void* const qspace = ROUND == 0 ? (void*)&lds[ROUND == 0 ? 2 : 0][0] : (void*)qextra;
    // the round's work list: round 0 one workgroup per entry (the grid is sized for it: AUTO_FIRST segments per tile
    // without a hint); the catch-up rounds a fixed grid striding over a list whose length only the device knows
    const int ntiles = cam.gx * cam.gy;
    // chain mode (CHAIN: round 0 of a forward without rounds; the segment table's comment): this view's heavy tiles
    const uint32_t nheavy = CHAIN ? seg_off[seg_counts_offset(ntiles, cap) + SEG_COUNT_CHAIN_HEAVY] : 0u;
    uint32_t* const chain_flags = (CHAIN && nheavy) ? seg_off + chain_flags_offset(ntiles, cap) : nullptr;
    const uint32_t count = seg_off[seg_counts_offset(ntiles, cap) + ROUND];
    const uint32_t* const list = seg_off + seg_list_offset(ntiles, cap, ROUND);
    uint32_t item0 = blockIdx.x - (ROUND == 0 && first_fused ? (uint32_t)ntiles : 0u);
    bool publish = false;
    if (CHAIN && nheavy) {
        // the launch (the segment table's comment): heavy tiles' first segments | region A | CHAIN_MAX_HEAVY walkers | the other
        // tiles' first segments | regions B, C
        const uint32_t size_a = seg_off[seg_counts_offset(ntiles, cap) + SEG_COUNT_CHAIN_A];
        const uint32_t size_b = seg_off[seg_counts_offset(ntiles, cap) + SEG_COUNT_CHAIN_B];
        const uint32_t* const order = seg_off + chain_heavy_offset(ntiles, cap);
        uint32_t b = blockIdx.x;
        int first_of = -1;
        if (b < nheavy) first_of = (int)order[b];
        else if ((b -= nheavy) < size_a) item0 = b;
        else if ((b -= size_a) < (uint32_t)CHAIN_MAX_HEAVY) {
            if (b < nheavy) {
                uint32_t* const wsh = reinterpret_cast<uint32_t*>(masks);      // (16 x u64 of LDS nobody else uses here)
                seg_chain_walk(cam, (int)order[b], seg_off, cap, Pbuf, Tbuf, seg_needed, wsh, chain_patience);
            }
            return;
        }
        else if ((b -= (uint32_t)CHAIN_MAX_HEAVY) < (uint32_t)ntiles - nheavy) first_of = (int)order[nheavy + b];
        else item0 = size_a + (b - ((uint32_t)ntiles - nheavy));
        if (first_of >= 0) {
            seg_first_body<FAST>(cam, first_of, ranges, seg_off, point_list, rec, Pbuf, segmask, part, lds, lds_s, masks, fwd_out, chain_flags);
            return;
        }
        publish = item0 < size_a || item0 >= size_a + size_b;
    } else if (ROUND == 0 && first_fused && (int)blockIdx.x < ntiles) {      // the tiles' FIRST segments: alpha and blend in one pass
        seg_first_body<FAST>(cam, (int)blockIdx.x, ranges, seg_off, point_list, rec, Pbuf, segmask, part, lds, lds_s, masks, fwd_out,
                             chain_flags);
        return;
    }
    for (uint32_t item = item0; item < count; item += gridDim.x) {
        SegCtx c;
        if (seg_setup_at(cam, ranges, seg_off, list[item], threadIdx.x >> 6, c)) {
            if (CHAIN && publish) {
                // a segment behind the end of its tile's chain: nobody needs it (all four waves take the same decision)
                uint32_t* const dsh = reinterpret_cast<uint32_t*>(masks);          // (free until seg_build_masks)
                if (threadIdx.x == 0) dsh[0] = chain_load(seg_off + chain_dead_offset(ntiles, cap) + c.tile);
                __syncthreads();
                const uint32_t dead = dsh[0];
                __syncthreads();
                // (NOT counted in a word next to the launch's counts: one atomic per skipped workgroup on the line that every
                // workgroup reads first took the kernel from 161 to 234 us)
                if ((uint32_t)c.sl >= dead) return;
            }
            const float p = seg_alpha_body<FAST>(c, point_list, rec, lds, masks, segmask, qspace);
            if (CHAIN && publish) {
                chain_store(&Pbuf[(size_t)c.seg * SEG + threadIdx.x], p);
                chain_publish(chain_flags, c.seg);
            } else {
                Pbuf[(size_t)c.seg * SEG + threadIdx.x] = p;
            }
        }
        if (ROUND == 0) break;       // (grid >= count in round 0)
        __syncthreads();             // the staging buffers are reused by the next entry
    }
}

// ---- B: per tile, boundary transmittances.  Tbuf[seg][pix] = Tb at the segment start, or -1 when
// the pixel is finished before that segment; seg_needed[tile] = number of segments any pixel needs.
//
// With a needed-segment hint k_seg_alpha computed only the first `limit` segments of the tile.  If a pixel is still
// alive at the end of that prefix the tile is SHORT (a stale or foreign hint, a scene that changed): this launch
// (ROUND 0) then leaves the tile unfinished -- it parks every pixel's state in the Tbuf row of the first missing
// segment and marks the tile (top bit of seg_needed) -- and up to two catch-up rounds finish it IN PARALLEL:
// k_seg_alpha<ROUND> runs once more over the window of segments catchup_end() assigns to the round (only short tiles
// have segments flagged 3 left), then this kernel with the same ROUND picks the chains of the short tiles up where they
// stopped; round 1 may leave a tile short again, round 2 walks to the end of the list.  (Round 2 computed the missing segments inside this kernel, one
// after the other per tile: ~6 us each, and a model that had trained for an epoch since the hint was recorded missed
// by hundreds of segments in the heavy tiles -- 2.98 ms per view instead of 1.42 without hints.)  The result never
// depends on the hint either way.
template <int ROUND>
__global__ void __launch_bounds__(256)
k_seg_scan(Camera cam, uint32_t* __restrict__ seg_off, uint32_t cap, uint32_t second, const float* __restrict__ Pbuf,
           float* __restrict__ Tbuf, uint32_t* __restrict__ seg_needed, uint32_t* __restrict__ hint, int chain)
{
    constexpr bool PASS2 = ROUND > 0;
    __shared__ uint32_t wneed[4];
    __shared__ uint32_t walive[4];
    __shared__ uint32_t wbase;
    const int tile = blockIdx.x;
    // chain mode: the tile's walker has done all of this inside k_seg_alpha's launch
    if (ROUND == 0 && chain && seg_off[chain_dead_offset(cam.gx * cam.gy, cap) + tile] != CHAIN_OPEN) return;
    const uint32_t lim_word = seg_needed[tile];   // k_seg_offsets' snapshot (ROUND > 0: what the previous round left)
    if (PASS2 && !(lim_word & TILE_SHORT)) return;
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int px = tx * TILE + region_x(w, lane), py = ty * TILE + region_y(w, lane);
    const uint32_t s0 = seg_off[tile], s1 = seg_off[tile + 1];
    // segments [first, last) of the tile are walked by this launch
    const uint32_t first = PASS2 ? (lim_word & ~TILE_SHORT) : 0u;
    const uint32_t last = PASS2 ? catchup_end(ROUND, first, s1 - s0, second) : min(s1 - s0, lim_word);
    bool alive = px < cam.W && py < cam.H;
    float Tb = 1.0f;
    if (PASS2) {   // the state parked by the first pass
        Tb = Tbuf[(size_t)(s0 + first) * SEG + threadIdx.x];
        alive = !(Tb < 0.0f);
    }
    // Each wave walks the segment chain of its own 64 pixels without block barriers; the P values of
    // UNROLL segments are fetched together so the chain is not bound by one memory latency per segment.
    constexpr int UNROLL = 24;
    uint32_t mine = first;  // segments this wave needs (some pixel alive at the segment start)
    bool wave_alive = __ballot(alive) != 0ull;
    const uint32_t sa = s0 + first, sb = s0 + last;
    for (uint32_t s = sa; s < sb && wave_alive; s += UNROLL) {
        float Pv[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k)
            Pv[k] = (s + k < sb) ? Pbuf[(size_t)(s + k) * SEG + threadIdx.x] : 1.0f;
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            if (s + k < sb && wave_alive) {
                mine = s + k - s0 + 1;
                Tbuf[(size_t)(s + k) * SEG + threadIdx.x] = alive ? Tb : -1.0f;
                const float Tn = Tb * Pv[k];
                if (alive && Tn < T_EPS) alive = false;  // the stop test fires inside this segment
                else if (alive) Tb = Tn;
                wave_alive = __ballot(alive) != 0ull;
            }
        }
    }
    if (lane == 0) { wneed[w] = mine; walive[w] = wave_alive ? 1u : 0u; }
    __syncthreads();
    const bool any_alive = (walive[0] | walive[1] | walive[2] | walive[3]) != 0u;
    if (ROUND < 2 && any_alive && last < s1 - s0) {
        // SHORT tile: rows [mine, last) of the waves that finished earlier, then every pixel's state in row `last`
        for (uint32_t s = s0 + mine; s < s0 + last; ++s) Tbuf[(size_t)s * SEG + threadIdx.x] = -1.0f;
        Tbuf[(size_t)(s0 + last) * SEG + threadIdx.x] = alive ? Tb : -1.0f;
        // the next round's window of this tile goes onto that round's work list (order among tiles: whoever comes first)
        const uint32_t wend = catchup_end(ROUND + 1, last, s1 - s0, second);
        const int ntiles = cam.gx * cam.gy;
        if (threadIdx.x == 0) {
            seg_needed[tile] = last | TILE_SHORT;
            wbase = atomicAdd(&seg_off[seg_counts_offset(ntiles, cap) + ROUND + 1], wend - last);
        }
        __syncthreads();
        uint32_t* const next_list = seg_off + seg_list_offset(ntiles, cap, ROUND + 1) + wbase;
        for (uint32_t k = threadIdx.x; k < wend - last; k += 256) next_list[k] = s0 + last + k;
        return;   // the segment flags of this tile stay as they are (3 behind the prefix)
    }
    const uint32_t needed = max(max(wneed[0], wneed[1]), max(wneed[2], wneed[3]));
    seg_tile_finish(cam, seg_off, cap, tile, s0, s1, mine, needed, Tbuf, seg_needed, hint, &wbase);
}

// The eight queue lists dealt round-robin into the needed list: entry k of queue q goes behind the entries < k of every
// queue and the entries k of the queues before q.  One thread per (queue, k) slot.
// Chain mode (slice SEG_QUEUES of the grid): the needed segments of the tiles whose walkers finished go BEHIND the queues'
// entries, level-major (every walked tile's segment k before any tile's segment k + 1: a tile's front segments -- every pixel
// alive, the expensive ones -- first, consecutive workgroups on different tiles; vr_segment.h on why the order matters).
__global__ void __launch_bounds__(256)
k_seg_merge(int ntiles, uint32_t* __restrict__ seg_off, uint32_t cap, const uint32_t* __restrict__ seg_needed, int chain)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    const int q = (int)blockIdx.y;
    uint32_t len[SEG_QUEUES], total = 0;
#pragma unroll
    for (int i = 0; i < SEG_QUEUES; ++i) { len[i] = seg_off[seg_qcount_offset(ntiles, cap, i)]; total += len[i]; }
    const uint32_t nheavy = chain ? seg_off[seg_counts_offset(ntiles, cap) + SEG_COUNT_CHAIN_HEAVY] : 0u;
    if (q == SEG_QUEUES) {
        __shared__ uint32_t wn[CHAIN_MAX_HEAVY], wfirst[CHAIN_MAX_HEAVY], wpre[CHAIN_MAX_HEAVY + 1];
        if (nheavy == 0u) return;
        if (threadIdx.x < CHAIN_MAX_HEAVY) {
            uint32_t n = 0, f = 0;
            if (threadIdx.x < nheavy) {
                const uint32_t t = seg_off[chain_heavy_offset(ntiles, cap) + threadIdx.x];
                if (seg_off[chain_dead_offset(ntiles, cap) + t] != CHAIN_OPEN) n = seg_needed[t];    // (else: k_seg_scan walked and enqueued it)
                f = seg_off[t];
            }
            wn[threadIdx.x] = n;
            wfirst[threadIdx.x] = f;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (uint32_t h = 0; h < nheavy; ++h) { wpre[h] = run; run += wn[h]; }
            wpre[nheavy] = run;
        }
        __syncthreads();
        const uint32_t late = wpre[nheavy];
        if (k == 0) seg_off[seg_counts_offset(ntiles, cap) + SEG_LIST_NEEDED] = total + late;
        if (k >= late) return;
        uint32_t h = 0;                              // entry k in tile-major order: walked tile h, its segment `lvl`
        while (h + 1 < nheavy && wpre[h + 1] <= k) ++h;
        const uint32_t lvl = k - wpre[h];
        uint32_t before = 0;
        for (uint32_t j = 0; j < nheavy; ++j) before += min(wn[j], lvl) + ((j < h && wn[j] > lvl) ? 1u : 0u);
        seg_off[seg_list_offset(ntiles, cap, SEG_LIST_NEEDED) + total + before] = wfirst[h] + lvl;
        return;
    }
    if (k == 0 && q == 0 && nheavy == 0u) seg_off[seg_counts_offset(ntiles, cap) + SEG_LIST_NEEDED] = total;
    if (k >= len[q]) return;
    uint32_t pos = 0;
#pragma unroll
    for (int i = 0; i < SEG_QUEUES; ++i) pos += min(len[i], k) + ((i < q && len[i] > k) ? 1u : 0u);
    seg_off[seg_list_offset(ntiles, cap, SEG_LIST_NEEDED) + pos] = seg_off[seg_qlist_offset(ntiles, cap, q) + k];
}

// ---- C: blend one segment from its boundary transmittance into segment-local sums.
// part[seg][k][pix], k = 0..10 channel sums, 11 = local product p, 12 = local last-contributor | done<<31 (NPART, above)

template <bool FAST>
__global__ void __launch_bounds__(64)
k_seg_blend(Camera cam, const int2* __restrict__ ranges, const uint32_t* __restrict__ seg_off, uint32_t cap,
            const uint32_t* __restrict__ seg_needed, const uint32_t* __restrict__ point_list,
            const Splat* __restrict__ rec, const float* __restrict__ Tbuf, float* __restrict__ part,
            const unsigned long long* __restrict__ segmask, int first_fused)
{
    // ONE WAVE (8x8 region = "strip") PER WORKGROUP, like k_seg_bwd: the four strips of a segment see very different numbers
    // of relevant entries and live pixels; as independent 64-thread workgroups they are scheduled and retire
    // individually (0.229 -> 0.205 ms).  The strip's relevant entries are staged 64 at a time.  (k_seg_alpha stays a
    // 256-thread workgroup: its relevance test is shared by the four strips, per-strip workgroups repeat it 4x.)
    __shared__ float4 lds[5][64];
    __shared__ unsigned char rel_j[SEG];
    __shared__ uint32_t rel_gid[SEG];
    SegCtx c;
    const int w = (int)(blockIdx.x & 3u);
    const int ntiles = cam.gx * cam.gy;
    if ((blockIdx.x >> 2) >= seg_off[seg_counts_offset(ntiles, cap) + SEG_LIST_NEEDED]) return;    // beyond the needed list
    if (!seg_setup_at(cam, ranges, seg_off, seg_off[seg_list_offset(ntiles, cap, SEG_LIST_NEEDED) + (blockIdx.x >> 2)], w, c)) return;
    if (c.sl == 0 && first_fused) return;                // a tile's first segment was blended by k_seg_first
    const int lane = threadIdx.x;
    const int pixslot = w * 64 + lane;
    // one batch of independent loads right after the segment descriptor (boundary transmittance, relevance masks,
    // all four 64-entry parts of the segment's list): one memory round trip instead of three dependent ones
    const float Tb = Tbuf[(size_t)c.seg * SEG + pixslot];
    const unsigned long long* masks = segmask + (size_t)c.seg * 16 + w * 4;
    const unsigned long long mraw[4] = {masks[0], masks[1], masks[2], masks[3]};
    uint32_t gid_q[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) gid_q[q] = q * 64 + lane < c.count ? point_list[c.first + q * 64 + lane] : 0u;
    const bool done = Tb < 0.0f;          // pixel finished before this segment
    if (__ballot(!done) == 0ull) return;  // nothing alive in this strip
    const unsigned long long mm[4] = {uniform64(mraw[0]), uniform64(mraw[1]), uniform64(mraw[2]), uniform64(mraw[3])};
    int nrel = 0;
    {
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if ((mm[q] >> lane) & 1ull) {
                const int pos = nrel + __popcll(mm[q] & lt);
                rel_j[pos] = (unsigned char)(q * 64 + lane);
                rel_gid[pos] = gid_q[q];
            }
            nrel += __popcll(mm[q]);
        }
    }
    __syncthreads();
    const float pyf = (float)c.py;
    // A FINISHED pixel (before this segment, or from the entry at which its stop test fires) is switched off by moving
    // it out of reach: with x = 1e30 every exponent is -inf (or NaN), never >= a threshold, so `pre` is false for the
    // pixel from then on -- no per-entry test of a loop-carried flag, which the compiler kept as a 0/1 register with
    // four conversions per entry.  A switched-off pixel changes nothing, so the results are what they were.
    constexpr float GATED = 1e30f;
    float gx = done ? GATED : (float)c.px;
    float p = 1.0f;
    f2 Cp[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) Cp[k] = f2_splat(0.0f);
    float Cd = 0.0f;
    int lastpos = -1;
    // Branch-free per lane (predicated); only wave-uniform branches: skip the exp when no pixel of the
    // strip can reach alpha >= 1/255, skip the channel update when no pixel applies the splat.
    for (int b0 = 0; b0 < nrel; b0 += 64) {
        const int n = min(64, nrel - b0);
        if (lane < n) {
            const float4* src = reinterpret_cast<const float4*>(rec + rel_gid[b0 + lane]);
            {   // geometry with the conic and the threshold in log2 units (splat_k2): x y kA kB | kC opacity thr2 depth
                float4 s0 = src[0], s1 = src[1];
                splat_k2(s0.z, s0.w, s1.x, s1.z, s0.z, s0.w, s1.x, s1.z);
                lds[0][lane] = s0;
                lds[1][lane] = s1;
            }
            lds[2][lane] = src[2];
            lds[3][lane] = src[3];
            lds[4][lane] = src[4];        // s1 s2 (clamp bits, pad)
        }
        __syncthreads();
        // four entries per trip: their geometry broadcasts are issued before any of them is processed (LDS latency
        // once per batch); the entries are still applied strictly in list order
        auto blend_one = [&](const int i, const float4 a, const float4 b) {
            float dx, dy;
            // (gx: the pixel's x coordinate, or GATED once the pixel is finished -- see below: no `done` test per entry)
            const float power = splat_power2(a.x, a.y, a.z, a.w, b.x, gx, pyf, dx, dy);
            const bool pre = !(power > 0.0f) && power >= b.z;
            if (__builtin_amdgcn_ballot_w64(pre) == 0ull) return;
            const float alpha = fminf(ALPHA_MAX, b.y * exp2_sel<FAST>(power));
            const bool valid = pre && !(alpha < ALPHA_MIN);
            const float pn = p * (1.0f - alpha);
            const bool stop = valid && (Tb * pn < T_EPS);
            const bool apply = valid && !stop;
            gx = stop ? GATED : gx;
            // (no "nobody applies it" exit: seg_first_body's comment)
            const float wgt = apply ? alpha * (Tb * p) : 0.0f;
            const float4 cc = lds[2][i];  // r g b qw
            const float4 d = lds[3][i];   // qx qy qz s0
            const float2 e2 = *reinterpret_cast<const float2*>(&lds[4][i]);  // s1 s2
            // the eleven channel sums as the record lays the attributes out: five adjacent pairs (packed fma straight
            // from the loaded registers -- summed channel by channel the compiler packed them too, after five register
            // moves per entry) and the depth; each sum is the same fma as before
            const f2 w2 = f2_splat(wgt);
            Cp[0] = f2_fma((f2){cc.x, cc.y}, w2, Cp[0]);   // r g
            Cp[1] = f2_fma((f2){cc.z, cc.w}, w2, Cp[1]);   // b qw
            Cp[2] = f2_fma((f2){d.x, d.y}, w2, Cp[2]);     // qx qy
            Cp[3] = f2_fma((f2){d.z, d.w}, w2, Cp[3]);     // qz s0
            Cp[4] = f2_fma((f2){e2.x, e2.y}, w2, Cp[4]);   // s1 s2
            Cd = fmaf(b.w, wgt, Cd);                       // depth
            p = apply ? pn : p;
            lastpos = apply ? b0 + i : lastpos;            // position in the compacted list of the last entry applied
        };
        int i = 0;
        for (; i + 3 < n; i += 4) {
            const float4 a0 = lds[0][i], b0v = lds[1][i], a1 = lds[0][i + 1], b1v = lds[1][i + 1];
            const float4 a2 = lds[0][i + 2], b2v = lds[1][i + 2], a3 = lds[0][i + 3], b3v = lds[1][i + 3];
            blend_one(i, a0, b0v);
            blend_one(i + 1, a1, b1v);
            blend_one(i + 2, a2, b2v);
            blend_one(i + 3, a3, b3v);
        }
        for (; i < n; ++i) blend_one(i, lds[0][i], lds[1][i]);
        if (__builtin_amdgcn_ballot_w64(gx != GATED) == 0ull) break;  // every pixel of the strip is finished
        __syncthreads();
    }
    if (!(Tb < 0.0f)) {
        float* dst = part + (size_t)c.seg * (NPART * SEG) + pixslot;
        // channel order of `part`: r g b depth qw qx qy qz s0 s1 s2
        const float Cs[NCH] = {Cp[0].x, Cp[0].y, Cp[1].x, Cd, Cp[1].y, Cp[2].x, Cp[2].y, Cp[3].x, Cp[3].y, Cp[4].x, Cp[4].y};
#pragma unroll
        for (int k = 0; k < NCH; ++k) dst[k * SEG] = Cs[k];
        dst[11 * SEG] = p;
        const bool stopped = gx == GATED;     // (this branch: the pixel was alive at the segment start)
        // last contributor: tile-relative list index + 1 of the last entry applied in this segment (0 = none)
        const uint32_t last = lastpos >= 0 ? (uint32_t)(c.sl * SEG + (int)rel_j[lastpos] + 1) : 0u;
        dst[12 * SEG] = __uint_as_float(last | (stopped ? 0x80000000u : 0u));
    }
}

// ---- D: per tile, add the segment sums in order and write the images.
// The planes are split over three workgroups per tile, each adding ITS planes in the spec's order with 13-24 segments
// of loads in flight instead of 6 (all 13 planes of 6 segments fill the registers of one workgroup):
//   group 0: rgb + local product + last contributor  -> colour, alpha, final T, contributor count
//   group 1: depth + rotation (+ local product: normalised depth and the identity fill need the final T)
//   group 2: scale
// Every group reads the boundary transmittances (a pixel's sums stop at the segment in which it finished).  69 -> 65 us;
// the kernel streams ~240 MB of `part` at ~3.8 TB/s behind a ~20 us fixed cost (measured by capping the segments per
// tile), it is not bound by the longest tile's chain: four waves fetching four runs of a heavy tile's segments at
// once and adding them in turn (sums handed on through LDS) took 75 us.
template <int GROUP>
__device__ __forceinline__ void seg_combine_group(const Camera& cam, uint32_t cap, const uint32_t* __restrict__ seg_off,
                                                  const uint32_t* __restrict__ seg_needed, const float* __restrict__ Tbuf,
                                                  const float* __restrict__ part, float* __restrict__ out_color,
                                                  float* __restrict__ out_depth, float* __restrict__ out_quat,
                                                  float* __restrict__ out_scale, float* __restrict__ out_alpha,
                                                  float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                  float* __restrict__ dsum, int first_fused)
{
    // planes of `part` this group loads: [P0, P0 + NP) channel sums, then (WITH_P) the local product and (GROUP 0) the
    // last-contributor word
    constexpr int P0 = GROUP == 0 ? 0 : (GROUP == 1 ? 3 : 8);
    constexpr int NP = GROUP == 0 ? 3 : (GROUP == 1 ? 5 : 3);
    constexpr bool WITH_P = GROUP != 2;
    constexpr int NL = NP + (WITH_P ? 1 : 0) + (GROUP == 0 ? 1 : 0);   // loads per segment besides Tbuf
    constexpr int CU = GROUP == 0 ? 16 : (GROUP == 1 ? 13 : 24);       // segments in flight (~96 registers)
    // the first `ntiles` workgroups of the launch take the tiles of the heavy list (long chains first, vr_segment.h), the
    // others their own tile unless it is on that list
    const int ntiles_c = cam.gx * cam.gy;
    int tile;
    if ((int)blockIdx.x < ntiles_c) {
        const uint32_t hcount = seg_off[seg_counts_offset(ntiles_c, cap) + SEG_COUNT_HEAVY];
        if (blockIdx.x >= hcount) return;
        tile = (int)seg_off[seg_actoff_offset(ntiles_c, cap) + (hcount - 1u - blockIdx.x)];   // the list is in finishing order: longest last
    } else {
        tile = xcd_tile(blockIdx.x - ntiles_c, ntiles_c);
        if ((seg_needed[tile] & 0x7FFFFFFFu) >= HEAVY_TILE) return;
    }
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int px = tx * TILE + region_x(w, lane), py = ty * TILE + region_y(w, lane);
    if (!(px < cam.W && py < cam.H)) return;
    const uint32_t s0 = seg_off[tile];
    if (first_fused && seg_off[tile + 1] - s0 == 1u) return;      // a tile with ONE segment was finished by seg_first_body
    const uint32_t needed = seg_needed[tile];
    float C[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) C[k] = 0.0f;
    float T = 1.0f;
    uint32_t last = 0;
    // every load of a trip is issued before the first use; the adds keep the spec's order.  Values of segments in
    // which the pixel is already finished are ignored.
    bool dead = false;
    for (uint32_t s = 0; s < needed && !dead; s += CU) {
        float Tbv[CU], v[CU][NL];
#pragma unroll
        for (int j = 0; j < CU; ++j) {
            if (s + j < needed) {   // tile-uniform: most tiles need 2-6 segments, their trip issues only those loads
                Tbv[j] = Tbuf[(size_t)(s0 + s + j) * SEG + threadIdx.x];
                const float* src = part + (size_t)(s0 + s + j) * (NPART * SEG) + threadIdx.x;
#pragma unroll
                for (int k = 0; k < NP; ++k) v[j][k] = src[(P0 + k) * SEG];
                if (WITH_P) v[j][NP] = src[11 * SEG];
                if (GROUP == 0) v[j][NP + 1] = src[12 * SEG];
            }
        }
#pragma unroll
        for (int j = 0; j < CU; ++j) {
            if (dead || s + j >= needed) continue;
            if (Tbv[j] < 0.0f) { dead = true; continue; }
#pragma unroll
            for (int k = 0; k < NP; ++k) C[k] += v[j][k];
            if (WITH_P) T = Tbv[j] * v[j][NP];
            if (GROUP == 0) {
                const uint32_t l = __float_as_uint(v[j][NP + 1]) & 0x7FFFFFFFu;
                if (l) last = l;
            }
        }
    }
    const size_t N = (size_t)cam.H * cam.W;
    const size_t pix = (size_t)py * cam.W + px;
    if (GROUP == 0) {
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = fmaf(T, cam.bg[0], C[0]);
        out_color[N + pix] = fmaf(T, cam.bg[1], C[1]);
        out_color[2 * N + pix] = fmaf(T, cam.bg[2], C[2]);
        out_alpha[pix] = 1.0f - T;
    } else if (GROUP == 1) {
        // fork switches (include/vegs_rast.h VrFlags): normalised depth, identity fill of the rotation image
        float depth_out = C[0];
        if (cam.flags & FLAG_DEPTH_NORMALIZED) {
            const float A = 1.0f - T;
            dsum[pix] = C[0];                       // the backward needs the un-normalised sum
            depth_out = A > 0.0f ? C[0] / A : 0.0f;
        }
        out_depth[pix] = depth_out;
        if (cam.flags & FLAG_FILL_EMPTY) C[1] += T;
#pragma unroll
        for (int k = 0; k < 4; ++k) out_quat[k * N + pix] = C[1 + k];
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) out_scale[k * N + pix] = C[k];
    }
}

__global__ void __launch_bounds__(256)
k_seg_combine(Camera cam, uint32_t cap, const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ seg_needed,
              const float* __restrict__ Tbuf, const float* __restrict__ part, float* __restrict__ out_color,
              float* __restrict__ out_depth, float* __restrict__ out_quat, float* __restrict__ out_scale,
              float* __restrict__ out_alpha, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
              float* __restrict__ dsum, int first_fused)
{
    if (blockIdx.y == 0)
        seg_combine_group<0>(cam, cap, seg_off, seg_needed, Tbuf, part, out_color, out_depth, out_quat, out_scale, out_alpha,
                             final_T, n_contrib, dsum, first_fused);
    else if (blockIdx.y == 1)
        seg_combine_group<1>(cam, cap, seg_off, seg_needed, Tbuf, part, out_color, out_depth, out_quat, out_scale, out_alpha,
                             final_T, n_contrib, dsum, first_fused);
    else
        seg_combine_group<2>(cam, cap, seg_off, seg_needed, Tbuf, part, out_color, out_depth, out_quat, out_scale, out_alpha,
                             final_T, n_contrib, dsum, first_fused);
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

// blended (pixel, splat) pairs: every valid entry in front of the pixel's last contributor was applied
__global__ void __launch_bounds__(256)
k_count_blended(Camera cam, const int2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                const Splat* __restrict__ rec, const uint32_t* __restrict__ n_contrib, unsigned long long* __restrict__ out)
{
    const int tile = blockIdx.x;
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const int px = tx * TILE + (int)(threadIdx.x % TILE), py = ty * TILE + (int)(threadIdx.x / TILE);
    unsigned long long cnt = 0;
    if (px < cam.W && py < cam.H) {
        const int start = ranges[tile].x;
        const int nc = (int)n_contrib[(size_t)py * cam.W + px];
        const float pxf = (float)px, pyf = (float)py;
        for (int j = 0; j < nc; ++j) {
            const float4* src = reinterpret_cast<const float4*>(rec + point_list[start + j]);
            const float4 a = src[0], b = src[1];
            float dx, dy;
            float kA, kB, kC, thr2;
            splat_k2(a.z, a.w, b.x, b.z, kA, kB, kC, thr2);
            const float power = splat_power2(a.x, a.y, kA, kB, kC, pxf, pyf, dx, dy);
            if (power > 0.0f) continue;
            const float alpha = fminf(ALPHA_MAX, b.y * ((cam.flags & FLAG_FAST_EXP) ? __builtin_amdgcn_exp2f(power) : vr_exp2(power)));
            if (!(alpha < ALPHA_MIN)) ++cnt;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(out, cnt);
}

int launch_count_blended(const Camera& cam, const int2* ranges, const uint32_t* point_list, const Splat* rec,
                         const uint32_t* n_contrib, unsigned long long* out_dev, hipStream_t s)
{
    VR_HIP(hipMemsetAsync(out_dev, 0, sizeof(unsigned long long), s));
    const int ntiles = cam.gx * cam.gy;
    if (ntiles > 0) {
        hipLaunchKernelGGL(k_count_blended, dim3(ntiles), dim3(256), 0, s, cam, ranges, point_list, rec, n_contrib, out_dev);
        VR_KERNEL_CHECK("count_blended", s, false);
    }
    return 0;
}

// (entry, region) pairs of the NEEDED segments whose relevance bit is set: each is one flush of k_seg_bwd = 17 global
// fp32 atomics (the L2-atomic figure of SURVEY 8d)
__global__ void __launch_bounds__(256)
k_count_flushes(const uint32_t* __restrict__ seg_off, int ntiles, uint32_t nseg, const unsigned long long* __restrict__ segmask,
                unsigned long long* __restrict__ out)
{
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    unsigned long long cnt = 0;
    if (b < nseg) {
        const int4 si = reinterpret_cast<const int4*>(seg_off + seg_tile_offset(ntiles))[b];
        if (si.x >= 0 && ((uint32_t)si.w >> 30) != 0u)
#pragma unroll
            for (int k = 0; k < 16; ++k) cnt += (unsigned long long)__popcll(segmask[(size_t)b * 16 + k]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(out, cnt);
}

int launch_count_flushes(const Camera& cam, long R, const uint32_t* seg_off, const unsigned long long* segmask,
                         unsigned long long* out_dev, hipStream_t s)
{
    VR_HIP(hipMemsetAsync(out_dev, 0, sizeof(unsigned long long), s));
    const int ntiles = cam.gx * cam.gy;
    const size_t nseg = seg_capacity(R, ntiles);
    if (ntiles > 0 && nseg > 0) {
        hipLaunchKernelGGL(k_count_flushes, dim3(cdiv((long)nseg, 256)), dim3(256), 0, s, seg_off, ntiles, (uint32_t)nseg, segmask, out_dev);
        VR_KERNEL_CHECK("count_flushes", s, false);
    }
    return 0;
}

size_t render_fwd_scratch_bytes(long R, int ntiles)
{
    const size_t nseg = seg_capacity(R, ntiles);
    return align_up(nseg * SEG * sizeof(float), 256);
}

int launch_render_fwd(const Camera& cam, long R, const int2* ranges, const uint32_t* point_list, const Splat* rec,
                      uint32_t* seg_off, uint32_t* seg_needed, float* Tbuf, float* part, unsigned long long* segmask,
                      void* scratch, float* out_color,
                      float* out_depth, float* out_quat, float* out_scale, float* out_alpha, float* final_T,
                      uint32_t* n_contrib, float* dsum, uint32_t* needed_hint, hipStream_t s, bool debug, int chain_patience)
{
    const int ntiles = cam.gx * cam.gy;
    if (ntiles == 0) return 0;
    const size_t nseg = seg_capacity(R, ntiles);
    float* Pbuf = (float*)scratch;
    constexpr int first_fused = 1;     // (0 = the round-3 structure: every first segment through k_seg_alpha and k_seg_blend; A/B on one box: same time)
    // Rounds without a hint ("auto"): used from 8.5 list segments per tile on (the table at AUTO_FIRST_SPARSE above).  On
    // the headline view (5 per tile) the deep tiles' catch-up rounds run at low parallelism behind everybody else's round 0
    // and cost 40 us more than the skipped segments save.  The host knows R when it gets here; VR_FLAG_ROUNDS_OFF / _ON
    // override.
    const size_t r_eff = (cam.flags & FLAG_FULL_TILE_LISTS) ? (size_t)R * 2 / 3 : (size_t)R;   // (full rectangles: a third are no-ops)
    const bool dense = 2 * r_eff / SEG >= (size_t)AUTO_DENSE_X2 * ntiles;
    const bool auto_rounds = !needed_hint && R > 0 && !(cam.flags & FLAG_ROUNDS_OFF) && ((cam.flags & FLAG_ROUNDS_ON) || dense);
    const bool rounds = needed_hint || auto_rounds;
    const uint32_t first = dense ? AUTO_FIRST_DENSE : AUTO_FIRST_SPARSE;
    const uint32_t second = dense ? AUTO_SECOND_DENSE : AUTO_SECOND_SPARSE;
    static const bool one_table = [] { const char* e = getenv("VEGS_SEG_TABLE"); return !(e && e[0] == '0'); }();   // (A/B switch)
    static const bool chain_env = [] { const char* e = getenv("VEGS_SEG_CHAIN"); return !(e && e[0] == '0'); }();   // (A/B switch)
    // chain mode (the segment table's comment): a forward without rounds whose table comes from k_seg_table; its arrays
    // (flags [cap] | dead and launch order [2 x tiles, padded]) live in the lists of the rounds
    const bool chain = chain_env && !rounds && first_fused && one_table && ntiles <= SEGTAB_MAX_TILES && R / SEG >= ntiles + 256;
    // A HINTED forward takes the two-kernel table (round 6, advisor finding): k_seg_offsets -- ONE workgroup -- reads the
    // caller's hint array once and leaves the limits in this call's own buffer; in k_seg_table every workgroup would read the
    // array itself, and another stream rendering the same camera (its forward ends by REWRITING the array) could make them
    // disagree on list positions and counts.  (Hints are an opt-in of no_grad forwards; the second launch costs ~4 us there.)
    if (one_table && !needed_hint && ntiles <= SEGTAB_MAX_TILES && nseg > 0) {
        hipLaunchKernelGGL(k_seg_table, dim3(cdiv((long)nseg, 256)), dim3(256), 0, s, ranges, ntiles, seg_off,
                           (const uint32_t*)needed_hint, seg_needed, (uint32_t)nseg, auto_rounds ? first : 0x3FFFFFFFu, first_fused,
                           chain ? 1 : 0);
    } else {
        hipLaunchKernelGGL(k_seg_offsets, dim3(1), dim3(SEGOFF_THREADS), 0, s, ranges, ntiles, seg_off,
                           (const uint32_t*)needed_hint, seg_needed, (uint32_t)nseg, auto_rounds ? first : 0x3FFFFFFFu, first_fused);
        hipLaunchKernelGGL(k_seg_tiles, dim3(cdiv((long)nseg, 256)), dim3(256), 0, s, ntiles, ranges, seg_off,
                           (const uint32_t*)seg_needed, (uint32_t)nseg, first_fused);
    }
    VR_KERNEL_CHECK("seg_offsets", s, debug);
    // Three rounds over work lists (vr_segment.h).  Round 0: the first AUTO_FIRST segments of every tile (with a hint:
    // the hinted prefix -- its length is only known on the device, so the grid covers every slot); k_seg_scan walks them
    // and puts the next window of every tile that still has a live pixel on the next round's list; round 2 takes what is
    // left of the tiles that are short even then.  The catch-up rounds run a fixed grid over lists whose length only the
    // device knows.
    const size_t bound0 = (size_t)first * ntiles;      // round 0 of the automatic rounds: at most AUTO_FIRST segments per tile
    const unsigned grid0 = (unsigned)(!auto_rounds || nseg < bound0 ? nseg : bound0);
    const unsigned gridc = (unsigned)(nseg < 4096 ? nseg : 4096);
    const bool fast = (cam.flags & FLAG_FAST_EXP) != 0u;
    const FwdOut fwd_out{out_color, out_depth, out_quat, out_scale, out_alpha, final_T, n_contrib, dsum};
#define VR_ALPHA(RD, FST, CHN, GRID)                                                                                      \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_seg_alpha<RD, FST, CHN>), dim3(GRID), dim3(256), 0, s, cam, ranges,              \
                       seg_off, (uint32_t)nseg, point_list, rec, Pbuf, segmask, part, first_fused, fwd_out, Tbuf, seg_needed,  \
                       chain_patience < 0 ? CHAIN_MAX_POLLS : chain_patience)
#define VR_ROUND(RD, CHN, GRID)                                                                                           \
    if (R > 0) { if (fast) VR_ALPHA(RD, true, CHN, GRID); else VR_ALPHA(RD, false, CHN, GRID); }                          \
    if (R > 0 || RD == 0) hipLaunchKernelGGL(k_seg_scan<RD>, dim3(ntiles), dim3(256), 0, s, cam, seg_off, (uint32_t)nseg, second, \
                                             (const float*)Pbuf, Tbuf, seg_needed, needed_hint, (RD == 0 && CHN) ? 1 : 0)
    // (+ the tiles' first segments: the launch's first `ntiles` workgroups; chain mode: + the walkers)
    if (chain) { VR_ROUND(0, true, grid0 + (unsigned)ntiles + (unsigned)CHAIN_MAX_HEAVY); }
    else { VR_ROUND(0, false, grid0 + (first_fused ? (unsigned)ntiles : 0u)); }
    if (rounds) {
        VR_ROUND(1, false, gridc);
        VR_ROUND(2, false, gridc);
    }
#undef VR_ROUND
#undef VR_ALPHA
    VR_KERNEL_CHECK("seg_alpha / seg_scan rounds", s, debug);
    hipLaunchKernelGGL(k_seg_merge, dim3(cdiv((long)nseg, 256), SEG_QUEUES + (chain ? 1 : 0)), dim3(256), 0, s, ntiles, seg_off,
                       (uint32_t)nseg, (const uint32_t*)seg_needed, chain ? 1 : 0);
    if (R > 0) {
        if (fast)
            hipLaunchKernelGGL(k_seg_blend<true>, dim3((unsigned)nseg * 4), dim3(64), 0, s, cam, ranges, (const uint32_t*)seg_off,
                               (uint32_t)nseg, (const uint32_t*)seg_needed, point_list, rec, (const float*)Tbuf, part,
                               (const unsigned long long*)segmask, first_fused);
        else
            hipLaunchKernelGGL(k_seg_blend<false>, dim3((unsigned)nseg * 4), dim3(64), 0, s, cam, ranges, (const uint32_t*)seg_off,
                               (uint32_t)nseg, (const uint32_t*)seg_needed, point_list, rec, (const float*)Tbuf, part,
                               (const unsigned long long*)segmask, first_fused);
        VR_KERNEL_CHECK("seg_blend", s, debug);
    }
    hipLaunchKernelGGL(k_seg_combine, dim3(2 * ntiles, 3), dim3(256), 0, s, cam, (uint32_t)nseg, (const uint32_t*)seg_off,
                       (const uint32_t*)seg_needed, (const float*)Tbuf, (const float*)part, out_color, out_depth,
                       out_quat, out_scale, out_alpha, final_T, n_contrib, dsum, first_fused);
    VR_KERNEL_CHECK("seg_combine", s, debug);
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
