// vr_segment.h -- device helpers shared by the segmented compositing kernels (render_fwd.hip,
// render_bwd.hip).  Unit of work: (tile, segment of SEG = 256 consecutive tile-list entries);
// workgroup = 256 threads = 4 wave64; wave w owns the 8x8 pixel region w of the tile (a "strip" in the comments; see REGION_W/REGION_H), lane = pixel.
#pragma once
#include "vr_device.h"

namespace vr {

constexpr int SEG = 256;

// The segment table lives in one array: seg_off[0..T] (first global segment id of every tile, seg_off[T] =
// number of segments) followed, at element seg_tile_offset(T), by one int4 per segment of the launch grid:
//   { tile (-1 beyond the last segment), first list entry, entries, sl | flag << 30 }
// sl = index of the segment inside its tile; flag (written by k_seg_scan): 0 = no pixel needs the segment,
// 1 = needed and the last needed one of its tile, 2 = needed and so is the next.  A workgroup finds ALL of its
// work description with ONE 16-byte load indexed by its block id: workgroups are short-lived, so the number of
// dependent memory round trips before the first useful instruction sets their lifetime, and with it (Little) how
// many of them it takes to keep the SIMDs busy.  (Before: binary search over seg_off -> seg_off[tile] ->
// ranges[tile] -> seg_needed[tile].)
__host__ __device__ inline int seg_tile_offset(int ntiles) { return (ntiles + 1 + 63) & ~63; }

// WORK LISTS behind the seg_info entries of the `cap` segment slots (all uint32; each list entry a global segment id):
//   counts[64] (counts[r] = entries of list r) | 8 queue counters, one 128-byte line each | act_off[T+1, padded]
//   (round 0: first list position of every tile) | lists 0, 1, 2 [cap each]: the forward's three alpha rounds |
//   list 3 [cap]: the NEEDED segments | 8 queue lists [cap each].
// Round 0's list is written in tile order by k_seg_tiles, the catch-up rounds' lists by the k_seg_scan that leaves a
// tile short.  A launch over "all segment slots, most of which return at once" costs ~0.5 us of a CU per empty
// 256-thread workgroup -- 30 us per round on the headline view; the lists make a round cost what it computes.
//
// NEEDED list: what k_seg_blend and k_seg_bwd run over (workgroup b takes entry b >> 2, region b & 3).  Its ORDER is
// worth 8 % of k_seg_bwd and 5 % of k_seg_blend: in the COMPLETION order of the k_seg_scan workgroups (the tile that
// finishes its chain appends its run) they take 320 / 127 us, in tile order 350 / 134 -- same instructions, same wave
// cycles, 8 % more waves in flight on average (PMC).  What a good order has (profiles/experiments/README.md): the
// expensive workgroups -- a tile's FRONT segments, every pixel still alive -- first, the cheap deep ones last, and
// consecutive workgroups on different tiles; a constructed depth-index-major list gets 325 / 129 us, orders that sort or
// permute TILES get nothing.  The completion order has both properties by itself and is cheaper to produce: a finishing tile appends to the list of
// its dispatch queue (blockIdx % 8; one atomicAdd on that queue's own cache line -- 2064 appends to ONE counter cost
// the scan 16 us), and k_seg_merge deals the eight queue lists round-robin into the needed list.  The order varies
// from run to run; no result depends on it (the images are per-segment sums added in segment order by k_seg_combine,
// the gradient sums are atomic -- or, in deterministic mode, written to per-(entry, region) slots).
constexpr int SEG_LIST_NEEDED = 3;
constexpr int SEG_QUEUES = 8;
// HEAVY tiles (>= HEAVY_TILE needed segments: the vanishing-point tiles of a street view need up to ~200): the per-tile
// chain kernels behind k_seg_scan (k_seg_combine, k_seg_suffix) start them FIRST -- a chain of 200 segments takes ~20 us
// whoever else is running, so it had better not start in the middle of the launch (k_seg_combine 63.5 -> 54 us).  The
// finishing k_seg_scan workgroup appends the tile id (counts[SEG_COUNT_HEAVY]; list in act_off's space, which is free once
// k_seg_tiles has run); the first `ntiles` workgroups of those launches take the list from its END (finishing order =
// shortest first), the others their own tile unless it is on the list.
constexpr uint32_t HEAVY_TILE = 64u;
constexpr int SEG_COUNT_HEAVY = 16;
__host__ __device__ inline size_t seg_counts_offset(int ntiles, size_t cap) { return (size_t)seg_tile_offset(ntiles) + 4 * cap; }
__host__ __device__ inline size_t seg_qcount_offset(int ntiles, size_t cap, int q) { return seg_counts_offset(ntiles, cap) + 64 + 32 * (size_t)q; }
__host__ __device__ inline size_t seg_actoff_offset(int ntiles, size_t cap) { return seg_counts_offset(ntiles, cap) + 64 + 32 * SEG_QUEUES; }
__host__ __device__ inline size_t seg_list_offset(int ntiles, size_t cap, int r)
{
    return seg_actoff_offset(ntiles, cap) + (size_t)seg_tile_offset(ntiles) + (size_t)r * cap;
}
__host__ __device__ inline size_t seg_qlist_offset(int ntiles, size_t cap, int q) { return seg_list_offset(ntiles, cap, 4 + q); }
__host__ __device__ inline size_t seg_table_words(int ntiles, size_t cap) { return seg_list_offset(ntiles, cap, 4 + SEG_QUEUES); }

struct SegCtx {
    uint32_t seg;      // global segment id handled by this workgroup
    int tile, sl;      // tile id, segment index inside the tile
    int first, count;  // first list entry of the segment (absolute index into point_list), entries
    uint32_t flag;     // 0 not needed / 1 needed, last of its tile / 2 needed and the next one too (after k_seg_scan)
    int px, py;        // this lane's pixel
    float x0, y0;      // tile origin (pixel coordinates of its first column / row)
    bool inside;
    size_t pix;
};

// (b, w): segment id and strip handled by the calling wave; lane = threadIdx.x & 63
__device__ __forceinline__ bool seg_setup_at(const Camera& cam, const int2* __restrict__ ranges,
                                             const uint32_t* __restrict__ seg_off, uint32_t b, int w, SegCtx& c);

__device__ __forceinline__ bool seg_setup(const Camera& cam, const int2* __restrict__ ranges,
                                          const uint32_t* __restrict__ seg_off, SegCtx& c)
{
    return seg_setup_at(cam, ranges, seg_off, blockIdx.x, threadIdx.x >> 6, c);
}

__device__ __forceinline__ bool seg_setup_at(const Camera& cam, const int2* __restrict__ ranges,
                                             const uint32_t* __restrict__ seg_off, uint32_t b, int w, SegCtx& c)
{
    const int ntiles = cam.gx * cam.gy;
    // Identity block -> segment map on purpose: the dispatcher deals consecutive blocks round-robin to
    // the 8 XCDs, which spreads the (contiguous) segments of the heavy vanishing-point tiles over the
    // whole chip.  Giving each XCD a contiguous run of segments for L2 locality was measured 1.4-2x
    // SLOWER (one XCD ends up with all the long tiles).
    const int4 si = reinterpret_cast<const int4*>(seg_off + seg_tile_offset(ntiles))[b];
    if (si.x < 0) return false;
    c.seg = b;
    c.tile = si.x;
    c.first = si.y;
    c.count = si.z;
    c.sl = si.w & 0x3FFFFFFF;
    c.flag = (uint32_t)si.w >> 30;
    const int tx = c.tile % cam.gx, ty = c.tile / cam.gx;
    const int lane = threadIdx.x & 63;
    c.px = tx * TILE + region_x(w, lane);
    c.py = ty * TILE + region_y(w, lane);
    c.x0 = (float)(tx * TILE);
    c.y0 = (float)(ty * TILE);
    c.inside = c.px < cam.W && c.py < cam.H;
    c.pix = (size_t)c.py * cam.W + c.px;
    return true;
}

// Relevance masks of the segment's entries: mask[strip*4 + part] bit j set <=> entry part*64+j can reach
// alpha >= 1/255 somewhere in that region (conservative ellipse-vs-rectangle test).  Built once per
// segment by k_seg_alpha (thread j tests entry j; q0 = x y A B, q1 = C opacity thr depth), kept in the
// binning buffer (16 x u64 per segment) and reused by the blend and backward kernels.
__device__ __forceinline__ void seg_build_masks(const SegCtx& c, bool have, float4 q0, float4 q1,
                                                unsigned long long* __restrict__ gmask)
{
    const uint32_t bits = have ? strips_relevant_exact(q0.x, q0.y, q0.z, q0.w, q1.x, q1.z, c.x0, c.y0) : 0u;
    const int part = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const unsigned long long m = __ballot((bits >> s) & 1u);
        if ((threadIdx.x & 63) == 0) gmask[s * 4 + part] = m;
    }
}

// wave-uniform 64-bit value held in SGPRs
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

}  // namespace vr
