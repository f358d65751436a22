// binning.hip -- builds the per-tile, depth-ordered splat lists.
//
// Contract (what the reference's native binning stage produces): the list of
// (tile, Gaussian) pairs ordered by (tile id, fp32 depth bits, Gaussian id) plus [start,end) per tile.
//
// MI355X design (not the upstream 64-bit global radix sort): the order is produced by two short,
// stable, LSD radix sorts on 32-bit keys:
//   1. compact the V visible Gaussians in id order (wave scan + block scan),
//   2. stable-sort them by depth bits              (4 passes x 8 bit over V  <<  R elements),
//   3. exclusive-scan tiles_touched in that order and emit (tile, id) pairs: emission order is
//      already (depth, id)-sorted within every tile,
//   4. stable-sort the R pairs by tile id only     (ceil(log2(T)/8) = 2 passes for 2064 tiles),
//   5. tile ranges from key boundaries.
// R-sized traffic is 2 passes of 8-byte pairs instead of upstream's 6 passes of 12-byte pairs.
// Ranking inside a pass uses wave64 ballots (match-by-digit), no per-element atomics.
#include "../../include/vegs_rast_debug.h"
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "vr_host.h"

namespace vr {

constexpr int SCAN_ITEMS = 8;                   // per thread
constexpr int SCAN_BLOCK = 256 * SCAN_ITEMS;    // elements per block
constexpr int RADIX_ITEMS = 16;                 // per thread
constexpr int RADIX_BLOCK = 256 * RADIX_ITEMS;  // elements per block

// ---------------------------------------------------------------- wave / block primitives

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread block; returns block total via `total`
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t& total, uint32_t* lds4)
{
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) lds4[w] = incl;
    __syncthreads();
    uint32_t s0 = lds4[0], s1 = lds4[1], s2 = lds4[2], s3 = lds4[3];
    uint32_t woff = (w > 0 ? s0 : 0) + (w > 1 ? s1 : 0) + (w > 2 ? s2 : 0);
    total = s0 + s1 + s2 + s3;
    __syncthreads();
    return woff + incl - v;
}

// ---------------------------------------------------------------- digit plan of the radix sorts

// digit width for sorting `nbits` key bits: fewest passes first, then the narrowest digit (cheaper ranking)
__host__ __device__ inline int radix_digit(int nbits)
{
    if (nbits <= 12) return 6;   // 1-2 passes
    if (nbits <= 16) return 8;   // 2
    if (nbits <= 18) return 9;   // 2
    if (nbits <= 24) return 8;   // 3
    if (nbits <= 27) return 9;   // 3
    return 8;                    // 4
}
__host__ __device__ inline int radix_passes(int nbits)
{
    return nbits <= 0 ? 0 : (nbits + radix_digit(nbits) - 1) / radix_digit(nbits);
}
__host__ __device__ inline int bits_of(uint32_t span)   // bits needed to tell 0..span apart
{
    int b = 0;
    while (span) { ++b; span >>= 1; }
    return b;
}
constexpr int HIST_WORDS = 4 * 512;   // digit totals of one sort: passes << digit words, at most this many

// ---------------------------------------------------------------- generic 3-kernel scan

// tiles a Gaussian's list entries go to: the tiles of the reachable cells of its rectangle (vr_device.h: TIGHT TILE LISTS).
// Up to 64 tiles a cell is a tile, both words are mask (bit j = tile j, row-major) and the count is its population count;
// beyond, r.z is the mask of at most 32 cells and r.w the number of kept tiles.
__device__ __forceinline__ unsigned long long rect_mask(uint4 r) { return ((unsigned long long)r.w << 32) | r.z; }
__device__ __forceinline__ uint32_t rect_area(uint4 r)
{
    const uint32_t area = (r.y & 0xFFFFu) * (r.y >> 16);
    return area > (uint32_t)TIGHT_MAX_TILES ? r.w : (uint32_t)(__popc(r.z) + __popc(r.w));
}

struct SrcFlagTiles {  // 1 for Gaussians with list entries; secondary value = their number (summed only);
    const uint32_t* tile_count;  // also the min / max depth key of those Gaussians (range of the depth sort)
    const uint32_t* depth_key;
    static constexpr bool MINMAX = true;
    __device__ uint32_t operator()(long i) const { return tile_count[i] ? 1u : 0u; }
    __device__ uint32_t second(long i) const { return tile_count[i] & TILE_COUNT_MASK; }
    __device__ uint32_t third(long i) const { return tile_count[i] >> 31; }      // rectangles of more than 64 tiles (counted)
    __device__ uint32_t key(long i) const { return depth_key[i]; }
};
struct SrcRectSorted {  // tiles touched, in depth-sorted order (rectangles already gathered: coalesced reads)
    const uint4* rect_sorted;
    static constexpr bool MINMAX = false;
    __device__ uint32_t operator()(long i) const { return rect_area(rect_sorted[i]); }
    __device__ uint32_t second(long) const { return 0; }
    __device__ uint32_t third(long) const { return 0; }
    __device__ uint32_t key(long) const { return 0; }
};
struct SrcPlain {
    const uint32_t* v;
    static constexpr bool MINMAX = false;
    __device__ uint32_t operator()(long i) const { return v[i]; }
    __device__ uint32_t second(long) const { return 0; }
    __device__ uint32_t third(long) const { return 0; }
    __device__ uint32_t key(long) const { return 0; }
};

struct SinkStore {
    uint32_t* out;
    __device__ void operator()(long i, uint32_t, uint32_t excl) const { out[i] = excl; }
};

// bsum2 (optional): per-block sums of the secondary value; for MINMAX sources bsum2[nb..3nb) also
// receives the per-block min / max of key(i) over the elements with a non-zero value.
template <class Src>
__global__ void __launch_bounds__(256) k_scan_reduce(Src src, long n, uint32_t* bsum, uint32_t* bsum2)
{
    __shared__ uint32_t lds[20];
    // sums, minimum and maximum do not care which thread takes which element of the block: consecutive lanes take
    // consecutive elements (coalesced; with SCAN_ITEMS consecutive elements per thread, as the ordered apply kernels must,
    // every load instruction of a wave touches 64 lines: 70 -> 2x us at 5 M Gaussians, where the arrays no longer sit in L2)
    const long base = (long)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    uint32_t a = 0, b = 0, c3 = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long e = base + (long)k * 256;
        if (e < n) {
            const uint32_t v = src(e);
            a += v;
            b += src.second(e);
            if (Src::MINMAX && v) { const uint32_t key = src.key(e); kmin = min(kmin, key); kmax = max(kmax, key); c3 += src.third(e); }
        }
    }
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        a += __shfl_down(a, d, 64);
        b += __shfl_down(b, d, 64);
        if (Src::MINMAX) {
            kmin = min(kmin, (uint32_t)__shfl_down((int)kmin, d, 64));
            kmax = max(kmax, (uint32_t)__shfl_down((int)kmax, d, 64));
            c3 += __shfl_down(c3, d, 64);
        }
    }
    if (lane == 0) { lds[w] = a; lds[4 + w] = b; lds[8 + w] = kmin; lds[12 + w] = kmax; lds[16 + w] = c3; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bsum[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
        if (bsum2) {
            bsum2[blockIdx.x] = lds[4] + lds[5] + lds[6] + lds[7];
            if (Src::MINMAX) {
                bsum2[gridDim.x + blockIdx.x] = min(min(lds[8], lds[9]), min(lds[10], lds[11]));
                bsum2[2 * gridDim.x + blockIdx.x] = max(max(lds[12], lds[13]), max(lds[14], lds[15]));
                bsum2[3 * gridDim.x + blockIdx.x] = lds[16] + lds[17] + lds[18] + lds[19];
            }
        }
    }
}

// block-wide sum of one value per thread (256 threads)
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* lds4)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// Second (last) kernel of a scan.  Every workgroup derives its own exclusive prefix from the per-block
// sums of the reduce kernel (a few KB, L2-resident) instead of waiting for a separate single-block scan
// kernel.
template <class Src, class Sink>
__global__ void __launch_bounds__(256)
k_scan_apply(Src src, Sink sink, long n, const uint32_t* __restrict__ bsum)
{
    __shared__ uint32_t lds4[4];
    uint32_t acc = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) acc += bsum[j];
    const uint32_t prefix = block_sum(acc, lds4);
    long base = (long)blockIdx.x * SCAN_BLOCK + (long)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t tsum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? src(base + k) : 0;
        tsum += v[k];
    }
    uint32_t total;
    uint32_t ex = block_excl_scan(tsum, total, lds4) + prefix;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) sink(base + k, v[k], ex);
        ex += v[k];
    }
}

// Grand totals of a reduce pass on their own (one workgroup): lets the host read them back while the apply
// kernel is still running.  totals = {sum, secondary sum, min key, max key}.
__global__ void __launch_bounds__(256)
k_scan_totals(const uint32_t* __restrict__ bsum, const uint32_t* __restrict__ bsum2, int nb, uint32_t* __restrict__ totals,
              uint32_t* __restrict__ err_clear, uint32_t* __restrict__ host_mail, uint32_t seq)
{
    __shared__ uint32_t lds4[4];
    __shared__ uint32_t mm[8];
    uint32_t t0 = 0, t1 = 0, t2 = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
    for (int j = threadIdx.x; j < nb; j += 256) {
        t0 += bsum[j];
        t1 += bsum2[j];
        t2 += bsum2[3 * nb + j];
        kmin = min(kmin, bsum2[nb + j]);
        kmax = max(kmax, bsum2[2 * nb + j]);
    }
    t0 = block_sum(t0, lds4);
    t1 = block_sum(t1, lds4);
    t2 = block_sum(t2, lds4);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, d, 64));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d, 64));
    }
    if ((threadIdx.x & 63) == 0) { mm[threadIdx.x >> 6] = kmin; mm[4 + (threadIdx.x >> 6)] = kmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        totals[0] = t0;
        totals[1] = t1;
        totals[2] = min(min(mm[0], mm[1]), min(mm[2], mm[3]));
        totals[3] = max(max(mm[4], mm[5]), max(mm[6], mm[7]));
        // this forward's guard word (api.hip: one per forward in flight) starts clean: the binning kernels that may raise it
        // are behind this kernel on the stream
        if (err_clear) __hip_atomic_store(err_clear, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        totals[4] = 0u;
        totals[5] = t2;      // Gaussians whose rectangle has more than 64 tiles (who emits them: emit_big_inline)
        // The host's copy: written straight into its pinned, coherent mailbox as six 64-bit {value, sequence number} words --
        // every word says by itself which forward it belongs to, so the host's view of the six does not depend on the order
        // in which the stores arrive (round 6; before: six values and then the sequence number, with a release in between).
        // No copy command and no event on the stream: the apply kernel follows this one without the ~10 us the two used
        // to put between them.
        if (host_mail) {
            unsigned long long* const m64 = reinterpret_cast<unsigned long long*>(host_mail);
#pragma unroll
            for (int k = 0; k < 6; ++k)
                __hip_atomic_store(&m64[k], ((unsigned long long)seq << 32) | totals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <class Src, class Sink>
static int run_scan(Src src, Sink sink, long n, uint32_t* bsum, hipStream_t s, bool debug, const char* what)
{
    if (n <= 0) return 0;
    int nb = cdiv(n, SCAN_BLOCK);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_reduce<Src>), dim3(nb), dim3(256), 0, s, src, n, bsum, (uint32_t*)nullptr);
    VR_KERNEL_CHECK(what, s, debug);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_apply<Src, Sink>), dim3(nb), dim3(256), 0, s, src, sink, n,
                       (const uint32_t*)bsum);
    VR_KERNEL_CHECK(what, s, debug);
    return 0;
}

// ---------------------------------------------------------------- stage 1: compaction

// block sums of the compaction scan (4 words per block), then one partial digit histogram of the depth keys per block
// stage-1 scratch: five per-block values of the compaction's reduce pass (count, list entries, min key, max key, large
// rectangles), then the per-block digit histograms of the depth keys, then the per-block id checksums (below)
static inline size_t stage1_partial_offset(size_t nb) { return align_up(5 * nb * sizeof(uint32_t), 256); }
static inline size_t stage1_chk_offset(size_t nb) { return stage1_partial_offset(nb) + align_up(nb * HIST_WORDS * sizeof(uint32_t), 256); }
size_t binning_stage1_scratch_bytes(int P)
{
    size_t nb = (size_t)cdiv(P > 0 ? P : 1, SCAN_BLOCK);
    return stage1_chk_offset(nb) + align_up(nb * sizeof(uint2), 256);
}

// PERMUTATION CHECK of the depth sort (always on, round 6).  The sort's passes move (key, id) pairs to positions computed
// from sums that OTHER workgroups posted; should any of that ever go wrong -- a stale status word, a lost workgroup, a
// hardware hiccup -- the value buffer stops being a permutation of the visible ids (slots written twice, others holding
// whatever was there before) and everything downstream would index with it.  So the ids are summed on the way in (the
// compaction: one pair of sums per workgroup, no atomics) and on the way out (the emission), both as a plain sum and as the
// sum of a hash, and the last binning kernel compares the two: a mismatch fails the view like a timed-out wait (guard word,
// empty tile ranges).  Two multisets of V ids with equal sums and equal hashed sums that are NOT equal would need a
// coincidence of 2^-64.
__device__ __forceinline__ uint32_t perm_mix(uint32_t id) { return (id ^ (id >> 11)) * 0x9E3779B1u + 0x7F4A7C15u; }
// block-wide sums of two values per thread (256 threads) -> thread 0 stores them
__device__ __forceinline__ void block_store_pair(uint32_t a, uint32_t b, uint2* dst, uint32_t* lds8)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if ((threadIdx.x & 63) == 0) { lds8[threadIdx.x >> 6] = a; lds8[4 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    if (threadIdx.x == 0) *dst = make_uint2(lds8[0] + lds8[1] + lds8[2] + lds8[3], lds8[4] + lds8[5] + lds8[6] + lds8[7]);
}

// Two halves: the totals {V, R, min key, max key} reach the host (host_mail, see k_scan_totals) while the apply kernel
// runs, so the round trip overlaps with that kernel instead of idling the GPU.
int launch_compact_reduce(int P, const uint32_t* tile_count, const uint32_t* depth_key, void* scratch, uint32_t* totals_dev,
                          uint32_t* err_clear, uint32_t* host_mail, uint32_t seq, hipStream_t s, bool debug)
{
    int nb = cdiv(P > 0 ? P : 1, SCAN_BLOCK);
    uint32_t* bsum = (uint32_t*)scratch;
    uint32_t* bsum2 = bsum + nb;
    SrcFlagTiles src{tile_count, depth_key};
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_reduce<SrcFlagTiles>), dim3(nb), dim3(256), 0, s, src, (long)P, bsum, bsum2);
    VR_KERNEL_CHECK("compact_reduce", s, debug);
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(256), 0, s, (const uint32_t*)bsum, (const uint32_t*)bsum2, nb, totals_dev,
                       err_clear, host_mail, seq);
    VR_KERNEL_CHECK("compact_totals", s, debug);
    return 0;
}

// ---------------------------------------------------------------- radix sort pass (stable LSD)

// lanes of the wave (restricted to `valid`) holding the same BITS-bit digit as this lane
template <int BITS>
__device__ __forceinline__ unsigned long long match_digit(uint32_t d, unsigned long long valid)
{
    unsigned long long m = valid;
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        unsigned long long bal = __ballot((d >> b) & 1u);
        m &= ((d >> b) & 1u) ? bal : ~bal;
    }
    return m;
}

__device__ __forceinline__ unsigned long long lanemask_lt()
{
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

// digit of a key: keys are first shifted down by `kmin` (monotone), so only the bits that actually
// vary across the input have to be sorted
template <int BITS>
__device__ __forceinline__ uint32_t digit_of(uint32_t key, uint32_t kmin, int shift)
{
    return ((key - kmin) >> shift) & ((1u << BITS) - 1u);
}

// hist layout: digit-major [1<<BITS][nblk]
template <int BITS>
__global__ void __launch_bounds__(256)
k_radix_hist(const uint32_t* __restrict__ keys, long n, uint32_t kmin, int shift, uint32_t* __restrict__ hist,
             int nblk)
{
    constexpr int SIZE = 1 << BITS;
    __shared__ uint32_t h[SIZE];
    for (int d = threadIdx.x; d < SIZE; d += 256) h[d] = 0;
    __syncthreads();
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    long wbase = (long)blockIdx.x * RADIX_BLOCK + (long)w * (64 * RADIX_ITEMS);
    uint32_t key[RADIX_ITEMS];
#pragma unroll
    for (int i = 0; i < RADIX_ITEMS; ++i) {
        long idx = wbase + i * 64 + lane;
        key[i] = idx < n ? keys[idx] : 0u;
    }
#pragma unroll
    for (int i = 0; i < RADIX_ITEMS; ++i) {
        long idx = wbase + i * 64 + lane;
        bool ok = idx < n;
        uint32_t d = ok ? digit_of<BITS>(key[i], kmin, shift) : 0;
        unsigned long long valid = __ballot(ok);
        unsigned long long m = match_digit<BITS>(d, valid);
        if (ok && (m & lanemask_lt()) == 0) atomicAdd(&h[d], (uint32_t)__popcll(m));
    }
    __syncthreads();
    for (int d = threadIdx.x; d < SIZE; d += 256) hist[(size_t)d * nblk + blockIdx.x] = h[d];
}

// GATHER (last pass of the depth sort): the value is a Gaussian id; its packed tile rectangle is gathered into sorted
// order on the way out (the only gather of the binning stage, formerly its own kernel).  (Tried and rejected:
// accumulating the block sums of the following scans with global atomics here and in the histogram kernel, to save
// the scans' reduce launches -- a few hundred hot words serialise at ~12 ns per atomic: 9 -> 358 us.)
template <int BITS, bool GATHER>
__global__ void __launch_bounds__(256)
k_radix_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, long n, uint32_t kmin, int shift,
                const uint32_t* __restrict__ hist_scanned, int nblk, const uint4* __restrict__ rect,
                uint4* __restrict__ rect_sorted)
{
    constexpr int SIZE = 1 << BITS;
    __shared__ uint32_t cnt[4][SIZE];
    __shared__ uint32_t off[4][SIZE];
    __shared__ uint32_t loc[SIZE], gl[SIZE];
    __shared__ uint32_t skey[RADIX_BLOCK], sval[RADIX_BLOCK];
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int d = threadIdx.x; d < SIZE; d += 256) {
#pragma unroll
        for (int k = 0; k < 4; ++k) cnt[k][d] = 0;
    }
    __syncthreads();
    long wbase = (long)blockIdx.x * RADIX_BLOCK + (long)w * (64 * RADIX_ITEMS);
    uint32_t key[RADIX_ITEMS], val[RADIX_ITEMS], rank[RADIX_ITEMS];
    unsigned long long lt = lanemask_lt();
    // all loads first (32 in flight per lane), ranking afterwards: the pass is latency-bound otherwise
#pragma unroll
    for (int i = 0; i < RADIX_ITEMS; ++i) {
        long idx = wbase + i * 64 + lane;
        key[i] = idx < n ? keys_in[idx] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < RADIX_ITEMS; ++i) {
        long idx = wbase + i * 64 + lane;
        val[i] = idx < n ? vals_in[idx] : 0u;
    }
#pragma unroll
    for (int i = 0; i < RADIX_ITEMS; ++i) {
        long idx = wbase + i * 64 + lane;
        bool ok = idx < n;
        uint32_t d = digit_of<BITS>(key[i], kmin, shift);
        unsigned long long valid = __ballot(ok);
        unsigned long long m = match_digit<BITS>(d, valid);
        uint32_t prior = ok ? cnt[w][d] : 0;
        rank[i] = prior + (uint32_t)__popcll(m & lt);
        if (ok && (m & lt) == 0) cnt[w][d] = prior + (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- block-local bucket layout: items are first put in digit order in LDS and then written out by
    // consecutive threads, so every bucket's run of this block is one contiguous, coalesced global write
    // (a direct scatter of 8-byte pairs to up to 512 destinations per wave wrote mostly 1/8-used lines)
    for (int d = threadIdx.x; d < SIZE; d += 256) {
        const uint32_t c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d], c3 = cnt[3][d];
        gl[d] = hist_scanned[(size_t)d * nblk + blockIdx.x];
        off[0][d] = 0;
        off[1][d] = c0;
        off[2][d] = c0 + c1;
        off[3][d] = c0 + c1 + c2;
        loc[d] = c0 + c1 + c2 + c3;
    }
    __syncthreads();
    if (w == 0) {   // exclusive scan of the bucket totals: lane handles DPL consecutive digits
        constexpr int DPL = SIZE / 64;
        uint32_t t[DPL], sum = 0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) { t[k] = loc[lane * DPL + k]; sum += t[k]; }
        uint32_t incl = sum;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, dd, 64);
            if (lane >= dd) incl += o;
        }
        uint32_t run = incl - sum;
#pragma unroll
        for (int k = 0; k < DPL; ++k) { loc[lane * DPL + k] = run; run += t[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RADIX_ITEMS; ++i) {
        long idx = wbase + i * 64 + lane;
        if (idx < n) {
            uint32_t d = digit_of<BITS>(key[i], kmin, shift);
            uint32_t p = loc[d] + off[w][d] + rank[i];
            skey[p] = key[i];
            sval[p] = val[i];
        }
    }
    __syncthreads();
    const long bbase = (long)blockIdx.x * RADIX_BLOCK;
    const int nvalid = (int)((n - bbase) < (long)RADIX_BLOCK ? (n - bbase) : (long)RADIX_BLOCK);
    for (int j = threadIdx.x; j < nvalid; j += 256) {
        const uint32_t k = skey[j];
        const uint32_t d = digit_of<BITS>(k, kmin, shift);
        const uint32_t dst = gl[d] + ((uint32_t)j - loc[d]);
        const uint32_t v = sval[j];
        keys_out[dst] = k;
        vals_out[dst] = v;
        if (GATHER) rect_sorted[dst] = rect[v];
    }
}

// One pass = histogram -> scan (reduce + apply) -> scatter.
struct RadixGather { const uint4* rect; uint4* rect_sorted; };
template <int BITS>
static int radix_pass(const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, long n, uint32_t kmin,
                      int shift, uint32_t* hist, uint32_t* bsum, const RadixGather* gather, hipStream_t s, bool debug)
{
    int nblk = cdiv(n, RADIX_BLOCK);
    long hn = (long)(1 << BITS) * nblk;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_radix_hist<BITS>), dim3(nblk), dim3(256), 0, s, kin, n, kmin, shift, hist,
                       nblk);
    VR_KERNEL_CHECK("radix_hist", s, debug);
    int rc = run_scan(SrcPlain{hist}, SinkStore{hist}, hn, bsum, s, debug, "radix_scan");
    if (rc) return rc;
    if (gather)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_radix_scatter<BITS, true>), dim3(nblk), dim3(256), 0, s, kin, vin, kout, vout,
                           n, kmin, shift, (const uint32_t*)hist, nblk, gather->rect, gather->rect_sorted);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_radix_scatter<BITS, false>), dim3(nblk), dim3(256), 0, s, kin, vin, kout, vout,
                           n, kmin, shift, (const uint32_t*)hist, nblk, (const uint4*)nullptr, (uint4*)nullptr);
    VR_KERNEL_CHECK("radix_scatter", s, debug);
    return 0;
}

// Stable LSD sort of (key, val) pairs on the low `nbits` bits of (key - kmin).  Ping-pongs between
// (k0,v0) and (k1,v1); returns in *res which pair holds the result (0 or 1).  Digit width is chosen
// per call: few wide passes for many bits, narrow digits (cheaper ballot ranking) for few bits.
// number of scan block sums one pass over n elements needs (any digit width: 512 digits is the maximum)
static inline size_t radix_bsum_words(long n) { return (size_t)cdiv((long)cdiv(n, RADIX_BLOCK) * 512, SCAN_BLOCK) + 1; }

// `bsum`: radix_bsum_words(n) words of scratch.  `gather`: rectangle gather on the LAST pass (depth sort).
static int radix_sort(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, long n, uint32_t kmin, int nbits,
                      uint32_t* hist, uint32_t* bsum, const RadixGather* gather, hipStream_t s, bool debug, int* res)
{
    const int digit = radix_digit(nbits);
    const int passes = nbits <= 0 ? 0 : cdiv(nbits, digit);
    uint32_t *ka = k0, *va = v0, *kb = k1, *vb = v1;
    int where = 0;
    for (int pass = 0; pass < passes; ++pass) {
        int rc;
        const RadixGather* g = pass == passes - 1 ? gather : nullptr;
        if (digit == 6) rc = radix_pass<6>(ka, va, kb, vb, n, kmin, pass * digit, hist, bsum, g, s, debug);
        else if (digit == 9) rc = radix_pass<9>(ka, va, kb, vb, n, kmin, pass * digit, hist, bsum, g, s, debug);
        else rc = radix_pass<8>(ka, va, kb, vb, n, kmin, pass * digit, hist, bsum, g, s, debug);
        if (rc) return rc;
        uint32_t* t = ka; ka = kb; kb = t;
        t = va; va = vb; vb = t;
        where ^= 1;
    }
    *res = where;
    return 0;
}
int radix_sort_passes(int nbits) { return radix_passes(nbits); }

// generic entry for other translation units (knn.hip): sort (key,val) pairs on the low nbits of key - kmin
size_t sort_pairs_scratch_bytes(long n)
{
    const size_t nblk = (size_t)cdiv(n > 0 ? n : 1, RADIX_BLOCK);
    return align_up(nblk * 512 * 4, 256) + align_up(radix_bsum_words(n > 0 ? n : 1) * 4 + 256, 256);
}
int launch_sort_pairs(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, long n, uint32_t kmin, int nbits,
                      void* scratch, hipStream_t s, bool debug, int* where)
{
    const size_t nblk = (size_t)cdiv(n > 0 ? n : 1, RADIX_BLOCK);
    uint32_t* hist = (uint32_t*)scratch;
    uint32_t* bsum = (uint32_t*)((char*)scratch + align_up(nblk * 512 * 4, 256));
    *where = 0;
    if (n <= 0) return 0;
    return radix_sort(k0, v0, k1, v1, n, kmin, nbits, hist, bsum, nullptr, s, debug, where);
}

// ---------------------------------------------------------------- single-launch radix passes (posted block sums)
//
// The rasterizer's two sorts do not use the 4-launch pass above but one launch per pass: the digit totals of ALL
// passes are order-independent, so one histogram kernel computes them up front; a pass then ranks its block locally,
// posts the block's digit counts and obtains the counts of the blocks before it from what those blocks posted,
// instead of a separate device-wide scan.  All workgroups of a pass are resident at once at these sizes, so a chained
// look-back would walk hundreds of predecessors at ~2.5 us per dependent device-scope load on this 8-XCD part (first
// attempt: 88 us per pass).  The sums are therefore posted on three fixed levels of fan-in 16:
//   level 1  [block][digit]         the block's own count
//   level 2  [block/16][digit]      sum over a complete group of 16 blocks, posted by the group's last block
//   level 3  [block/256][digit]     sum over 256 blocks, posted by the last block of the group's last sub-group
// and a block adds at most 15 entries of each level: three dependent load rounds, whatever the grid size (up to
// 4096 blocks; longer inputs take the 4-launch passes).  A block only waits for blocks with a smaller index, which the
// in-order dispatcher has started before it; every wait is bounded all the same (2 s of wall clock, SpinClock) and reports
// through `err` rather than hanging the queue.  Status words are read and written with device-scope atomics (the 8
// XCD L2s are not coherent with each other); flag and count share the word, so no fence is needed.
constexpr uint32_t ST_POSTED = 1u << 31, ST_VALUE = ST_POSTED - 1u;
constexpr int FAN = 16;                                        // fan-in of a level
constexpr long ONESWEEP_MAX_N = (long)FAN * FAN * FAN * RADIX_BLOCK;   // 16.7 M elements
// Bound of every wait for another workgroup's posted sum: WALL-CLOCK, not a poll count -- a predecessor that is merely
// not running yet (CUs held by another process: several ranks sharing one GPU) must not trip it.  s_memrealtime ticks
// at 100 MHz; the clock is only read every SPIN_CHECK polls.  2 s is far beyond anything but a lost workgroup; the
// alternative to a bound would be a hung queue.
constexpr int SPIN_CHECK = 256;
constexpr unsigned long long SPIN_TICKS = 200000000ull;   // 2 s at 100 MHz
// ... AND the waiter must itself have polled for about as long: the wall clock also runs while the whole process is
// switched out (several processes time-slicing one GPU), when its predecessors cannot post either.  A poll is a round of
// dependent device-scope loads (1 ... 2.5 us): 2^20 of them are 1 ... 2.5 s of the waiter's own execution.
constexpr int SPIN_MIN_POLLS = 1 << 20;
struct SpinClock {
    unsigned long long t0 = 0;
    // true once the wait has lasted longer than SPIN_TICKS (call once per poll)
    __device__ __forceinline__ bool expired(int polls)
    {
        if ((polls & (SPIN_CHECK - 1)) != SPIN_CHECK - 1) return false;
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (t0 == 0) { t0 = now | 1ull; return false; }
        return now - t0 > SPIN_TICKS && polls >= SPIN_MIN_POLLS;
    }
};

__device__ __forceinline__ uint32_t st_load(const uint32_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_store(uint32_t* p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum of `count` (< FAN) posted entries st[(first + i) * stride + d[k]], for the NB digits of this thread at once (all
// loads of a round in flight together); polls until every entry has been posted.
template <int NB>
__device__ __forceinline__ void sum_posted(const uint32_t* st, long first, int count, long stride, const int (&d)[NB],
                                           const bool (&on)[NB], uint32_t (&acc)[NB], uint32_t* err)
{
    if (count <= 0) return;
    SpinClock clk;
    for (int polls = 0;; ++polls) {
        uint32_t v[NB][FAN - 1];
#pragma unroll
        for (int k = 0; k < NB; ++k)
#pragma unroll
            for (int i = 0; i < FAN - 1; ++i)
                v[k][i] = (on[k] && i < count) ? st_load(&st[(first + i) * stride + d[k]]) : ST_POSTED;
        uint32_t all = ST_POSTED, s[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            s[k] = 0;
#pragma unroll
            for (int i = 0; i < FAN - 1; ++i) { all &= v[k][i]; s[k] += v[k][i] & ST_VALUE; }
        }
        // (a guard word that is already up -- another workgroup's wait ran out, or it found the view broken -- ends this
        // wait too: the view is lost, its workgroups must not sit out their own two seconds one after the other)
        const bool lost = (polls & (SPIN_CHECK - 1)) == SPIN_CHECK - 1 && st_load(err) != 0u;
        if (all || lost || clk.expired(polls)) {
            if (!all) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < NB; ++k) acc[k] += s[k];
            return;
        }
    }
}

// status words of one pass (levels 1-3) and of one sort (digit totals first, then every pass)
__host__ __device__ inline size_t onesweep_pass_words(long n, int digit)
{
    const size_t nblk = (size_t)((n + RADIX_BLOCK - 1) / RADIX_BLOCK), n2 = (nblk + FAN - 1) / FAN, n3 = (n2 + FAN - 1) / FAN;
    return (nblk + n2 + n3) << digit;
}
__host__ __device__ inline size_t onesweep_status_words(long n, int nbits)
{
    return HIST_WORDS + (size_t)radix_passes(nbits) * onesweep_pass_words(n, radix_digit(nbits));
}
__host__ __device__ inline size_t emit_status_words(long V)   // levels 1-3 of the emission scan (64-bit words)
{
    const size_t nblk = (size_t)((V + 255) / 256), n2 = (nblk + 63) / 64, n3 = (n2 + 63) / 64;   // (sized for a fan-in of 64: enough for the kernel's 128)
    return nblk + n2 + n3;
}
// The packed status region of one view: depth sort | emission scan | tile sort (bytes, multiples of 16).  Computed
// with the same arithmetic by the host (pointers) and by the compaction kernel (which clears the region).
struct StatusPlan { size_t depth, emit, tile; };
__host__ __device__ inline StatusPlan status_plan(long V, long R, int key_bits, int tile_bits)
{
    StatusPlan p;
    p.depth = (onesweep_status_words(V, key_bits) * 4 + 15) / 16 * 16;
    p.emit = (emit_status_words(V) * 8 + 15) / 16 * 16 + 16;     // (+ the counter of the big-rectangle list, emit_rects)
    p.tile = (onesweep_status_words(R, tile_bits) * 4 + 15) / 16 * 16;
    return p;
}

// h[d] += 1 for every lane with ok; lanes sharing the digit of the first one or two live lanes are counted with one
// add (sorted or clustered keys put most of a wave on one digit, and same-address LDS atomics serialise)
__device__ __forceinline__ void wave_hist_add(uint32_t* h, uint32_t d, bool ok, int lane)
{
    unsigned long long act = __ballot(ok);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        if (!act) break;
        const int leader = __builtin_ctzll(act);
        const uint32_t d0 = (uint32_t)__shfl((int)d, leader, 64);
        const unsigned long long m = __ballot(ok && d == d0) & act;
        if (lane == leader) atomicAdd(&h[d0], (uint32_t)__popcll(m));
        act &= ~m;
    }
    if ((act >> lane) & 1ull) atomicAdd(&h[d], 1u);
}
// all passes' digits of one key into the block's LDS histogram h[pass << digit | value]; the top digit is the
// clustered one (depth exponent, tile row), the lower ones are spread
__device__ __forceinline__ void hist_key(uint32_t* h, uint32_t rel_key, bool ok, int digit, int passes, int lane)
{
    const uint32_t mask = (1u << digit) - 1u;
    for (int p = 0; p + 1 < passes; ++p)
        if (ok) atomicAdd(&h[(p << digit) + ((rel_key >> (p * digit)) & mask)], 1u);
    if (passes > 0) {
        const int p = passes - 1;
        wave_hist_add(h + (p << digit), (rel_key >> (p * digit)) & mask, ok, lane);
    }
}

// Partial digit histograms of keys that were not counted where they were produced (the tile keys: counting them
// inside the emission kernel put LDS atomics -- ~1 lane per cycle per CU, ~90 cycles per instruction however few lanes
// are live -- on that kernel's critical path: 44 -> 97 us, whether in its per-lane loops or in a second dense sweep).
constexpr int HIST_THREADS = 1024;
constexpr int HIST_BLOCKS = 256;   // one per CU
__global__ void __launch_bounds__(HIST_THREADS)
k_digit_hist(const uint32_t* __restrict__ keys, long n, uint32_t kmin, int digit, int passes, uint32_t* __restrict__ partial)
{
    constexpr int BATCH = 8;   // keys in flight per thread
    __shared__ uint32_t h[HIST_WORDS];
    const int words = passes << digit;
    const long stride = (long)gridDim.x * HIST_THREADS;
    for (int d = threadIdx.x; d < words; d += HIST_THREADS) h[d] = 0;
    __syncthreads();
    const long n_up = (n + 63) / 64 * 64;   // whole waves enter an iteration together (hist_key uses ballots)
    for (long i0 = (long)blockIdx.x * HIST_THREADS + threadIdx.x; i0 < n_up; i0 += stride * BATCH) {
        uint32_t k[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const long i = i0 + j * stride;
            k[j] = i < n ? keys[i] : 0u;
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) hist_key(h, k[j] - kmin, i0 + j * stride < n, digit, passes, threadIdx.x & 63);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < words; d += HIST_THREADS) partial[(size_t)blockIdx.x * words + d] = h[d];
}

// Adds the G partial histograms up: workgroup (x, y) sums partials [32 y, 32 y + 32) for the 64 entries
// [64 x, 64 x + 64) of the passes << digit totals and adds the result to `tot` (cleared with the status region):
// G / 32 atomics per word instead of G.
__global__ void __launch_bounds__(256)
k_digit_sum(const uint32_t* __restrict__ partial, int G, int words, uint32_t* __restrict__ tot)
{
    __shared__ uint32_t red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + lane;
    const int g0 = blockIdx.y * 32 + w * 8;
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (d < words && g0 + j < G) ? partial[(size_t)(g0 + j) * words + d] : 0u;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += t[j];
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0 && d < words) {
        const uint32_t v = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        if (v) atomicAdd(&tot[d], v);
    }
}

template <int BITS>
__global__ void __launch_bounds__(256)
k_onesweep(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
           uint32_t* __restrict__ vals_out, long n, uint32_t kmin, int pass,
           const uint32_t* __restrict__ totals, uint32_t* __restrict__ status, uint32_t* __restrict__ err, int lose_block0)
{
    constexpr int SIZE = 1 << BITS;
    constexpr int BPT = (SIZE + 255) / 256;   // digits per thread in the per-digit phases
    __shared__ uint32_t cnt[4][SIZE];
    __shared__ uint32_t off[4][SIZE];
    __shared__ uint32_t loc[SIZE], gl[SIZE];
    __shared__ uint32_t skey[RADIX_BLOCK], sval[RADIX_BLOCK];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int shift = pass * BITS;
    const long b = blockIdx.x;
    // A wait that ran out in an EARLIER launch of this view (a pass before this one, visible in stream order) left keys and
    // values that are not a permutation of the input any more -- slots written twice, others holding stale or uninitialised
    // memory.  This pass would rank keys whose digit counts no longer add up to the totals it scatters by: destinations
    // beyond the buffers.  The view is lost either way (k_tile_ranges leaves it empty and reports it): nothing is touched.
    // The decision is the WORKGROUP's (one load, through LDS): were it every thread's own, a guard word going up while the
    // workgroup starts could send some of its waves home and leave the others ranking against half-initialised counters.
    __shared__ uint32_t s_tripped;
    if (threadIdx.x == 0) s_tripped = st_load(err);
    for (int d = threadIdx.x; d < SIZE; d += 256) {
#pragma unroll
        for (int k = 0; k < 4; ++k) cnt[k][d] = 0;
    }
    __syncthreads();
    if (s_tripped) return;
    const long wbase = b * RADIX_BLOCK + (long)w * (64 * RADIX_ITEMS);
    uint32_t key[RADIX_ITEMS], val[RADIX_ITEMS], rank[RADIX_ITEMS];
    const unsigned long long lt = lanemask_lt();
#pragma unroll
    for (int i = 0; i < RADIX_ITEMS; ++i) {
        long idx = wbase + i * 64 + lane;
        key[i] = idx < n ? keys_in[idx] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < RADIX_ITEMS; ++i) {
        long idx = wbase + i * 64 + lane;
        val[i] = idx < n ? vals_in[idx] : 0u;
    }
#pragma unroll
    for (int i = 0; i < RADIX_ITEMS; ++i) {
        long idx = wbase + i * 64 + lane;
        bool ok = idx < n;
        uint32_t d = digit_of<BITS>(key[i], kmin, shift);
        unsigned long long valid = __ballot(ok);
        unsigned long long m = match_digit<BITS>(d, valid);
        uint32_t prior = ok ? cnt[w][d] : 0;
        rank[i] = prior + (uint32_t)__popcll(m & lt);
        if (ok && (m & lt) == 0) cnt[w][d] = prior + (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- publish this block's digit counts, lay the block out in digit order
    uint32_t mine[BPT];
#pragma unroll
    for (int k = 0; k < BPT; ++k) {
        const int d = threadIdx.x + k * 256;
        mine[k] = 0;
        if (d < SIZE) {
            const uint32_t c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d], c3 = cnt[3][d];
            mine[k] = c0 + c1 + c2 + c3;
            // (lose_block0: test hook -- workgroup 0 never posts, as a lost workgroup would not: its successors' waits run
            // out FOR REAL, vr_debug_raise_guard(2))
            if (!(lose_block0 && b == 0)) st_store(&status[b * SIZE + d], mine[k] | ST_POSTED);
            off[0][d] = 0;
            off[1][d] = c0;
            off[2][d] = c0 + c1;
            off[3][d] = c0 + c1 + c2;
            loc[d] = mine[k];
            gl[d] = totals[pass * SIZE + d];
        }
    }
    __syncthreads();
    if (w < 2) {   // wave 0: exclusive scan of the block's bucket sizes; wave 1: of the global digit totals
        uint32_t* arr = w == 0 ? loc : gl;
        constexpr int DPL = SIZE / 64;
        uint32_t t[DPL], sum = 0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) { t[k] = arr[lane * DPL + k]; sum += t[k]; }
        uint32_t incl = sum;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, dd, 64);
            if (lane >= dd) incl += o;
        }
        uint32_t run = incl - sum;
#pragma unroll
        for (int k = 0; k < DPL; ++k) { arr[lane * DPL + k] = run; run += t[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RADIX_ITEMS; ++i) {
        long idx = wbase + i * 64 + lane;
        if (idx < n) {
            uint32_t d = digit_of<BITS>(key[i], kmin, shift);
            uint32_t p = loc[d] + off[w][d] + rank[i];
            skey[p] = key[i];
            sval[p] = val[i];
        }
    }
    // ---- elements with each digit in all earlier blocks: <= 15 posted sums from each of the three levels
    {
        const long nblk = gridDim.x, n2 = (nblk + FAN - 1) / FAN;
        uint32_t* const level2 = status + nblk * SIZE;
        uint32_t* const level3 = level2 + n2 * SIZE;
        int d[BPT];
        bool on[BPT];
        uint32_t before[BPT];
#pragma unroll
        for (int k = 0; k < BPT; ++k) { d[k] = threadIdx.x + k * 256; on[k] = d[k] < SIZE; before[k] = 0; }
        const long g1 = b / FAN, g2 = g1 / FAN;
        const int r1 = (int)(b % FAN), r2 = (int)(g1 % FAN), r3 = (int)g2;
        sum_posted<BPT>(status, b - r1, r1, SIZE, d, on, before, err);
        if (r1 == FAN - 1) {
#pragma unroll
            for (int k = 0; k < BPT; ++k)
                if (on[k]) st_store(&level2[g1 * SIZE + d[k]], (before[k] + mine[k]) | ST_POSTED);
        }
        uint32_t upper[BPT];
#pragma unroll
        for (int k = 0; k < BPT; ++k) upper[k] = 0;
        sum_posted<BPT>(level2, g1 - r2, r2, SIZE, d, on, upper, err);
        if (r1 == FAN - 1 && r2 == FAN - 1) {
#pragma unroll
            for (int k = 0; k < BPT; ++k)
                if (on[k]) st_store(&level3[g2 * SIZE + d[k]], (before[k] + mine[k] + upper[k]) | ST_POSTED);
        }
        sum_posted<BPT>(level3, 0, r3, SIZE, d, on, upper, err);
#pragma unroll
        for (int k = 0; k < BPT; ++k)
            if (on[k]) gl[d[k]] += before[k] + upper[k];
    }
    __syncthreads();
    const long bbase = b * RADIX_BLOCK;
    const int nvalid = (int)((n - bbase) < (long)RADIX_BLOCK ? (n - bbase) : (long)RADIX_BLOCK);
    for (int j = threadIdx.x; j < nvalid; j += 256) {
        const uint32_t k = skey[j];
        const uint32_t d = digit_of<BITS>(k, kmin, shift);
        const uint32_t dst = gl[d] + ((uint32_t)j - loc[d]);
        const uint32_t v = sval[j];
        // (sums that do not add up -- a wait that gave up in THIS launch -- must not become a store beyond the buffers)
        if (dst < (uint32_t)n) {
            keys_out[dst] = k;
            vals_out[dst] = v;
        }
    }
}

// Stable LSD sort like radix_sort(), one launch per pass plus one that adds the partial digit histograms up
// (`partial`: `rows` rows of passes << digit words, written by the kernel that produced the keys).  `status`:
// onesweep_status_words(n, nbits) words, all zero.
static int onesweep_sort(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, long n, uint32_t kmin, int nbits,
                         const uint32_t* partial, int rows, uint32_t* status, uint32_t* err, hipStream_t s, bool debug,
                         int* res, bool lose_block0 = false)
{
    const int digit = radix_digit(nbits);
    const int passes = radix_passes(nbits);
    const int nblk = cdiv(n, RADIX_BLOCK);
    const size_t per_pass = onesweep_pass_words(n, digit);
    *res = 0;
    if (passes == 0) return 0;
    uint32_t* const totals = status;   // digit totals at the head of the status region
    status += HIST_WORDS;
    hipLaunchKernelGGL(k_digit_sum, dim3(cdiv((long)passes << digit, 64), cdiv(rows, 32)), dim3(256), 0, s, partial, rows,
                       passes << digit, totals);
    VR_KERNEL_CHECK("digit_sum", s, debug);
    uint32_t *ka = k0, *va = v0, *kb = k1, *vb = v1;
    for (int pass = 0; pass < passes; ++pass) {
        uint32_t* st = status + (size_t)pass * per_pass;
#define VR_SWEEP(B)                                                                                           \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_onesweep<B>), dim3(nblk), dim3(256), 0, s, (const uint32_t*)ka,      \
                       (const uint32_t*)va, kb, vb, n, kmin, pass, (const uint32_t*)totals, st, err,        \
                       (lose_block0 && pass == 0) ? 1 : 0)
        if (digit == 6) VR_SWEEP(6);
        else if (digit == 9) VR_SWEEP(9);
        else VR_SWEEP(8);
#undef VR_SWEEP
        VR_KERNEL_CHECK("onesweep", s, debug);
        uint32_t* t = ka; ka = kb; kb = t;
        t = va; va = vb; vb = t;
        *res ^= 1;
    }
    return 0;
}

// ---------------------------------------------------------------- stage 1, second half: compaction

// Writes the visible Gaussians' (depth key, id) in id order.  On the side (each saves a launch): the partial digit
// histogram of this block's depth keys -- the digit plan follows from the totals the previous kernel left in device
// memory, the host learns them in parallel --, and the clearing of the tile ranges and of the view's status region.
__global__ void __launch_bounds__(256)
k_compact_apply(const uint4* __restrict__ rect, const uint32_t* __restrict__ depth_key, long n,
                const uint32_t* __restrict__ bsum, const uint32_t* __restrict__ totals, int tile_bits,
                uint32_t* __restrict__ vis_key, uint32_t* __restrict__ vis_id, uint32_t* __restrict__ partial,
                uint32_t* __restrict__ zero_a, long zero_na, uint4* __restrict__ status, long status_bytes,
                uint2* __restrict__ chk_in)
{
    __shared__ uint32_t lds4[4];
    __shared__ uint32_t chk_lds[8];
    __shared__ uint32_t h[HIST_WORDS];
    const uint32_t V = totals[0], R = totals[1], kmin = totals[2];
    const int key_bits = V ? bits_of(totals[3] - kmin) : 0;
    const int digit = radix_digit(key_bits), passes = radix_passes(key_bits), words = passes << digit;
    const long tid = (long)blockIdx.x * 256 + threadIdx.x, nthr = (long)gridDim.x * 256;
    for (long i = tid; i < zero_na; i += nthr) zero_a[i] = 0u;
    if (status && (long)V <= ONESWEEP_MAX_N && (long)R <= ONESWEEP_MAX_N) {
        const StatusPlan sp = status_plan(V, R, key_bits, tile_bits);
        const long n16 = (long)((sp.depth + sp.emit + sp.tile) / 16);
        // `status` was sized from the caller's capacity hint BEFORE R was known: when the actual R outgrows it the
        // region is not cleared here (the host sees R > capacity, allocates afresh and clears with a fill)
        if (n16 * 16 <= status_bytes)
            for (long i = tid; i < n16; i += nthr) status[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    for (int d = threadIdx.x; d < words; d += 256) h[d] = 0;
    uint32_t acc = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) acc += bsum[j];
    const uint32_t prefix = block_sum(acc, lds4);   // (has the barriers that also publish h)
    // Element (k, t) = block start + 256 k + t: every load of a wave is one contiguous run (with SCAN_ITEMS consecutive
    // elements per thread it touched 64 lines).  The values are FLAGS, so the ordered scan is ballots and popcounts: the
    // position of a visible element = the visible ones before it in its wave's row + the counts of the (row, wave) pairs
    // in front, 32 numbers in LDS.
    __shared__ uint32_t rowcnt[SCAN_ITEMS][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long base = (long)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    uint32_t v[SCAN_ITEMS], key[SCAN_ITEMS], before[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {      // (who has list entries is in the key itself: 4 bytes per Gaussian, not the rectangle's 16)
        const long e = base + (long)k * 256;
        key[k] = e < n ? depth_key[e] : DEPTH_KEY_NONE;
        v[k] = key[k] != DEPTH_KEY_NONE ? 1u : 0u;
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) key[k] = v[k] ? key[k] : 0u;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const unsigned long long m = __ballot(v[k] != 0u);
        before[k] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) rowcnt[k][w] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    uint32_t run = prefix, c1 = 0u, c2 = 0u;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            if (ww == w && v[k]) {
                const uint32_t ex = run + before[k];
                const uint32_t id = (uint32_t)(base + (long)k * 256);
                vis_key[ex] = key[k];
                vis_id[ex] = id;
                c1 += id; c2 += perm_mix(id);
            }
            run += rowcnt[k][ww];
        }
        hist_key(h, key[k] - kmin, v[k] != 0u, digit, passes, lane);
    }
    block_store_pair(c1, c2, chk_in + blockIdx.x, chk_lds);     // (its barrier also completes h)
    for (int d = threadIdx.x; d < words; d += 256) partial[(size_t)blockIdx.x * words + d] = h[d];
}

int launch_compact_apply(int P, const uint4* rect, const uint32_t* depth_key, void* scratch, const uint32_t* totals_dev,
                         int tile_bits, uint32_t* vis_key, uint32_t* vis_id, uint32_t* zero_a, long zero_na,
                         void* status, size_t status_bytes, hipStream_t s, bool debug)
{
    int nb = cdiv(P > 0 ? P : 1, SCAN_BLOCK);
    uint32_t* bsum = (uint32_t*)scratch;
    uint32_t* partial = (uint32_t*)((char*)scratch + stage1_partial_offset((size_t)nb));
    uint2* chk_in = (uint2*)((char*)scratch + stage1_chk_offset((size_t)nb));
    hipLaunchKernelGGL(k_compact_apply, dim3(nb), dim3(256), 0, s, rect, depth_key, (long)P, (const uint32_t*)bsum,
                       totals_dev, tile_bits, vis_key, vis_id, partial, zero_a, zero_na, (uint4*)status,
                       (long)status_bytes, chk_in);
    VR_KERNEL_CHECK("compact_apply", s, debug);
    return 0;
}

// ---------------------------------------------------------------- emission + ranges

// One block per 256 depth-sorted Gaussians; lanes are spread over OUTPUT entries (binary search in
// the block's LDS prefix), so long rectangles do not serialise a lane and writes are coalesced.
// the only gather of the binning stage: rectangles of the depth-sorted Gaussians (8 B each)
__global__ void __launch_bounds__(256)
k_gather_rect(int V, const uint32_t* __restrict__ sorted_id, const uint4* __restrict__ rect,
              uint4* __restrict__ rect_sorted)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < V) rect_sorted[r] = rect[sorted_id[r]];
}

// Emission of the (tile, id) pairs of 256 depth-sorted Gaussians per workgroup.  Small rectangles
// (<= EMIT_SMALL tiles: the vast majority, far splats touch 1-4 tiles) are written by their own lane --
// consecutive lanes own consecutive output ranges, so the stores stay nearly coalesced; a large
// rectangle is written by its whole wave, 64 entries per step.
constexpr int EMIT_SMALL = 8;

// Rectangles of more than 64 tiles are NOT emitted by the wave that owns them: the depth order puts the nearest Gaussians
// -- the ones with screen-sized rectangles -- side by side, and the first wave of a street view walked 200 ... 300 64-tile
// steps while the average wave has 0.2 (k_emit_scan waited for that one wave).  They are appended to a list, and
// k_emit_big gives every one of them a wave of its own.
struct BigRects {
    uint32_t* count;     // entries so far (cleared with the status words of the view)
    uint2* items;        // {first output position, Gaussian id}
    int inline_big;      // few such rectangles in the view: their own waves emit them, no list, no k_emit_big
};

// The (tile, id) pairs of one wave's 64 rectangles, `cnt` kept tiles each (rect_area) from output position `off` on.
// Small ones (<= EMIT_SMALL kept tiles: the vast majority) are written by their own lane, one of 9 ... 64 tiles by the whole
// wave, larger ones are listed for k_emit_big.  Within a rectangle the kept tiles go out in row-major order (the reference's
// emission order, minus the tiles the mask drops).
__device__ __forceinline__ void emit_rects(uint4 rc, uint32_t cnt, uint32_t off, uint32_t id, int gx, int lane,
                                           uint32_t* __restrict__ tkeys, uint32_t* __restrict__ tvals, BigRects big_list)
{
    const int x0 = (int)(rc.x & 0xFFFFu), y0 = (int)(rc.x >> 16), w = max((int)(rc.y & 0xFFFFu), 1), h = (int)(rc.y >> 16);
    const uint32_t area = (uint32_t)(w * h);
    const unsigned long long mask = rect_mask(rc);
    const bool huge = cnt > 0 && area > (uint32_t)TIGHT_MAX_TILES;
    const bool small = cnt > 0 && cnt <= (uint32_t)EMIT_SMALL && !huge;
    if (small) {
        int rx = 0, ry = 0;
        uint32_t k = 0;
        for (uint32_t j = 0; k < cnt; ++j) {
            if ((mask >> j) & 1ull) {
                tkeys[off + k] = (uint32_t)((y0 + ry) * gx + x0 + rx);
                tvals[off + k] = id;
                ++k;
            }
            if (++rx == w) { rx = 0; ++ry; }
        }
    }
    if (big_list.inline_big) {
    // rectangles of more than 64 tiles by their own wave (few of them: emit_big_inline), with division-free index arithmetic:
    // floor((k + 1/2) / w) in fp32 is exact for k < 2^22 -- used below 2^21 -- (the error of the product, (k / w) 2^-23, stays below the 1 / (2 w)
    // that separates (k + 1/2) / w from the nearest integer)
    {
        int kc_l = 1, cw_l = w, ch_l = h;
        if (huge) tile_cells(w, h, kc_l, cw_l, ch_l);
        for (unsigned long long hm = __ballot(huge); hm; hm &= hm - 1) {
            const int src = __builtin_ctzll(hm);
            const uint32_t b_off = (uint32_t)__builtin_amdgcn_readlane((int)off, src);
            const uint32_t b_id = (uint32_t)__builtin_amdgcn_readlane((int)id, src);
            const int b_x0 = __builtin_amdgcn_readlane(x0, src), b_y0 = __builtin_amdgcn_readlane(y0, src);
            const int b_w = __builtin_amdgcn_readlane(w, src);
            const uint32_t b_area = (uint32_t)__builtin_amdgcn_readlane((int)area, src);
            const uint32_t b_mask = (uint32_t)__builtin_amdgcn_readlane((int)rc.z, src);
            const int kc = __builtin_amdgcn_readlane(kc_l, src), cw = __builtin_amdgcn_readlane(cw_l, src);
            const float inv_w = 1.0f / (float)b_w, inv_k = 1.0f / (float)kc;
            uint32_t done = 0;
            for (uint32_t base = 0; base < b_area; base += 64) {
                const uint32_t k = base + (uint32_t)lane;
                int ry, rx, cell;
                if (b_area < (1u << 21)) {
                    ry = (int)(((float)k + 0.5f) * inv_w);
                    rx = (int)k - ry * b_w;
                    cell = (int)(((float)ry + 0.5f) * inv_k) * cw + (int)(((float)rx + 0.5f) * inv_k);
                } else {
                    ry = (int)(k / (uint32_t)b_w);
                    rx = (int)k - ry * b_w;
                    cell = (ry / kc) * cw + rx / kc;
                }
                const bool keep = k < b_area && ((b_mask >> (cell & 31)) & 1u);
                const unsigned long long km = __ballot(keep);
                if (keep) {
                    const uint32_t pos = b_off + done + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
                    tkeys[pos] = (uint32_t)((b_y0 + ry) * gx + b_x0 + rx);
                    tvals[pos] = b_id;
                }
                done += (uint32_t)__popcll(km);
            }
        }
    }
    } else {
        const unsigned long long hm = __ballot(huge);
        if (hm) {
            uint32_t base = 0;
            if (lane == __builtin_ctzll(hm)) base = atomicAdd(big_list.count, (uint32_t)__popcll(hm));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(hm));
            if (huge) big_list.items[base + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = make_uint2(off, id);
        }
    }
    // 9 ... 64 tiles: one at a time, all 64 lanes of the wave (a cell is a tile: positions from the mask itself)
    for (unsigned long long big = __ballot(cnt > 0 && !small && !huge); big; big &= big - 1) {
        const int src = __builtin_ctzll(big);
        const uint32_t b_off = (uint32_t)__builtin_amdgcn_readlane((int)off, src);
        const uint32_t b_id = (uint32_t)__builtin_amdgcn_readlane((int)id, src);
        const int b_x0 = __builtin_amdgcn_readlane(x0, src), b_y0 = __builtin_amdgcn_readlane(y0, src);
        const int b_w = __builtin_amdgcn_readlane(w, src);
        const uint32_t b_area = (uint32_t)__builtin_amdgcn_readlane((int)area, src);
        const unsigned long long b_mask = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)rc.w, src) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane((int)rc.z, src);
        const uint32_t k = (uint32_t)lane;
        if (k < b_area && ((b_mask >> k) & 1ull)) {
            const uint32_t pos = b_off + (uint32_t)__popcll(b_mask & ((1ull << k) - 1ull));
            const int ry = (int)(k / (uint32_t)b_w), rx = (int)(k - (uint32_t)ry * (uint32_t)b_w);
            tkeys[pos] = (uint32_t)((b_y0 + ry) * gx + b_x0 + rx);
            tvals[pos] = b_id;
        }
    }
}
// Who emits the rectangles of more than 64 tiles?  Listed for k_emit_big they cost a launch (7 us); emitted by their own
// waves they cost nothing while there are few of them (a street view: ~300, all in the first workgroups) and a lot when most
// waves hold some (emit stage, us, inline / listed -- discs x1: 63 / 70, x1.5: 88 / 74, x2: 121 / 79, x3: 177 / 101; the four
// views have 120 ... 290 / ~1.1 k / ~5.8 k / 26 k of them).  Their number comes back with the list sizes (k_scan_totals):
// inline below 384.
static inline int emit_big_inline(uint32_t n_huge) { return n_huge < 384u; }
// One wave per listed rectangle (more than 64 tiles): 64 tiles per step, those of the mask's kept cells written.
constexpr int EMIT_BIG_GRID = 2048;
__global__ void __launch_bounds__(64)
k_emit_big(BigRects big_list, const uint4* __restrict__ rect, int gx, uint32_t* __restrict__ tkeys,
           uint32_t* __restrict__ tvals)
{
    const uint32_t n = *big_list.count;
    const int lane = threadIdx.x;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint2 it = big_list.items[i];
        const uint4 rc = rect[it.y];
        const int x0 = (int)(rc.x & 0xFFFFu), y0 = (int)(rc.x >> 16), w = max((int)(rc.y & 0xFFFFu), 1), h = (int)(rc.y >> 16);
        const uint32_t area = (uint32_t)(w * h), mask = rc.z;
        int kc, cw, ch;
        tile_cells(w, h, kc, cw, ch);
        uint32_t done = 0;
        for (uint32_t base = 0; base < area; base += 64) {
            const uint32_t k = base + (uint32_t)lane;
            const int ry = (int)(k / (uint32_t)w), rx = (int)(k - (uint32_t)ry * (uint32_t)w);
            const int cell = (ry / kc) * cw + rx / kc;
            const bool keep = k < area && ((mask >> (cell & 31)) & 1u);
            const unsigned long long km = __ballot(keep);
            if (keep) {
                const uint32_t pos = it.x + done + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
                tkeys[pos] = (uint32_t)((y0 + ry) * gx + x0 + rx);
                tvals[pos] = it.y;
            }
            done += (uint32_t)__popcll(km);
        }
    }
}
__global__ void __launch_bounds__(256)
k_emit(int V, int gx, const uint32_t* __restrict__ sorted_id, const uint32_t* __restrict__ offs,
       const uint4* __restrict__ rect_sorted, uint32_t* __restrict__ tkeys, uint32_t* __restrict__ tvals, BigRects big_list)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t off = 0, cnt = 0, id = 0;
    uint4 rc = make_uint4(0u, 0u, 0u, 0u);
    if (r < V) {
        rc = rect_sorted[r];
        off = offs[r];
        cnt = rect_area(rc);
        id = sorted_id[r];
    }
    emit_rects(rc, cnt, off, id, gx, lane, tkeys, tvals, big_list);
}

// The same emission with the offset scan folded in: a workgroup sums its 256 rectangle areas, posts the sum and adds
// what the workgroups before it posted -- the single-value form of the three-level scheme of the radix passes, fan-in
// 128 (two status words per lane of wave 0 and level).  64-bit words, top bit = posted.
constexpr unsigned long long SE_POSTED = 1ull << 63;
constexpr int EFAN = 128;   // (64: 6400 workgroups need all three levels; 128: two cover 16384)
constexpr long EMIT_SCAN_MAX_V = (long)EFAN * EFAN * EFAN * 256;

// wave-wide: sum of `count` (< 64) posted entries st[first + lane]; polls until all are posted
__device__ __forceinline__ unsigned long long sum_posted_wave(const unsigned long long* st, long first, int count, int lane,
                                                              uint32_t* err)
{
    if (count <= 0) return 0ull;
    constexpr int PER = EFAN / 64;   // status words per lane
    SpinClock clk;
    for (int polls = 0;; ++polls) {
        unsigned long long v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j)
            v[j] = lane + 64 * j < count
                       ? __hip_atomic_load(&st[first + lane + 64 * j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                       : SE_POSTED;
        unsigned long long both = SE_POSTED;
#pragma unroll
        for (int j = 0; j < PER; ++j) both &= v[j];
        const bool all = __ballot((both & SE_POSTED) == 0ull) == 0ull;
        const bool lost = (polls & (SPIN_CHECK - 1)) == SPIN_CHECK - 1 && st_load(err) != 0u;     // (see sum_posted)
        if (all || lost || clk.expired(polls)) {
            if (!all && lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long s = 0ull;
#pragma unroll
            for (int j = 0; j < PER; ++j) s += v[j] & ~SE_POSTED;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
            return s;
        }
    }
}

// ITEMS batches of 256 depth-sorted Gaussians per workgroup (round 5).  The kernel is bound by the latency of its dependent
// round trips -- id -> rectangle gather -> posted sums of the workgroups before it -- and a view's 6.4 k workgroups of 256
// came in three resident sets; with 2 batches per workgroup the gathers of both are in flight together, half as many
// workgroups post and look back, and the sets go from three to two.  Emission order and positions are what they were: batch
// k of the workgroup lies behind batch k - 1.
template <int ITEMS>
__global__ void __launch_bounds__(256)
k_emit_scan(int V, int P, long R, int gx, const uint32_t* __restrict__ sorted_id, const uint4* __restrict__ rect,
            unsigned long long* __restrict__ status, uint32_t* __restrict__ err, uint32_t* __restrict__ tkeys,
            uint32_t* __restrict__ tvals, BigRects big_list, uint2* __restrict__ chk_out)
{
    __shared__ uint32_t lds4[4];
    __shared__ uint32_t chk_lds[8];
    __shared__ unsigned long long s_before;
    const int lane = threadIdx.x & 63;
    uint32_t cnt[ITEMS], id[ITEMS];
    uint4 rc[ITEMS];
    // A wait of this view's depth sort ran out: `sorted_id` may hold anything.  The workgroup does NOT leave (threads deciding
    // one by one could leave half a workgroup behind its barriers): it takes part in everything with empty rectangles.
    const bool dead = st_load(err) != 0u;
    uint32_t c1 = 0u, c2 = 0u;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int r = (blockIdx.x * ITEMS + k) * 256 + threadIdx.x;
        cnt[k] = 0; id[k] = 0;
        if (r < V && !dead) { id[k] = sorted_id[r]; c1 += id[k]; c2 += perm_mix(id[k]); }
    }
    // (the ids as they came out of the sort, whatever they are: the compaction's sums must come back -- PERMUTATION CHECK)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { c1 += __shfl_xor(c1, d, 64); c2 += __shfl_xor(c2, d, 64); }
    if (lane == 0) { chk_lds[threadIdx.x >> 6] = c1; chk_lds[4 + (threadIdx.x >> 6)] = c2; }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int r = (blockIdx.x * ITEMS + k) * 256 + threadIdx.x;
        rc[k] = make_uint4(0u, 0u, 0u, 0u);
        if (dead) continue;
        // the packed rectangle + tile mask is gathered by id HERE (1.65 M random 16-byte reads): this kernel is bound by
        // the latency of its waits, not by bandwidth, and hides them; in the last depth-sort pass they cost 22 us
        // An id beyond the model cannot come out of a sound depth sort: it is not used as an index (the gather would
        // fault the process) but raises the view's guard word -- the view is reported as failed, like a timed-out wait.
        if (r < V && id[k] >= (uint32_t)P) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); id[k] = 0u; }
        else if (r < V) { rc[k] = rect[id[k]]; cnt[k] = rect_area(rc[k]); }
    }
    uint32_t ex[ITEMS], total = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        uint32_t t;
        ex[k] = block_excl_scan(cnt[k], t, lds4) + total;
        total += t;
    }
    if (threadIdx.x < 64) {
        const long b = blockIdx.x, nblk = gridDim.x, n2 = (nblk + EFAN - 1) / EFAN;
        unsigned long long* const level2 = status + nblk;
        unsigned long long* const level3 = level2 + n2;
        if (lane == 0)
            __hip_atomic_store(&status[b], (unsigned long long)total | SE_POSTED, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        const long g1 = b / EFAN, g2 = g1 / EFAN;
        const int r1 = (int)(b % EFAN), r2 = (int)(g1 % EFAN), r3 = (int)g2;
        const unsigned long long within = sum_posted_wave(status, b - r1, r1, lane, err);
        if (r1 == EFAN - 1 && lane == 0)
            __hip_atomic_store(&level2[g1], (within + total) | SE_POSTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long upper = sum_posted_wave(level2, g1 - r2, r2, lane, err);
        if (r1 == EFAN - 1 && r2 == EFAN - 1 && lane == 0)
            __hip_atomic_store(&level3[g2], (within + total + upper) | SE_POSTED, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        upper += sum_posted_wave(level3, 0, r3, lane, err);
        if (lane == 0) s_before = within + upper;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        chk_out[blockIdx.x] = make_uint2(chk_lds[0] + chk_lds[1] + chk_lds[2] + chk_lds[3], chk_lds[4] + chk_lds[5] + chk_lds[6] + chk_lds[7]);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        // Entries that would land beyond the R the lists were sized for: the ids are not the compaction's (repeated ones
        // repeat their rectangles) or a posted sum was wrong.  Nothing is written there; the view fails.
        const unsigned long long first = s_before + ex[k];
        if (cnt[k] != 0u && first + cnt[k] > (unsigned long long)R) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cnt[k] = 0u;
        }
        emit_rects(rc[k], cnt[k], (uint32_t)first, id[k], gx, lane, tkeys, tvals, big_list);
    }
}

// The PERMUTATION CHECK's verdict (see perm_mix): sums of the compaction's per-workgroup pairs against the emission's.  One
// workgroup; on a mismatch the guard word goes up.  Returns (to every thread) whether the view is failed.
__device__ __forceinline__ bool perm_check_failed(const uint2* __restrict__ chk_in, int n_in, const uint2* __restrict__ chk_out,
                                                  int n_out, uint32_t* err, uint32_t* lds /* [2 * waves + 1] */)
{
    uint32_t a1 = 0u, a2 = 0u;
    for (int j = threadIdx.x; j < n_in; j += blockDim.x) { const uint2 c = chk_in[j]; a1 += c.x; a2 += c.y; }
    for (int j = threadIdx.x; j < n_out; j += blockDim.x) { const uint2 c = chk_out[j]; a1 -= c.x; a2 -= c.y; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { a1 += __shfl_xor(a1, d, 64); a2 += __shfl_xor(a2, d, 64); }
    const int nw = (int)blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { lds[threadIdx.x >> 6] = a1; lds[nw + (threadIdx.x >> 6)] = a2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t1 = 0u, t2 = 0u;
        for (int q = 0; q < nw; ++q) { t1 += lds[q]; t2 += lds[nw + q]; }
        const uint32_t bad = (t1 | t2) != 0u ? 1u : 0u;
        if (bad) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds[2 * nw] = bad;
    }
    __syncthreads();
    return lds[2 * nw] != 0u;
}
// ... as a launch of its own in front of k_tile_ranges (frames of more than SPLIT_MAX_TILES tiles; k_split_base does it itself)
__global__ void __launch_bounds__(1024)
k_perm_check(const uint2* __restrict__ chk_in, int n_in, const uint2* __restrict__ chk_out, int n_out, uint32_t* __restrict__ err)
{
    __shared__ uint32_t lds[33];
    (void)perm_check_failed(chk_in, n_in, chk_out, n_out, err, lds);
}
// test hook (vr_debug_raise_guard(4)): the sorted ids stop being a permutation -- one of them is written over its neighbour
__global__ void k_debug_break_permutation(uint32_t* sorted_id, int V)
{
    if (V > 1) sorted_id[V / 2] = sorted_id[V / 2 - 1];
}

// Also (thread 0 of the launch): the look-back guard word as it stands after ALL waiting passes of this view, posted
// with this forward's sequence number into the host's pinned ring slot -- the matching vr_backward (or any later call
// that uses the lists) reads it there, so a timed-out wait fails the SAME view, not a later one.
__global__ void __launch_bounds__(256)
k_tile_ranges(const uint32_t* __restrict__ tkeys, long R, int2* __restrict__ ranges, const uint32_t* __restrict__ err,
              uint32_t* __restrict__ post, uint32_t seq)
{
    long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j == 0 && post) {
        const uint32_t g = err ? __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        __hip_atomic_store(&post[1], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&post[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (j >= R) return;
    // A timed-out wait left the lists of THIS view invalid (keys may be stale memory): leave every tile empty -- the ranges
    // were cleared by the compaction -- so that nothing downstream indexes with them; the view is reported through the ring.
    if (err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    uint32_t k = tkeys[j];
    if (j == 0) ranges[k].x = 0;
    else {
        uint32_t kp = tkeys[j - 1];
        if (kp != k) { ranges[kp].y = (int)j; ranges[k].x = (int)j; }
    }
    if (j == R - 1) ranges[k].y = (int)R;
}

// ---------------------------------------------------------------- tile split: the stable sort by tile id in ONE scatter
//
// Until round 5 the (tile, id) pairs were sorted by two 6-bit onesweep passes (+ the digit histogram of the keys, its sum, and
// k_tile_ranges over the sorted keys: 10 + 5 + 2 x 19 + 6.5 us on the headline view, all of it latency per launch).  A frame
// has at most a few thousand tiles, so the whole key fits ONE digit -- too many bins for posted per-bin sums (4,096 words per
// workgroup and level), but not for a count table:
//   k_split_count    workgroup b counts the tiles of its SPLIT_BLOCK consecutive pairs in LDS -> table[b][tile]
//   k_split_scan     per strip of 16 tiles, down the table: table[b][t] <- pairs of tile t in the workgroups before b; totals[t]
//   k_split_base     one workgroup: base[t] = pairs of the tiles before t; ranges[t] = [base, base + total) -- k_tile_ranges'
//                    output without the sorted keys, which nobody else reads -- and the view's guard word for the host
//   k_split_scatter  workgroup b ranks its pairs again (per-wave counters in LDS, ballot matches on the 12 bits) and writes
//                    each id to base[t] + table[b][t] + (pairs of t in the waves before) + (rank in the wave): stable.
// No workgroup waits for another.  Used for frames of up to SPLIT_MAX_TILES tiles; larger ones keep the two passes.
constexpr int SPLIT_BLOCK = 4096;                 // pairs per workgroup (16 per thread)
constexpr int SPLIT_ITEMS = SPLIT_BLOCK / 256;
constexpr int SPLIT_MAX_TILES = 3264;             // (k_split_scatter: 5 x 4 bytes of LDS per tile, 64 KB)
__host__ __device__ inline int split_stride(int ntiles) { return (ntiles + 63) & ~63; }      // table row length (words)

__global__ void __launch_bounds__(256)
k_split_count(const uint32_t* __restrict__ tkeys, long R, int ntiles, int stride, uint32_t* __restrict__ table,
              uint32_t* __restrict__ err)
{
    extern __shared__ uint32_t split_lds[];
    uint32_t* const h = split_lds;                 // [stride]
    __shared__ uint32_t s_tripped;                 // (a wait of this view's depth sort / emission ran out: the keys may be anything;
    if (threadIdx.x == 0) s_tripped = st_load(err);   //  the workgroup decides as one, see k_onesweep)
    for (int d = threadIdx.x; d < stride; d += 256) h[d] = 0u;
    __syncthreads();
    if (s_tripped) return;
    const long base = (long)blockIdx.x * SPLIT_BLOCK;
    uint32_t k[SPLIT_ITEMS];
#pragma unroll
    for (int i = 0; i < SPLIT_ITEMS; ++i) {
        const long idx = base + i * 256 + threadIdx.x;
        k[i] = idx < R ? tkeys[idx] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < SPLIT_ITEMS; ++i) {
        if (k[i] < (uint32_t)ntiles) atomicAdd(&h[k[i]], 1u);
        else if (k[i] != 0xFFFFFFFFu) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // not a tile: fail the view
    }
    __syncthreads();
    uint32_t* const row = table + (size_t)blockIdx.x * stride;
    for (int d = threadIdx.x; d < stride; d += 256) row[d] = h[d];
}

// 16 tiles x 16 row-lanes per step: thread (r, c) takes SCAN_U consecutive rows of column c, the 16 row-lanes are scanned
// through LDS, the column's running total carries over to the next step.  In place: counts -> exclusive prefixes.
// (16 rows per thread and step instead of 8: the same 9 us.  k_split_base run by the LAST workgroup of this launch -- a
// counter, device-scope totals -- instead of a launch of its own: 13.6 against 9.2 + 4.9 us; two launches it stays.)
constexpr int SPLIT_SCAN_U = 8;
constexpr int PERM_BLOCKS = 8;      // extra workgroups of k_split_scan that add up the permutation check's checksums
__global__ void __launch_bounds__(256)
k_split_scan(uint32_t* __restrict__ table, int nblk, int stride, uint32_t* __restrict__ totals, const uint2* __restrict__ chk_in,
             int n_in, const uint2* __restrict__ chk_out, int n_out, uint2* __restrict__ perm_part)
{
    __shared__ uint32_t part[16][17];
    // PERM_BLOCKS workgroups more than there are strips: the depth sort's PERMUTATION CHECK rides along -- each of them adds up
    // an eighth of the two checksum arrays (in - out) under the strips' column walks and leaves its pair in perm_part[];
    // k_split_base, the next launch, adds the eight pairs and raises the guard word if they do not cancel.  (As a step of
    // k_split_base alone -- one workgroup on the critical path, ~8 k loads -- the check cost 3.5 us; as ONE extra workgroup
    // here 2.8 us: the launch waited for it.)
    if ((int)blockIdx.x >= stride / 16) {
        __shared__ uint32_t chk_lds[8];
        const int part = (int)blockIdx.x - stride / 16;
        uint32_t a1 = 0u, a2 = 0u;
        if (chk_in) {
            for (int j = part * 256 + threadIdx.x; j < n_in; j += PERM_BLOCKS * 256) { const uint2 c = chk_in[j]; a1 += c.x; a2 += c.y; }
            for (int j = part * 256 + threadIdx.x; j < n_out; j += PERM_BLOCKS * 256) { const uint2 c = chk_out[j]; a1 -= c.x; a2 -= c.y; }
        }
        block_store_pair(a1, a2, perm_part + part, chk_lds);
        return;
    }
    const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + c;
    uint32_t carry = 0;
    for (int row0 = 0; row0 < nblk; row0 += 16 * SPLIT_SCAN_U) {
        const int mine = row0 + r * SPLIT_SCAN_U;
        uint32_t v[SPLIT_SCAN_U], sum = 0;
#pragma unroll
        for (int u = 0; u < SPLIT_SCAN_U; ++u) {
            v[u] = (mine + u < nblk) ? table[(size_t)(mine + u) * stride + col] : 0u;
            sum += v[u];
        }
        part[r][c] = sum;
        __syncthreads();
        uint32_t before = carry, all = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t t = part[q][c];
            if (q < r) before += t;
            all += t;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SPLIT_SCAN_U; ++u) {
            if (mine + u < nblk) table[(size_t)(mine + u) * stride + col] = before;
            before += v[u];
        }
        carry += all;
    }
    if (r == 0) totals[col] = carry;
}

// base[t] = pairs of the tiles before t; ranges[t] = [base, base + total) for the tiles that have pairs (the others stay
// cleared) -- and (thread 0) the look-back guard word as it stands after ALL waiting passes of this view, posted with this
// forward's sequence number into the host's pinned ring slot (what k_tile_ranges does on the two-pass path).
__global__ void __launch_bounds__(1024)
k_split_base(const uint32_t* __restrict__ totals, int ntiles, uint32_t* __restrict__ base, int2* __restrict__ ranges,
             uint32_t* __restrict__ err, uint32_t* __restrict__ post, uint32_t seq, const uint2* __restrict__ perm_part)
{
    __shared__ uint32_t wsum[16];
    // the guard word as it stands after the waits of the depth sort and the emission -- and the verdict of the PERMUTATION
    // CHECK: the eight partial (in - out) checksum pairs of k_split_scan must cancel (every thread adds them up: 8 cached loads)
    uint32_t tripped = 0u;
    if (err) tripped = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (perm_part) {
        uint32_t t1 = 0u, t2 = 0u;
#pragma unroll
        for (int q = 0; q < PERM_BLOCKS; ++q) { const uint2 c = perm_part[q]; t1 += c.x; t2 += c.y; }
        if ((t1 | t2) != 0u) {
            tripped = 1u;
            if (threadIdx.x == 0 && err) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (k_split_scatter obeys it)
        }
    }
    if (threadIdx.x == 0 && post) {
        __hip_atomic_store(&post[1], tripped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&post[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // A timed-out wait left the lists of THIS view invalid: every tile stays empty (the ranges were cleared by the compaction)
    if (tripped) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int PER = 4;                         // (1024 threads: up to 4096 tiles)
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int t = threadIdx.x * PER + k;
        v[k] = t < ntiles ? totals[t] : 0u;
        sum += v[k];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (int q = 0; q < w; ++q) run += wsum[q];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int t = threadIdx.x * PER + k;
        if (t < ntiles) {
            base[t] = run;
            if (v[k]) ranges[t] = make_int2((int)run, (int)(run + v[k]));
        }
        run += v[k];
    }
}

__global__ void __launch_bounds__(256)
k_split_scatter(const uint32_t* __restrict__ tkeys, const uint32_t* __restrict__ tvals, long R, int ntiles, int stride,
                const uint32_t* __restrict__ table, const uint32_t* __restrict__ base, uint32_t* __restrict__ out,
                const uint32_t* __restrict__ err)
{
    extern __shared__ uint32_t split_lds[];
    uint32_t* const cnt = split_lds;                    // [4][stride]: per wave, pairs of each tile so far -> pairs in the waves before
    uint32_t* const rowoff = split_lds + 4 * stride;    // [stride]: base[t] + pairs of t in the workgroups before this one
    __shared__ uint32_t s_tripped;
    if (threadIdx.x == 0) s_tripped = st_load(err);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t* const row = table + (size_t)blockIdx.x * stride;
    for (int d = threadIdx.x; d < stride; d += 256) {
#pragma unroll
        for (int q = 0; q < 4; ++q) cnt[q * stride + d] = 0u;
        rowoff[d] = (d < ntiles ? base[d] : 0u) + row[d];
    }
    __syncthreads();
    if (s_tripped) return;
    const long wbase = (long)blockIdx.x * SPLIT_BLOCK + (long)w * (64 * SPLIT_ITEMS);
    uint32_t key[SPLIT_ITEMS], val[SPLIT_ITEMS], rank[SPLIT_ITEMS];
#pragma unroll
    for (int i = 0; i < SPLIT_ITEMS; ++i) {
        const long idx = wbase + i * 64 + lane;
        key[i] = idx < R ? tkeys[idx] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < SPLIT_ITEMS; ++i) {
        const long idx = wbase + i * 64 + lane;
        val[i] = idx < R ? tvals[idx] : 0u;
    }
    const unsigned long long lt = lanemask_lt();
    uint32_t* const mycnt = cnt + w * stride;
#pragma unroll
    for (int i = 0; i < SPLIT_ITEMS; ++i) {
        const bool ok = key[i] < (uint32_t)ntiles;
        const unsigned long long valid = __ballot(ok);
        const unsigned long long m = match_digit<12>(key[i], valid);
        const uint32_t prior = ok ? mycnt[key[i]] : 0u;
        rank[i] = prior + (uint32_t)__popcll(m & lt);
        if (ok && (m & lt) == 0ull) mycnt[key[i]] = prior + (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int d = threadIdx.x; d < stride; d += 256) {       // counts of the four waves -> pairs in the waves before each
        const uint32_t c0 = cnt[d], c1 = cnt[stride + d], c2 = cnt[2 * stride + d];
        cnt[d] = 0u;
        cnt[stride + d] = c0;
        cnt[2 * stride + d] = c0 + c1;
        cnt[3 * stride + d] = c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SPLIT_ITEMS; ++i)
        if (key[i] < (uint32_t)ntiles) {
            const uint32_t dst = rowoff[key[i]] + mycnt[key[i]] + rank[i];
            if (dst < (uint32_t)R) out[dst] = val[i];
        }
}

// ---------------------------------------------------------------- stage 2 driver

struct Stage2Layout {
    size_t status, tmp_key, tmp_id, offs, rect_sorted, tkeysA, tkeysB, tvalsB, hist, bsum, tpartial, big, split, split_tot, chk, total;
};

static inline int tile_bits_of(int ntiles)
{
    int bits = 0;
    while ((1 << bits) < ntiles) ++bits;
    return bits;
}

static Stage2Layout stage2_layout(int V, long R, int ntiles)
{
    Stage2Layout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes, 256); return at; };
    size_t v = (size_t)(V > 0 ? V : 1), r = (size_t)(R > 0 ? R : 1);
    const int tbits = tile_bits_of(ntiles);
    // posted block sums of the depth sort | emission scan | tile sort, packed at run time (status_plan); FIRST, so
    // that the compaction kernel can clear it knowing only the scratch pointer.  Sized for any depth-key span.
    L.status = take(onesweep_status_words((long)v, 27) * 4 + emit_status_words((long)v) * 8 + 16 +
                    onesweep_status_words((long)r, tbits) * 4 + 64);
    L.tmp_key = take(v * 4);
    L.tmp_id = take(v * 4);
    L.offs = take(v * 4);
    L.rect_sorted = take(v * 16);
    L.tkeysA = take(r * 4);
    L.tkeysB = take(r * 4);
    L.tvalsB = take(r * 4);
    size_t nmax = r > v ? r : v;
    size_t nblk = (size_t)cdiv((long)nmax, RADIX_BLOCK);
    L.hist = take(nblk * 512 * 4);
    L.bsum = take(radix_bsum_words((long)nmax) * 4 + 256);
    // partial digit histograms of the tile keys
    L.tpartial = take((size_t)HIST_BLOCKS * HIST_WORDS * 4);
    L.big = take(v * 8 + 16);       // rectangles of more than 64 tiles: {output position, id} (+ the list's counter, multi-launch path)
    // tile split: the count table (one row per SPLIT_BLOCK pairs), then totals[stride] | base[stride]
    const bool can_split = ntiles <= SPLIT_MAX_TILES;
    L.split = take(can_split ? (size_t)cdiv((long)r, SPLIT_BLOCK) * split_stride(ntiles) * 4 : 0);
    L.split_tot = take(can_split ? (size_t)split_stride(ntiles) * 8 : 0);
    L.chk = take(((size_t)cdiv((long)v, 256) + 8) * sizeof(uint2));      // the emission's id checksums, one pair per workgroup (+ the check's 8 partial pairs)
    L.total = o;
    return L;
}

size_t binning_stage2_scratch_bytes(int V, long R, int ntiles) { return stage2_layout(V, R, ntiles).total; }
void* binning_stage2_status(void* scratch) { return (char*)scratch + 0; }   // Stage2Layout::status comes first
size_t binning_stage2_status_bytes(int V, long R, int ntiles)
{
    const Stage2Layout L = stage2_layout(V, R, ntiles);
    return L.tmp_key - L.status;
}
int binning_tile_bits(int ntiles) { return tile_bits_of(ntiles); }

// The 4-launch passes and the separate offset scan (inputs too long for 30-bit look-back counts; also the shape of
// the generic launch_sort_pairs used by knn.hip and the deterministic backward).
static int binning_multi_launch(const Camera& cam, int V, long R, uint32_t key_min, int key_bits, uint32_t* vis_key,
                                uint32_t* vis_id, const uint4* rect, char* base, const Stage2Layout& L,
                                uint32_t* point_list, uint32_t** tile_keys, uint32_t n_huge, hipStream_t s, bool debug)
{
    uint32_t* tmp_key = (uint32_t*)(base + L.tmp_key);
    uint32_t* tmp_id = (uint32_t*)(base + L.tmp_id);
    uint32_t* offs = (uint32_t*)(base + L.offs);
    uint4* rect_sorted = (uint4*)(base + L.rect_sorted);
    uint32_t* tkeysA = (uint32_t*)(base + L.tkeysA);
    uint32_t* tkeysB = (uint32_t*)(base + L.tkeysB);
    uint32_t* tvalsB = (uint32_t*)(base + L.tvalsB);
    uint32_t* hist = (uint32_t*)(base + L.hist);
    uint32_t* bsum = (uint32_t*)(base + L.bsum);
    uint32_t* sorted_id = vis_id;
    const int depth_passes = radix_sort_passes(key_bits);
    {
        ProfScope ps(VR_STAGE_DEPTH_SORT, s);
        int where = 0;
        const RadixGather g{rect, rect_sorted};
        int rc = radix_sort(vis_key, vis_id, tmp_key, tmp_id, V, key_min, key_bits, hist, bsum, &g, s, debug, &where);
        if (rc) return rc;
        sorted_id = where ? tmp_id : vis_id;
    }
    prof_begin(VR_STAGE_EMIT, s);
    {
        if (depth_passes == 0) {   // all depth keys equal (or one Gaussian): nothing was scattered, gather here
            hipLaunchKernelGGL(k_gather_rect, dim3(cdiv(V, 256)), dim3(256), 0, s, V, (const uint32_t*)sorted_id, rect,
                               rect_sorted);
            VR_KERNEL_CHECK("gather_rect", s, debug);
        }
        int rc = run_scan(SrcRectSorted{rect_sorted}, SinkStore{offs}, V, bsum, s, debug, "offset_scan");
        if (rc) return rc;
    }
    int bits = 0;
    while ((1 << bits) < cam.gx * cam.gy) ++bits;
    const int passes = radix_sort_passes(bits);
    uint32_t* va = (passes % 2 == 0) ? point_list : tvalsB;   // the last pass must land in point_list
    uint32_t* vb = (passes % 2 == 0) ? tvalsB : point_list;
    uint32_t *ka = tkeysA, *kb = tkeysB;
    BigRects big_list;
    big_list.items = (uint2*)(base + L.big);
    big_list.count = (uint32_t*)(base + L.big + (size_t)(V > 0 ? V : 1) * 8);
    big_list.inline_big = emit_big_inline(n_huge);
    VR_HIP(hipMemsetAsync(big_list.count, 0, 4, s));
    hipLaunchKernelGGL(k_emit, dim3(cdiv(V, 256)), dim3(256), 0, s, V, cam.gx, (const uint32_t*)sorted_id,
                       (const uint32_t*)offs, (const uint4*)rect_sorted, ka, va, big_list);
    if (!big_list.inline_big)
        hipLaunchKernelGGL(k_emit_big, dim3(EMIT_BIG_GRID), dim3(64), 0, s, big_list, rect, cam.gx, ka, va);
    VR_KERNEL_CHECK("emit", s, debug);
    prof_end(VR_STAGE_EMIT, s);
    ProfScope ps(VR_STAGE_TILE_SORT, s);
    int where = 0;
    int rc = radix_sort(ka, va, kb, vb, R, 0u, bits, hist, bsum, nullptr, s, debug, &where);
    if (rc) return rc;
    *tile_keys = where ? kb : ka;
    return 0;
}

// ---------------------------------------------------------------- post-mortem of one view's depth sort (VEGS_DEBUG_BINNING)
//
// A debugging aid, not part of the product path: api.hip calls it at the END of vr_forward (after the render stage has
// been queued, so the forward's own launch sequence runs as always) when the environment asks for it.  It synchronises the
// stream, copies the binning's intermediate state to the host and checks every invariant the depth sort rests on; the
// first violated one is reported with enough context to tell a lost workgroup from a stale status word from wrong digit
// totals.  Returns 0 when everything holds.  (What survives until the end of the forward: the compaction's inputs, the
// output of the LAST pass and -- the ping-pong partner -- the output of the pass before it, the status words of all passes.)
int debug_verify_binning(int P, int V, long R, uint32_t key_min, int key_bits, const uint32_t* vis_key, const uint32_t* vis_id,
                         const uint32_t* depth_key, const uint32_t* tile_count, const void* stage1_scratch, const void* scratch,
                         const uint32_t* totals_dev, const uint32_t* pinned, const uint32_t* err, int ntiles, hipStream_t s)
{
    VR_HIP(hipStreamSynchronize(s));
    int bad = 0;
    auto say = [&](const char* fmt, auto... a) { fprintf(stderr, "[vegs debug binning] "); fprintf(stderr, fmt, a...); fputc('\n', stderr); ++bad; };
    uint32_t tot[6] = {0, 0, 0, 0, 0, 0}, errw = 0;
    VR_HIP(hipMemcpy(tot, totals_dev, sizeof tot, hipMemcpyDeviceToHost));
    if (err) VR_HIP(hipMemcpy(&errw, err, 4, hipMemcpyDeviceToHost));
    if (errw) say("guard word = %u", errw);
    if (tot[0] != (uint32_t)V || tot[1] != (uint32_t)R) say("host V,R = %d,%ld but device totals = %u,%u", V, R, tot[0], tot[1]);
    if (V > 0 && tot[2] != key_min) say("host key_min = 0x%x, device 0x%x", key_min, tot[2]);
    if (V > 0 && bits_of(tot[3] - tot[2]) != key_bits) say("host key_bits = %d, device span says %d", key_bits, bits_of(tot[3] - tot[2]));
    if (pinned && (pinned[0] != tot[0] || pinned[1] != tot[1] || (V > 0 && (pinned[2] != tot[2] || pinned[3] != tot[3]))))
        say("mailbox {%u,%u,0x%x,0x%x} != device totals {%u,%u,0x%x,0x%x}", pinned[0], pinned[1], pinned[2], pinned[3], tot[0],
            tot[1], tot[2], tot[3]);
    if (V <= 0 || R <= 0 || (long)V > ONESWEEP_MAX_N || R > ONESWEEP_MAX_N) return bad;
    std::vector<uint32_t> dk(P), tc(P);
    VR_HIP(hipMemcpy(dk.data(), depth_key, (size_t)P * 4, hipMemcpyDeviceToHost));
    VR_HIP(hipMemcpy(tc.data(), tile_count, (size_t)P * 4, hipMemcpyDeviceToHost));
    long nvis = 0, nent = 0, mism = 0;
    for (int i = 0; i < P; ++i) {
        const bool a = dk[i] != DEPTH_KEY_NONE, b = tc[i] != 0u;
        nvis += a; nent += tc[i] & TILE_COUNT_MASK; mism += a != b;
    }
    if (nvis != V || nent != R || mism) say("preprocess arrays: %ld keys (V = %d), %ld entries (R = %ld), %ld rows where key and count disagree", nvis, V, nent, R, mism);
    const Stage2Layout L = stage2_layout(V, R, ntiles);
    const char* base = (const char*)scratch;
    const int digit = radix_digit(key_bits), passes = radix_passes(key_bits), SIZE = 1 << digit;
    const int nblk = cdiv(V, RADIX_BLOCK);
    std::vector<uint32_t> kA(V), iA(V), kB(V), iB(V);
    VR_HIP(hipMemcpy(kA.data(), vis_key, (size_t)V * 4, hipMemcpyDeviceToHost));
    VR_HIP(hipMemcpy(iA.data(), vis_id, (size_t)V * 4, hipMemcpyDeviceToHost));
    VR_HIP(hipMemcpy(kB.data(), base + L.tmp_key, (size_t)V * 4, hipMemcpyDeviceToHost));
    VR_HIP(hipMemcpy(iB.data(), base + L.tmp_id, (size_t)V * 4, hipMemcpyDeviceToHost));
    // output of pass q (0-based) lies in B when q is even, in A when odd; q = -1 is the compaction (A)
    auto check_pairs = [&](const std::vector<uint32_t>& K, const std::vector<uint32_t>& I, int done, const char* what) {
        std::vector<uint8_t> seen(P, 0);
        long oob = 0, keymis = 0, dup = 0, order = 0, first_bad = -1;
        const int sb = done * digit;
        const uint32_t mask = sb >= 32 ? 0xFFFFFFFFu : ((1u << sb) - 1u);
        for (long j = 0; j < V; ++j) {
            const uint32_t id = I[j];
            bool b = false;
            if (id >= (uint32_t)P) { ++oob; b = true; }
            else {
                if (dk[id] != K[j] || dk[id] == DEPTH_KEY_NONE) { ++keymis; b = true; }
                if (seen[id]) { ++dup; b = true; }
                seen[id] = 1;
            }
            if (j > 0) {
                const uint32_t a0 = (K[j - 1] - key_min) & mask, a1 = (K[j] - key_min) & mask;
                if (a0 > a1 || (a0 == a1 && I[j - 1] >= id)) { ++order; b = true; }
            }
            if (b && first_bad < 0) first_bad = j;
        }
        if (oob || keymis || dup || order) {
            say("%s (sorted on %d bits): %ld ids beyond P, %ld pairs whose key is not the id's depth key, %ld duplicate ids, %ld order violations; first at %ld (block %ld)",
                what, sb, oob, keymis, dup, order, first_bad, first_bad / RADIX_BLOCK);
            // which destination blocks hold the bad pairs
            std::vector<long> per(nblk, 0);
            for (long j = 0; j < V; ++j) {
                const uint32_t id = I[j];
                if (id >= (uint32_t)P || dk[id] != K[j]) ++per[j / RADIX_BLOCK];
            }
            int shown = 0;
            for (int b = 0; b < nblk && shown < 12; ++b)
                if (per[b]) { fprintf(stderr, "[vegs debug binning]    destination block %d: %ld bad pairs\n", b, per[b]); ++shown; }
            for (long j = first_bad; j < first_bad + 4 && j < V; ++j)
                fprintf(stderr, "[vegs debug binning]    [%ld] key 0x%08x id 0x%08x\n", j, K[j], I[j]);
        }
        return oob + keymis + dup + order;
    };
    const std::vector<uint32_t>& kLast = (passes & 1) ? kB : kA, & iLast = (passes & 1) ? iB : iA;
    const std::vector<uint32_t>& kPrev = (passes & 1) ? kA : kB, & iPrev = (passes & 1) ? iA : iB;
    if (passes >= 1) check_pairs(kPrev, iPrev, passes - 1, passes == 1 ? "compaction output" : "output of the pass before the last");
    check_pairs(kLast, iLast, passes, "output of the last pass");
    if (passes == 0) return bad;
    // status region: digit totals, then per pass level 1 | 2 | 3
    const StatusPlan sp = status_plan(V, R, key_bits, tile_bits_of(ntiles));
    std::vector<uint32_t> st(sp.depth / 4);
    VR_HIP(hipMemcpy(st.data(), base + L.status, sp.depth, hipMemcpyDeviceToHost));
    // digit totals against the keys themselves
    {
        std::vector<uint32_t> h((size_t)passes << digit, 0u);
        for (int i = 0; i < P; ++i)
            if (dk[i] != DEPTH_KEY_NONE)
                for (int p = 0; p < passes; ++p) ++h[((size_t)p << digit) + (((dk[i] - key_min) >> (p * digit)) & (SIZE - 1))];
        long wrong = 0, firstw = -1;
        for (size_t d = 0; d < h.size(); ++d)
            if (h[d] != st[d]) { ++wrong; if (firstw < 0) firstw = (long)d; }
        if (wrong) say("digit totals: %ld of %zu words differ from the keys' histogram (first: word %ld holds %u, should be %u)", wrong,
                       h.size(), firstw, st[firstw], h[firstw]);
    }
    const size_t per_pass = onesweep_pass_words(V, digit);
    const long n2 = (nblk + FAN - 1) / FAN, n3 = (n2 + FAN - 1) / FAN;
    for (int p = 0; p < passes; ++p) {
        const uint32_t* l1 = st.data() + HIST_WORDS + (size_t)p * per_pass;
        const uint32_t* l2 = l1 + (size_t)nblk * SIZE;
        const uint32_t* l3 = l2 + (size_t)n2 * SIZE;
        long unposted = 0, sum_bad = 0, l2_bad = 0, l3_bad = 0;
        for (int d = 0; d < SIZE; ++d) {
            uint64_t sum = 0;
            for (int b = 0; b < nblk; ++b) {
                const uint32_t w = l1[(size_t)b * SIZE + d];
                if (!(w & ST_POSTED)) ++unposted;
                sum += w & ST_VALUE;
            }
            if (sum != st[((size_t)p << digit) + d]) ++sum_bad;
            for (long g = 0; g < n2; ++g) {
                if ((g + 1) * FAN > nblk) continue;     // incomplete group: never posted
                uint64_t s2 = 0;
                for (int b = (int)g * FAN; b < (int)(g + 1) * FAN; ++b) s2 += l1[(size_t)b * SIZE + d] & ST_VALUE;
                const uint32_t w = l2[(size_t)g * SIZE + d];
                if (!(w & ST_POSTED) || (w & ST_VALUE) != s2) ++l2_bad;
            }
            for (long g = 0; g < n3; ++g) {
                if ((g + 1) * FAN * FAN > nblk) continue;
                uint64_t s3 = 0;
                for (int b = (int)g * FAN * FAN; b < (int)(g + 1) * FAN * FAN; ++b) s3 += l1[(size_t)b * SIZE + d] & ST_VALUE;
                const uint32_t w = l3[(size_t)g * SIZE + d];
                if (!(w & ST_POSTED) || (w & ST_VALUE) != s3) ++l3_bad;
            }
        }
        if (unposted || sum_bad || l2_bad || l3_bad)
            say("pass %d status: %ld level-1 words not posted, %ld digits whose posted counts do not add up to the total, %ld / %ld bad level-2 / level-3 words",
                p, unposted, sum_bad, l2_bad, l3_bad);
    }
    // the last pass once more on the host, from its intact input: where SHOULD every pair have gone?
    {
        const int p = passes - 1, shift = p * digit;
        std::vector<uint32_t> start(SIZE, 0u);
        uint32_t run = 0;
        for (int d = 0; d < SIZE; ++d) { start[d] = run; run += st[((size_t)p << digit) + d]; }
        std::vector<uint32_t> next(start);
        long moved_wrong = 0, firstm = -1;
        for (long j = 0; j < V; ++j) {
            const uint32_t d = ((kPrev[j] - key_min) >> shift) & (SIZE - 1);
            const uint32_t dst = next[d]++;
            if (dst >= (uint32_t)V || kLast[dst] != kPrev[j] || iLast[dst] != iPrev[j]) { ++moved_wrong; if (firstm < 0) firstm = j; }
        }
        if (moved_wrong)
            say("last pass replayed on the host: %ld of %d pairs are not where the pass should have put them (first: source %ld, source block %ld)",
                moved_wrong, V, firstm, firstm / RADIX_BLOCK);
        if (moved_wrong) {
            std::vector<long> per(nblk, 0);
            std::vector<uint32_t> nx(start);
            for (long j = 0; j < V; ++j) {
                const uint32_t d = ((kPrev[j] - key_min) >> shift) & (SIZE - 1);
                const uint32_t dst = nx[d]++;
                if (dst >= (uint32_t)V || kLast[dst] != kPrev[j] || iLast[dst] != iPrev[j]) ++per[j / RADIX_BLOCK];
            }
            int shown = 0;
            for (int b = 0; b < nblk && shown < 16; ++b)
                if (per[b]) { fprintf(stderr, "[vegs debug binning]    source block %d: %ld pairs misplaced\n", b, per[b]); ++shown; }
        }
    }
    return bad;
}

int launch_binning(const Camera& cam, int P, int V, long R, uint32_t key_min, int key_bits, uint32_t* vis_key,
                   uint32_t* vis_id, const uint4* rect, const void* stage1_scratch, void* scratch, uint32_t* point_list,
                   int2* ranges, bool ranges_zeroed, bool status_zeroed, uint32_t* err, uint32_t* guard_post,
                   uint32_t guard_seq, int debug_raise_guard, uint32_t n_huge, hipStream_t s, bool debug)
{
    // debug_raise_guard (tests): 1 = raise the guard word by hand after a VALID binning; 2 = lose workgroup 0 of the depth
    // sort's first pass, so that real waits run out and the lists that follow are built from a short prefix
    int ntiles = cam.gx * cam.gy;
    if (!ranges_zeroed) VR_HIP(hipMemsetAsync(ranges, 0, sizeof(int2) * (size_t)ntiles, s));
    if (V == 0 || R == 0) return 0;
    Stage2Layout L = stage2_layout(V, R, ntiles);
    char* base = (char*)scratch;
    uint32_t* tile_keys = nullptr;
    if ((cam.flags & FLAG_SCAN_BINNING) || (long)V > ONESWEEP_MAX_N || R > ONESWEEP_MAX_N || (long)V > EMIT_SCAN_MAX_V) {
        int rc = binning_multi_launch(cam, V, R, key_min, key_bits, vis_key, vis_id, rect, base, L, point_list,
                                      &tile_keys, n_huge, s, debug);
        if (rc) return rc;
    } else {
        uint32_t* tmp_key = (uint32_t*)(base + L.tmp_key);
        uint32_t* tmp_id = (uint32_t*)(base + L.tmp_id);
        uint32_t* tkeysA = (uint32_t*)(base + L.tkeysA);
        uint32_t* tkeysB = (uint32_t*)(base + L.tkeysB);
        uint32_t* tvalsB = (uint32_t*)(base + L.tvalsB);
        uint32_t* tpartial = (uint32_t*)(base + L.tpartial);
        const int bits = tile_bits_of(ntiles);
        // posted block sums of the three stages, packed; cleared by the compaction kernel when the scratch existed
        // by then (the caller's capacity hint), by a fill otherwise
        const StatusPlan sp = status_plan(V, R, key_bits, bits);
        char* st = base + L.status;
        if (!status_zeroed) VR_HIP(hipMemsetAsync(st, 0, sp.depth + sp.emit + sp.tile, s));
        // the compaction kernel's partial digit histograms of the depth keys, one row per scan block
        const int rows = cdiv(P > 0 ? P : 1, SCAN_BLOCK);
        const uint32_t* dpartial =
            (const uint32_t*)((const char*)stage1_scratch + stage1_partial_offset((size_t)rows));
        // 2. depth sort of the visible Gaussians on the bits of (key - kmin) that vary
        uint32_t* sorted_id = vis_id;
        {
            ProfScope ps(VR_STAGE_DEPTH_SORT, s);
            int where = 0;
            int rc = onesweep_sort(vis_key, vis_id, tmp_key, tmp_id, V, key_min, key_bits, dpartial, rows, (uint32_t*)st,
                                   err, s, debug, &where, debug_raise_guard == 2);
            if (rc) return rc;
            sorted_id = where ? tmp_id : vis_id;
        }
        // 3. emission; the rectangles are gathered and the offsets scanned on the way
        prof_begin(VR_STAGE_EMIT, s);
        const int passes = radix_passes(bits);
        static const bool split_env = [] { const char* e = getenv("VEGS_TILE_SPLIT"); return !(e && e[0] == '0'); }();   // (A/B switch)
        const bool split = split_env && ntiles <= SPLIT_MAX_TILES;      // the tile sort as one scatter (k_split_*)
        uint32_t* va = (split || passes % 2 != 0) ? tvalsB : point_list;   // the last pass (or the scatter) must land in point_list
        uint32_t* vb = (passes % 2 == 0) ? tvalsB : point_list;
        BigRects big_list;
        big_list.items = (uint2*)(base + L.big);
        big_list.count = (uint32_t*)(st + sp.depth + sp.emit - 16);        // (cleared with the status words)
        big_list.inline_big = emit_big_inline(n_huge);
        // (one batch of 256 Gaussians per workgroup: with 2 / 4 batches -- half / a quarter of the posted sums and look-backs,
        // the gathers of all batches in flight together -- the stage went 66 -> 81 / 93 us: the serial emission of a
        // workgroup's batches outweighs what the shorter look-back saves; round 5, profiles/experiments/README.md)
        uint2* const chk_out = (uint2*)(base + L.chk);
        const uint2* const chk_in = (const uint2*)((const char*)stage1_scratch + stage1_chk_offset((size_t)rows));
        if (debug_raise_guard == 4) hipLaunchKernelGGL(k_debug_break_permutation, dim3(1), dim3(1), 0, s, sorted_id, V);
        hipLaunchKernelGGL(k_emit_scan<1>, dim3(cdiv(V, 256)), dim3(256), 0, s, V, P, R, cam.gx, (const uint32_t*)sorted_id,
                           rect, (unsigned long long*)(st + sp.depth), err, tkeysA, va, big_list, chk_out);
        if (!big_list.inline_big)
            hipLaunchKernelGGL(k_emit_big, dim3(EMIT_BIG_GRID), dim3(64), 0, s, big_list, rect, cam.gx, tkeysA, va);
        VR_KERNEL_CHECK("emit_scan", s, debug);
        prof_end(VR_STAGE_EMIT, s);
        // 4. stable sort by tile id
        if (split) {
            ProfScope ps(VR_STAGE_TILE_SORT, s);
            const int nblk = (int)cdiv(R, SPLIT_BLOCK), stride = split_stride(ntiles);
            uint32_t* const table = (uint32_t*)(base + L.split);
            uint32_t* const totals = (uint32_t*)(base + L.split_tot);
            uint32_t* const tbase = totals + stride;
            hipLaunchKernelGGL(k_split_count, dim3(nblk), dim3(256), (size_t)stride * 4, s, (const uint32_t*)tkeysA, R, ntiles, stride,
                               table, err);
            uint2* const perm_part = chk_out + cdiv(V, 256);
            hipLaunchKernelGGL(k_split_scan, dim3(stride / 16 + PERM_BLOCKS), dim3(256), 0, s, table, nblk, stride, totals, chk_in, rows,
                               (const uint2*)chk_out, cdiv(V, 256), perm_part);
            if (debug_raise_guard == 1) VR_HIP(hipMemsetD32Async((hipDeviceptr_t)err, 1, 1, s));   // test hook: "a wait of THIS view timed out"
            hipLaunchKernelGGL(k_split_base, dim3(1), dim3(1024), 0, s, (const uint32_t*)totals, ntiles, tbase, ranges,
                               err, guard_post, guard_seq, (const uint2*)perm_part);
            hipLaunchKernelGGL(k_split_scatter, dim3(nblk), dim3(256), (size_t)stride * 20, s, (const uint32_t*)tkeysA,
                               (const uint32_t*)va, R, ntiles, stride, (const uint32_t*)table, (const uint32_t*)tbase, point_list,
                               (const uint32_t*)err);
            VR_KERNEL_CHECK("tile split", s, debug);
            return 0;
        }
        ProfScope ps(VR_STAGE_TILE_SORT, s);
        hipLaunchKernelGGL(k_perm_check, dim3(1), dim3(1024), 0, s, chk_in, rows, (const uint2*)chk_out, cdiv(V, 256), err);
        const int hist_rows = (int)(cdiv(R, 4096) < HIST_BLOCKS ? cdiv(R, 4096) : HIST_BLOCKS);
        hipLaunchKernelGGL(k_digit_hist, dim3(hist_rows), dim3(HIST_THREADS), 0, s, (const uint32_t*)tkeysA, R, 0u,
                           radix_digit(bits), passes, tpartial);
        VR_KERNEL_CHECK("digit_hist", s, debug);
        int where = 0;
        int rc = onesweep_sort(tkeysA, va, tkeysB, vb, R, 0u, bits, tpartial, hist_rows,
                               (uint32_t*)(st + sp.depth + sp.emit), err, s, debug, &where);
        if (rc) return rc;
        tile_keys = where ? tkeysB : tkeysA;
    }
    // 5. ranges
    prof_begin(VR_STAGE_RANGES, s);
    if (debug_raise_guard == 1) VR_HIP(hipMemsetD32Async((hipDeviceptr_t)err, 1, 1, s));   // test hook: "a wait of THIS view timed out"
    hipLaunchKernelGGL(k_tile_ranges, dim3(cdiv(R, 256)), dim3(256), 0, s, (const uint32_t*)tile_keys, R, ranges,
                       (const uint32_t*)err, guard_post, guard_seq);
    VR_KERNEL_CHECK("tile_ranges", s, debug);
    prof_end(VR_STAGE_RANGES, s);
    return 0;
}

}  // namespace vr
