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
#include "../../include/vegs_rast.h"
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

// ---------------------------------------------------------------- generic 3-kernel scan

__device__ __forceinline__ uint32_t rect_area(uint2 r) { return (r.y & 0xFFFFu) * (r.y >> 16); }

struct SrcFlagTiles {  // 1 for visible Gaussians; secondary value = tiles touched (summed only);
    const uint2* rect;  // also the min / max depth key of the visible ones (range of the depth sort)
    const uint32_t* depth_key;
    static constexpr bool MINMAX = true;
    __device__ uint32_t operator()(long i) const { return rect_area(rect[i]) ? 1u : 0u; }
    __device__ uint32_t second(long i) const { return rect_area(rect[i]); }
    __device__ uint32_t key(long i) const { return depth_key[i]; }
};
struct SrcRectSorted {  // tiles touched, in depth-sorted order (rectangles already gathered: coalesced reads)
    const uint2* rect_sorted;
    static constexpr bool MINMAX = false;
    __device__ uint32_t operator()(long i) const { return rect_area(rect_sorted[i]); }
    __device__ uint32_t second(long) const { return 0; }
    __device__ uint32_t key(long) const { return 0; }
};
struct SrcPlain {
    const uint32_t* v;
    static constexpr bool MINMAX = false;
    __device__ uint32_t operator()(long i) const { return v[i]; }
    __device__ uint32_t second(long) const { return 0; }
    __device__ uint32_t key(long) const { return 0; }
};

struct SinkCompact {
    const uint32_t* depth_key;
    uint32_t* vis_key;
    uint32_t* vis_id;
    __device__ void operator()(long i, uint32_t v, uint32_t excl) const
    {
        if (v) { vis_key[excl] = depth_key[i]; vis_id[excl] = (uint32_t)i; }
    }
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
    __shared__ uint32_t lds[16];
    long base = (long)blockIdx.x * SCAN_BLOCK + (long)threadIdx.x * SCAN_ITEMS;
    uint32_t a = 0, b = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) {
            const uint32_t v = src(base + k);
            a += v;
            b += src.second(base + k);
            if (Src::MINMAX && v) { const uint32_t key = src.key(base + k); kmin = min(kmin, key); kmax = max(kmax, key); }
        }
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        a += __shfl_down(a, d, 64);
        b += __shfl_down(b, d, 64);
        if (Src::MINMAX) {
            kmin = min(kmin, (uint32_t)__shfl_down((int)kmin, d, 64));
            kmax = max(kmax, (uint32_t)__shfl_down((int)kmax, d, 64));
        }
    }
    if (lane == 0) { lds[w] = a; lds[4 + w] = b; lds[8 + w] = kmin; lds[12 + w] = kmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bsum[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
        if (bsum2) {
            bsum2[blockIdx.x] = lds[4] + lds[5] + lds[6] + lds[7];
            if (Src::MINMAX) {
                bsum2[gridDim.x + blockIdx.x] = min(min(lds[8], lds[9]), min(lds[10], lds[11]));
                bsum2[2 * gridDim.x + blockIdx.x] = max(max(lds[12], lds[13]), max(lds[14], lds[15]));
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
// kernel; workgroup 0 also publishes the grand totals (sum, secondary sum, min/max key) when asked.
template <class Src, class Sink>
__global__ void __launch_bounds__(256)
k_scan_apply(Src src, Sink sink, long n, const uint32_t* __restrict__ bsum, const uint32_t* __restrict__ bsum2,
             int nb, uint32_t* __restrict__ totals, uint32_t* __restrict__ zero_a = nullptr, long zero_na = 0,
             uint32_t* __restrict__ zero_b = nullptr, long zero_nb = 0)
{
    __shared__ uint32_t lds4[4];
    // side duty (saves two fill launches): clear small arrays that LATER kernels of the same stream accumulate into
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < zero_na; i += (long)gridDim.x * 256) zero_a[i] = 0u;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < zero_nb; i += (long)gridDim.x * 256) zero_b[i] = 0u;
    uint32_t acc = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) acc += bsum[j];
    const uint32_t prefix = block_sum(acc, lds4);
    if (totals && blockIdx.x == 0) {
        uint32_t t0 = 0, t1 = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
        for (int j = threadIdx.x; j < nb; j += 256) {
            t0 += bsum[j];
            if (bsum2) {
                t1 += bsum2[j];
                if (Src::MINMAX) { kmin = min(kmin, bsum2[nb + j]); kmax = max(kmax, bsum2[2 * nb + j]); }
            }
        }
        t0 = block_sum(t0, lds4);
        t1 = block_sum(t1, lds4);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, d, 64));
            kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d, 64));
        }
        __shared__ uint32_t mm[8];
        if ((threadIdx.x & 63) == 0) { mm[threadIdx.x >> 6] = kmin; mm[4 + (threadIdx.x >> 6)] = kmax; }
        __syncthreads();
        if (threadIdx.x == 0) {
            totals[0] = t0;
            totals[1] = t1;
            if (Src::MINMAX) {
                totals[2] = min(min(mm[0], mm[1]), min(mm[2], mm[3]));
                totals[3] = max(max(mm[4], mm[5]), max(mm[6], mm[7]));
            }
        }
        __syncthreads();
    }
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
k_scan_totals(const uint32_t* __restrict__ bsum, const uint32_t* __restrict__ bsum2, int nb, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t lds4[4];
    __shared__ uint32_t mm[8];
    uint32_t t0 = 0, t1 = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
    for (int j = threadIdx.x; j < nb; j += 256) {
        t0 += bsum[j];
        t1 += bsum2[j];
        kmin = min(kmin, bsum2[nb + j]);
        kmax = max(kmax, bsum2[2 * nb + j]);
    }
    t0 = block_sum(t0, lds4);
    t1 = block_sum(t1, lds4);
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
    }
}

template <class Src, class Sink>
static int run_scan(Src src, Sink sink, long n, uint32_t* bsum, uint32_t* bsum2, uint32_t* totals, hipStream_t s,
                    bool debug, const char* what)
{
    if (n <= 0) return 0;
    int nb = cdiv(n, SCAN_BLOCK);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_reduce<Src>), dim3(nb), dim3(256), 0, s, src, n, bsum, bsum2);
    VR_KERNEL_CHECK(what, s, debug);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_apply<Src, Sink>), dim3(nb), dim3(256), 0, s, src, sink, n,
                       (const uint32_t*)bsum, (const uint32_t*)bsum2, nb, totals);
    VR_KERNEL_CHECK(what, s, debug);
    return 0;
}

// ---------------------------------------------------------------- stage 1: compaction

size_t binning_stage1_scratch_bytes(int P)
{
    size_t nb = (size_t)cdiv(P > 0 ? P : 1, SCAN_BLOCK);
    return align_up(4 * nb * sizeof(uint32_t), 256);
}

// Two halves, so that the caller can start reading the totals {V, R, min key, max key} back to the host between
// them: the device->host round trip then overlaps with the apply kernel instead of idling the GPU.
int launch_compact_reduce(int P, const uint2* rect, const uint32_t* depth_key, void* scratch, uint32_t* totals_dev,
                          hipStream_t s, bool debug)
{
    int nb = cdiv(P > 0 ? P : 1, SCAN_BLOCK);
    uint32_t* bsum = (uint32_t*)scratch;
    uint32_t* bsum2 = bsum + nb;
    SrcFlagTiles src{rect, depth_key};
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_reduce<SrcFlagTiles>), dim3(nb), dim3(256), 0, s, src, (long)P, bsum, bsum2);
    VR_KERNEL_CHECK("compact_reduce", s, debug);
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(256), 0, s, (const uint32_t*)bsum, (const uint32_t*)bsum2, nb, totals_dev);
    VR_KERNEL_CHECK("compact_totals", s, debug);
    return 0;
}

int launch_compact_apply(int P, const uint2* rect, const uint32_t* depth_key, void* scratch, uint32_t* vis_key,
                         uint32_t* vis_id, uint32_t* zero_a, long zero_na, uint32_t* zero_b, long zero_nb, hipStream_t s,
                         bool debug)
{
    int nb = cdiv(P > 0 ? P : 1, SCAN_BLOCK);
    uint32_t* bsum = (uint32_t*)scratch;
    uint32_t* bsum2 = bsum + nb;
    SrcFlagTiles src{rect, depth_key};
    SinkCompact sink{depth_key, vis_key, vis_id};
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_apply<SrcFlagTiles, SinkCompact>), dim3(nb), dim3(256), 0, s, src, sink,
                       (long)P, (const uint32_t*)bsum, (const uint32_t*)bsum2, nb, (uint32_t*)nullptr, zero_a, zero_na,
                       zero_b, zero_nb);
    VR_KERNEL_CHECK("compact_apply", s, debug);
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
                const uint32_t* __restrict__ hist_scanned, int nblk, const uint2* __restrict__ rect,
                uint2* __restrict__ rect_sorted)
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
struct RadixGather { const uint2* rect; uint2* rect_sorted; };
template <int BITS>
static int radix_pass(const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, long n, uint32_t kmin,
                      int shift, uint32_t* hist, uint32_t* bsum, const RadixGather* gather, hipStream_t s, bool debug)
{
    int nblk = cdiv(n, RADIX_BLOCK);
    long hn = (long)(1 << BITS) * nblk;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_radix_hist<BITS>), dim3(nblk), dim3(256), 0, s, kin, n, kmin, shift, hist,
                       nblk);
    VR_KERNEL_CHECK("radix_hist", s, debug);
    int rc = run_scan(SrcPlain{hist}, SinkStore{hist}, hn, bsum, (uint32_t*)nullptr, (uint32_t*)nullptr, s, debug,
                      "radix_scan");
    if (rc) return rc;
    if (gather)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_radix_scatter<BITS, true>), dim3(nblk), dim3(256), 0, s, kin, vin, kout, vout,
                           n, kmin, shift, (const uint32_t*)hist, nblk, gather->rect, gather->rect_sorted);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_radix_scatter<BITS, false>), dim3(nblk), dim3(256), 0, s, kin, vin, kout, vout,
                           n, kmin, shift, (const uint32_t*)hist, nblk, (const uint2*)nullptr, (uint2*)nullptr);
    VR_KERNEL_CHECK("radix_scatter", s, debug);
    return 0;
}

// digit width for sorting `nbits` key bits: fewest passes first, then the narrowest digit (cheaper ranking)
static int radix_digit(int nbits)
{
    if (nbits <= 12) return 6;   // 1-2 passes
    if (nbits <= 16) return 8;   // 2
    if (nbits <= 18) return 9;   // 2
    if (nbits <= 24) return 8;   // 3
    if (nbits <= 27) return 9;   // 3
    return 8;                    // 4
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
int radix_sort_passes(int nbits)
{
    return nbits <= 0 ? 0 : cdiv(nbits, radix_digit(nbits));
}

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

// ---------------------------------------------------------------- emission + ranges

// One block per 256 depth-sorted Gaussians; lanes are spread over OUTPUT entries (binary search in
// the block's LDS prefix), so long rectangles do not serialise a lane and writes are coalesced.
// the only gather of the binning stage: rectangles of the depth-sorted Gaussians (8 B each)
__global__ void __launch_bounds__(256)
k_gather_rect(int V, const uint32_t* __restrict__ sorted_id, const uint2* __restrict__ rect,
              uint2* __restrict__ rect_sorted)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < V) rect_sorted[r] = rect[sorted_id[r]];
}

// Emission of the (tile, id) pairs of 256 depth-sorted Gaussians per workgroup.  Small rectangles
// (<= EMIT_SMALL tiles: the vast majority, far splats touch 1-4 tiles) are written by their own lane --
// consecutive lanes own consecutive output ranges, so the stores stay nearly coalesced; a large
// rectangle is written by its whole wave, 64 entries per step.
constexpr int EMIT_SMALL = 8;
__global__ void __launch_bounds__(256)
k_emit(int V, int gx, const uint32_t* __restrict__ sorted_id, const uint32_t* __restrict__ offs,
       const uint2* __restrict__ rect_sorted, uint32_t* __restrict__ tkeys, uint32_t* __restrict__ tvals)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t off = 0, cnt = 0, id = 0;
    int x0 = 0, y0 = 0, w = 1;
    if (r < V) {
        const uint2 rc = rect_sorted[r];
        off = offs[r];
        cnt = rect_area(rc);
        id = sorted_id[r];
        x0 = (int)(rc.x & 0xFFFFu);
        y0 = (int)(rc.x >> 16);
        w = max((int)(rc.y & 0xFFFFu), 1);
    }
    if (cnt <= EMIT_SMALL) {
        int rx = 0, ry = 0;
        for (uint32_t k = 0; k < cnt; ++k) {
            tkeys[off + k] = (uint32_t)((y0 + ry) * gx + x0 + rx);
            tvals[off + k] = id;
            if (++rx == w) { rx = 0; ++ry; }
        }
    }
    // large rectangles: one at a time, all 64 lanes of the wave
    for (unsigned long long big = __ballot(cnt > EMIT_SMALL); big; big &= big - 1) {
        const int src = __builtin_ctzll(big);
        const uint32_t b_off = (uint32_t)__shfl((int)off, src, 64), b_cnt = (uint32_t)__shfl((int)cnt, src, 64);
        const uint32_t b_id = (uint32_t)__shfl((int)id, src, 64);
        const int b_x0 = __shfl(x0, src, 64), b_y0 = __shfl(y0, src, 64), b_w = __shfl(w, src, 64);
        for (uint32_t k = lane; k < b_cnt; k += 64) {
            const int ry = (int)(k / (uint32_t)b_w), rx = (int)(k - (uint32_t)ry * (uint32_t)b_w);
            tkeys[b_off + k] = (uint32_t)((b_y0 + ry) * gx + b_x0 + rx);
            tvals[b_off + k] = b_id;
        }
    }
}

__global__ void __launch_bounds__(256)
k_tile_ranges(const uint32_t* __restrict__ tkeys, long R, int2* __restrict__ ranges)
{
    long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= R) return;
    uint32_t k = tkeys[j];
    if (j == 0) ranges[k].x = 0;
    else {
        uint32_t kp = tkeys[j - 1];
        if (kp != k) { ranges[kp].y = (int)j; ranges[k].x = (int)j; }
    }
    if (j == R - 1) ranges[k].y = (int)R;
}

// ---------------------------------------------------------------- stage 2 driver

struct Stage2Layout {
    size_t tmp_key, tmp_id, offs, rect_sorted, tkeysA, tkeysB, tvalsB, hist, bsum, total;
};

static Stage2Layout stage2_layout(int V, long R)
{
    Stage2Layout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes, 256); return at; };
    size_t v = (size_t)(V > 0 ? V : 1), r = (size_t)(R > 0 ? R : 1);
    L.tmp_key = take(v * 4);
    L.tmp_id = take(v * 4);
    L.offs = take(v * 4);
    L.rect_sorted = take(v * 8);
    L.tkeysA = take(r * 4);
    L.tkeysB = take(r * 4);
    L.tvalsB = take(r * 4);
    size_t nmax = r > v ? r : v;
    size_t nblk = (size_t)cdiv((long)nmax, RADIX_BLOCK);
    L.hist = take(nblk * 512 * 4);
    L.bsum = take(radix_bsum_words((long)nmax) * 4 + 256);
    L.total = o;
    return L;
}

size_t binning_stage2_scratch_bytes(int V, long R, int) { return stage2_layout(V, R).total; }

int launch_binning(const Camera& cam, int P, int V, long R, uint32_t key_min, int key_bits, uint32_t* vis_key,
                   uint32_t* vis_id, const uint2* rect, void* scratch, uint32_t* point_list, int2* ranges,
                   bool ranges_zeroed, hipStream_t s, bool debug)
{
    int ntiles = cam.gx * cam.gy;
    if (!ranges_zeroed) VR_HIP(hipMemsetAsync(ranges, 0, sizeof(int2) * (size_t)ntiles, s));
    if (V == 0 || R == 0) return 0;
    Stage2Layout L = stage2_layout(V, R);
    char* base = (char*)scratch;
    uint32_t* tmp_key = (uint32_t*)(base + L.tmp_key);
    uint32_t* tmp_id = (uint32_t*)(base + L.tmp_id);
    uint32_t* offs = (uint32_t*)(base + L.offs);
    uint2* rect_sorted = (uint2*)(base + L.rect_sorted);
    uint32_t* tkeysA = (uint32_t*)(base + L.tkeysA);
    uint32_t* tkeysB = (uint32_t*)(base + L.tkeysB);
    uint32_t* tvalsB = (uint32_t*)(base + L.tvalsB);
    uint32_t* hist = (uint32_t*)(base + L.hist);
    uint32_t* bsum = (uint32_t*)(base + L.bsum);
    (void)P;

    // 2. depth sort of the visible Gaussians on the bits of (key - kmin) that vary; its last scatter also gathers the
    // tile rectangles into sorted order
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
    // 3. offsets in depth order, then emission
    prof_begin(VR_STAGE_EMIT, s);
    {
        if (depth_passes == 0) {   // all depth keys equal (or one Gaussian): nothing was scattered, gather here
            hipLaunchKernelGGL(k_gather_rect, dim3(cdiv(V, 256)), dim3(256), 0, s, V, (const uint32_t*)sorted_id, rect,
                               rect_sorted);
            VR_KERNEL_CHECK("gather_rect", s, debug);
        }
        int rc = run_scan(SrcRectSorted{rect_sorted}, SinkStore{offs}, V, bsum, (uint32_t*)nullptr,
                          (uint32_t*)nullptr, s, debug, "offset_scan");
        if (rc) return rc;
    }
    int bits = 0;
    while ((1 << bits) < ntiles) ++bits;
    const int passes = radix_sort_passes(bits);
    // choose the starting value buffer so that the last pass lands in point_list
    uint32_t* va = (passes % 2 == 0) ? point_list : tvalsB;
    uint32_t* vb = (passes % 2 == 0) ? tvalsB : point_list;
    uint32_t *ka = tkeysA, *kb = tkeysB;
    hipLaunchKernelGGL(k_emit, dim3(cdiv(V, 256)), dim3(256), 0, s, V, cam.gx, (const uint32_t*)sorted_id,
                       (const uint32_t*)offs, (const uint2*)rect_sorted, ka, va);
    VR_KERNEL_CHECK("emit", s, debug);
    prof_end(VR_STAGE_EMIT, s);
    // 4. stable sort by tile id
    prof_begin(VR_STAGE_TILE_SORT, s);
    {
        int where = 0;
        int rc = radix_sort(ka, va, kb, vb, R, 0u, bits, hist, bsum, nullptr, s, debug, &where);
        if (rc) return rc;
        if (where) { uint32_t* t = ka; ka = kb; kb = t; }
    }
    prof_end(VR_STAGE_TILE_SORT, s);
    // 5. ranges
    prof_begin(VR_STAGE_RANGES, s);
    hipLaunchKernelGGL(k_tile_ranges, dim3(cdiv(R, 256)), dim3(256), 0, s, (const uint32_t*)ka, R, ranges);
    VR_KERNEL_CHECK("tile_ranges", s, debug);
    prof_end(VR_STAGE_RANGES, s);
    return 0;
}

}  // namespace vr
