// xgmi.hip -- the direct gradient exchange of the view-sharded job (include/vegs_xgmi.h).
//
// One process per GPU.  Every rank owns a WINDOW (device memory + flag words) that its peers map with hipIpc; data moves by
// plain stores into the peers' windows, i.e. over the point-to-point xGMI link between the two GPUs -- all 7 links of a
// GPU carry traffic at the same time, where a ring collective keeps one link pair busy per step:
//
//   all-reduce of D floats (two-shot):  push D/N floats to each peer, reduce my shard from N local slots in RANK ORDER
//                                       (a fixed order: the result does not depend on timing), push the reduced D/N floats
//                                       to each peer's result buffer.  Per link 2 D/N; 88 MB at N = 8: 2 x 11 MB per link.
//   all-gather of a block B:            push B to each peer's slot.  Per link B.
//
// Synchronisation is three arrays of epoch words per window (written by the peer with a system-scope release store after a
// system-scope fence behind its data stores; polled with system-scope acquire loads).  A kernel's LAST workgroup (device
// counter) posts the epoch to all peers.  Every wait is wall-clock bounded (10 s by default, vr_xgmi_set_wait_bound) and
// raises the window's error word instead of hanging the queue -- the word in the window AND its mirror in pinned host
// memory, which every vr_xgmi_* call reads first (no synchronisation): the call after a timed-out exchange fails.  The
// error is sticky: a reduce whose wait ran out writes NOTHING into the peers' result buffers and posts nothing (the peers'
// own waits then run out as well -- every rank learns), and so does every later one on this window.
// Buffers are reused every iteration without further handshakes:
//   recv[]    is written by a peer's push k+1 only after that peer has seen my flag_b(k), which I post after my reduce(k)
//             has read recv[] completely;
//   result[]  is written by a peer's reduce k+1 only after my flag_a(k+1), which my push k+1 posts -- stream-ordered behind
//             whatever consumed result[] of iteration k on my stream;
//   gather[]  has two parities: a peer may push block k+1 while I still read block k; its push k+2 needs my flag_b(k+1).
// (The caller alternates the gather parity and issues one all-reduce per iteration; vegs_amd/xgmi.py.)
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "../../include/vegs_xgmi.h"
#include "vr_host.h"

namespace vr {

constexpr int XG_MAXR = VR_XGMI_MAX_RANKS;
constexpr int XG_MAXS = VR_XGMI_MAX_SEGMENTS;
constexpr int XG_FLAG_STRIDE = 8;                       // uint64 words per flag: one 64-byte line each
constexpr size_t XG_FLAG_BYTES = 16384;                 // flags + counters + error word, at the start of the window
constexpr unsigned long long XG_TICKS_PER_S = 100000000ull;    // s_memrealtime: 100 MHz
constexpr double XG_WAIT_SECONDS = 10.0;                       // default bound of a wait (ranks may be far apart at start-up)
// word indices (uint64) inside the flag region
constexpr int XG_FLAG_A = 0;                            // [N] all-reduce: peer j's push has landed
constexpr int XG_FLAG_B = XG_MAXR * XG_FLAG_STRIDE;     // [N] all-reduce: peer j's reduced shard has landed
constexpr int XG_FLAG_G = 2 * XG_MAXR * XG_FLAG_STRIDE; // [2][N] all-gather per parity
constexpr int XG_DONE = 4 * XG_MAXR * XG_FLAG_STRIDE;   // [4] last-workgroup counters (push, reduce, gather 0, gather 1)
constexpr int XG_ERR = XG_DONE + 4 * XG_FLAG_STRIDE;    // error word
constexpr int XG_NONCE = XG_ERR + XG_FLAG_STRIDE;       // identity of this window (checked by the peers after mapping it)

struct XgSeg { const float* src; long n, shard, recv_off, result_off; };   // shard = floats per rank (multiple of 4)
struct XgArgs {
    float* peer[XG_MAXR];          // window bases (own window at [rank])
    XgSeg seg[XG_MAXS];
    int nseg, rank, world;
    long recv_offset, result_offset, slot_stride;      // floats; slot_stride = sum of the segments' shard lengths
    unsigned long long epoch;
    float scale;
    unsigned long long wait_ticks;                     // bound of every wait of this launch
    unsigned long long* err_host;                      // mirror of the window's error word in pinned host memory
};

__device__ __forceinline__ unsigned long long* flags_of(float* window) { return reinterpret_cast<unsigned long long*>(window); }

__device__ __forceinline__ void post(unsigned long long* word, unsigned long long epoch)
{
    __hip_atomic_store(word, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// thread j < world of the calling workgroup waits for word[j * stride] >= epoch; raises *err (and its host mirror) after
// the bound.  Returns (to every thread of the workgroup) whether the window's error word is set -- by this wait or earlier.
__device__ __forceinline__ bool wait_all(const unsigned long long* words, int world, unsigned long long epoch,
                                         unsigned long long* err, unsigned long long* err_host, unsigned long long ticks)
{
    if ((int)threadIdx.x < world) {
        const unsigned long long* w = words + (size_t)threadIdx.x * XG_FLAG_STRIDE;
        unsigned long long t0 = 0;
        // (polled RELAXED, ONE acquire fence behind the loop: an acquire at system scope is a `buffer_inv` -- every poll of
        // every waiting workgroup would drop the XCD's L2 under the backward kernels this exchange is overlapped with; what
        // that costs was measured on the render forward's walkers in round 5)
        for (int polls = 0;; ++polls) {
            if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= epoch) break;
            if ((polls & 255) == 255) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (t0 == 0) t0 = now | 1ull;
                else if (now - t0 > ticks) {
                    __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (err_host) __hip_atomic_store(err_host, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
    }
    __syncthreads();
    // ONE acquire fence per thread, BEHIND the barrier (round 6, advisor finding: it used to sit inside the branch of the
    // polling threads, the others relying on the barrier alone): system scope -- what the peers wrote before posting is
    // visible to every thread of the workgroup from here on
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    return __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull;
}

// all stores of this workgroup are visible system-wide; the LAST workgroup to get here posts `epoch` into word
// [slot + rank] of every peer's flag region
__device__ __forceinline__ void finish_and_post(const XgArgs& a, int done_idx, int flag_base)
{
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* mine = flags_of(a.peer[a.rank]);
        const unsigned long long n = __hip_atomic_fetch_add(mine + XG_DONE + done_idx * XG_FLAG_STRIDE, 1ull, __ATOMIC_ACQ_REL,
                                                            __HIP_MEMORY_SCOPE_AGENT);
        if (n == gridDim.x - 1) {
            __hip_atomic_store(mine + XG_DONE + done_idx * XG_FLAG_STRIDE, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            for (int p = 0; p < a.world; ++p)
                post(flags_of(a.peer[p]) + flag_base + (size_t)a.rank * XG_FLAG_STRIDE, a.epoch);
        }
    }
}

constexpr int XG_THREADS = 256;
constexpr int XG_CHUNK = XG_THREADS * 4;       // floats per workgroup step

// ---- all-reduce, shot 1: my contribution to shard s goes to peer s's recv[rank]
__global__ void __launch_bounds__(XG_THREADS) k_xg_push(XgArgs a)
{
    for (int t = 0; t < a.nseg; ++t) {
        const XgSeg s = a.seg[t];
        const long padded = s.shard * a.world;
        const bool vec = (((uintptr_t)s.src & 15) == 0);
        for (long q = ((long)blockIdx.x * XG_THREADS + threadIdx.x) * 4; q < padded; q += (long)gridDim.x * XG_CHUNK) {
            const int dst = (int)(q / s.shard);
            const long o = q - (long)dst * s.shard;                 // (shard % 4 == 0: the four floats share a shard)
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q + 3 < s.n && vec) v = nt_load4(reinterpret_cast<const float4*>(s.src + q));
            else {
                if (q < s.n) v.x = s.src[q];
                if (q + 1 < s.n) v.y = s.src[q + 1];
                if (q + 2 < s.n) v.z = s.src[q + 2];
                if (q + 3 < s.n) v.w = s.src[q + 3];
            }
            float* out = a.peer[dst] + a.recv_offset + (long)a.rank * a.slot_stride + s.recv_off + o;
            *reinterpret_cast<float4*>(out) = v;
        }
    }
    finish_and_post(a, 0, XG_FLAG_A);
}

// ---- all-reduce, shot 2: sum my shard over the N slots in rank order, scale, write it into every peer's result[]
__global__ void __launch_bounds__(XG_THREADS) k_xg_reduce(XgArgs a)
{
    unsigned long long* mine = flags_of(a.peer[a.rank]);
    // a push that never arrived (or an earlier failure on this window): recv[] is not the peers' data -- this workgroup sums
    // nothing, writes nothing and the kernel posts no epoch.  (The decision is the WORKGROUP's: one whose poll succeeded just
    // before another's ran out may already have written its part of the shard -- result[] of a failed exchange is PARTIAL,
    // never to be consumed; every exchange call tests the window's sticky error word, vegs_xgmi.h.)
    if (wait_all(mine + XG_FLAG_A, a.world, a.epoch, mine + XG_ERR, a.err_host, a.wait_ticks)) return;
    const float* recv = a.peer[a.rank] + a.recv_offset;
    for (int t = 0; t < a.nseg; ++t) {
        const XgSeg s = a.seg[t];
        for (long o = ((long)blockIdx.x * XG_THREADS + threadIdx.x) * 4; o < s.shard; o += (long)gridDim.x * XG_CHUNK) {
            const long e = (long)a.rank * s.shard + o;              // element of the tensor
            if (e >= s.n) break;
            float4 acc = *reinterpret_cast<const float4*>(recv + s.recv_off + o);
            for (int j = 1; j < a.world; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(recv + (long)j * a.slot_stride + s.recv_off + o);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            acc.x *= a.scale; acc.y *= a.scale; acc.z *= a.scale; acc.w *= a.scale;
            const long left = s.n - e;
            for (int p = 0; p < a.world; ++p) {
                float* out = a.peer[p] + a.result_offset + s.result_off + e;      // (result_off % 4 == 0, e % 4 == 0)
                if (left >= 4) *reinterpret_cast<float4*>(out) = acc;
                else {
                    out[0] = acc.x;
                    if (left > 1) out[1] = acc.y;
                    if (left > 2) out[2] = acc.z;
                }
            }
        }
    }
    finish_and_post(a, 1, XG_FLAG_B);
}

// ---- waits (one workgroup): every peer's word has reached `epoch`
__global__ void __launch_bounds__(64) k_xg_wait(float* window, int flag_base, int world, unsigned long long epoch,
                                                unsigned long long* err_host, unsigned long long ticks)
{
    unsigned long long* mine = flags_of(window);
    (void)wait_all(mine + flag_base, world, epoch, mine + XG_ERR, err_host, ticks);
}

// ---- all-gather: my block into slot `rank` of every peer's gather buffer
struct XgGatherArgs {
    float* peer[XG_MAXR];
    const float* src[XG_MAXS];
    long n[XG_MAXS], off[XG_MAXS];
    int nseg, rank, world, parity;
    long slot_base;                 // float offset of slot `rank` inside a window
    unsigned long long epoch;
};

__global__ void __launch_bounds__(XG_THREADS) k_xg_gather(XgGatherArgs g)
{
    for (int t = 0; t < g.nseg; ++t) {
        const float* src = g.src[t];
        const long n = g.n[t];
        const bool vec = (((uintptr_t)src & 15) == 0);
        for (long q = ((long)blockIdx.x * XG_THREADS + threadIdx.x) * 4; q < n; q += (long)gridDim.x * XG_CHUNK) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q + 3 < n && vec) v = nt_load4(reinterpret_cast<const float4*>(src + q));
            else {
                v.x = src[q];
                if (q + 1 < n) v.y = src[q + 1];
                if (q + 2 < n) v.z = src[q + 2];
                if (q + 3 < n) v.w = src[q + 3];
            }
            for (int p = 0; p < g.world; ++p)                          // (slots are padded to whole float4s)
                *reinterpret_cast<float4*>(g.peer[p] + g.slot_base + g.off[t] + q) = v;
        }
    }
    XgArgs a;            // (finish_and_post reads peer / rank / world / epoch only)
    for (int p = 0; p < XG_MAXR; ++p) a.peer[p] = g.peer[p];
    a.rank = g.rank; a.world = g.world; a.epoch = g.epoch;
    finish_and_post(a, 2 + g.parity, XG_FLAG_G + g.parity * XG_MAXR * XG_FLAG_STRIDE);
}

static long align4(long v) { return (v + 3) / 4 * 4; }

}  // namespace vr

using namespace vr;

struct VrXgmi {
    int rank = 0, world = 1, device = 0;
    long reduce_floats = 0, gather_floats = 0;
    VrXgmiLayout lay{};
    float* window = nullptr;
    float* peer[XG_MAXR] = {};
    bool attached = false;
    unsigned long long epoch_r = 0, epoch_g[2] = {0, 0};
    unsigned long long nonce = 0;
    unsigned long long wait_ticks = (unsigned long long)(XG_WAIT_SECONDS * (double)XG_TICKS_PER_S);
    unsigned long long* err_host = nullptr;       // pinned, coherent mirror of the error word (host pointer) ...
    unsigned long long* err_host_dev = nullptr;   // ... and its device address
};

// a timed-out wait of an EARLIER call on this window, read from the pinned mirror: no synchronisation
static int xg_failed(const VrXgmi* x)
{
    if (x->err_host && __atomic_load_n(x->err_host, __ATOMIC_ACQUIRE) != 0ull) {
        set_error("xgmi: a peer did not arrive within the wait bound in an earlier exchange on this window (rank %d of %d): "
                  "its results are invalid and the window is unusable", x->rank, x->world);
        return VR_ERR_HIP;
    }
    return 0;
}


extern "C" int vr_xgmi_create(int32_t rank, int32_t world, int64_t reduce_floats, int64_t gather_floats, VrXgmi** out)
{
    if (!out || world < 1 || world > XG_MAXR || rank < 0 || rank >= world || reduce_floats < 0 || gather_floats < 0)
        { set_error("xgmi: need 0 <= rank < world <= %d and non-negative capacities", XG_MAXR); return VR_ERR_INVALID_ARGUMENT; }
    VrXgmi* x = new VrXgmi();
    x->rank = rank; x->world = world; x->reduce_floats = reduce_floats; x->gather_floats = gather_floats;
    VR_HIP(hipGetDevice(&x->device));
    const long pad = 4L * XG_MAXS;
    long off = (long)(XG_FLAG_BYTES / sizeof(float));
    x->lay.recv_offset = off;       off += align4(reduce_floats) + (long)world * pad;       // N slots of <= D/N + padding
    x->lay.result_offset = off;     off += align4(reduce_floats) + pad;
    x->lay.gather_slot = align4(gather_floats) + pad;
    for (int p = 0; p < 2; ++p) { x->lay.gather_offset[p] = off; off += (long)world * x->lay.gather_slot; }
    x->lay.total_floats = off;
    // fine-grained device memory: the flag words are polled while peers write them, and peers' data stores must not sit in
    // a non-coherent cache of the writer (what RCCL allocates for its own buffers); plain hipMalloc if the runtime refuses
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, (size_t)off * sizeof(float), hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        hipError_t e = hipMalloc(&p, (size_t)off * sizeof(float));
        if (e != hipSuccess) { delete x; set_error("xgmi: window allocation of %ld bytes failed: %s", off * 4, hipGetErrorString(e)); return VR_ERR_HIP; }
    }
    x->window = (float*)p;
    hipError_t e = hipMemset(p, 0, XG_FLAG_BYTES);
    if (e != hipSuccess) { (void)hipFree(p); delete x; set_error("xgmi: hipMemset failed: %s", hipGetErrorString(e)); return VR_ERR_HIP; }
    {   // the window's identity: a peer verifies after mapping that it sees THIS allocation (a stale mapping of a freed
        // window would otherwise swallow its pushes silently)
        static unsigned long long counter = 0;
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        x->nonce = ((unsigned long long)getpid() << 40) ^ ((unsigned long long)ts.tv_nsec << 8) ^ (unsigned long long)ts.tv_sec ^ (++counter << 56) ^ 0x9E3779B97F4A7C15ull;
        if (e == hipSuccess)
            e = hipMemcpy(reinterpret_cast<unsigned long long*>(p) + XG_NONCE, &x->nonce, sizeof(x->nonce), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();      // the clear must have HAPPENED before any peer can post into this window
    if (e != hipSuccess) { (void)hipFree(p); delete x; set_error("xgmi: hipDeviceSynchronize failed: %s", hipGetErrorString(e)); return VR_ERR_HIP; }
    if (hipHostMalloc((void**)&x->err_host, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void**)&x->err_host_dev, x->err_host, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (x->err_host) (void)hipHostFree(x->err_host);
        (void)hipFree(p); delete x;
        set_error("xgmi: pinned error word could not be allocated");
        return VR_ERR_HIP;
    }
    *x->err_host = 0ull;
    x->peer[rank] = x->window;
    x->attached = (world == 1);
    *out = x;
    return VR_OK;
}

extern "C" int vr_xgmi_detach(VrXgmi* x)
{
    if (!x) return VR_OK;
    (void)hipDeviceSynchronize();
    for (int p = 0; p < x->world; ++p)
        if (p != x->rank && x->peer[p]) { (void)hipIpcCloseMemHandle(x->peer[p]); x->peer[p] = nullptr; }
    x->attached = (x->world == 1);
    return VR_OK;
}

extern "C" int vr_xgmi_destroy(VrXgmi* x)
{
    if (!x) return VR_OK;
    vr_xgmi_detach(x);
    if (x->window) (void)hipFree(x->window);
    if (x->err_host) (void)hipHostFree(x->err_host);
    delete x;
    return VR_OK;
}

extern "C" int vr_xgmi_layout(const VrXgmi* x, VrXgmiLayout* out)
{
    if (!x || !out) { set_error("xgmi: NULL argument"); return VR_ERR_INVALID_ARGUMENT; }
    *out = x->lay;
    return VR_OK;
}

extern "C" void* vr_xgmi_window(VrXgmi* x) { return x ? x->window : nullptr; }

extern "C" int vr_xgmi_handle(VrXgmi* x, void* handle_out)
{
    if (!x || !handle_out) { set_error("xgmi: NULL argument"); return VR_ERR_INVALID_ARGUMENT; }
    static_assert(sizeof(hipIpcMemHandle_t) + 8 == VR_XGMI_HANDLE_BYTES, "hipIpcMemHandle_t size");
    hipIpcMemHandle_t h;
    VR_HIP(hipIpcGetMemHandle(&h, x->window));
    memcpy(handle_out, &h, sizeof(h));
    memcpy((char*)handle_out + sizeof(h), &x->nonce, 8);
    return VR_OK;
}

extern "C" int vr_xgmi_attach(VrXgmi* x, const void* handles)
{
    if (!x || !handles) { set_error("xgmi: NULL argument"); return VR_ERR_INVALID_ARGUMENT; }
    if (x->attached) return VR_OK;
    for (int p = 0; p < x->world; ++p) {
        if (p == x->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + (size_t)p * VR_XGMI_HANDLE_BYTES, sizeof(h));
        void* ptr = nullptr;
        VR_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
        x->peer[p] = (float*)ptr;
        unsigned long long want = 0, seen = 0;
        memcpy(&want, (const char*)handles + (size_t)p * VR_XGMI_HANDLE_BYTES + sizeof(h), 8);
        VR_HIP(hipMemcpy(&seen, reinterpret_cast<unsigned long long*>(ptr) + XG_NONCE, sizeof(seen), hipMemcpyDeviceToHost));
        if (seen != want) {
            set_error("xgmi: the mapping of rank %d's window does not show that window (stale IPC mapping?)", p);
            return VR_ERR_HIP;
        }
    }
    x->attached = true;
    return VR_OK;
}

static int xg_grid(long floats)
{
    const long g = (floats + XG_CHUNK - 1) / XG_CHUNK;
    return (int)(g < 1 ? 1 : (g > 128 ? 128 : g));      // link-bound kernels: a modest grid leaves the CUs to compute kernels
                                                        // (and, with N processes on ONE GPU, room for every process's pushes
                                                        // beside the others' spinning reduce workgroups: 8 x 128 x 256 threads)
}

extern "C" int vr_xgmi_allreduce(VrXgmi* x, const VrXgmiSegment* segs, int32_t count, float scale,
                                 int64_t* result_floats_offset, void* stream)
{
    if (!x || !x->attached) { set_error("xgmi: window not attached"); return VR_ERR_INVALID_ARGUMENT; }
    if (count < 1 || count > XG_MAXS || !segs || !result_floats_offset) { set_error("xgmi: 1 .. %d segments", XG_MAXS); return VR_ERR_INVALID_ARGUMENT; }
    if (int rf = xg_failed(x)) return rf;
    XgArgs a;
    memset(&a, 0, sizeof(a));
    long total = 0, slot = 0, res = 0, most = 0;
    for (int i = 0; i < count; ++i) {
        if (segs[i].n < 0 || (segs[i].n > 0 && !segs[i].src)) { set_error("xgmi: segment %d has a NULL source or a negative size", i); return VR_ERR_INVALID_ARGUMENT; }
        XgSeg& s = a.seg[i];
        s.src = segs[i].src; s.n = (long)segs[i].n;
        s.shard = align4((s.n + x->world - 1) / x->world);
        s.recv_off = slot; s.result_off = res;
        result_floats_offset[i] = res;
        slot += s.shard; res += align4(s.n); total += s.n;
        if (s.shard * x->world > most) most = s.shard * x->world;
    }
    if (total > x->reduce_floats) { set_error("xgmi: %ld floats exceed the window's all-reduce capacity (%ld)", total, x->reduce_floats); return VR_ERR_INVALID_ARGUMENT; }
    for (int p = 0; p < x->world; ++p) a.peer[p] = x->peer[p];
    a.nseg = count; a.rank = x->rank; a.world = x->world;
    a.recv_offset = x->lay.recv_offset; a.result_offset = x->lay.result_offset; a.slot_stride = slot;
    a.epoch = ++x->epoch_r;
    a.scale = scale;
    a.wait_ticks = x->wait_ticks;
    a.err_host = x->err_host_dev;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_xg_push, dim3(xg_grid(most)), dim3(XG_THREADS), 0, s, a);
    hipLaunchKernelGGL(k_xg_reduce, dim3(xg_grid(most / x->world)), dim3(XG_THREADS), 0, s, a);
    hipLaunchKernelGGL(k_xg_wait, dim3(1), dim3(64), 0, s, x->window, XG_FLAG_B, x->world, a.epoch, x->err_host_dev, x->wait_ticks);
    if (hipGetLastError() != hipSuccess) { set_error("xgmi: all-reduce launch failed"); return VR_ERR_HIP; }
    return VR_OK;
}

extern "C" int vr_xgmi_allgather_begin(VrXgmi* x, const VrXgmiSegment* segs, int32_t count, int32_t parity,
                                       int64_t* slot_floats_offset, void* stream)
{
    if (!x || !x->attached) { set_error("xgmi: window not attached"); return VR_ERR_INVALID_ARGUMENT; }
    if (count < 1 || count > XG_MAXS || !segs || !slot_floats_offset || (parity != 0 && parity != 1))
        { set_error("xgmi: 1 .. %d segments, parity 0 or 1", XG_MAXS); return VR_ERR_INVALID_ARGUMENT; }
    if (int rf = xg_failed(x)) return rf;
    XgGatherArgs g;
    memset(&g, 0, sizeof(g));
    long off = 0;
    for (int i = 0; i < count; ++i) {
        if (segs[i].n < 0 || (segs[i].n > 0 && !segs[i].src)) { set_error("xgmi: segment %d has a NULL source or a negative size", i); return VR_ERR_INVALID_ARGUMENT; }
        g.src[i] = segs[i].src; g.n[i] = (long)segs[i].n; g.off[i] = off;
        slot_floats_offset[i] = off;
        off += align4(g.n[i]);
    }
    if (off > x->lay.gather_slot) { set_error("xgmi: %ld floats exceed the window's all-gather capacity (%ld)", off, (long)x->gather_floats); return VR_ERR_INVALID_ARGUMENT; }
    for (int p = 0; p < x->world; ++p) g.peer[p] = x->peer[p];
    g.nseg = count; g.rank = x->rank; g.world = x->world; g.parity = parity;
    g.slot_base = x->lay.gather_offset[parity] + (long)x->rank * x->lay.gather_slot;
    g.epoch = ++x->epoch_g[parity];
    hipLaunchKernelGGL(k_xg_gather, dim3(xg_grid(off)), dim3(XG_THREADS), 0, (hipStream_t)stream, g);
    if (hipGetLastError() != hipSuccess) { set_error("xgmi: all-gather launch failed"); return VR_ERR_HIP; }
    return VR_OK;
}

extern "C" int vr_xgmi_allgather_wait(VrXgmi* x, int32_t parity, void* stream)
{
    if (!x || !x->attached || (parity != 0 && parity != 1)) { set_error("xgmi: window not attached / bad parity"); return VR_ERR_INVALID_ARGUMENT; }
    if (int rf = xg_failed(x)) return rf;
    hipLaunchKernelGGL(k_xg_wait, dim3(1), dim3(64), 0, (hipStream_t)stream, x->window,
                       XG_FLAG_G + parity * XG_MAXR * XG_FLAG_STRIDE, x->world, x->epoch_g[parity], x->err_host_dev, x->wait_ticks);
    if (hipGetLastError() != hipSuccess) { set_error("xgmi: wait launch failed"); return VR_ERR_HIP; }
    return VR_OK;
}

extern "C" int vr_xgmi_check(VrXgmi* x, void* stream)
{
    if (!x) { set_error("xgmi: NULL argument"); return VR_ERR_INVALID_ARGUMENT; }
    VR_HIP(hipStreamSynchronize((hipStream_t)stream));
    unsigned long long err = 0;
    VR_HIP(hipMemcpy(&err, reinterpret_cast<unsigned long long*>(x->window) + XG_ERR, sizeof(err), hipMemcpyDeviceToHost));
    if (err && x->err_host) __atomic_store_n(x->err_host, 1ull, __ATOMIC_RELEASE);
    return xg_failed(x);
}

extern "C" int vr_xgmi_failed(const VrXgmi* x)
{
    return x && x->err_host && __atomic_load_n(x->err_host, __ATOMIC_ACQUIRE) != 0ull ? 1 : 0;
}

extern "C" int vr_xgmi_set_wait_bound(VrXgmi* x, double seconds)
{
    if (!x || !(seconds > 0.0) || seconds > 3600.0) { set_error("xgmi: the wait bound must lie in (0, 3600] seconds"); return VR_ERR_INVALID_ARGUMENT; }
    x->wait_ticks = (unsigned long long)(seconds * (double)XG_TICKS_PER_S);
    if (x->wait_ticks == 0ull) x->wait_ticks = 1ull;
    return VR_OK;
}
