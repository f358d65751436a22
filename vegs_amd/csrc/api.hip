// api.hip -- the C ABI of libvegsrast.so (include/vegs_rast.h): argument validation, buffer
// layout, and the kernel sequence of forward / backward / mark_visible.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <vector>

#include "../../include/vegs_loss.h"
#include "vr_host.h"
#include "vr_segment.h"

namespace vr {

static_assert(FLAG_SCALE_MODIFIED == VR_FLAG_SCALE_MODIFIED && FLAG_DEPTH_NORMALIZED == VR_FLAG_DEPTH_NORMALIZED &&
              FLAG_EXTRA_NO_ALPHA_GRAD == VR_FLAG_EXTRA_NO_ALPHA_GRAD && FLAG_FILL_EMPTY == VR_FLAG_FILL_EMPTY &&
              FLAG_DETERMINISTIC == VR_FLAG_DETERMINISTIC && FLAG_SCAN_BINNING == VR_FLAG_SCAN_BINNING &&
              FLAG_ROUNDS_OFF == VR_FLAG_ROUNDS_OFF && FLAG_ROUNDS_ON == VR_FLAG_ROUNDS_ON &&
              FLAG_RAW_PARAMS == VR_FLAG_RAW_PARAMS && FLAG_FAST_EXP == VR_FLAG_FAST_EXP &&
              FLAG_VERIFY_BINNING == VR_FLAG_VERIFY_BINNING && FLAG_FULL_TILE_LISTS == VR_FLAG_FULL_TILE_LISTS &&
              FLAG_ACCUMULATE_GRADS == VR_FLAG_ACCUMULATE_GRADS,
              "device-side flag constants must match include/vegs_rast.h");
constexpr uint32_t KNOWN_FLAGS = FLAG_SCALE_MODIFIED | FLAG_DEPTH_NORMALIZED | FLAG_EXTRA_NO_ALPHA_GRAD | FLAG_FILL_EMPTY |
                                 FLAG_DETERMINISTIC | FLAG_SCAN_BINNING | FLAG_ROUNDS_OFF | FLAG_ROUNDS_ON | FLAG_RAW_PARAMS | FLAG_FAST_EXP | FLAG_VERIFY_BINNING | FLAG_FULL_TILE_LISTS | FLAG_ACCUMULATE_GRADS;

static thread_local char g_err[512] = "";
static thread_local VrCounters g_counters = {0, 0, 0, 0, 0};
// host-pinned, coherent mailbox that the totals kernel writes (V, R, min key, max key, guard word; then a sequence
// number the host polls) and the device-side guard word: one per (host thread, device)
constexpr int MAX_DEVICES = 64;
// pinned words: [0..11] six 64-bit {value, sequence number} pairs: the totals of the forward in flight (V, R, min key, max
// key, 0, large rectangles), each stamped with that forward's sequence number; from word RING_AT a ring of RING_SLOTS
// {seq, guard} pairs: slot seq % RING_SLOTS is filled by the LAST binning kernel of forward `seq` with the guard word as
// it stands after all of that view's waiting passes (k_tile_ranges).  vr_backward / the export calls find the slot
// through VrSaved.ticket = (mailbox id + 1) << 32 | seq -- PyTorch runs the op's backward on its autograd thread, so
// the mailboxes are also registered process-wide.
// RING_SLOTS bounds the forwards that may lie between a forward and ITS backward on one thread and device (each holds its
// saved buffers -- GBs at the headline size -- so 1024 outstanding forwards is beyond any batch; 8 KB of pinned memory).
// Beyond that the slot has been reused: check_ticket then says so instead of guessing.
constexpr int RING_AT = 64, RING_SLOTS = 1024, MAIL_BYTES = (RING_AT + 2 * RING_SLOTS) * 4;
// DEVICE guard words: one per forward IN FLIGHT (ABI v9; until v8 one per thread and device, so that two views in flight on
// two streams failed each other).  Forward `seq` owns word seq % GUARD_SLOTS: its totals kernel clears it, its binning
// kernels raise it, its last binning kernel copies it into the forward's ring slot.  64 forwards of one thread cannot be
// in their binning at once (each blocks its host thread once, behind its own compaction).
constexpr int GUARD_SLOTS = 64;
// ring slot word 1: 0 = clean, GUARD_RAISED = a wait of that forward's binning ran out, GUARD_REPORTED = ... and a later
// forward of the thread has already reported it (the view's own backward still fails)
constexpr uint32_t GUARD_RAISED = 1u, GUARD_REPORTED = 2u;
struct Mailbox {
    uint32_t* pinned; uint32_t* pinned_dev; uint32_t seq; uint32_t* guard; uint32_t id;
    uint32_t unchecked;      // oldest forward whose ring slot no later forward has looked at yet (report_earlier)
};
static thread_local Mailbox g_mail[MAX_DEVICES] = {};
struct MailRef { uint32_t* pinned; uint32_t* guard; int dev; };
static std::mutex g_mail_mu;
static std::vector<MailRef> g_mail_reg;
static thread_local int g_raise_guard = 0;        // test hook: vr_debug_raise_guard (1 = raise the word, 2 = lose a workgroup, 3 = impatient walkers, 4 = break the sorted ids)
static thread_local int g_rebinned = 0;           // views re-binned under VR_FLAG_VERIFY_BINNING (vr_debug_rebinned)

// the calling thread's mailbox for the current device, created on first use
static int get_mailbox(int dev_id, hipStream_t s, Mailbox** out);
static int fail(int code, const char* fmt, ...);

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

static int get_mailbox(int dev_id, hipStream_t s, Mailbox** out)
{
    Mailbox& mail = g_mail[dev_id];
    if (!mail.pinned) {
        VR_HIP(hipHostMalloc((void**)&mail.pinned, MAIL_BYTES, hipHostMallocMapped | hipHostMallocCoherent));
        memset(mail.pinned, 0, MAIL_BYTES);
        VR_HIP(hipHostGetDevicePointer((void**)&mail.pinned_dev, mail.pinned, 0));
    }
    if (!mail.guard) {   // device words raised by a binning kernel whose bounded wait ran out (binning.hip): GUARD_SLOTS of them
        VR_HIP(hipMalloc((void**)&mail.guard, GUARD_SLOTS * sizeof(uint32_t)));
        VR_HIP(hipMemsetAsync(mail.guard, 0, GUARD_SLOTS * sizeof(uint32_t), s));
        std::lock_guard<std::mutex> lk(g_mail_mu);
        g_mail_reg.push_back({mail.pinned, mail.guard, dev_id});
        mail.id = (uint32_t)g_mail_reg.size();   // 1-based
    }
    *out = &mail;
    return 0;
}

// The guard word of the forward behind `ticket` (see Mailbox).  Returns VR_OK when that forward's binning finished
// without a timed-out wait (or nothing can be said: no ticket, slot long overwritten), VR_ERR_HIP when one timed out --
// the view's lists, images and everything derived from them are invalid.
//   wait = true  (the calls that read the lists back and synchronise anyway): waits until the slot is posted -- normally
//                not at all, the caller comes after the forward's launches --, with a stream synchronisation after 20 ms;
//   wait = false (vr_backward, ABI v9): NEVER blocks the host.  A slot that is not posted yet says nothing: *pending is set
//                and the caller asks again once it has queued its work (a tripped view's tile ranges are all empty, so the
//                backward's kernels are harmless on it); if the answer is still missing then, the thread's next forward
//                reports the view (report_earlier).
static int check_ticket(uint64_t ticket, hipStream_t s, bool wait = true, bool* pending = nullptr)
{
    if (pending) *pending = false;
    if (ticket == 0) return VR_OK;
    const uint32_t id = (uint32_t)(ticket >> 32), seq = (uint32_t)ticket;
    MailRef ref;
    {
        std::lock_guard<std::mutex> lk(g_mail_mu);
        if (id == 0 || id > g_mail_reg.size()) return VR_OK;
        ref = g_mail_reg[id - 1];
    }
    uint32_t* slot = ref.pinned + RING_AT + 2 * (seq % RING_SLOTS);
    const auto t0 = std::chrono::steady_clock::now();
    bool synced = false;
    for (unsigned spins = 1;; ++spins) {
        const uint32_t got = __atomic_load_n(&slot[0], __ATOMIC_ACQUIRE);
        if (got == seq) break;
        if (synced) return VR_OK;                                // never posted (a forward that failed before its binning)
        if ((int32_t)(got - seq) > 0) {                          // reused: more than RING_SLOTS forwards since this one
            set_error("more than %d forwards lie between this call and the forward it belongs to: the record of that "
                      "forward's binning guard is gone (run the backward closer to its forward)", RING_SLOTS);
            return VR_ERR_INVALID_ARGUMENT;
        }
        if (!wait) { if (pending) *pending = true; return VR_OK; }
        __builtin_ia32_pause();
        if ((spins & 4095u) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
            VR_HIP(hipStreamSynchronize(s));   // the forward ran on this stream (same-stream contract of the op)
            synced = true;
        }
    }
    if (__atomic_load_n(&slot[1], __ATOMIC_RELAXED) == 0u) return VR_OK;
    __atomic_store_n(&slot[1], GUARD_REPORTED, __ATOMIC_RELAXED);   // (the device word is the forward's own: nothing to clear)
    set_error("this view's binning failed (a look-back wait timed out, or the depth sort's output was not a permutation of "
              "the visible ids): its lists, images and gradients are invalid (re-render the view; VR_FLAG_SCAN_BINNING "
              "avoids inter-workgroup waits)");
    return VR_ERR_HIP;
}

// Forwards that never get a backward (evaluation under no_grad) cannot fail "their own" later call: every vr_forward first
// looks at the ring slots of the thread's earlier forwards that nobody has looked at yet and reports the first one whose
// binning tripped (once).  Host memory only; a slot that is not posted yet is left for the next call.
static int report_earlier(Mailbox& mail)
{
    while ((int32_t)(mail.seq - mail.unchecked) >= 0 && mail.unchecked != 0u) {
        const uint32_t q = mail.unchecked;
        uint32_t* slot = mail.pinned + RING_AT + 2 * (q % RING_SLOTS);
        const uint32_t got = __atomic_load_n(&slot[0], __ATOMIC_ACQUIRE);
        if (got != q) {
            // not posted (yet): still running, or a forward that returned an error before its binning.  Old ones are given up.
            if ((int32_t)(got - q) > 0 || mail.seq - q >= 32u) { mail.unchecked = q + 1u ? q + 1u : 1u; continue; }
            break;
        }
        mail.unchecked = q + 1u ? q + 1u : 1u;
        uint32_t want = GUARD_RAISED;
        if (__atomic_compare_exchange_n(&slot[1], &want, GUARD_REPORTED, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED))
            return fail(VR_ERR_HIP, "a look-back wait in an earlier binning pass timed out; that view's output is invalid");
    }
    return VR_OK;
}

// Waits until the totals kernel has published this call's sequence number in the mailbox.  Polls host memory (the
// kernel's system-scope release store); every ~50 us of polling it also asks the stream for errors, and after 20 ms it
// falls back to a blocking stream synchronisation and a plain copy of the device-side totals.
// The six totals of forward `seq`, once all six mailbox words carry that sequence number (each word is one 8-byte store of
// the kernel: value and stamp cannot be torn apart, and no assumption is made about the order the stores arrive in).
static bool read_mailbox(const uint32_t* pinned, uint32_t seq, uint32_t out[6])
{
    const unsigned long long* m64 = reinterpret_cast<const unsigned long long*>(pinned);
    for (int k = 0; k < 6; ++k) {
        const unsigned long long w = __atomic_load_n(&m64[k], __ATOMIC_ACQUIRE);
        if ((uint32_t)(w >> 32) != seq) return false;
        out[k] = (uint32_t)w;
    }
    return true;
}
static int wait_mailbox(const uint32_t* pinned, uint32_t seq, const uint32_t* totals_dev, hipStream_t s, uint32_t out[6])
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; ++spins) {
        if (read_mailbox(pinned, seq, out)) return 0;
        __builtin_ia32_pause();
        if ((spins & 4095u) == 0u) {
            const hipError_t q = hipStreamQuery(s);
            if (q != hipSuccess && q != hipErrorNotReady) { set_error("%s while waiting for the list sizes", hipGetErrorString(q)); return VR_ERR_HIP; }
            const auto dt = std::chrono::steady_clock::now() - t0;
            if (q == hipSuccess || dt > std::chrono::milliseconds(20)) {
                VR_HIP(hipStreamSynchronize(s));
                if (read_mailbox(pinned, seq, out)) return 0;
                VR_HIP(hipMemcpy(out, totals_dev, 6 * sizeof(uint32_t), hipMemcpyDeviceToHost));
                return 0;
            }
        }
    }
}

// ---- stage profiler: pairs of events on the caller's stream, resolved in vr_profile_collect
// (process-wide, not thread-local: PyTorch runs the op's backward on its autograd thread)
struct ProfRec { int stage; hipEvent_t a, b; };
static std::atomic<int> g_prof_level{0};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;
static thread_local hipEvent_t g_prof_open[VR_STAGE_COUNT];

static bool prof_on(int stage)
{
    int lvl = g_prof_level.load(std::memory_order_relaxed);
    if (lvl >= 2) return true;
    return lvl == 1 && stage == VR_STAGE_K_SEG_BWD;   // level 1: only the roofline kernel (every event pair costs a ~5 us bubble)
}
static hipEvent_t prof_event()
{
    hipEvent_t e = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (!g_prof_pool.empty()) { e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    }
    if (hipEventCreate(&e) != hipSuccess) e = nullptr;
    return e;
}
void prof_begin(int stage, hipStream_t s)
{
    if (!prof_on(stage)) return;
    if (g_prof_level.load(std::memory_order_relaxed) == 1) {   // level 1 samples one launch in four (12 us of bubbles
        static thread_local unsigned n = 0;                     // per view otherwise: 0.8 % of the headline)
        if ((n++ & 3u) != 0u) return;
    }
    hipEvent_t e = prof_event();
    if (e) (void)hipEventRecord(e, s);
    g_prof_open[stage] = e;
}
void prof_end(int stage, hipStream_t s)
{
    if (!prof_on(stage) || !g_prof_open[stage]) return;
    hipEvent_t e = prof_event();
    if (e) {
        (void)hipEventRecord(e, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof_recs.push_back({stage, g_prof_open[stage], e});
    }
    g_prof_open[stage] = nullptr;
}

static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

static int make_camera(const VrSettings* st, int M, Camera* cam)
{
    if (!st) return fail(VR_ERR_INVALID_ARGUMENT, "settings is NULL");
    if (st->image_height <= 0 || st->image_width <= 0)
        return fail(VR_ERR_INVALID_ARGUMENT, "image size must be positive (got %d x %d)", st->image_height,
                    st->image_width);
    if (!st->bg || !st->viewmatrix || !st->projmatrix || !st->campos)
        return fail(VR_ERR_INVALID_ARGUMENT, "bg, viewmatrix, projmatrix and campos must be device pointers");
    if (st->sh_degree < 0 || st->sh_degree > 3)
        return fail(VR_ERR_INVALID_ARGUMENT, "sh_degree must be in 0..3 (got %d)", st->sh_degree);
    cam->H = st->image_height;
    cam->W = st->image_width;
    cam->gx = (cam->W + TILE - 1) / TILE;
    cam->gy = (cam->H + TILE - 1) / TILE;
    cam->tanfovx = st->tanfovx;
    cam->tanfovy = st->tanfovy;
    cam->fx = (float)cam->W / (2.0f * st->tanfovx);
    cam->fy = (float)cam->H / (2.0f * st->tanfovy);
    if (st->flags & ~KNOWN_FLAGS)
        return fail(VR_ERR_INVALID_ARGUMENT, "unknown bits in settings.flags (0x%x)", st->flags & ~KNOWN_FLAGS);
    cam->flags = st->flags;
    cam->mod = st->scale_modifier;
    cam->deg = st->sh_degree;
    cam->M = M;
    cam->view = st->viewmatrix;
    cam->proj = st->projmatrix;
    cam->campos = st->campos;
    cam->bg = st->bg;
    return 0;
}

static int check_inputs(const VrSettings* st, const VrInputs* in)
{
    if (!in) return fail(VR_ERR_INVALID_ARGUMENT, "inputs is NULL");
    if (in->P < 0) return fail(VR_ERR_INVALID_ARGUMENT, "means3D must have dimensions (num_points, 3)");
    if (in->P > 0 && (!in->means3D || !in->opacities))
        return fail(VR_ERR_INVALID_ARGUMENT, "means3D and opacities are required");
    if ((in->shs == nullptr) == (in->colors_precomp == nullptr) && in->P > 0)
        return fail(VR_ERR_INVALID_ARGUMENT, "exactly one of shs and colors_precomp must be given");
    bool sr = in->scales != nullptr || in->rotations != nullptr;
    if (in->P > 0) {
        if (sr == (in->cov3D_precomp != nullptr) || (sr && (!in->scales || !in->rotations)))
            return fail(VR_ERR_INVALID_ARGUMENT,
                        "exactly one of (scales, rotations) and cov3D_precomp must be given");
        if ((st->flags & FLAG_RAW_PARAMS) && !sr)
            return fail(VR_ERR_INVALID_ARGUMENT, "VR_FLAG_RAW_PARAMS needs scales and rotations (not cov3D_precomp)");
        if (in->shs) {
            int K = (st->sh_degree + 1) * (st->sh_degree + 1);
            if (in->M < K)
                return fail(VR_ERR_INVALID_ARGUMENT, "shs holds %d coefficients but sh_degree %d needs %d", in->M,
                            st->sh_degree, K);
            if (in->shs_rest && (in->M < 2 || in->M > 16))
                return fail(VR_ERR_INVALID_ARGUMENT, "split SH storage needs 2..16 coefficients in total (got %d)", in->M);
            if (in->shs_tail) {
                if (in->tail_start < 0 || in->tail_start > in->P)
                    return fail(VR_ERR_INVALID_ARGUMENT, "tail_start must lie in [0, P] (got %lld, P = %d)",
                                (long long)in->tail_start, in->P);
                if ((in->M * 3) % 4 != 0 || in->M > 16 ||
                    ((reinterpret_cast<uintptr_t>(in->shs) | reinterpret_cast<uintptr_t>(in->shs_rest) |
                      reinterpret_cast<uintptr_t>(in->shs_tail)) & 15u))
                    return fail(VR_ERR_INVALID_ARGUMENT,
                                "an SH tail needs 4 | 3 M (M <= 16) and 16-byte aligned SH arrays (M = %d)", in->M);
            }
        } else if (in->shs_rest || in->shs_tail) {
            return fail(VR_ERR_INVALID_ARGUMENT, "shs_rest / shs_tail given without shs");
        }
    }
    return 0;
}

// per-pixel state kept for the backward: final transmittance, contributor count and (written only with
// VR_FLAG_DEPTH_NORMALIZED) the un-normalised depth sum
struct ImageLayout { size_t counters, final_T, n_contrib, dsum, total; };
static ImageLayout image_layout(size_t N)
{
    ImageLayout L;
    L.counters = 0;
    L.final_T = 256;
    L.n_contrib = L.final_T + align_up(N * 4, 256);
    L.dsum = L.n_contrib + align_up(N * 4, 256);
    L.total = L.dsum + align_up(N * 4, 256);
    return L;
}
// binning buffer: tile ranges | segment table (seg_off[T+1], seg_info[S] int4) | point list | boundary transmittances per (segment, pixel) |
// segment-local channel sums per (segment, channel, pixel) -- the backward derives its w*u sums from them
struct BinLayout { size_t ranges, seg_off, seg_needed, point_list, tbuf, part, segmask, total; };
static BinLayout bin_layout(size_t T, size_t R)
{
    BinLayout L;
    L.ranges = 0;
    L.seg_off = align_up(T * 8, 256);
    L.seg_needed = L.seg_off + align_up(seg_table_words((int)T, seg_capacity((long)R, (int)T)) * 4, 256);
    L.point_list = L.seg_needed + align_up(T * 4, 256);
    L.tbuf = L.point_list + align_up((R > 0 ? R : 1) * 4, 256);
    L.part = L.tbuf + align_up(seg_capacity((long)R, (int)T) * 256 * sizeof(float), 256);
    L.segmask = L.part + align_up(seg_capacity((long)R, (int)T) * 13 * 256 * sizeof(float), 256);
    L.total = L.segmask + align_up(seg_capacity((long)R, (int)T) * 16 * sizeof(unsigned long long), 256);
    return L;
}

}  // namespace vr

using namespace vr;

extern "C" {

int vr_abi_version(void) { return VR_ABI_VERSION; }

const char* vr_last_error(void) { return g_err; }

void vr_get_counters(VrCounters* out)
{
    if (out) *out = g_counters;
}

int vr_forward(const VrSettings* st, const VrInputs* in, const VrOutputs* out, VrAllocFn alloc, void* user,
               void* stream, VrSaved* saved)
{
    g_err[0] = 0;
    if (!out || !saved || !alloc) return fail(VR_ERR_INVALID_ARGUMENT, "outputs, saved and alloc are required");
    int rc = check_inputs(st, in);
    if (rc) return rc;
    Camera cam;
    rc = make_camera(st, in->M, &cam);
    if (rc) return rc;
    if (!out->color || !out->depth || !out->cov_quat || !out->cov_scale || !out->alpha || (in->P > 0 && !out->radii))
        return fail(VR_ERR_INVALID_ARGUMENT, "all six output arrays are required");
    hipStream_t s = (hipStream_t)stream;
    const bool debug = st->debug != 0;
    const int P = in->P;
    const size_t N = (size_t)cam.H * cam.W, T = (size_t)cam.gx * cam.gy;

    int dev_id = 0;
    VR_HIP(hipGetDevice(&dev_id));
    if (dev_id < 0 || dev_id >= MAX_DEVICES) return fail(VR_ERR_NO_DEVICE, "device ordinal %d out of range", dev_id);
    Mailbox* mailp = nullptr;
    rc = get_mailbox(dev_id, s, &mailp);
    if (rc) return rc;
    Mailbox& mail = *mailp;
    uint32_t* const g_pinned = mail.pinned;
    rc = report_earlier(mail);        // an EARLIER forward of this thread whose binning tripped and that nobody has reported
    if (rc) return rc;
    uint32_t* guard_word = mail.guard;       // this forward's device guard word (set with its sequence number below)

    // ---- buffers that survive until backward
    const size_t p1 = (size_t)(P > 0 ? P : 1);
    // records | clamp bits | d colour / d direction (9 floats per Gaussian, written for visible ones in the SH modes)
    const size_t geom_clamp = align_up(p1 * sizeof(Splat), 256), geom_shd = geom_clamp + align_up(p1, 256);
    void* geom = alloc(user, VR_BUF_GEOM, geom_shd + align_up(p1 * 9 * sizeof(float), 256));
    const ImageLayout IL = image_layout(N);
    void* image = alloc(user, VR_BUF_IMAGE, IL.total);
    // ---- transient, P-sized
    const size_t s1 = binning_stage1_scratch_bytes(P);
    const size_t arr = align_up(p1 * 4, 256);
    void* scr = alloc(user, VR_BUF_SCRATCH, 8 * arr + s1 + 256);
    if (!geom || !image || !scr) return fail(VR_ERR_ALLOC, "allocator returned NULL");
    Splat* rec = (Splat*)geom;
    float* final_T = (float*)((char*)image + IL.final_T);
    uint32_t* n_contrib = (uint32_t*)((char*)image + IL.n_contrib);
    uint4* rect = (uint4*)scr;                                   // 16 B per Gaussian: rectangle + tile mask
    uint32_t* depth_key = (uint32_t*)((char*)scr + 4 * arr);
    uint32_t* vis_key = (uint32_t*)((char*)scr + 5 * arr);
    uint32_t* vis_id = (uint32_t*)((char*)scr + 6 * arr);
    uint32_t* tile_count = (uint32_t*)((char*)scr + 7 * arr);      // list entries per Gaussian (what the rectangle + mask say, as one word)
    void* scan_scr = (char*)scr + 8 * arr;
    uint32_t* totals_dev = (uint32_t*)((char*)scr + 8 * arr + s1);
    bool ranges_zeroed = false, status_zeroed = false;

    uint32_t V = 0, R = 0, key_min = 0, n_huge = 0;
    int key_bits = 0;
    // R-sized buffers: requested up front when the caller supplied a capacity hint (see VrSaved)
    size_t Rcap = saved->binning_capacity > 0 ? (size_t)saved->binning_capacity : 0;
    void *binning = nullptr, *scr2 = nullptr, *scr3 = nullptr;
    if (Rcap > 0 && P > 0) {
        binning = alloc(user, VR_BUF_BINNING, bin_layout(T, Rcap).total);
        scr2 = alloc(user, VR_BUF_SCRATCH, binning_stage2_scratch_bytes(P, (long)Rcap, (int)T));
        scr3 = alloc(user, VR_BUF_SCRATCH, render_fwd_scratch_bytes((long)Rcap, (int)T) + 256);
        if (!binning || !scr2 || !scr3) return fail(VR_ERR_ALLOC, "allocator returned NULL");
    } else {
        Rcap = 0;
    }
    if (P > 0) {
        prof_begin(VR_STAGE_PREPROCESS, s);
        rc = launch_preprocess(cam, P, in->means3D, in->shs, in->shs_rest, in->shs_tail,
                               in->shs_tail ? (int)in->tail_start : P, in->colors_precomp, in->opacities, in->scales,
                               in->rotations, in->cov3D_precomp, rec, out->radii, rect, depth_key, tile_count,
                               (uint8_t*)geom + geom_clamp, (float*)((char*)geom + geom_shd), s, debug);
        prof_end(VR_STAGE_PREPROCESS, s);
        if (rc) return rc;
        prof_begin(VR_STAGE_COMPACT, s);
        // the one host<->device round trip of the forward pass: sizes of the data-dependent lists.  The totals kernel
        // writes them into the pinned mailbox itself; the host polls for this call's sequence number after it has
        // queued the compaction's apply kernel, so the round trip overlaps with that kernel instead of idling the GPU.
        const uint32_t seq = ++mail.seq ? mail.seq : ++mail.seq;   // never 0 (the mailbox's initial content)
        if (mail.unchecked == 0u) mail.unchecked = seq;
        guard_word = mail.guard + (seq % GUARD_SLOTS);             // cleared by the totals kernel, ahead of the binning
        rc = launch_compact_reduce(P, tile_count, depth_key, scan_scr, totals_dev, guard_word, mail.pinned_dev, seq, s, debug);
        if (rc) return rc;
        // the apply kernel also clears the tile ranges (when the binning buffer already exists): no fill launch
        // and the status words of the binning passes (when their scratch exists)
        uint32_t* rz = binning ? (uint32_t*)((char*)binning + bin_layout(T, Rcap).ranges) : nullptr;
        ranges_zeroed = rz != nullptr;
        status_zeroed = scr2 != nullptr;
        rc = launch_compact_apply(P, rect, depth_key, scan_scr, totals_dev, binning_tile_bits((int)T), vis_key, vis_id, rz,
                                  rz ? (long)(2 * T) : 0L, scr2 ? binning_stage2_status(scr2) : nullptr,
                                  scr2 ? binning_stage2_status_bytes(P, (long)Rcap, (int)T) : 0, s, debug);
        prof_end(VR_STAGE_COMPACT, s);
        if (rc) return rc;
        uint32_t tot[6];
        rc = wait_mailbox(g_pinned, seq, totals_dev, s, tot);
        if (rc) return rc;
        V = tot[0];
        R = tot[1];
        n_huge = tot[5];
        if (V > (uint32_t)P) return fail(VR_ERR_HIP, "the compaction reports %u visible Gaussians of %d", V, P);
        if (V > 0) {
            key_min = tot[2];
            uint32_t span = tot[3] - tot[2];
            while (span) { ++key_bits; span >>= 1; }
        }
    }
    if (Rcap == 0 || R > Rcap) {   // no hint, or the hint was too small: size for the actual R
        Rcap = R;
        ranges_zeroed = status_zeroed = false;   // fresh buffers
        binning = alloc(user, VR_BUF_BINNING, bin_layout(T, Rcap).total);
        scr2 = alloc(user, VR_BUF_SCRATCH, binning_stage2_scratch_bytes(P, (long)Rcap, (int)T));
        scr3 = alloc(user, VR_BUF_SCRATCH, render_fwd_scratch_bytes((long)Rcap, (int)T) + 256);
    }
    if (!binning || !scr2 || !scr3) return fail(VR_ERR_ALLOC, "allocator returned NULL");
    const BinLayout BL = bin_layout(T, Rcap);
    int2* ranges = (int2*)((char*)binning + BL.ranges);
    uint32_t* point_list = (uint32_t*)((char*)binning + BL.point_list);
    const bool lists = V > 0 && R > 0;
    const int raise = g_raise_guard;
    g_raise_guard = 0;
    rc = launch_binning(cam, P, (int)V, (long)R, key_min, key_bits, vis_key, vis_id, rect, scan_scr, scr2, point_list,
                        ranges, ranges_zeroed, status_zeroed, guard_word,
                        mail.pinned_dev + RING_AT + 2 * (mail.seq % RING_SLOTS), mail.seq, raise == 3 ? 0 : raise, n_huge, s, debug);
    if (rc) return rc;
    if (!lists && P > 0) {   // no binning kernel will post this forward's slot: the host does (nothing can have tripped)
        uint32_t* slot = g_pinned + RING_AT + 2 * (mail.seq % RING_SLOTS);
        __atomic_store_n(&slot[1], 0u, __ATOMIC_RELAXED);
        __atomic_store_n(&slot[0], mail.seq, __ATOMIC_RELEASE);
    }
    if ((st->flags & FLAG_VERIFY_BINNING) && lists && !(cam.flags & FLAG_SCAN_BINNING)) {
        // VR_FLAG_VERIFY_BINNING: the host waits for this view's guard word (posted by the last binning kernel) BEFORE it
        // queues the render stage; a view whose look-back wait gave up is binned once more with the wait-free multi-launch
        // passes -- same lists bit for bit -- instead of being failed at its backward.  Costs the forward its run-ahead
        // over the binning (a launch-latency bubble per view): opt-in.
        uint32_t* slot = g_pinned + RING_AT + 2 * (mail.seq % RING_SLOTS);
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 1; __atomic_load_n(&slot[0], __ATOMIC_ACQUIRE) != mail.seq; ++spins) {
            __builtin_ia32_pause();
            if ((spins & 4095u) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
                VR_HIP(hipStreamSynchronize(s));
                break;
            }
        }
        if (__atomic_load_n(&slot[1], __ATOMIC_RELAXED) != 0u) {
            // the slot is posted a second time by the re-run's last kernel: until then it reads "not posted yet"
            __atomic_store_n(&slot[1], 0u, __ATOMIC_RELAXED);
            __atomic_store_n(&slot[0], 0u, __ATOMIC_RELEASE);
            VR_HIP(hipMemsetAsync(guard_word, 0, 4, s));
            // The first attempt's depth sort ping-pongs between (vis_key, vis_id) and its scratch pair, and after a wait that
            // really ran out its passes scattered from a short prefix: slots collide, others keep stale memory -- the pairs
            // the second attempt would sort are NOT the compaction's any more.  They are written again from what the first
            // attempt only read: the depth keys, the compaction's block sums and the totals (round-4 advisor finding).
            rc = launch_compact_apply(P, rect, depth_key, scan_scr, totals_dev, binning_tile_bits((int)T), vis_key, vis_id,
                                      nullptr, 0L, nullptr, 0, s, debug);
            if (rc) return rc;
            Camera again = cam;
            again.flags |= FLAG_SCAN_BINNING;
            rc = launch_binning(again, P, (int)V, (long)R, key_min, key_bits, vis_key, vis_id, rect, scan_scr, scr2, point_list,
                                ranges, false, false, guard_word, mail.pinned_dev + RING_AT + 2 * (mail.seq % RING_SLOTS),
                                mail.seq, 0, n_huge, s, debug);
            if (rc) return rc;
            ++g_rebinned;
        }
    }
    prof_begin(VR_STAGE_RENDER_FWD, s);
    rc = launch_render_fwd(cam, (long)R, ranges, point_list, rec, (uint32_t*)((char*)binning + BL.seg_off),
                           (uint32_t*)((char*)binning + BL.seg_needed), (float*)((char*)binning + BL.tbuf),
                           (float*)((char*)binning + BL.part),
                           (unsigned long long*)((char*)binning + BL.segmask), scr3,
                           out->color, out->depth, out->cov_quat, out->cov_scale, out->alpha, final_T, n_contrib,
                           (float*)((char*)image + IL.dsum), saved->needed_hint, s, debug, raise == 3 ? 0 : -1);
    prof_end(VR_STAGE_RENDER_FWD, s);
    if (rc) return rc;
    // VEGS_DEBUG_BINNING=n: every n-th forward ends with a post-mortem of its depth sort (binning.hip: debug_verify_binning)
    static const int dbg_every = [] { const char* e = getenv("VEGS_DEBUG_BINNING"); return e ? atoi(e) : 0; }();
    if (dbg_every > 0 && lists && !(cam.flags & FLAG_SCAN_BINNING)) {
        static std::atomic<unsigned> dbg_n{0};
        if (++dbg_n % (unsigned)dbg_every == 0u) {
            const int nbad = debug_verify_binning(P, (int)V, (long)R, key_min, key_bits, vis_key, vis_id, depth_key, tile_count,
                                                  scan_scr, scr2, totals_dev, nullptr, guard_word, (int)T, s);
            if (nbad != 0) return fail(VR_ERR_HIP, "VEGS_DEBUG_BINNING: forward %u failed its depth-sort post-mortem (%d findings, stderr)", dbg_n.load(), nbad);
        }
    }

    saved->geom = geom;
    saved->binning = binning;
    saved->image = image;
    saved->num_rendered = (int64_t)R;
    saved->num_visible = (int64_t)V;
    saved->binning_capacity = (int64_t)Rcap;
    saved->ticket = lists ? ((uint64_t)mail.id << 32) | mail.seq : 0ull;
    g_counters.P = P;
    g_counters.num_visible = V;
    g_counters.num_rendered = R;
    g_counters.num_tiles = (int64_t)T;
    g_counters.num_pixels = (int64_t)N;
    return VR_OK;
}

// ---- backward, in two halves.  `first` = zero the accumulators, render backward (k_seg_u, k_seg_suffix, k_seg_bwd) and,
// with the factored SH gradient, its factor; `second` = k_preprocess_bwd.  gacc ([P][16] floats) carries the per-Gaussian
// sums from one to the other.
static int backward_checks(const VrSettings* st, const VrInputs* in, const int32_t* radii, const VrSaved* saved,
                           const VrInGrads* gin, Camera* cam, bool* sh_factored)
{
    int rc = check_inputs(st, in);
    if (rc) return rc;
    rc = make_camera(st, in->M, cam);
    if (rc) return rc;
    if (in->P == 0) return VR_OK;
    if (!radii || !saved->geom || !saved->binning || !saved->image)
        return fail(VR_ERR_INVALID_ARGUMENT, "radii and the saved forward buffers are required");
    if (!gin->dL_dmeans3D || !gin->dL_dmeans2D || !gin->dL_dopacities)
        return fail(VR_ERR_INVALID_ARGUMENT, "dL_dmeans3D, dL_dmeans2D and dL_dopacities are required");
    *sh_factored = in->shs && gin->dL_dcolors_sh;     // factored SH gradient: dL_dshs is not written
    if ((in->shs && !gin->dL_dshs && !*sh_factored) || (in->colors_precomp && !gin->dL_dcolors_precomp) ||
        (in->scales && (!gin->dL_dscales || !gin->dL_drotations)) || (in->cov3D_precomp && !gin->dL_dcov3D_precomp))
        return fail(VR_ERR_INVALID_ARGUMENT, "a gradient array is missing for a provided input");
    if (!*sh_factored && in->shs_rest && (!gin->dL_dshs || !gin->dL_dshs_rest))
        return fail(VR_ERR_INVALID_ARGUMENT, "split SH storage needs both dL_dshs and dL_dshs_rest");
    if (!*sh_factored && in->shs_tail) {
        if (!gin->dL_dshs_tail) return fail(VR_ERR_INVALID_ARGUMENT, "an SH tail needs dL_dshs_tail");
        if ((reinterpret_cast<uintptr_t>(gin->dL_dshs) | reinterpret_cast<uintptr_t>(gin->dL_dshs_rest) |
             reinterpret_cast<uintptr_t>(gin->dL_dshs_tail)) & 15u)
            return fail(VR_ERR_INVALID_ARGUMENT, "with an SH tail the SH gradient arrays must be 16-byte aligned");
    }
    return VR_OK;
}

static int backward_first(const Camera& cam, const VrSettings* st, const VrInputs* in, const int32_t* radii,
                          const VrSaved* saved, const VrOutGrads* gout, const VrInGrads* gin, bool sh_factored,
                          bool factor_now, float* gacc, VrAllocFn alloc, void* user, hipStream_t s)
{
    const bool debug = st->debug != 0;
    const int P = in->P;
    // a timed-out wait in THIS view's binning fails its own backward.  Asked without blocking the host (ABI v9): when the
    // forward's last binning kernel has not run yet the answer comes after this call's kernels are queued (below).
    bool guard_pending = false;
    int rc = check_ticket(saved->ticket, s, false, &guard_pending);
    if (rc) return rc;
    const size_t N = (size_t)cam.H * cam.W, T = (size_t)cam.gx * cam.gy;
    const ImageLayout IL = image_layout(N);
    const size_t Rcap = saved->binning_capacity > 0 ? (size_t)saved->binning_capacity : (size_t)saved->num_rendered;
    const BinLayout BL = bin_layout(T, Rcap);
    const Splat* rec = (const Splat*)saved->geom;
    const float* final_T = (const float*)((const char*)saved->image + IL.final_T);
    const uint32_t* n_contrib = (const uint32_t*)((const char*)saved->image + IL.n_contrib);
    const int2* ranges = (const int2*)((const char*)saved->binning + BL.ranges);
    const uint32_t* point_list = (const uint32_t*)((const char*)saved->binning + BL.point_list);

    prof_begin(VR_STAGE_BWD_ZERO, s);
    // the accumulators of the render backward are cleared by its first kernel (k_seg_u) when there is one
    const bool zero_in_kernel = saved->num_rendered > 0 && cam.gx * cam.gy > 0;
    if (!zero_in_kernel) {
        VR_HIP(hipMemsetAsync(gacc, 0, (size_t)P * 16 * sizeof(float), s));
        VR_HIP(hipMemsetAsync(gin->dL_dmeans2D, 0, (size_t)P * 3 * sizeof(float), s));
    }
    if (sh_factored) {
        // nothing to clear: every row of the [P,3] factor array is written
    } else if (in->shs_rest) {
        // split SH storage: the kernel writes every row of both gradient arrays
    } else if (in->shs_tail) {
        // SH tail: validated for the staged paths, which write every row of all the SH gradient arrays
    } else if (cam.flags & FLAG_ACCUMULATE_GRADS) {
        // accumulate mode: the caller's arrays hold the sum so far; nothing is cleared
    } else if (gin->dL_dshs && !preprocess_bwd_writes_all_sh(in->M, in->shs, gin->dL_dshs)) {
        VR_HIP(hipMemsetAsync(gin->dL_dshs, 0, (size_t)P * in->M * 3 * sizeof(float), s));
    }
    prof_end(VR_STAGE_BWD_ZERO, s);
    if (saved->num_rendered > 0) {
        void* scr = alloc(user, VR_BUF_SCRATCH, render_bwd_scratch_bytes((long)Rcap, (int)T) + 256);
        if (!scr) return fail(VR_ERR_ALLOC, "allocator returned NULL");
        void* det_scr = nullptr;
        if (cam.flags & FLAG_DETERMINISTIC) {
            det_scr = alloc(user, VR_BUF_SCRATCH, render_bwd_det_bytes((long)saved->num_rendered, P));
            if (!det_scr) return fail(VR_ERR_ALLOC, "allocator returned NULL");
        }
        ProfScope ps(VR_STAGE_RENDER_BWD, s);
        rc = launch_render_bwd(cam, (long)saved->num_rendered, ranges, point_list, rec,
                               (const uint32_t*)((const char*)saved->binning + BL.seg_off),
                               (const uint32_t*)((const char*)saved->binning + BL.seg_needed),
                               (const float*)((const char*)saved->binning + BL.tbuf),
                               (const float*)((const char*)saved->binning + BL.part),
                               (const unsigned long long*)((const char*)saved->binning + BL.segmask), scr, final_T,
                               n_contrib,
                               gout->dL_dcolor, gout->dL_ddepth, gout->dL_dcov_quat, gout->dL_dcov_scale,
                               gout->dL_dalpha, gacc, gin->dL_dmeans2D,
                               (const float*)((const char*)saved->image + IL.dsum), det_scr, P, zero_in_kernel, s, debug);
        if (rc) return rc;
    }
    if (sh_factored && factor_now) {
        rc = launch_sh_factor(P, radii, (const uint8_t*)saved->geom + align_up((size_t)P * sizeof(Splat), 256), gacc,
                              gin->dL_dcolors_sh, s, debug);
        if (rc) return rc;
    }
    // The second look at the guard (see above), now with this call's kernels queued.  If the forward's last binning kernel
    // STILL has not run, the host waits for it here (round 6, advisor finding): it lies in front of everything this call
    // queued, so the GPU has work for the whole wait, and "a failed view fails its OWN backward" holds without exception --
    // the gradients of a tripped view (zeros: its ranges are empty) never reach an optimizer step unnoticed.
    if (guard_pending) return check_ticket(saved->ticket, s, true, nullptr);
    return VR_OK;
}

static int backward_second(const Camera& cam, const VrSettings* st, const VrInputs* in, const int32_t* radii,
                           const VrSaved* saved, const VrInGrads* gin, bool sh_factored, bool store_factor,
                           const float* gacc, hipStream_t s)
{
    const int P = in->P;
    ProfScope ps2(VR_STAGE_PREPROCESS_BWD, s);
    return launch_preprocess_bwd(cam, P, in->means3D, in->shs, in->shs_rest, in->shs_tail ? (int)in->tail_start : P,
                                 in->colors_precomp, in->opacities, in->scales, in->rotations, in->cov3D_precomp, radii,
                                 (const uint8_t*)saved->geom + align_up((size_t)P * sizeof(Splat), 256),
                                 (const float*)((const char*)saved->geom + align_up((size_t)P * sizeof(Splat), 256) + align_up((size_t)P, 256)),
                                 gacc, gin->dL_dmeans2D, gin->dL_dmeans3D, gin->dL_dshs, gin->dL_dshs_rest,
                                 (in->shs_tail && !sh_factored) ? gin->dL_dshs_tail : nullptr, gin->dL_dcolors_precomp, gin->dL_dopacities, gin->dL_dscales, gin->dL_drotations,
                                 gin->dL_dcov3D_precomp, sh_factored ? gin->dL_dcolors_sh : nullptr, store_factor, s,
                                 st->debug != 0);
}

int vr_backward(const VrSettings* st, const VrInputs* in, const int32_t* radii, const VrSaved* saved,
                const VrOutGrads* gout, const VrInGrads* gin, VrAllocFn alloc, void* user, void* stream)
{
    g_err[0] = 0;
    if (!saved || !gout || !gin || !alloc) return fail(VR_ERR_INVALID_ARGUMENT, "saved, grads and alloc are required");
    Camera cam;
    bool sh_factored = false;
    int rc = backward_checks(st, in, radii, saved, gin, &cam, &sh_factored);
    if (rc || in->P == 0) return rc;
    hipStream_t s = (hipStream_t)stream;
    float* gacc = (float*)alloc(user, VR_BUF_SCRATCH, (size_t)in->P * 16 * sizeof(float));
    if (!gacc) return fail(VR_ERR_ALLOC, "allocator returned NULL");
    rc = backward_first(cam, st, in, radii, saved, gout, gin, sh_factored, false, gacc, alloc, user, s);
    if (rc) return rc;
    return backward_second(cam, st, in, radii, saved, gin, sh_factored, true, gacc, s);
}

int vr_backward_render(const VrSettings* st, const VrInputs* in, const int32_t* radii, const VrSaved* saved,
                       const VrOutGrads* gout, const VrInGrads* gin, VrAllocFn alloc, void* user, void* stream,
                       void** state)
{
    g_err[0] = 0;
    if (!saved || !gout || !gin || !alloc || !state)
        return fail(VR_ERR_INVALID_ARGUMENT, "saved, grads, alloc and state are required");
    *state = nullptr;
    Camera cam;
    bool sh_factored = false;
    int rc = backward_checks(st, in, radii, saved, gin, &cam, &sh_factored);
    if (rc || in->P == 0) return rc;
    float* gacc = (float*)alloc(user, VR_BUF_BACKWARD, (size_t)in->P * 16 * sizeof(float));
    if (!gacc) return fail(VR_ERR_ALLOC, "allocator returned NULL");
    rc = backward_first(cam, st, in, radii, saved, gout, gin, sh_factored, true, gacc, alloc, user, (hipStream_t)stream);
    if (rc) return rc;
    *state = gacc;
    return VR_OK;
}

int vr_backward_preprocess(const VrSettings* st, const VrInputs* in, const int32_t* radii, const VrSaved* saved,
                           const VrInGrads* gin, void* state, void* stream)
{
    g_err[0] = 0;
    if (!saved || !gin) return fail(VR_ERR_INVALID_ARGUMENT, "saved and grads are required");
    Camera cam;
    bool sh_factored = false;
    int rc = backward_checks(st, in, radii, saved, gin, &cam, &sh_factored);
    if (rc || in->P == 0) return rc;
    if (!state) return fail(VR_ERR_INVALID_ARGUMENT, "state of vr_backward_render is required");
    return backward_second(cam, st, in, radii, saved, gin, sh_factored, false, (const float*)state, (hipStream_t)stream);
}

int vr_profile_level(int level)
{
    return g_prof_level.exchange(level < 0 ? 0 : (level > 2 ? 2 : level));
}

int vr_profile_collect(double* ms, int64_t* count)
{
    g_err[0] = 0;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof_recs) {
        float t = 0.f;
        VR_HIP(hipEventSynchronize(r.b));
        VR_HIP(hipEventElapsedTime(&t, r.a, r.b));
        if (ms) ms[r.stage] += (double)t;
        if (count) count[r.stage] += 1;
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof_recs.clear();
    return VR_OK;
}

int vr_mark_visible(const float* xyz, int32_t P, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                    void* stream)
{
    g_err[0] = 0;
    (void)projmatrix;  // accepted for API parity; the test is on view-space depth only
    if (P < 0 || (P > 0 && (!xyz || !viewmatrix || !present)))
        return fail(VR_ERR_INVALID_ARGUMENT, "mark_visible: bad arguments");
    return launch_mark_visible(xyz, P, viewmatrix, present, (hipStream_t)stream);
}

int vr_knn3_mean_dist2(const float* points, int32_t N, float* out, VrAllocFn alloc, void* user, void* stream)
{
    g_err[0] = 0;
    if (N < 0 || (N > 0 && (!points || !out || !alloc))) return fail(VR_ERR_INVALID_ARGUMENT, "knn3: bad arguments");
    if (N == 0) return VR_OK;
    void* scr = alloc(user, VR_BUF_SCRATCH, knn3_scratch_bytes(N));
    if (!scr) return fail(VR_ERR_ALLOC, "allocator returned NULL");
    return launch_knn3(points, N, out, scr, (hipStream_t)stream, false);
}

int vr_photometric_forward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, float* sums, float* dmaps,
                           VrAllocFn alloc, void* user, void* stream)
{
    g_err[0] = 0;
    if (C <= 0 || H <= 0 || W <= 0) return fail(VR_ERR_INVALID_ARGUMENT, "image must be [C,H,W] with positive sizes");
    if (!image || !gt || !sums || !alloc) return fail(VR_ERR_INVALID_ARGUMENT, "image, gt, sums and alloc are required");
    void* scr = alloc(user, VR_BUF_SCRATCH, photometric_scratch_bytes(C, H, W));
    if (!scr) return fail(VR_ERR_ALLOC, "allocator returned NULL");
    return launch_photometric_fwd(image, gt, C, H, W, sums, dmaps, scr, (hipStream_t)stream, false);
}

int vr_photometric_backward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, const float* dmaps,
                            const float* g_l1, const float* g_ssim, float* dL_dimage, void* stream)
{
    g_err[0] = 0;
    if (C <= 0 || H <= 0 || W <= 0) return fail(VR_ERR_INVALID_ARGUMENT, "image must be [C,H,W] with positive sizes");
    if (!image || !gt || !dmaps || !dL_dimage)
        return fail(VR_ERR_INVALID_ARGUMENT, "image, gt, dmaps and dL_dimage are required");
    return launch_photometric_bwd(image, gt, C, H, W, dmaps, g_l1, g_ssim, dL_dimage, (hipStream_t)stream, false);
}

int vr_normal_guidance_forward(const float* cov_quat, const float* cov_scale, const float* normal, const float* R,
                               int32_t H, int32_t W, float* loss, VrAllocFn alloc, void* user, void* stream)
{
    g_err[0] = 0;
    if (H <= 0 || W <= 0) return fail(VR_ERR_INVALID_ARGUMENT, "maps must be [k,H,W] with positive sizes");
    if (!cov_quat || !cov_scale || !normal || !R || !loss || !alloc)
        return fail(VR_ERR_INVALID_ARGUMENT, "cov_quat, cov_scale, normal, R_cam2world, loss and alloc are required");
    void* scr = alloc(user, VR_BUF_SCRATCH, normal_guidance_scratch_bytes(H, W));
    if (!scr) return fail(VR_ERR_ALLOC, "allocator returned NULL");
    return launch_normal_guidance_fwd(cov_quat, cov_scale, normal, R, H, W, loss, scr, (hipStream_t)stream, false);
}

int vr_normal_guidance_backward(const float* cov_quat, const float* cov_scale, const float* normal, const float* R,
                                int32_t H, int32_t W, const float* g, float* dL_dquat, float* dL_dscale, void* stream)
{
    g_err[0] = 0;
    if (H <= 0 || W <= 0) return fail(VR_ERR_INVALID_ARGUMENT, "maps must be [k,H,W] with positive sizes");
    if (!cov_quat || !cov_scale || !normal || !R || !g || !dL_dquat || !dL_dscale)
        return fail(VR_ERR_INVALID_ARGUMENT, "all pointers are required");
    return launch_normal_guidance_bwd(cov_quat, cov_scale, normal, R, H, W, g, dL_dquat, dL_dscale, (hipStream_t)stream,
                                      false);
}

int vr_training_loss_forward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, const float* cov_quat,
                             const float* cov_scale, const float* normal, const float* R, float lambda_dssim,
                             float lambda_dnormal, int32_t guard_empty, float* loss, float* aux, float* dmaps,
                             VrAllocFn alloc, void* user, void* stream)
{
    g_err[0] = 0;
    if (C <= 0 || H <= 0 || W <= 0) return fail(VR_ERR_INVALID_ARGUMENT, "image must be [C,H,W] with positive sizes");
    if (!image || !gt || !cov_quat || !cov_scale || !normal || !R || !loss || !aux || !alloc)
        return fail(VR_ERR_INVALID_ARGUMENT, "training_loss: image, gt, the maps, R_cam2world, loss, aux and alloc are required");
    void* scr = alloc(user, VR_BUF_SCRATCH, training_loss_scratch_bytes(C, H, W));
    if (!scr) return fail(VR_ERR_ALLOC, "allocator returned NULL");
    return launch_training_loss_fwd(image, gt, C, H, W, cov_quat, cov_scale, normal, R, lambda_dssim, lambda_dnormal,
                                    guard_empty != 0, loss, aux, dmaps, scr, (hipStream_t)stream, false);
}

int vr_training_loss_backward(const float* image, const float* gt, int32_t C, int32_t H, int32_t W, const float* dmaps,
                              const float* cov_quat, const float* cov_scale, const float* normal, const float* R,
                              float lambda_dssim, float lambda_dnormal, int32_t guard_empty, const float* g,
                              float* dL_dimage, float* dL_dquat, float* dL_dscale, void* stream)
{
    g_err[0] = 0;
    if (C <= 0 || H <= 0 || W <= 0) return fail(VR_ERR_INVALID_ARGUMENT, "image must be [C,H,W] with positive sizes");
    if (!image || !gt || !dmaps || !cov_quat || !cov_scale || !normal || !R || !g || !dL_dimage || !dL_dquat || !dL_dscale)
        return fail(VR_ERR_INVALID_ARGUMENT, "training_loss backward: all pointers are required");
    return launch_training_loss_bwd(image, gt, C, H, W, dmaps, cov_quat, cov_scale, normal, R, lambda_dssim, lambda_dnormal,
                                    guard_empty != 0, g, dL_dimage, dL_dquat, dL_dscale, (hipStream_t)stream, false);
}

int vr_count_fragments(const VrSaved* saved, int32_t H, int32_t W, void* stream, int64_t* fragments)
{
    g_err[0] = 0;
    if (!saved || !saved->image || !fragments || H <= 0 || W <= 0)
        return fail(VR_ERR_INVALID_ARGUMENT, "count_fragments: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (int rt = check_ticket(saved->ticket, s)) return rt;
    const size_t N = (size_t)H * W;
    const ImageLayout IL = image_layout(N);
    unsigned long long* ctr = (unsigned long long*)((char*)saved->image + IL.counters);
    int rc = launch_count_fragments((const uint32_t*)((const char*)saved->image + IL.n_contrib), (long)N, ctr, s);
    if (rc) return rc;
    unsigned long long host = 0;
    VR_HIP(hipMemcpyAsync(&host, ctr, sizeof host, hipMemcpyDeviceToHost, s));
    VR_HIP(hipStreamSynchronize(s));
    *fragments = (int64_t)host;
    return VR_OK;
}

int vr_count_blended(const VrSaved* saved, int32_t H, int32_t W, void* stream, int64_t* blended)
{
    g_err[0] = 0;
    if (!saved || !saved->image || !saved->geom || !saved->binning || !blended || H <= 0 || W <= 0)
        return fail(VR_ERR_INVALID_ARGUMENT, "count_blended: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (int rt = check_ticket(saved->ticket, s)) return rt;
    Camera cam = {};
    cam.H = H; cam.W = W;
    cam.gx = (W + TILE - 1) / TILE;
    cam.gy = (H + TILE - 1) / TILE;
    const size_t N = (size_t)H * W, T = (size_t)cam.gx * cam.gy;
    const ImageLayout IL = image_layout(N);
    const BinLayout BL = bin_layout(T, saved->binning_capacity > 0 ? (size_t)saved->binning_capacity : (size_t)saved->num_rendered);
    unsigned long long* ctr = (unsigned long long*)((char*)saved->image + IL.counters) + 1;
    *blended = 0;
    if (saved->num_rendered == 0) return VR_OK;
    int rc = launch_count_blended(cam, (const int2*)((const char*)saved->binning + BL.ranges),
                                  (const uint32_t*)((const char*)saved->binning + BL.point_list), (const Splat*)saved->geom,
                                  (const uint32_t*)((const char*)saved->image + IL.n_contrib), ctr, s);
    if (rc) return rc;
    unsigned long long host = 0;
    VR_HIP(hipMemcpyAsync(&host, ctr, sizeof host, hipMemcpyDeviceToHost, s));
    VR_HIP(hipStreamSynchronize(s));
    *blended = (int64_t)host;
    return VR_OK;
}

int vr_count_flushes(const VrSaved* saved, int32_t H, int32_t W, void* stream, int64_t* flushes)
{
    g_err[0] = 0;
    if (!saved || !saved->image || !saved->binning || !flushes || H <= 0 || W <= 0)
        return fail(VR_ERR_INVALID_ARGUMENT, "count_flushes: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (int rt = check_ticket(saved->ticket, s)) return rt;
    Camera cam = {};
    cam.H = H; cam.W = W;
    cam.gx = (W + TILE - 1) / TILE;
    cam.gy = (H + TILE - 1) / TILE;
    const size_t N = (size_t)H * W, T = (size_t)cam.gx * cam.gy;
    const ImageLayout IL = image_layout(N);
    const size_t cap = saved->binning_capacity > 0 ? (size_t)saved->binning_capacity : (size_t)saved->num_rendered;
    const BinLayout BL = bin_layout(T, cap);
    unsigned long long* ctr = (unsigned long long*)((char*)saved->image + IL.counters) + 1;
    *flushes = 0;
    if (saved->num_rendered == 0) return VR_OK;
    int rc = launch_count_flushes(cam, (long)cap, (const uint32_t*)((const char*)saved->binning + BL.seg_off),
                                  (const unsigned long long*)((const char*)saved->binning + BL.segmask), ctr, s);
    if (rc) return rc;
    unsigned long long host = 0;
    VR_HIP(hipMemcpyAsync(&host, ctr, sizeof host, hipMemcpyDeviceToHost, s));
    VR_HIP(hipStreamSynchronize(s));
    *flushes = (int64_t)host;
    return VR_OK;
}

int vr_export_needed(const VrSaved* saved, int32_t H, int32_t W, uint32_t* out, void* stream)
{
    g_err[0] = 0;
    if (!saved || !saved->binning || !out || H <= 0 || W <= 0)
        return fail(VR_ERR_INVALID_ARGUMENT, "export_needed: bad arguments");
    if (int rt = check_ticket(saved->ticket, (hipStream_t)stream)) return rt;
    const size_t T = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    const BinLayout BL = bin_layout(T, saved->binning_capacity > 0 ? (size_t)saved->binning_capacity : (size_t)saved->num_rendered);
    VR_HIP(hipMemcpyAsync(out, (const char*)saved->binning + BL.seg_needed, T * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                          (hipStream_t)stream));
    return VR_OK;
}

int vr_debug_set_guard(uint32_t value, void* stream)
{
    g_err[0] = 0;
    int dev_id = 0;
    VR_HIP(hipGetDevice(&dev_id));
    if (dev_id < 0 || dev_id >= MAX_DEVICES) return fail(VR_ERR_NO_DEVICE, "device ordinal %d out of range", dev_id);
    Mailbox* mail = nullptr;
    int rc = get_mailbox(dev_id, (hipStream_t)stream, &mail);
    if (rc) return rc;
    if (mail->seq == 0u) return fail(VR_ERR_INVALID_ARGUMENT, "vr_debug_set_guard: no forward has run on this thread and device");
    // as if the thread's most recent forward had (value != 0) or had not (0) tripped: its ring slot, once its own post is in
    VR_HIP(hipStreamSynchronize((hipStream_t)stream));
    uint32_t* slot = mail->pinned + RING_AT + 2 * (mail->seq % RING_SLOTS);
    __atomic_store_n(&slot[1], value ? GUARD_RAISED : 0u, __ATOMIC_RELAXED);
    __atomic_store_n(&slot[0], mail->seq, __ATOMIC_RELEASE);
    if (mail->unchecked == 0u || (int32_t)(mail->unchecked - mail->seq) > 0) mail->unchecked = mail->seq;
    return VR_OK;
}

int vr_debug_raise_guard(int on)
{
    g_err[0] = 0;
    g_raise_guard = (on >= 2 && on <= 4) ? on : (on != 0 ? 1 : 0);
    return VR_OK;
}

int vr_debug_rebinned(void) { return g_rebinned; }

int vr_debug_export_binning(const VrSaved* saved, int32_t H, int32_t W, uint32_t* point_list, int32_t* ranges,
                            void* stream)
{
    g_err[0] = 0;
    if (!saved || !saved->binning || H <= 0 || W <= 0)
        return fail(VR_ERR_INVALID_ARGUMENT, "debug_export_binning: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (int rt = check_ticket(saved->ticket, s)) return rt;
    const size_t T = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    const BinLayout BL = bin_layout(T, saved->binning_capacity > 0 ? (size_t)saved->binning_capacity : (size_t)saved->num_rendered);
    if (ranges)
        VR_HIP(hipMemcpyAsync(ranges, (const char*)saved->binning + BL.ranges, T * 8, hipMemcpyDeviceToDevice, s));
    if (point_list && saved->num_rendered > 0)
        VR_HIP(hipMemcpyAsync(point_list, (const char*)saved->binning + BL.point_list,
                              (size_t)saved->num_rendered * 4, hipMemcpyDeviceToDevice, s));
    return VR_OK;
}

}  // extern "C"
