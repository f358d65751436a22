/* vegs_xgmi.h -- C ABI of the DIRECT gradient exchange of the view-sharded job ("direct
 * reduce-scatter + all-gather over all links"; section 5, last row).  Same library and conventions as vegs_rast.h.
 *
 * The reference has no distributed code (it trains one view per iteration on one GPU, train.py:126-150); what this
 * replaces is the exchange step of the build's own multi-GPU extension -- torch.distributed.all_reduce /
 * all_gather_into_tensor on RCCL, whose ring algorithms drive ONE xGMI link pair per step.  On an 8 x MI355X node every
 * GPU has a direct link to each of its 7 peers; here every rank moves its 1/N shards to (and from) ALL peers concurrently
 * through `hipIpc` mappings of their windows, with flag words in the peers' memory as the only synchronisation:
 *
 *   all-reduce (two-shot, push):   1  rank r writes shard s of its gradients into peer s's  recv[r]        (N-1 links out)
 *                                  2  rank r adds recv[0..N-1] in rank order (a FIXED order: deterministic), scales,
 *                                     and writes the reduced shard r into every peer's  result[]            (N-1 links out)
 *                                  3  wait until every peer's shard has arrived
 *   all-gather (one-shot, push):   rank r writes its block into slot r of every peer's  gather[parity][]
 *
 * `result[]` and `gather[][]` ARE the tensors the optimizer reads afterwards (vr_xgmi_window + the offsets below): the
 * reduced gradients are never copied out of the exchange buffer ("fused with gradient un-bucketing").
 * One process per GPU; the same code runs with several processes on ONE GPU (the IPC mappings then point into the same
 * device), which is how tests/test_gpu_xgmi.py checks it on a single-GPU box.
 */
#ifndef VEGS_XGMI_H
#define VEGS_XGMI_H

#include "vegs_rast.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VR_XGMI_MAX_RANKS 16
#define VR_XGMI_HANDLE_BYTES 72      /* sizeof(hipIpcMemHandle_t) + the window's 8-byte identity, verified by attach() */
#define VR_XGMI_MAX_SEGMENTS 8

typedef struct VrXgmi VrXgmi;        /* opaque: one per process (rank) */

/* Sizes of the three regions of a window, in floats, for given capacities (the same on every rank):
 *   reduce_floats   = upper bound of the summed lengths of the tensors of one all-reduce (e.g. 11 P)
 *   gather_floats   = upper bound of ONE rank's all-gather block (e.g. 3 rows + 3) */
typedef struct VrXgmiLayout {
    int64_t recv_offset;      /* float offsets from the window base */
    int64_t result_offset;
    int64_t gather_offset[2]; /* two parities: a rank may already push iteration k+1 while a peer still reads k */
    int64_t gather_slot;      /* floats per rank slot inside a gather buffer */
    int64_t total_floats;
} VrXgmiLayout;

/* Allocates this rank's window (device memory of the CURRENT device + flag words) for the given capacities. */
int vr_xgmi_create(int32_t rank, int32_t world, int64_t reduce_floats, int64_t gather_floats, VrXgmi** out);
/* Tear-down is two steps so that the host side can put a barrier between them: detach() unmaps the peers' windows,
 * destroy() (which detaches if needed) frees this rank's -- no rank frees a window a peer still has mapped. */
int vr_xgmi_detach(VrXgmi* x);
int vr_xgmi_destroy(VrXgmi* x);
int vr_xgmi_layout(const VrXgmi* x, VrXgmiLayout* out);
void* vr_xgmi_window(VrXgmi* x);                       /* device pointer of this rank's window */

/* This rank's IPC handle (VR_XGMI_HANDLE_BYTES bytes); the host side all-gathers the handles of all ranks (any
 * transport) and passes the world x VR_XGMI_HANDLE_BYTES array to attach(), which maps the peers' windows. */
int vr_xgmi_handle(VrXgmi* x, void* handle_out);
int vr_xgmi_attach(VrXgmi* x, const void* handles);

/* One tensor of a collective: n floats at `src` (device; the gradient as the backward wrote it). */
typedef struct VrXgmiSegment {
    const float* src;
    int64_t n;
} VrXgmiSegment;

/* Mean (scale * sum over ranks, summed in rank order) of every segment, delivered on every rank at
 *     window + result_offset + result_floats_offset[i]      (result_floats_offset is an OUTPUT: segment i's place)
 * All ranks must call with the same segment lengths.  Kernels go onto `stream`; the call returns once they are
 * enqueued.
 * FAILURE CONTRACT (ABI v9).  Every device-side wait for a peer is bounded (10 s of wall clock by default;
 * vr_xgmi_set_wait_bound).  A wait that runs out raises the window's error word -- in the window and in a pinned host
 * mirror -- instead of hanging the queue, and the error is STICKY:
 *   * a reduce kernel whose wait ran out posts no epoch, so the peers waiting for its shard run into their own bound: every
 *     rank of the job learns, within two bounds.  Its workgroups decide one by one: those whose wait ran out add and write
 *     nothing, one whose poll succeeded just before may already have written its part -- result[] of a failed exchange may
 *     be PARTIAL (see the last paragraph);
 *   * vr_xgmi_allreduce / _allgather_begin / _allgather_wait read the host mirror FIRST (one host load, no
 *     synchronisation) and return VR_ERR_HIP once it is set: the call after the failed exchange fails, on every rank;
 *     vr_xgmi_failed() is the same test as a predicate, vr_xgmi_check() synchronises `stream` first.
 * What the failed exchange itself left in result[] / gather[] is the PREVIOUS iteration's content or a partial one: a
 * caller that must not consume it calls vr_xgmi_check() before it does; a caller that can afford to lose one step
 * (the trainer: the exception ends the run, the last checkpoint is the recovery point) relies on the next call.  A window
 * that has failed stays failed: destroy it and build a new one, collectively. */
int vr_xgmi_allreduce(VrXgmi* x, const VrXgmiSegment* segs, int32_t count, float scale, int64_t* result_floats_offset,
                      void* stream);

/* All-gather: the segments of this rank, back to back, land in slot `rank` of every rank's gather[parity] buffer; rank
 * j's block is then at  window + gather_offset[parity] + j * gather_slot  (+ slot_floats_offset[i] for segment i).
 * vr_xgmi_allgather_begin only PUSHES (call it as early as the data exists, on any stream); ..._wait enqueues the wait
 * for all peers' blocks on `stream` (the consumer's). */
int vr_xgmi_allgather_begin(VrXgmi* x, const VrXgmiSegment* segs, int32_t count, int32_t parity,
                            int64_t* slot_floats_offset, void* stream);
int vr_xgmi_allgather_wait(VrXgmi* x, int32_t parity, void* stream);

/* Synchronises `stream` and reports a timed-out wait (VR_ERR_HIP): for a caller that must not consume a failed result. */
int vr_xgmi_check(VrXgmi* x, void* stream);
/* 1 if a wait on this window has timed out (as far as the host can see without synchronising), else 0. */
int vr_xgmi_failed(const VrXgmi* x);
/* Bound of every device-side wait of the calls that follow, in seconds of wall clock (default 10; tests use less). */
int vr_xgmi_set_wait_bound(VrXgmi* x, double seconds);

#ifdef __cplusplus
}
#endif
#endif
