/* vegs_rast_debug.h -- the EXPERIMENTAL / TEST-ONLY part of libvegsrast.so's rasterizer interface.
 *
 * Nothing in here belongs to the drop-in boundary (include/vegs_rast.h is that): these entry
 * points exist for this repository's benchmarks and tests, may change or disappear without a bump of VR_ABI_VERSION, and no
 * product code should bind them.  They are exported by the same library (tests/test_capi_exports.py checks both headers).
 *   - tuning overrides of the forward's segment rounds (two bits of VrSettings.flags)
 *   - benchmark bookkeeping: vr_count_fragments / vr_count_blended / vr_count_flushes (block the host)
 *   - stage timers: vr_profile_level / vr_profile_collect
 *   - test hooks: vr_debug_export_binning, vr_debug_set_guard, vr_debug_raise_guard, vr_debug_rebinned
 * Environment switches of the same kind (read once per process): VEGS_DEBUG_BINNING=n -- every n-th vr_forward ends with a
 * host-side post-mortem of its depth sort (vegs_amd/csrc/binning.hip: debug_verify_binning; synchronises) and fails on the
 * first violated invariant; VEGS_LIB=<path> -- vegs_amd/_capi.py loads that library instead of the shipped one (reproducer
 * builds: python -m vegs_amd.build --variant <name>, profiles/tools/ab/build_at.sh). */
#ifndef VEGS_RAST_DEBUG_H
#define VEGS_RAST_DEBUG_H

#include "vegs_rast.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Further bits of VrSettings.flags: EXECUTION overrides that never change a result. */
typedef enum VrDebugFlags {
    /* The forward's segment rounds (ABI v6).  A tile's list is cut into 256-entry segments; most of them lie behind
     * the point where every pixel of the tile has stopped.  With ROUNDS the forward evaluates the first 6 segments of
     * every tile (2 on dense lists), then -- only for tiles that still have a live pixel -- the next 64 (8), then whatever is left; without,
     * every segment at once.  Results are identical bit for bit either way; the time is not: rounds win when lists
     * are long (discs three times larger than the street scene's: forward 0.49 -> 0.27 ms) and lose on short ones (the
     * deep tiles' later rounds run at low parallelism: +0.04 ms on the headline view).  Default: chosen per call from
     * the number of list entries (rounds from 8.5 segments per tile on).  A needed-segment hint implies rounds. */
    VR_FLAG_ROUNDS_OFF = 1u << 10,
    VR_FLAG_ROUNDS_ON = 1u << 11
} VrDebugFlags;

/* F = sum over pixels of n_contrib (fragments traversed by the forward blend loop) of the
 * forward whose state is `saved`; blocks the host. */
int vr_count_fragments(const VrSaved* saved, int32_t image_height, int32_t image_width, void* stream,
                       int64_t* fragments);

/* B = number of (pixel, splat) pairs actually BLENDED by the forward whose state is `saved` (alpha >= 1/255 and
 * in front of the pixel's stop), as opposed to the pairs merely traversed (vr_count_fragments); blocks the host.
 * Needs the forward's inputs only through `saved`; a slow one-thread-per-pixel walk, for benchmarks' bookkeeping. */
int vr_count_blended(const VrSaved* saved, int32_t image_height, int32_t image_width, void* stream,
                     int64_t* blended);

/* Flushes of the render backward for the forward whose state is `saved`: (list entry, 8x8 region) pairs of the NEEDED
 * segments whose relevance bit is set.  The backward issues 17 global fp32 atomics per flush (mean2D 2, conic 3, opacity 1,
 * colour 3, depth 1, quaternion 4, scale 3): atomics per view = 17 x this -- the L2-atomic figure benchmarks report next
 * to the HBM roofline; blocks the host. */
int vr_count_flushes(const VrSaved* saved, int32_t image_height, int32_t image_width, void* stream, int64_t* flushes);

/* ---- stage timing (HIP events recorded on `stream` around the kernels of each stage).
 * level 0 = off (default), 1 = only the k_seg_bwd kernel (the roofline kernel), one launch in four of each host
 * thread (an event pair costs a ~6 us bubble on the stream), 2 = every stage of every call.
 * vr_profile_collect synchronises the recorded events, ADDS the elapsed milliseconds and launch
 * counts of each stage to ms[VR_STAGE_COUNT] / count[VR_STAGE_COUNT], and clears the record. */
typedef enum VrStage {
    VR_STAGE_PREPROCESS = 0,
    VR_STAGE_COMPACT = 1,
    VR_STAGE_DEPTH_SORT = 2,
    VR_STAGE_EMIT = 3,
    VR_STAGE_TILE_SORT = 4,
    VR_STAGE_RANGES = 5,
    VR_STAGE_RENDER_FWD = 6,
    VR_STAGE_BWD_ZERO = 7,
    VR_STAGE_RENDER_BWD = 8,
    VR_STAGE_PREPROCESS_BWD = 9,
    VR_STAGE_K_SEG_BWD = 10, /* the single kernel k_seg_bwd inside RENDER_BWD (the dominant kernel) */
    VR_STAGE_COUNT = 11
} VrStage;
int vr_profile_level(int level);
int vr_profile_collect(double* ms, int64_t* count);

/* Introspection for tests: copies of the sorted tile lists kept in `saved` (device -> device).
 * point_list [R] uint32, ranges [T][2] int32 (start,end).  Either pointer may be NULL. */
int vr_debug_export_binning(const VrSaved* saved, int32_t image_height, int32_t image_width,
                            uint32_t* point_list, int32_t* ranges, void* stream);

/* Test hooks for the binning guard: the single-launch radix passes bound every wait for another workgroup's posted sum
 * (2 s of wall clock) and raise a device-side guard word if one runs out.  The view whose wait ran out is the one that
 * fails: its vr_backward (and every other call taking its VrSaved) returns VR_ERR_HIP through VrSaved.ticket; a forward
 * that never gets a backward (eval under no_grad) is reported by the NEXT vr_forward of the host thread instead.
 * vr_debug_set_guard sets the word (value != 0) or clears it (0) by hand, as if an earlier view had raised it;
 * vr_debug_raise_guard(1) makes the NEXT vr_forward of the calling thread raise it in the middle of its own binning, as
 * a timed-out wait would -- after lists that are in fact valid; vr_debug_raise_guard(2) makes that forward LOSE the first
 * workgroup of its depth sort (it never posts its counts): the waits of its successors run out for real (~2 s) and the
 * lists behind them are built from short prefixes -- what VR_FLAG_VERIFY_BINNING has to recover from.
 * vr_debug_raise_guard(3) has nothing to do with the guard word: the walker workgroups of that forward's render stage (they
 * follow the deep tiles' segment chains inside the alpha launch) give up at their first empty poll, as they would after
 * their bounded wait on a GPU that does not schedule the producers -- the tiles must then be finished by the kernel behind
 * the launch, with the same result. */
int vr_debug_set_guard(uint32_t value, void* stream);
int vr_debug_raise_guard(int on);
/* views of the calling thread that VR_FLAG_VERIFY_BINNING binned a second time (tests) */
int vr_debug_rebinned(void);

#ifdef __cplusplus
}
#endif
#endif /* VEGS_RAST_DEBUG_H */
