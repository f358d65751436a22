// vr_host.h -- host-side declarations shared by the translation units of libvegsrast.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vegs_rast_debug.h"   // (the stable header + the experimental / test-only entries the library also exports)
#include "vr_device.h"

namespace vr {

// thread-local error string behind vr_last_error()
void set_error(const char* fmt, ...);

#define VR_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            vr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -3;                                                                        \
        }                                                                                     \
    } while (0)

// after a kernel launch: always check the launch; in debug mode also synchronise
#define VR_KERNEL_CHECK(name, stream, debug)                                                  \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e == hipSuccess && (debug)) _e = hipStreamSynchronize(stream);                   \
        if (_e != hipSuccess) {                                                               \
            vr::set_error("kernel %s failed: %s", name, hipGetErrorString(_e));               \
            return -3;                                                                        \
        }                                                                                     \
    } while (0)

// stage timing (api.hip); no-ops unless vr_profile_level() enabled them
void prof_begin(int stage, hipStream_t s);
void prof_end(int stage, hipStream_t s);
struct ProfScope {
    int stage; hipStream_t s;
    ProfScope(int st, hipStream_t str) : stage(st), s(str) { prof_begin(stage, s); }
    ~ProfScope() { prof_end(stage, s); }
};

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- preprocess.hip
int launch_preprocess(const Camera& cam, int P, const float* means3D, const float* shs, const float* shs_rest,
                      const float* shs_tail, int tail_start, const float* colors_precomp,
                      const float* opacities, const float* scales, const float* rotations,
                      const float* cov3D_precomp, Splat* rec, int* radii, uint4* rect,
                      uint32_t* depth_key, uint32_t* tile_count, uint8_t* clampb, float* shd, hipStream_t s, bool debug);
int launch_mark_visible(const float* xyz, int P, const float* view, uint8_t* present, hipStream_t s);

// ---- binning.hip
// stage 1 (P-sized): compaction of visible Gaussians in id order + totals.  totals_dev[0]=V, [1]=R, [2],[3] = min / max
// depth key, [4] = *err_in (the look-back guard word, see launch_binning).
size_t binning_stage1_scratch_bytes(int P);
// host_mail: device address of the caller's pinned, coherent mailbox; receives the five words, then host_mail[8] = seq.
int launch_compact_reduce(int P, const uint32_t* tile_count, const uint32_t* depth_key, void* scratch, uint32_t* totals_dev,
                          uint32_t* err_clear, uint32_t* host_mail, uint32_t seq, hipStream_t s, bool debug);
// Second half of the compaction.  Side duties: the partial digit histograms of the depth keys (into `scratch`, for
// the depth sort), clearing zero_a (the tile ranges) and, when `status` = binning_stage2_status(stage-2 scratch) is
// given, the posted-sum status region of this view (else launch_binning clears it with a fill).
int launch_compact_apply(int P, const uint4* rect, const uint32_t* depth_key, void* scratch, const uint32_t* totals_dev,
                         int tile_bits, uint32_t* vis_key, uint32_t* vis_id, uint32_t* zero_a, long zero_na,
                         void* status, size_t status_bytes, hipStream_t s, bool debug);

// stage 2 (V- and R-sized): depth sort, emission, tile sort, ranges.
size_t binning_stage2_scratch_bytes(int V, long R, int ntiles);
void* binning_stage2_status(void* scratch);
size_t binning_stage2_status_bytes(int V, long R, int ntiles);   // bytes reserved for the status region
int binning_tile_bits(int ntiles);
// ranges_zeroed / status_zeroed: already cleared by launch_compact_apply.  stage1_scratch: the compaction's scratch
// (holds the depth keys' digit histograms).  err: device word that a kernel sets to 1 if a bounded wait for another
// workgroup's posted sum ran out (never observed; the alternative would be a hung queue).  guard_post / guard_seq: host-
// pinned {seq, guard} slot that the last binning kernel fills with this view's sequence number and the guard word as it
// stands after all waiting passes (NULL = none).  debug_raise_guard: test hook, raises the word as a timed-out wait would.
int launch_binning(const Camera& cam, int P, int V, long R, uint32_t key_min, int key_bits, uint32_t* vis_key,
                   uint32_t* vis_id, const uint4* rect, const void* stage1_scratch, void* scratch, uint32_t* point_list,
                   int2* ranges, bool ranges_zeroed, bool status_zeroed, uint32_t* err, uint32_t* guard_post,
                   uint32_t guard_seq, int debug_raise_guard, uint32_t n_huge, hipStream_t s, bool debug);

// VEGS_DEBUG_BINNING: post-mortem of one view's depth sort at the end of vr_forward (synchronises; a debugging aid)
int debug_verify_binning(int P, int V, long R, uint32_t key_min, int key_bits, const uint32_t* vis_key, const uint32_t* vis_id,
                         const uint32_t* depth_key, const uint32_t* tile_count, const void* stage1_scratch, const void* scratch,
                         const uint32_t* totals_dev, const uint32_t* pinned, const uint32_t* err, int ntiles, hipStream_t s);

// stable LSD radix sort of (key,val) u32 pairs on the low nbits of (key - kmin); ping-pongs between the two
// pairs, *where = 0/1 tells which pair holds the result
size_t sort_pairs_scratch_bytes(long n);
int launch_sort_pairs(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, long n, uint32_t kmin, int nbits,
                      void* scratch, hipStream_t s, bool debug, int* where);

// ---- knn.hip (simple-knn replacement)
size_t knn3_scratch_bytes(int N);
int launch_knn3(const float* points, int N, float* out, void* scratch, hipStream_t s, bool debug);

// losses.hip
size_t photometric_scratch_bytes(int C, int H, int W);
int launch_photometric_fwd(const float* image, const float* gt, int C, int H, int W, float* sums, float* dmaps,
                           void* scratch, hipStream_t s, bool debug);
int launch_photometric_bwd(const float* image, const float* gt, int C, int H, int W, const float* dmaps,
                           const float* g_l1, const float* g_ssim, float* dL_dimage, hipStream_t s, bool debug);
size_t normal_guidance_scratch_bytes(int H, int W);
int launch_normal_guidance_fwd(const float* cov_quat, const float* cov_scale, const float* normal, const float* R9, int H,
                               int W, float* loss, void* scratch, hipStream_t s, bool debug);
int launch_normal_guidance_bwd(const float* cov_quat, const float* cov_scale, const float* normal, const float* R9, int H,
                               int W, const float* g, float* dL_dquat, float* dL_dscale, hipStream_t s, bool debug);
size_t training_loss_scratch_bytes(int C, int H, int W);
int launch_training_loss_fwd(const float* image, const float* gt, int C, int H, int W, const float* cov_quat,
                             const float* cov_scale, const float* normal, const float* R9, float lambda_dssim,
                             float lambda_dnormal, bool guard, float* loss, float* aux, float* dmaps, void* scratch,
                             hipStream_t s, bool debug);
int launch_training_loss_bwd(const float* image, const float* gt, int C, int H, int W, const float* dmaps,
                             const float* cov_quat, const float* cov_scale, const float* normal, const float* R9,
                             float lambda_dssim, float lambda_dnormal, bool guard, const float* g, float* dL_dimage,
                             float* dL_dquat, float* dL_dscale, hipStream_t s, bool debug);

// ---- render_fwd.hip / render_bwd.hip (segmented compositing)
// upper bound on the number of 256-entry segments: sum_t ceil(n_t/256) <= R/256 + T
static inline size_t seg_capacity(long R, int ntiles) { return (size_t)(R / 256 + ntiles); }
size_t render_fwd_scratch_bytes(long R, int ntiles);
int launch_render_fwd(const Camera& cam, long R, const int2* ranges, const uint32_t* point_list, const Splat* rec,
                      uint32_t* seg_off, uint32_t* seg_needed, float* Tbuf, float* part, unsigned long long* segmask,
                      void* scratch, float* out_color,
                      float* out_depth, float* out_quat, float* out_scale, float* out_alpha, float* final_T,
                      uint32_t* n_contrib, float* dsum, uint32_t* needed_hint, hipStream_t s, bool debug,
                      int chain_patience = -1);      // (polls a chain-mode walker waits for a segment; < 0: the default.  0: test hook)
int launch_count_fragments(const uint32_t* n_contrib, long N, unsigned long long* out_dev, hipStream_t s);

// gacc: [P][16] floats = conic dA,dB,dC | opacity | attr[11] | pad ; gmean2D: [P][3] (x,y used)
size_t render_bwd_scratch_bytes(long R, int ntiles);
int launch_render_bwd(const Camera& cam, long R, const int2* ranges, const uint32_t* point_list, const Splat* rec,
                      const uint32_t* seg_off, const uint32_t* seg_needed, const float* Tbuf, const float* part,
                      const unsigned long long* segmask, void* scratch, const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                      const float* dL_ddepth, const float* dL_dquat, const float* dL_dscale,
                      const float* dL_dalpha, float* gacc, float* gmean2D, const float* dsum, void* det_scratch, int P,
                      bool zero_accumulators, hipStream_t s, bool debug);
// scratch of the deterministic backward mode (VR_FLAG_DETERMINISTIC): per-(entry, region) slots + the id sort
size_t render_bwd_det_bytes(long R, int P);
// B = blended (pixel, splat) pairs of a finished forward (one thread per pixel walks its list; bookkeeping only)
int launch_count_blended(const Camera& cam, const int2* ranges, const uint32_t* point_list, const Splat* rec,
                         const uint32_t* n_contrib, unsigned long long* out_dev, hipStream_t s);
int launch_count_flushes(const Camera& cam, long R, const uint32_t* seg_off, const unsigned long long* segmask,
                         unsigned long long* out_dev, hipStream_t s);

// ---- preprocess_bwd.hip
bool preprocess_bwd_writes_all_sh(int M, const float* shs, const float* dL_dshs);
int launch_preprocess_bwd(const Camera& cam, int P, const float* means3D, const float* shs, const float* shs_rest,
                          int tail_start, const float* colors_precomp, const float* opacities, const float* scales, const float* rotations,
                          const float* cov3D_precomp, const int* radii, const uint8_t* clampb, const float* shd, const float* gacc,
                          const float* gmean2D, float* dL_dmeans3D, float* dL_dshs, float* dL_dshs_rest, float* dL_dshs_tail,
                          float* dL_dcolors,
                          float* dL_dopacities, float* dL_dscales, float* dL_drots, float* dL_dcov3D,
                          float* dL_dcolors_sh, bool store_factor, hipStream_t s, bool debug);
int launch_sh_factor(int P, const int* radii, const uint8_t* clampb, const float* gacc, float* dL_dcolors_sh,
                     hipStream_t s, bool debug);

}  // namespace vr
