// preprocess_bwd.hip -- per-Gaussian backward: conic -> cov2D -> (cov3D, view-space point) ->
// (scale, rotation, mean); NDC gradient -> mean through the perspective divide; SH colour backward;
// plus the direct gradients of the fork's blended channels (depth -> mean, rotation row, scale row).
//
// Replaces the native preprocess-backward stage reached from loss.backward() (reference
// train.py:196); semantics SURVEY.md A.6.  Gradients must be right for means3D, scales AND
// rotations because render_all feeds box-transformed tensors (reference
// gaussian_renderer/__init__.py:123-153) whose gradients flow on into model/boxmodel.py:30-42.
// One lane per Gaussian; every row of the small outputs is written (zeros when culled), dL_dshs is
// pre-zeroed by the caller and only visible rows are written.
#include "vr_host.h"

namespace vr {

constexpr int SH_ROW_MAX = 48;  // floats per Gaussian staged through LDS (M <= 16 coefficients x 3)
constexpr int SH_LDS_STRIDE = 52;  // padded LDS row stride (13 x 16 bytes: conflict-free per-lane float4 rows, see preprocess.hip)

// One Gaussian's SH backward from the forward's cached d colour / d direction (D, see sh_ddir9):
// gsh[k][c] = basis[k] * dL/drgb_c (zero where the colour was clamped) and the gradient through the view
// direction into the mean.  The SH coefficients themselves are not needed any more.
__device__ __forceinline__ void sh_backward_row(const Camera& cam, float px3, float py3, float pz3, uint32_t clampbits,
                                                float g0, float g1, float g2, const float* D, float* gsh, float* dmean)
{
    const float d0 = px3 - cam.campos[0], d1 = py3 - cam.campos[1], d2v = pz3 - cam.campos[2];
    const float len = sqrtf(d0 * d0 + d1 * d1 + d2v * d2v);
    const float il = 1.0f / len;
    const float dir[3] = {d0 * il, d1 * il, d2v * il};
    float bas[16];
    sh_basis(cam.deg, dir[0], dir[1], dir[2], bas);
    const int K = (cam.deg + 1) * (cam.deg + 1);
    const float gc0 = (clampbits & 1u) ? 0.f : g0;
    const float gc1 = (clampbits & 2u) ? 0.f : g1;
    const float gc2 = (clampbits & 4u) ? 0.f : g2;
#pragma unroll
    for (int k = 0; k < 16; ++k) {       // constant trip count + guards: `gsh` may be a register array
        if (k < K) {
            gsh[3 * k + 0] = bas[k] * gc0;
            gsh[3 * k + 1] = bas[k] * gc1;
            gsh[3 * k + 2] = bas[k] * gc2;
        } else if (k < cam.M) {
            gsh[3 * k] = 0.f; gsh[3 * k + 1] = 0.f; gsh[3 * k + 2] = 0.f;
        }
    }
    float ddir[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) ddir[a] = fmaf(gc2, D[6 + a], fmaf(gc1, D[3 + a], gc0 * D[a]));
    const float dot = dir[0] * ddir[0] + dir[1] * ddir[1] + dir[2] * ddir[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) dmean[k] += (ddir[k] - dir[k] * dot) * il;
}

// Factored variant (VrInGrads.dL_dcolors_sh): only the clamp-masked dL/d(colour) leaves the kernel -- dL/dshs is the
// rank-1 product basis(dir) x that vector and is rebuilt (or consumed by the optimizer) elsewhere -- plus, as always,
// the gradient through the view direction into the mean.
__device__ __forceinline__ void sh_backward_factor(const Camera& cam, float px3, float py3, float pz3, uint32_t clampbits,
                                                   float g0, float g1, float g2, const float* D, float* gc, float* dmean)
{
    const float d0 = px3 - cam.campos[0], d1 = py3 - cam.campos[1], d2v = pz3 - cam.campos[2];
    const float len = sqrtf(d0 * d0 + d1 * d1 + d2v * d2v);
    const float il = 1.0f / len;
    const float dir[3] = {d0 * il, d1 * il, d2v * il};
    gc[0] = (clampbits & 1u) ? 0.f : g0;
    gc[1] = (clampbits & 2u) ? 0.f : g1;
    gc[2] = (clampbits & 4u) ? 0.f : g2;
    float ddir[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) ddir[a] = fmaf(gc[2], D[6 + a], fmaf(gc[1], D[3 + a], gc[0] * D[a]));
    const float dot = dir[0] * ddir[0] + dir[1] * ddir[1] + dir[2] * ddir[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) dmean[k] += (ddir[k] - dir[k] * dot) * il;
}

// dst[e] += lds[e] for e < n (the wave's contiguous block of gradient rows): VR_FLAG_ACCUMULATE_GRADS
template <int MAXV>
__device__ __forceinline__ void wave_add_from_lds(float* __restrict__ dst, const float* __restrict__ lds, int n, int lane)
{
    if ((reinterpret_cast<size_t>(dst) & 15) == 0) {
        const int nv = n >> 2;
        float4* d4 = reinterpret_cast<float4*>(dst);
        const float4* s4 = reinterpret_cast<const float4*>(lds);
        float4 old[MAXV];
#pragma unroll
        for (int j = 0; j < MAXV; ++j)
            if (lane + 64 * j < nv) old[j] = d4[lane + 64 * j];
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            if (lane + 64 * j < nv) {
                const float4 v = s4[lane + 64 * j];
                d4[lane + 64 * j] = make_float4(old[j].x + v.x, old[j].y + v.y, old[j].z + v.z, old[j].w + v.w);
            }
        }
        if (lane < (n & 3)) dst[(nv << 2) + lane] += lds[(nv << 2) + lane];
    } else {
        for (int e = lane; e < n; e += 64) dst[e] += lds[e];
    }
}

// RAW: VR_FLAG_RAW_PARAMS, ACC: VR_FLAG_ACCUMULATE_GRADS -- compile-time: the default instantiation is the kernel as it was.
// ACC: every gradient array except dL_dmeans2D (written by the render backward) and the SH factor RECEIVES this view's
// gradient, grad[i] += g_view[i] in fp32, for rows with radii > 0 only; culled rows are neither read nor written (their
// contribution is an exact zero: in a partly visible wave of the SH paths they get "+ 0" with their neighbours).  What a
// multi-view step otherwise pays per view -- this kernel's dense write of (56 + 12 K) bytes per Gaussian plus autograd's
// read-read-write accumulation of the same arrays -- becomes one read-modify-write of the visible rows.
template <bool RAW, bool ACC>
__global__ void __launch_bounds__(256)
k_preprocess_bwd(Camera cam, int P, int sh_staged, const float* __restrict__ means3D, const float* __restrict__ shs,
                 const float* __restrict__ shs_rest, float* __restrict__ dL_dshs_rest, float* __restrict__ dL_dshs_tail,
                 int tail_start, const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                 const float* __restrict__ scales,
                 const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
                 const int* __restrict__ radii, const uint8_t* __restrict__ clampb, const float* __restrict__ shd,
                 const float* __restrict__ gacc,
                 const float* __restrict__ gmean2D, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dshs,
                 float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacities, float* __restrict__ dL_dscales,
                 float* __restrict__ dL_drots, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dcolors_sh,
                 int store_factor)
{
    __shared__ __attribute__((aligned(16))) float sh_lds[4][64 * SH_LDS_STRIDE];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool in_range = i < P;
    const bool vis = in_range && radii[i] > 0;
    float dmean[3] = {0.f, 0.f, 0.f};
    float dsc[3] = {0.f, 0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f}, dcol[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dop = 0.f;

    // ---- SH backward, wave-cooperative: the wave's 64 SH rows are contiguous in memory, so they are
    // staged through LDS with coalesced dwordx4 loads, every lane turns its row into its dL/dsh row in
    // place, and the 64 rows are stored back coalesced (zeros for culled Gaussians: the kernel writes
    // every row of dL_dshs itself, no separate memset).
    if (shs && dL_dcolors_sh) {
        float gc[3] = {0.f, 0.f, 0.f};
        if (vis) {
            const float4 a1 = reinterpret_cast<const float4*>(gacc + (size_t)i * 16)[1];
            const float px3 = means3D[3 * (size_t)i], py3 = means3D[3 * (size_t)i + 1], pz3 = means3D[3 * (size_t)i + 2];
            float D[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) D[q] = shd[9 * (size_t)i + q];
            sh_backward_factor(cam, px3, py3, pz3, (uint32_t)clampb[i], a1.x, a1.y, a1.z, D, gc, dmean);
        }
        if (in_range && store_factor) {   // (0: already written by k_sh_factor -- the split backward, vr_backward_render)
#pragma unroll
            for (int c = 0; c < 3; ++c) dL_dcolors_sh[3 * (size_t)i + c] = gc[c];
        }
    } else if (shs && dL_dshs_tail && (long)(blockIdx.x * blockDim.x + w * 64) < (long)tail_start &&
               (long)(blockIdx.x * blockDim.x + w * 64) + 64 > (long)tail_start) {
        // SH TAIL (VrInputs.shs_tail), the one wave across the boundary: every lane writes its own row (zeros when
        // culled) to wherever it lives -- the static model's dL_dshs (/ dL_dshs_rest) or the tail's gradient
        if (in_range) {
            const int row = cam.M * 3;
            float srow[SH_ROW_MAX];
#pragma unroll
            for (int k = 0; k < SH_ROW_MAX; ++k) srow[k] = 0.0f;
            if (vis) {
                const float4 a1 = reinterpret_cast<const float4*>(gacc + (size_t)i * 16)[1];
                const float px3 = means3D[3 * (size_t)i], py3 = means3D[3 * (size_t)i + 1], pz3 = means3D[3 * (size_t)i + 2];
                float D[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) D[q] = shd[9 * (size_t)i + q];
                sh_backward_row(cam, px3, py3, pz3, (uint32_t)clampb[i], a1.x, a1.y, a1.z, D, srow, dmean);
            }
            const bool mine_tail = i >= tail_start;
            float* head = mine_tail ? dL_dshs_tail + (size_t)(i - tail_start) * row
                                    : (shs_rest ? dL_dshs + 3 * (size_t)i : dL_dshs + (size_t)i * row);
            float* rest = (!mine_tail && shs_rest) ? dL_dshs_rest + (size_t)i * (row - 3) : head + 3;
            if (ACC) {
                if (vis) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) head[c] += srow[c];
#pragma unroll
                    for (int k = 0; k < SH_ROW_MAX - 3; ++k)
                        if (k < row - 3) rest[k] += srow[3 + k];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) head[c] = srow[c];
#pragma unroll
                for (int k = 0; k < SH_ROW_MAX - 3; ++k)
                    if (k < row - 3) rest[k] = srow[3 + k];
            }
        }
    } else if (shs && shs_rest && !(dL_dshs_tail && (long)(blockIdx.x * blockDim.x + w * 64) >= (long)tail_start)) {
        // split storage: dL_dshs = [P,1,3] DC rows (written by their own lane: 12 contiguous bytes per lane),
        // dL_dshs_rest = [P,M-1,3]: every lane builds its gradient row in LDS (row stride = the memory's own, 45
        // floats at M = 16: odd, conflict-free) and the wave's 64 rows, one contiguous block, are copied out
        // linearly.  Every row of both gradients is written.
        const int rowr = cam.M * 3 - 3;
        const size_t wave_first = (size_t)(blockIdx.x * blockDim.x + w * 64);
        const int rows_here = max(0, min(64, (dL_dshs_tail ? tail_start : P) - (int)wave_first));
        const bool any = __ballot(vis) != 0ull;
        float dc[3] = {0.f, 0.f, 0.f};
        if (any) {
            float* my = sh_lds[w] + lane * rowr;
            if (vis) {
                const float4 a1 = reinterpret_cast<const float4*>(gacc + (size_t)i * 16)[1];
                const float px3 = means3D[3 * (size_t)i], py3 = means3D[3 * (size_t)i + 1], pz3 = means3D[3 * (size_t)i + 2];
                float D[9], srow[SH_ROW_MAX];
#pragma unroll
                for (int q = 0; q < 9; ++q) D[q] = shd[9 * (size_t)i + q];
                sh_backward_row(cam, px3, py3, pz3, (uint32_t)clampb[i], a1.x, a1.y, a1.z, D, srow, dmean);
#pragma unroll
                for (int c = 0; c < 3; ++c) dc[c] = srow[c];
#pragma unroll
                for (int k = 0; k < SH_ROW_MAX - 3; ++k)
                    if (k < rowr) my[k] = srow[3 + k];
            } else if (lane < rows_here) {
                for (int k = 0; k < rowr; ++k) my[k] = 0.0f;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (ACC) {
            if (vis) {
#pragma unroll
                for (int c = 0; c < 3; ++c) dL_dshs[3 * (size_t)i + c] += dc[c];
            }
            if (any) wave_add_from_lds<SH_ROW_MAX / 4>(dL_dshs_rest + wave_first * rowr, sh_lds[w], rows_here * rowr, lane);
        } else {
            if (in_range) {
#pragma unroll
                for (int c = 0; c < 3; ++c) dL_dshs[3 * (size_t)i + c] = dc[c];
            }
            wave_copy_from_lds<SH_ROW_MAX / 4>(dL_dshs_rest + wave_first * rowr, sh_lds[w], rows_here * rowr, lane, !any);
        }
    } else if (shs && (sh_staged || dL_dshs_tail)) {
        // whole rows: the only SH tensor, the static part in front of a tail, or the tail itself (indexed by Gaussian id)
        const int row = cam.M * 3;
        const size_t wave_first = (size_t)(blockIdx.x * blockDim.x + w * 64);
        const bool in_tail = dL_dshs_tail && (long)wave_first >= (long)tail_start;
        float* const whole_out = in_tail ? dL_dshs_tail - (size_t)tail_start * row : dL_dshs;
        const int rows_here = max(0, min(64, (in_tail || !dL_dshs_tail ? P : tail_start) - (int)wave_first));
        const int nvec = rows_here * row / 4;
        const int row4 = row >> 2;
        float4* lds4 = reinterpret_cast<float4*>(sh_lds[w]);
        const bool any = __ballot(vis) != 0ull;
        if (any) {
            float4* my4 = lds4 + lane * (SH_LDS_STRIDE / 4);
            if (vis) {
                const float4 a1 = reinterpret_cast<const float4*>(gacc + (size_t)i * 16)[1];
                const float px3 = means3D[3 * (size_t)i], py3 = means3D[3 * (size_t)i + 1], pz3 = means3D[3 * (size_t)i + 2];
                float D[9], srow[SH_ROW_MAX];    // the lane's gradient row, written to LDS as whole float4s
#pragma unroll
                for (int q = 0; q < 9; ++q) D[q] = shd[9 * (size_t)i + q];
                sh_backward_row(cam, px3, py3, pz3, (uint32_t)clampb[i], a1.x, a1.y, a1.z, D, srow, dmean);
#pragma unroll
                for (int j = 0; j < SH_ROW_MAX / 4; ++j)
                    if (j < row4) my4[j] = make_float4(srow[4 * j], srow[4 * j + 1], srow[4 * j + 2], srow[4 * j + 3]);
            } else if (lane < rows_here) {
                for (int j = 0; j < row4; ++j) my4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __builtin_amdgcn_wave_barrier();
        }
        float4* dst4 = reinterpret_cast<float4*>(whole_out + wave_first * row);
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ACC) {
            if (any) {      // (a wave without a visible Gaussian adds nothing: its rows are not touched)
                float4 old[SH_ROW_MAX / 4];
#pragma unroll
                for (int j = 0; j < SH_ROW_MAX / 4; ++j)
                    if (lane + 64 * j < nvec) old[j] = dst4[lane + 64 * j];
#pragma unroll
                for (int j = 0; j < SH_ROW_MAX / 4; ++j) {
                    const int v = lane + 64 * j;
                    if (v < nvec) {
                        const int r = row4 == 12 ? v / 12 : v / row4;
                        const float4 val = lds4[r * (SH_LDS_STRIDE / 4) + (v - r * row4)];
                        dst4[v] = make_float4(old[j].x + val.x, old[j].y + val.y, old[j].z + val.z, old[j].w + val.w);
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < SH_ROW_MAX / 4; ++j) {
                const int v = lane + 64 * j;
                if (v < nvec) {
                    const int r = row4 == 12 ? v / 12 : v / row4;
                    float4 val = zero4;
                    if (any) val = lds4[r * (SH_LDS_STRIDE / 4) + (v - r * row4)];
                    nt_store4(val, &dst4[v]);
                }
            }
        }
    } else if (shs && vis) {
        // unusual coefficient count / alignment: direct row access, dL_dshs pre-zeroed by the caller
        const float4 a1 = reinterpret_cast<const float4*>(gacc + (size_t)i * 16)[1];
        const float px3 = means3D[3 * (size_t)i], py3 = means3D[3 * (size_t)i + 1], pz3 = means3D[3 * (size_t)i + 2];
        float D[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) D[q] = shd[9 * (size_t)i + q];
        if (ACC) {
            float srow[SH_ROW_MAX];
            sh_backward_row(cam, px3, py3, pz3, (uint32_t)clampb[i], a1.x, a1.y, a1.z, D, srow, dmean);
            float* out = dL_dshs + (size_t)i * cam.M * 3;
#pragma unroll
            for (int k = 0; k < SH_ROW_MAX; ++k)
                if (k < cam.M * 3) out[k] += srow[k];
        } else {
            sh_backward_row(cam, px3, py3, pz3, (uint32_t)clampb[i], a1.x, a1.y, a1.z, D, dL_dshs + (size_t)i * cam.M * 3, dmean);
        }
    }

    if (vis) {
        const float* V = cam.view;
        const float* Pm = cam.proj;
        const float4* ga4 = reinterpret_cast<const float4*>(gacc + (size_t)i * 16);
        const float4 a0 = ga4[0], a1 = ga4[1], a2 = ga4[2], a3 = ga4[3];
        const float gA = a0.x, gB = a0.y, gC = a0.z;
        dop = a0.w;
        const float ga[NCH] = {a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z};
        const float px3 = means3D[3 * (size_t)i], py3 = means3D[3 * (size_t)i + 1], pz3 = means3D[3 * (size_t)i + 2];

        // ---- recompute the forward intermediates (identical expressions as preprocess)
        float t0, t1, t2;
        xform43(V, px3, py3, pz3, t0, t1, t2);
        float c6[6], q[4] = {0.f, 0.f, 0.f, 0.f}, sc[3] = {0.f, 0.f, 0.f}, qraw[4] = {0.f, 0.f, 0.f, 0.f};
        if (cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * (size_t)i + k];
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) sc[k] = scales[3 * (size_t)i + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = rotations[4 * (size_t)i + k];
            if (RAW && i < tail_start) {          // as in the forward; the raw quaternion is kept for the chain below
#pragma unroll
                for (int k = 0; k < 4; ++k) qraw[k] = q[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) sc[k] = expf(sc[k]);
                act_normalize(qraw, q);
            }
            cov3d_from_scale_rot(sc, cam.mod, q, c6);
        }
        Cov2D cv;
        cov2d(cam, V, t0, t1, t2, c6, cv);
        const float a = cv.a, b = cv.b, c = cv.c;
        const float det = a * c - b * b;
        float da = 0.f, db = 0.f, dc = 0.f;
        if (det != 0.0f) {
            const float d2 = 1.0f / (det * det);
            da = d2 * (-c * c * gA + b * c * gB - b * b * gC);
            db = d2 * (2.0f * b * c * gA - (a * c + b * b) * gB + 2.0f * a * b * gC);
            dc = d2 * (-b * b * gA + a * b * gB - a * a * gC);
        }
        const float* m0 = cv.m0; const float* m1 = cv.m1;
        float gS[6];
        gS[0] = da * m0[0] * m0[0] + db * m0[0] * m1[0] + dc * m1[0] * m1[0];
        gS[3] = da * m0[1] * m0[1] + db * m0[1] * m1[1] + dc * m1[1] * m1[1];
        gS[5] = da * m0[2] * m0[2] + db * m0[2] * m1[2] + dc * m1[2] * m1[2];
        gS[1] = 2.f * da * m0[0] * m0[1] + db * (m0[0] * m1[1] + m0[1] * m1[0]) + 2.f * dc * m1[0] * m1[1];
        gS[2] = 2.f * da * m0[0] * m0[2] + db * (m0[0] * m1[2] + m0[2] * m1[0]) + 2.f * dc * m1[0] * m1[2];
        gS[4] = 2.f * da * m0[1] * m0[2] + db * (m0[1] * m1[2] + m0[2] * m1[1]) + 2.f * dc * m1[1] * m1[2];
        const float u[3] = {c6[0] * m0[0] + c6[1] * m0[1] + c6[2] * m0[2], c6[1] * m0[0] + c6[3] * m0[1] + c6[4] * m0[2],
                            c6[2] * m0[0] + c6[4] * m0[1] + c6[5] * m0[2]};
        const float wv[3] = {c6[0] * m1[0] + c6[1] * m1[1] + c6[2] * m1[2], c6[1] * m1[0] + c6[3] * m1[1] + c6[4] * m1[2],
                             c6[2] * m1[0] + c6[4] * m1[1] + c6[5] * m1[2]};
        float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float dm0 = 2.f * da * u[k] + db * wv[k];
            const float dm1 = db * u[k] + 2.f * dc * wv[k];
            dJ00 += dm0 * V[4 * k + 0];
            dJ02 += dm0 * V[4 * k + 2];
            dJ11 += dm1 * V[4 * k + 1];
            dJ12 += dm1 * V[4 * k + 2];
        }
        const float tzi = 1.0f / cv.tz, tzi2 = tzi * tzi, tzi3 = tzi2 * tzi;
        float dt[3];
        dt[0] = cv.clampx ? 0.f : -cam.fx * tzi2 * dJ02;
        dt[1] = cv.clampy ? 0.f : -cam.fy * tzi2 * dJ12;
        dt[2] = -cam.fx * tzi2 * dJ00 - cam.fy * tzi2 * dJ11 + 2.f * cam.fx * cv.tx * tzi3 * dJ02 +
                2.f * cam.fy * cv.ty * tzi3 * dJ12;
        dt[2] += ga[3];  // blended depth channel: depth_i = t.z
#pragma unroll
        for (int k = 0; k < 3; ++k) dmean[k] += V[4 * k + 0] * dt[0] + V[4 * k + 1] * dt[1] + V[4 * k + 2] * dt[2];

        // ---- NDC gradient -> mean
        float h0, h1, h2;
        xform43(Pm, px3, py3, pz3, h0, h1, h2);
        const float hw = xform_w(Pm, px3, py3, pz3);
        const float mw = 1.0f / (hw + 0.0000001f);
        const float gx2 = gmean2D[3 * (size_t)i], gy2 = gmean2D[3 * (size_t)i + 1];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float mul1 = Pm[4 * k + 0] * mw - Pm[4 * k + 3] * h0 * mw * mw;
            const float mul2 = Pm[4 * k + 1] * mw - Pm[4 * k + 3] * h1 * mw * mw;
            dmean[k] += mul1 * gx2 + mul2 * gy2;
        }

        // ---- colour (the SH part was handled wave-cooperatively above)
        if (colors_precomp) {
            dcol[0] = ga[0]; dcol[1] = ga[1]; dcol[2] = ga[2];
        }

        // ---- cov3D -> scale / rotation, plus the directly blended rows
        if (cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; ++k) dcov[k] = gS[k];
        } else {
            float R[9];
            quat_to_R(q[0], q[1], q[2], q[3], R);
            const float Gf[9] = {gS[0], 0.5f * gS[1], 0.5f * gS[2], 0.5f * gS[1], gS[3], 0.5f * gS[4],
                                 0.5f * gS[2], 0.5f * gS[4], gS[5]};
            const float sp[3] = {cam.mod * sc[0], cam.mod * sc[1], cam.mod * sc[2]};
            float D[9];
            const float sm = (cam.flags & FLAG_SCALE_MODIFIED) ? cam.mod : 1.0f;   // blended row = sm * scales
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float accs = 0.f;
#pragma unroll
                for (int ii = 0; ii < 3; ++ii) {
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc += Gf[3 * ii + j] * (R[3 * j + k] * sp[k]);
                    const float dLm = 2.f * acc;
                    accs += dLm * R[3 * ii + k];
                    D[3 * ii + k] = dLm * sp[k];
                }
                dsc[k] = cam.mod * accs + sm * ga[8 + k];
            }
            const float r = q[0], x = q[1], y = q[2], z = q[3];
            drot[0] = 2.f * (z * (D[3] - D[1]) + y * (D[2] - D[6]) + x * (D[7] - D[5])) + ga[4];
            drot[1] = 2.f * (y * (D[1] + D[3]) + z * (D[2] + D[6]) + r * (D[7] - D[5])) - 4.f * x * (D[4] + D[8]) + ga[5];
            drot[2] = 2.f * (x * (D[1] + D[3]) + r * (D[2] - D[6]) + z * (D[5] + D[7])) - 4.f * y * (D[0] + D[8]) + ga[6];
            drot[3] = 2.f * (r * (D[3] - D[1]) + x * (D[2] + D[6]) + y * (D[5] + D[7])) - 4.f * z * (D[0] + D[4]) + ga[7];
            if (RAW && i < tail_start) {          // chain through exp and F.normalize (vr_activations_backward's arithmetic)
#pragma unroll
                for (int k = 0; k < 3; ++k) dsc[k] = dsc[k] * sc[k];
                const float g[4] = {drot[0], drot[1], drot[2], drot[3]};
                act_normalize_bwd(qraw, g, drot);
            }
        }
        if (RAW && i < tail_start) {              // ... and through the sigmoid
            const float y = act_sigmoid(opacities[i]);
            dop = dop * (1.0f - y) * y;
        }
    }
    if (!in_range) return;
    if (ACC) {
        if (!vis) return;
#pragma unroll
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * (size_t)i + k] += dmean[k];
        dL_dopacities[i] += dop;
        if (dL_dcolors) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dL_dcolors[3 * (size_t)i + k] += dcol[k];
        }
        if (dL_dscales) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dL_dscales[3 * (size_t)i + k] += dsc[k];
        }
        if (dL_drots) {
#pragma unroll
            for (int k = 0; k < 4; ++k) dL_drots[4 * (size_t)i + k] += drot[k];
        }
        if (dL_dcov3D) {
#pragma unroll
            for (int k = 0; k < 6; ++k) dL_dcov3D[6 * (size_t)i + k] += dcov[k];
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * (size_t)i + k] = dmean[k];
    dL_dopacities[i] = dop;
    if (dL_dcolors) {
#pragma unroll
        for (int k = 0; k < 3; ++k) dL_dcolors[3 * (size_t)i + k] = dcol[k];
    }
    if (dL_dscales) {
#pragma unroll
        for (int k = 0; k < 3; ++k) dL_dscales[3 * (size_t)i + k] = dsc[k];
    }
    if (dL_drots) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dL_drots[4 * (size_t)i + k] = drot[k];
    }
    if (dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; ++k) dL_dcov3D[6 * (size_t)i + k] = dcov[k];
    }
}

// true when k_preprocess_bwd takes the LDS-staged SH path and therefore writes EVERY row of dL_dshs
bool preprocess_bwd_writes_all_sh(int M, const float* shs, const float* dL_dshs)
{
    const int row = 3 * M;
    return shs && dL_dshs && row <= SH_ROW_MAX && (row & 3) == 0 && (reinterpret_cast<size_t>(shs) & 15) == 0 &&
           (reinterpret_cast<size_t>(dL_dshs) & 15) == 0;
}

int launch_preprocess_bwd(const Camera& cam, int P, const float* means3D, const float* shs, const float* shs_rest,
                          int tail_start, const float* colors_precomp, const float* opacities, const float* scales,
                          const float* rotations,
                          const float* cov3D_precomp, const int* radii, const uint8_t* clampb, const float* shd, const float* gacc,
                          const float* gmean2D, float* dL_dmeans3D, float* dL_dshs, float* dL_dshs_rest, float* dL_dshs_tail,
                          float* dL_dcolors,
                          float* dL_dopacities, float* dL_dscales, float* dL_drots, float* dL_dcov3D,
                          float* dL_dcolors_sh, bool store_factor, hipStream_t s, bool debug)
{
    if (P == 0) return 0;
    const int sh_staged = preprocess_bwd_writes_all_sh(cam.M, shs, dL_dshs) ? 1 : 0;
#define VR_PBWD(RAWP, ACCP)                                                                                                   \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_preprocess_bwd<RAWP, ACCP>), dim3(cdiv(P, 256)), dim3(256), 0, s, cam, P, sh_staged, \
                       means3D, shs, shs_rest, dL_dshs_rest, dL_dshs_tail, tail_start, colors_precomp, opacities, scales,     \
                       rotations, cov3D_precomp, radii, clampb, shd, gacc, gmean2D, dL_dmeans3D, dL_dshs, dL_dcolors,         \
                       dL_dopacities, dL_dscales, dL_drots, dL_dcov3D, dL_dcolors_sh, store_factor ? 1 : 0)
    const bool raw = (cam.flags & FLAG_RAW_PARAMS) != 0u, acc = (cam.flags & FLAG_ACCUMULATE_GRADS) != 0u;
    if (raw) { if (acc) VR_PBWD(true, true); else VR_PBWD(true, false); }
    else { if (acc) VR_PBWD(false, true); else VR_PBWD(false, false); }
#undef VR_PBWD
    VR_KERNEL_CHECK("preprocess_bwd", s, debug);
    return 0;
}

// The factor of the factored SH gradient on its own: dL/d(colour) of every Gaussian, zero where the colour was clamped
// and for culled rows -- complete as soon as the render backward is (its inputs are the accumulators only), i.e. BEFORE
// k_preprocess_bwd: a multi-GPU job starts exchanging it while that kernel runs (vr_backward_render).
__global__ void __launch_bounds__(256)
k_sh_factor(int P, const int* __restrict__ radii, const uint8_t* __restrict__ clampb, const float* __restrict__ gacc,
            float* __restrict__ dL_dcolors_sh)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (radii[i] > 0) {
        const float4 a1 = reinterpret_cast<const float4*>(gacc + (size_t)i * 16)[1];
        const uint32_t cb = clampb[i];
        g0 = (cb & 1u) ? 0.f : a1.x;
        g1 = (cb & 2u) ? 0.f : a1.y;
        g2 = (cb & 4u) ? 0.f : a1.z;
    }
    dL_dcolors_sh[3 * (size_t)i] = g0;
    dL_dcolors_sh[3 * (size_t)i + 1] = g1;
    dL_dcolors_sh[3 * (size_t)i + 2] = g2;
}

int launch_sh_factor(int P, const int* radii, const uint8_t* clampb, const float* gacc, float* dL_dcolors_sh,
                     hipStream_t s, bool debug)
{
    if (P == 0) return 0;
    hipLaunchKernelGGL(k_sh_factor, dim3(cdiv(P, 256)), dim3(256), 0, s, P, radii, clampb, gacc, dL_dcolors_sh);
    VR_KERNEL_CHECK("sh_factor", s, debug);
    return 0;
}

}  // namespace vr
