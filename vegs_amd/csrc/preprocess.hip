// preprocess.hip -- per-Gaussian forward: near-plane cull, projection, cov3D -> EWA cov2D -> conic,
// integer radius + tile rectangle, SH colour.  One lane per Gaussian; visible Gaussians write an
// 80-byte Splat record that the render kernels gather with dwordx4 loads.
//
// Replaces the native "preprocess" stage behind GaussianRasterizer.forward (reference call site
// gaussian_renderer/__init__.py:86-94).  In-repo definitions of the same math: SH colour
// utils/sh_utils.py:57-112 and the +0.5 / clamp at gaussian_renderer/__init__.py:79-80; cov3D
// utils/general_utils.py:97-129; camera matrices scene/cameras.py:76-88.
//
// Memory: the SH block is the bulk of the bytes (192 B per Gaussian at 16 coefficients).  A lane
// reading its own row would stride 192 B across the wave and over-fetch 2.5x (measured with
// FETCH_SIZE), so each wave stages the 64 contiguous rows of its Gaussians through LDS with fully
// coalesced dwordx4 loads and every lane then reads its row back from LDS.
#include <stdlib.h>

#include "vr_host.h"

namespace vr {

constexpr int SH_ROW_MAX = 48;  // floats per Gaussian staged through LDS (M <= 16 coefficients x 3)
// LDS row stride: 52 floats = 13 x 16 bytes.  Lanes read their own row, so with the natural 48-float stride
// eight consecutive lanes fell on two 16-byte bank groups (8-way conflicts on every access, 16-way for scalar
// accesses: SQ_WAIT_INST_LDS was 28 % of the backward kernel's wave cycles); 13 is odd, so eight consecutive
// lanes now cover all eight groups, and rows are moved as whole float4s.
constexpr int SH_LDS_STRIDE = 52;

// colour_c = sum_k basis[k] * sh[k][c]   (fixed fma order: part of the numerics contract)
__device__ __forceinline__ void sh_dot(const float* bas, int K, const float* sh, float* acc)
{
    float a0 = bas[0] * sh[0], a1 = bas[0] * sh[1], a2 = bas[0] * sh[2];
#pragma unroll
    for (int k = 1; k < 16; ++k) {       // constant trip count + guard: `sh` may be a register array
        if (k < K) {
            a0 = fmaf(bas[k], sh[3 * k + 0], a0);
            a1 = fmaf(bas[k], sh[3 * k + 1], a1);
            a2 = fmaf(bas[k], sh[3 * k + 2], a2);
        }
    }
    acc[0] = a0; acc[1] = a1; acc[2] = a2;
}

// LDS row stride of the HALF path: half an SH row (8 coefficients = 24 floats) + one float4 of padding = 7 x 16 bytes
// (odd: the per-lane float4 row reads are conflict-free, as with 13 above)
constexpr int SH_HALF_STRIDE = 28;
#ifndef VR_REC_LDS
#define VR_REC_LDS 1
#endif

// RAW (VR_FLAG_RAW_PARAMS) is a compile-time switch: the default instantiation is the kernel as it was.
// HALF (round 5): the whole-tensor, 16-coefficient SH layout (the op's plain `shs` argument: the headline path) staged
// through LDS in two HALVES of 8 coefficients and evaluated coefficient by coefficient (sh_accumulate: the operations of
// sh_dot / sh_ddir9 in their order, bit for bit) instead of 48-float rows against whole basis tables: 7 instead of 13 KB
// of LDS per wave and ~60 registers fewer -- the kernel is bound by the latency of its three dependent memory round trips
// per wave (means -> scales / rotations -> SH rows) at whatever occupancy it gets, and both resources held it to 3 waves
// per SIMD.
template <bool RAW, int HALF>      // HALF: 0 = rows of up to 48 floats, 1 = whole [P,16,3] tensor in coefficient halves, 2 = split storage in row halves
#ifndef PRE_THREADS
#define PRE_THREADS 128      // (two waves per workgroup: the same three waves per SIMD in finer grains, 159 -> 157.5 us; one wave: 170 VGPRs, 189 us)
#endif
// (the HALF paths run one wave per workgroup: 0.1318 -> 0.1288 ms against two, 0.134 with four; round 5)
#define PRE_WG(H) ((H) ? 64 : PRE_THREADS)
__global__ void __launch_bounds__(PRE_WG(HALF))
k_preprocess(Camera cam, int P, const float* __restrict__ means3D, const float* __restrict__ shs,
             const float* __restrict__ shs_rest, const float* __restrict__ shs_tail, int tail_start,
             const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
             const float* __restrict__ scales, const float* __restrict__ rotations,
             const float* __restrict__ cov3D_precomp, Splat* __restrict__ rec, int* __restrict__ radii,
             uint4* __restrict__ rect, uint32_t* __restrict__ depth_key, uint32_t* __restrict__ tile_count,
             uint8_t* __restrict__ clampb, float* __restrict__ shd)
{
    __shared__ __attribute__((aligned(16))) float sh_lds[PRE_WG(HALF) / 64][64 * (HALF ? SH_HALF_STRIDE : SH_LDS_STRIDE)];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool in_range = i < P;
    // culled unless proven visible
    bool vis = false;
    int rad = 0, ntiles = 0, rx0 = 0, ry0 = 0, rw = 0, rh = 0;
    float px = 0.f, py = 0.f, t2 = 0.f, conA = 0.f, conB = 0.f, conC = 0.f;
    float q[4] = {0.f, 0.f, 0.f, 0.f}, sc[3] = {0.f, 0.f, 0.f};
    float px3 = 0.f, py3 = 0.f, pz3 = 0.f;
    float t0 = 0.f, t1 = 0.f;
    if (in_range) {
        px3 = means3D[3 * (size_t)i];
        py3 = means3D[3 * (size_t)i + 1];
        pz3 = means3D[3 * (size_t)i + 2];
        xform43(cam.view, px3, py3, pz3, t0, t1, t2);
    }
#ifdef VR_EARLY_SH
    // REPRODUCER BUILD ONLY (python -m vegs_amd.build --variant early; profiles/experiments/README.md, round 5/6): the SH rows
    // of every Gaussian in front of the near plane requested a round trip earlier.  Same results, different timing and
    // register pressure: the build whose bench runs ended in a memory fault in k_emit_scan.
    float4 eA[6], eB[6];
    unsigned long long front_rows = 0ull;
    if (HALF == 1 && shs) {
        front_rows = __ballot(in_range && t2 > NEAR_Z);
        if (front_rows != 0ull) {
            const size_t wf = (size_t)(blockIdx.x * blockDim.x + w * 64);
            const float4* s4 = reinterpret_cast<const float4*>(shs + wf * 48);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int v = lane + 64 * j, r = v / 6, c = v - r * 6;
                if ((front_rows >> r) & 1ull) {
                    eA[j] = nt_load4(&s4[r * 12 + c]);
                    eB[j] = nt_load4(&s4[r * 12 + 6 + c]);
                }
            }
        }
    }
#endif
    if (in_range) {
        if (t2 > NEAR_Z) {
            float h0, h1, h2;
            xform43(cam.proj, px3, py3, pz3, h0, h1, h2);
            const float hw = xform_w(cam.proj, px3, py3, pz3);
            const float pw = 1.0f / (hw + 0.0000001f);
            const float ndcx = h0 * pw, ndcy = h1 * pw;
            float c6[6];
            if (cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * (size_t)i + k];
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) sc[k] = scales[3 * (size_t)i + k];
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = rotations[4 * (size_t)i + k];
                // (rows from tail_start on -- box instances behind the static model -- arrive activated and transformed)
                if (RAW && (cam.flags & FLAG_RAW_PARAMS) && i < tail_start) {      // raw parameters: the model's activations, here (the run-time
                                                                 // test keeps this instantiation at 160 VGPRs; without it: 170)
#pragma unroll
                    for (int k = 0; k < 3; ++k) sc[k] = expf(sc[k]);
                    act_normalize(q, q);      // (in place: element-wise after the norm)
                }
                cov3d_from_scale_rot(sc, cam.mod, q, c6);
            }
            Cov2D cv;
            cov2d(cam, cam.view, t0, t1, t2, c6, cv);
            const float det = cv.a * cv.c - cv.b * cv.b;
            if (det != 0.0f) {
                const float det_inv = 1.0f / det;
                const float mid = 0.5f * (cv.a + cv.c);
                const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
                const float lam = fmaxf(mid + sq, mid - sq);
                rad = (int)ceilf(3.0f * sqrtf(lam));
                px = ndc2pix(ndcx, cam.W);
                py = ndc2pix(ndcy, cam.H);
                int x0, y0, x1, y1;
                tile_rect(px, py, rad, cam.gx, cam.gy, x0, y0, x1, y1);
                ntiles = (x1 - x0) * (y1 - y0);
                rx0 = x0; ry0 = y0; rw = x1 - x0; rh = y1 - y0;
                vis = ntiles != 0;
                conA = cv.c * det_inv;
                conB = -cv.b * det_inv;
                conC = cv.a * det_inv;
            }
        }
    }

    // ---- tile mask (vr_device.h: TIGHT TILE LISTS): which tiles of the rectangle can the splat reach at all?
    const float opac = !vis ? 0.0f
                       : ((RAW && (cam.flags & FLAG_RAW_PARAMS) && i < tail_start) ? act_sigmoid(opacities[i]) : opacities[i]);
    unsigned long long tmask = 0ull;
    {
        const int area = vis ? rw * rh : 0;
        const bool full = (cam.flags & FLAG_FULL_TILE_LISTS) != 0u;
        // the reference's lists: every tile (beyond 64 tiles: low word = the cell mask, high word = the number of kept tiles)
        if (area > 0 && full)
            tmask = area > TIGHT_MAX_TILES ? (((unsigned long long)(uint32_t)area << 32) | 0xFFFFFFFFull)
                                           : (area == 64 ? ~0ull : ((1ull << area) - 1ull));
        const bool tested = area > 0 && !full;
        TileTest tt = {0, 0.f, 0.f, 0.f};
        if (tested) tt = tile_test_setup(conA, conB, conC, opac);
        // rectangles of up to 4 tiles (91 % of the Gaussians of a street view): by their own lane, four predicated steps
        if (tested && area <= 4) {
            int cx = 0, cy = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < area && tile_reachable(tt, px, py, conA, conB, conC, rx0 + cx, ry0 + cy)) tmask |= 1ull << j;
                if (++cx == rw) { cx = 0; ++cy; }
            }
        }
        // larger ones (two or three per wave): one at a time by the whole wave, lane j = tile j -- beyond 64 tiles: cell j of
        // k x k tiles, vr_device.h: tile_cells -- (a lane walking its own 64 tiles held its 63 neighbours up: k_preprocess 137 -> 183 us)
        int kc_l = 1, cw_l = rw, ch_l = rh;
        if (tested && area > TIGHT_MAX_TILES) tile_cells(rw, rh, kc_l, cw_l, ch_l);       // (rare: 0.02 % of a street view's Gaussians)
        for (unsigned long long big = __ballot(tested && area > 4); big; big &= big - 1) {
            const int src = __builtin_ctzll(big);
            auto bf = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); };
            const int b_w = __builtin_amdgcn_readlane(rw, src), b_h = __builtin_amdgcn_readlane(rh, src);
            const int b_x0 = __builtin_amdgcn_readlane(rx0, src), b_y0 = __builtin_amdgcn_readlane(ry0, src);
            const int kc = __builtin_amdgcn_readlane(kc_l, src), cw = __builtin_amdgcn_readlane(cw_l, src);
            const int chh = __builtin_amdgcn_readlane(ch_l, src);
            TileTest bt;
            bt.mode = __builtin_amdgcn_readlane(tt.mode, src);
            bt.lim = bf(tt.lim); bt.inv_A = bf(tt.inv_A); bt.inv_C = bf(tt.inv_C);
            const float b_px = bf(px), b_py = bf(py), b_A = bf(conA), b_B = bf(conB), b_C = bf(conC);
            const int cy = lane / cw, cx = lane - cy * cw;
            const int ntx = min(kc, b_w - cx * kc), nty = min(kc, b_h - cy * kc);
            const bool r = lane < cw * chh &&
                           tile_reachable(bt, b_px, b_py, b_A, b_B, b_C, b_x0 + cx * kc, b_y0 + cy * kc, ntx, nty);
            unsigned long long bm = __ballot(r);
            if (kc > 1) {        // (wave-uniform) the kept tiles are counted here, once: cells differ in size
                int kept = r ? ntx * nty : 0;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) kept += __shfl_xor(kept, d, 64);
                bm = (bm & 0xFFFFFFFFull) | ((unsigned long long)(uint32_t)kept << 32);
            }
            if (lane == src) tmask = bm;
        }
    }

    // ---- colour
    float rgb[3] = {0.f, 0.f, 0.f};
    uint32_t clampbits = 0;
    if (colors_precomp) {
        if (vis) {
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb[c] = colors_precomp[3 * (size_t)i + c];
        }
    } else if (HALF) {
        // HALF == 1 (host-checked): shs is the whole [P,16,3] tensor, 16-byte aligned; no split storage, no tail.
        // HALF == 2: split storage (shs = [P0,1,3] DC rows, shs_rest = [P0,15,3]), optionally with an SH TAIL (the box instances'
        // rows behind the static model, a whole [P - P0,16,3] tensor): a wave reads the split static rows, or -- behind
        // tail_start -- the tail tensor like HALF == 1 reads shs, or -- the one wave across the boundary -- its rows per lane.
        const unsigned long long vis_rows = __ballot(vis);
        if (vis_rows != 0ull) {
            const size_t wave_first = (size_t)(blockIdx.x * blockDim.x + w * 64);
            const int rows_here = min(64, P - (int)wave_first);
            const bool tailed = HALF == 2 && shs_tail != nullptr;
            const bool in_tail = tailed && (long)wave_first >= (long)tail_start;
            const bool straddle = tailed && !in_tail && (long)wave_first + 64 > (long)tail_start;
            float x = 0.f, y = 0.f, z = 0.f;
            if (vis) {
                x = px3 - cam.campos[0]; y = py3 - cam.campos[1]; z = pz3 - cam.campos[2];
                const float len = sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
                x = x / len; y = y / len; z = z / len;
            }
            const int K = (cam.deg + 1) * (cam.deg + 1);
            float acc[3] = {0.f, 0.f, 0.f}, D[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (HALF == 1 || in_tail) {
                // ---- whole 48-float rows in two coefficient HALVES
                float4* dst4 = reinterpret_cast<float4*>(sh_lds[w]);
                // Both halves of the wave's rows are requested at once (12 loads per lane in flight, rows of culled Gaussians
                // left out); the second half waits in registers while the first is evaluated.  (Requesting them a round trip
                // EARLIER -- for every Gaussian in front of the near plane, before the covariance and the tile test -- measured
                // slower, 0.152 against 0.148 ms.  That build's bench runs once ended in a memory fault in k_emit_scan on one box
                // (garbage ids out of the depth sort); it is kept as the reproducer build -DVR_EARLY_SH above and has not
                // faulted since, with or without the depth sort's post-mortem: profiles/experiments/README.md, round 6.)
                const float* rows = HALF == 1 ? shs + wave_first * 48 : shs_tail + (wave_first - (size_t)tail_start) * 48;
                const float4* src4 = reinterpret_cast<const float4*>(rows);
                float4 tA[6], tB[6];
                int at[6];
                bool ok[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int v = lane + 64 * j, r = v / 6, c = v - r * 6;
                    ok[j] = (vis_rows >> r) & 1ull;
                    at[j] = r * (SH_HALF_STRIDE / 4) + c;
#ifdef VR_EARLY_SH
                    if (HALF == 1) { tA[j] = eA[j]; tB[j] = eB[j]; continue; }
#endif
                    if (ok[j]) {
                        tA[j] = nt_load4(&src4[r * 12 + c]);
                        tB[j] = nt_load4(&src4[r * 12 + 6 + c]);
                    }
                }
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                const float4* my4 = dst4 + lane * (SH_HALF_STRIDE / 4);
#define VR_SH_K(KK, F) if (KK < K) sh_accumulate<KK>(x, y, z, xx, yy, zz, xy, yz, xz, &F[3 * ((KK) & 7)], acc, D)
#define VR_SH_HALF(BASE)                                                                                                  \
                if (vis) {                                                                                                \
                    float f[24];                                                                                          \
                    _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                                       \
                        const float4 t = my4[j];                                                                          \
                        f[4 * j] = t.x; f[4 * j + 1] = t.y; f[4 * j + 2] = t.z; f[4 * j + 3] = t.w;                       \
                    }                                                                                                     \
                    VR_SH_K(BASE + 0, f); VR_SH_K(BASE + 1, f); VR_SH_K(BASE + 2, f); VR_SH_K(BASE + 3, f);               \
                    VR_SH_K(BASE + 4, f); VR_SH_K(BASE + 5, f); VR_SH_K(BASE + 6, f); VR_SH_K(BASE + 7, f);               \
                }
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    if (ok[j]) dst4[at[j]] = tA[j];
                __builtin_amdgcn_wave_barrier();
                VR_SH_HALF(0)
                __builtin_amdgcn_wave_barrier();
                if (K > 8) {                                     // (wave-uniform: degree 0 and 1 need the first half only)
#pragma unroll
                    for (int j = 0; j < 6; ++j)
                        if (ok[j]) dst4[at[j]] = tB[j];
                    __builtin_amdgcn_wave_barrier();
                    VR_SH_HALF(8)
                    __builtin_amdgcn_wave_barrier();
                }
#undef VR_SH_HALF
#undef VR_SH_K
            } else if (HALF == 2 && straddle) {
                // ---- the one wave across tail_start: every lane reads its own coefficients from memory, one at a time
                if (vis) {
                    const bool mine_tail = i >= tail_start;
                    const float* p0 = mine_tail ? shs_tail + (size_t)(i - tail_start) * 48 : shs + 3 * (size_t)i;
                    const float* pr = mine_tail ? p0 + 3 : shs_rest + (size_t)i * 45;
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    sh_accumulate<0>(x, y, z, xx, yy, zz, xy, yz, xz, p0, acc, D);
#define VR_SH_G(KK) if (KK < K) sh_accumulate<KK>(x, y, z, xx, yy, zz, xy, yz, xz, pr + 3 * ((KK) - 1), acc, D)
                    VR_SH_G(1); VR_SH_G(2); VR_SH_G(3); VR_SH_G(4); VR_SH_G(5); VR_SH_G(6); VR_SH_G(7); VR_SH_G(8);
                    VR_SH_G(9); VR_SH_G(10); VR_SH_G(11); VR_SH_G(12); VR_SH_G(13); VR_SH_G(14); VR_SH_G(15);
#undef VR_SH_G
                }
            } else if (HALF == 2) {
                // ---- SPLIT storage, the model's own two tensors.  The wave's 64 rest rows are ONE linear block of 64 x 45
                // floats; a row is 180 bytes -- no float4 boundary inside it to cut coefficient halves at -- so the halves are
                // ROW halves: rows 0..31 are staged (1440 floats) and evaluated by lanes 0..31, then rows 32..63 by lanes
                // 32..63, coefficient by coefficient straight from LDS (row stride 45 floats: odd, the per-lane reads are
                // conflict-free).  Half the lanes idle during the evaluation -- 350 instructions twice in a kernel that waits
                // for memory -- for 5.6 instead of 11.3 KB of LDS per wave and the register budget of the whole-tensor path.
                const int rows_split = min(rows_here, (tailed ? tail_start : P) - (int)wave_first);
                float dc[3] = {0.f, 0.f, 0.f};
                if (vis) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) dc[c] = shs[3 * (size_t)i + c];
                }
                {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    if (vis) sh_accumulate<0>(x, y, z, xx, yy, zz, xy, yz, xz, dc, acc, D);
                }
                if (K > 1) {
                    const float* my = sh_lds[w] + (lane & 31) * 45;
#define VR_SH_R(KK) if (KK < K) sh_accumulate<KK>(x, y, z, xx, yy, zz, xy, yz, xz, my + 3 * ((KK) - 1), acc, D)
#pragma unroll 1
                    for (int h = 0; h < 2; ++h) {
                        const int first = 32 * h, nrows = max(0, min(32, rows_split - first));
                        const unsigned long long hrows = (vis_rows >> first) & 0xFFFFFFFFull;
                        if (hrows != 0ull) {      // (wave-uniform)
                            wave_copy_to_lds<6>(shs_rest + (wave_first + first) * 45, sh_lds[w], nrows * 45, lane, hrows, 45);
                            __builtin_amdgcn_wave_barrier();
                            if (vis && (lane >> 5) == h) {
                                // (the direction goes through an opaque move: otherwise the 15 basis values and their 45
                                // partial derivatives are loop invariants and are kept in registers across both halves -- 211 VGPRs)
                                asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
                                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                                VR_SH_R(1); VR_SH_R(2); VR_SH_R(3); VR_SH_R(4); VR_SH_R(5); VR_SH_R(6); VR_SH_R(7);
                                VR_SH_R(8); VR_SH_R(9); VR_SH_R(10); VR_SH_R(11); VR_SH_R(12); VR_SH_R(13); VR_SH_R(14); VR_SH_R(15);
                            }
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
#undef VR_SH_R
                }
            }
            {   // the wave's 64 D rows, one contiguous 2304-byte block: through LDS (row stride 9), out as float4s
                float* myd = sh_lds[w] + lane * 9;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < 9; ++q) myd[q] = vis ? D[q] : 0.0f;
                __builtin_amdgcn_wave_barrier();
                wave_copy_from_lds<3>(shd + wave_first * 9, sh_lds[w], rows_here * 9, lane, false);
            }
            if (vis) {
                acc[0] += 0.5f; acc[1] += 0.5f; acc[2] += 0.5f;
                clampbits = (acc[0] < 0.0f ? 1u : 0u) | (acc[1] < 0.0f ? 2u : 0u) | (acc[2] < 0.0f ? 4u : 0u);
                rgb[0] = fmaxf(acc[0], 0.0f); rgb[1] = fmaxf(acc[1], 0.0f); rgb[2] = fmaxf(acc[2], 0.0f);
            }
        }
    } else if (__ballot(vis) != 0ull) {
        // stage this wave's 64 SH rows (contiguous in memory) through LDS, coalesced
        const int row = cam.M * 3;                             // floats per Gaussian
        const size_t wave_first = (size_t)(blockIdx.x * blockDim.x + w * 64);
        // SH TAIL (VrInputs.shs_tail): rows tail_start.. live in a second, whole tensor.  A wave lies in the static part
        // (paths as without a tail, its rows ending at tail_start), in the tail (the whole-tensor path on the tail), or --
        // one wave of the launch at most -- across the boundary (every lane fetches its own row).
        const bool in_tail = shs_tail && (long)wave_first >= (long)tail_start;
        const bool straddle = shs_tail && !in_tail && (long)wave_first + 64 > (long)tail_start;
        const int seg_end = in_tail || !shs_tail ? P : tail_start;     // end of the rows this wave's SH source holds
        // whole-row source indexed by the Gaussian id (NULL: split storage)
        const float* const whole = in_tail ? shs_tail - (size_t)tail_start * row : (shs_rest ? nullptr : shs);
        const bool staged = row <= SH_ROW_MAX && (row & 3) == 0 && (reinterpret_cast<size_t>(whole) & 15) == 0;
        float acc[3] = {0.f, 0.f, 0.f};
        float bas[16], bx[16], by[16], bz[16];
        float D[9];            // d colour / d direction, kept for the backward (see sh_ddir9)
        const int K = (cam.deg + 1) * (cam.deg + 1);
        if (vis) {
            float dx = px3 - cam.campos[0], dy = py3 - cam.campos[1], dz = pz3 - cam.campos[2];
            const float len = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            dx = dx / len; dy = dy / len; dz = dz / len;
            sh_basis(cam.deg, dx, dy, dz, bas);
            sh_basis_grad(cam.deg, dx, dy, dz, bx, by, bz);
        }
        if (straddle) {
            if (vis) {
                float srow[SH_ROW_MAX];
                // (constant trip counts + guards: srow must stay a register array)
                const bool mine_tail = i >= tail_start;
                const float* head = mine_tail ? shs_tail + (size_t)(i - tail_start) * row
                                              : (shs_rest ? shs + 3 * (size_t)i : shs + (size_t)i * row);
                const float* rest = (!mine_tail && shs_rest) ? shs_rest + (size_t)i * (row - 3) : head + 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) srow[c] = head[c];
#pragma unroll
                for (int k = 0; k < SH_ROW_MAX - 3; ++k)
                    if (k < row - 3) srow[3 + k] = rest[k];
                sh_dot(bas, K, srow, acc);
                sh_ddir9(bx, by, bz, K, srow, D);
            }
        } else if (!whole) {
            // split storage (the model's own two tensors, no torch.cat): shs = [P,1,3] DC rows, shs_rest = [P,M-1,3].
            // The wave's 64 rest rows are one contiguous block: copied linearly into LDS, where the row stride is
            // the memory's own (45 floats at M = 16: odd, so the per-lane row reads are bank-conflict free).
            const int rowr = row - 3;
            const int rows_here = min(64, seg_end - (int)wave_first);
            float srow[SH_ROW_MAX];
            if (vis) {      // DC row first: its loads are in flight together with the staging loads below
#pragma unroll
                for (int c = 0; c < 3; ++c) srow[c] = shs[3 * (size_t)i + c];
            }
            wave_copy_to_lds<SH_ROW_MAX / 4>(shs_rest + wave_first * rowr, sh_lds[w], rows_here * rowr, lane, __ballot(vis),
                                             rowr);
            __builtin_amdgcn_wave_barrier();
            if (vis) {
                const float* my = sh_lds[w] + lane * rowr;
#pragma unroll
                for (int k = 0; k < SH_ROW_MAX - 3; ++k)
                    if (k < rowr) srow[3 + k] = my[k];
                sh_dot(bas, K, srow, acc);
                sh_ddir9(bx, by, bz, K, srow, D);
            }
        } else if (staged) {
            const float* src = whole + wave_first * row;
            const int rows_here = min(64, seg_end - (int)wave_first);
            const int nvec = rows_here * row / 4;               // float4s to move
            float4* dst4 = reinterpret_cast<float4*>(sh_lds[w]);
            const float4* src4 = reinterpret_cast<const float4*>(src);
            // all (up to 12) loads of the lane in flight before the first LDS store
            float4 tmp[SH_ROW_MAX / 4];
            const int row4 = row >> 2;
            // rows of culled Gaussians (18 % in the street scene, scattered) are not fetched: their lane never reads them
            const unsigned long long vis_rows = __ballot(vis);
            int rr[SH_ROW_MAX / 4];
#pragma unroll
            for (int j = 0; j < SH_ROW_MAX / 4; ++j) {
                const int v = lane + 64 * j;
                rr[j] = row4 == 12 ? v / 12 : v / row4;
                if (v < nvec && ((vis_rows >> rr[j]) & 1ull)) tmp[j] = nt_load4(&src4[v]);  // streamed once
            }
#pragma unroll
            for (int j = 0; j < SH_ROW_MAX / 4; ++j) {
                const int v = lane + 64 * j;
                if (v < nvec && ((vis_rows >> rr[j]) & 1ull)) dst4[rr[j] * (SH_LDS_STRIDE / 4) + (v - rr[j] * row4)] = tmp[j];
            }
            __builtin_amdgcn_wave_barrier();
            if (vis) {
                float srow[SH_ROW_MAX];
                const float4* my4 = dst4 + lane * (SH_LDS_STRIDE / 4);
#pragma unroll
                for (int j = 0; j < SH_ROW_MAX / 4; ++j) {
                    if (j < row4) {
                        const float4 t = my4[j];
                        srow[4 * j] = t.x; srow[4 * j + 1] = t.y; srow[4 * j + 2] = t.z; srow[4 * j + 3] = t.w;
                    }
                }
                sh_dot(bas, K, srow, acc);
                sh_ddir9(bx, by, bz, K, srow, D);
            }
        } else if (vis) {
            sh_dot(bas, K, whole + (size_t)i * row, acc);      // unusual M / alignment: direct row reads
            sh_ddir9(bx, by, bz, K, whole + (size_t)i * row, D);
        }
        {   // the wave's 64 D rows are one contiguous 2304-byte block: through LDS (row stride 9, odd), out as float4s
            const int rows_here = min(64, P - (int)wave_first);
            __builtin_amdgcn_wave_barrier();          // the SH rows in sh_lds[w] are no longer needed
            float* myd = sh_lds[w] + lane * 9;
#pragma unroll
            for (int q = 0; q < 9; ++q) myd[q] = vis ? D[q] : 0.0f;
            __builtin_amdgcn_wave_barrier();
            wave_copy_from_lds<3>(shd + wave_first * 9, sh_lds[w], rows_here * 9, lane, false);
        }
        if (vis) {
            acc[0] += 0.5f; acc[1] += 0.5f; acc[2] += 0.5f;
            clampbits = (acc[0] < 0.0f ? 1u : 0u) | (acc[1] < 0.0f ? 2u : 0u) | (acc[2] < 0.0f ? 4u : 0u);
            rgb[0] = fmaxf(acc[0], 0.0f); rgb[1] = fmaxf(acc[1], 0.0f); rgb[2] = fmaxf(acc[2], 0.0f);
        }
    }

    if (HALF && VR_REC_LDS) __builtin_amdgcn_wave_barrier();     // (the D rows have been read out of sh_lds[w])
    if (vis) {
        Splat s;
        s.x = px; s.y = py; s.conA = conA; s.conB = conB;
        s.conC = conC; s.opacity = opac; s.thr = splat_thr(opac); s.depth = t2;
        s.r = rgb[0]; s.g = rgb[1]; s.b = rgb[2]; s.qw = q[0];
        // blended scale row: the raw input row (A-3) or, with VR_FLAG_SCALE_MODIFIED, the row the covariance is built from
        const float sm = (cam.flags & FLAG_SCALE_MODIFIED) ? cam.mod : 1.0f;
        s.qx = q[1]; s.qy = q[2]; s.qz = q[3]; s.s0 = sc[0] * sm;
        s.s1 = sc[1] * sm; s.s2 = sc[2] * sm; s.clamped = clampbits; s.pad0 = 0;
        const float4* src = reinterpret_cast<const float4*>(&s);
        if (HALF && VR_REC_LDS) {
            float4* l4 = reinterpret_cast<float4*>(sh_lds[w]);     // (free: the SH rows and the D rows have left)
#pragma unroll
            for (int k = 0; k < 5; ++k) l4[lane * 5 + k] = src[k];
        } else {
            float4* dst = reinterpret_cast<float4*>(rec + i);
#pragma unroll
            for (int k = 0; k < 5; ++k) dst[k] = src[k];
        }
    }
    if (HALF && VR_REC_LDS) {
        // the wave's 64 records are one contiguous 5 KB block: out of LDS as whole float4 runs (a lane writing its own 80-byte
        // record made every store instruction touch 64 different lines); rows of culled Gaussians are left alone
        __builtin_amdgcn_wave_barrier();
        const unsigned long long vrows = __ballot(vis);
        if (vrows != 0ull) {
            const float4* l4 = reinterpret_cast<const float4*>(sh_lds[w]);
            float4* dst4 = reinterpret_cast<float4*>(rec + (size_t)(blockIdx.x * blockDim.x + w * 64));
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int v = lane + 64 * j, r = v / 5;
                if ((vrows >> r) & 1ull) dst4[v] = l4[v];
            }
        }
    }
    if (in_range) {
        radii[i] = vis ? rad : 0;
        // tile rectangle (x0, y0 | width, height), 16 bit each, and the mask of its reachable tiles (row-major bit j; with
        // more than 64 tiles: every tile): tiles touched = popcount, or width * height (0 when culled)
        rect[i] = vis ? make_uint4((uint32_t)rx0 | ((uint32_t)ry0 << 16), (uint32_t)rw | ((uint32_t)rh << 16),
                                   (uint32_t)tmask, (uint32_t)(tmask >> 32))
                      : make_uint4(0u, 0u, 0u, 0u);
        // depth key (bits of a positive float); DEPTH_KEY_NONE for a Gaussian without list entries (culled, or no tile of its
        // rectangle reachable): the compaction then needs only this array to know who is in (binning.hip: k_compact_apply)
        const uint32_t kept = rw * rh > TIGHT_MAX_TILES ? (uint32_t)(tmask >> 32) : (uint32_t)__popcll(tmask);
        depth_key[i] = vis && kept ? __float_as_uint(t2) : DEPTH_KEY_NONE;
        // (the same number rect_area reads off the rectangle -- one word for the totals' pass --, bit 31: more than 64 tiles)
        tile_count[i] = vis && kept ? (kept | (rw * rh > TIGHT_MAX_TILES ? ~TILE_COUNT_MASK : 0u)) : 0u;
        clampb[i] = (uint8_t)clampbits;   // dense copy for the backward (a 4-byte gather out of the 80-byte records costs a line each)
    }
}

__global__ void __launch_bounds__(256)
k_mark_visible(const float* __restrict__ xyz, int P, const float* __restrict__ view, uint8_t* __restrict__ present)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float t0, t1, t2;
    xform43(view, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], t0, t1, t2);
    present[i] = t2 > NEAR_Z ? 1 : 0;
}

int launch_preprocess(const Camera& cam, int P, const float* means3D, const float* shs, const float* shs_rest,
                      const float* shs_tail, int tail_start, const float* colors_precomp,
                      const float* opacities, const float* scales, const float* rotations,
                      const float* cov3D_precomp, Splat* rec, int* radii, uint4* rect,
                      uint32_t* depth_key, uint32_t* tile_count, uint8_t* clampb, float* shd, hipStream_t s, bool debug)
{
    if (P == 0) return 0;
    // HALF: the plain whole-tensor SH layout with all 16 coefficients stored (see k_preprocess); VEGS_PRE_HALF=0 keeps the
    // 48-float rows (A/B measurements)
    static const bool half_ok = [] { const char* e = getenv("VEGS_PRE_HALF"); return !(e && e[0] == '0'); }();
    const bool plain = half_ok && shs && !colors_precomp && cam.M == 16;
    const bool half = plain && !shs_rest && !shs_tail && (reinterpret_cast<size_t>(shs) & 15) == 0;   // whole [P,16,3] tensor
    // (features_dc, features_rest), with or without the instances' rows as an SH tail (16-byte aligned: check_inputs)
    const bool split = plain && shs_rest && (reinterpret_cast<size_t>(shs_rest) & 15) == 0 &&
                       (!shs_tail || (reinterpret_cast<size_t>(shs_tail) & 15) == 0);
#define VR_PRE(RAWP, HALFP)                                                                                               \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_preprocess<RAWP, HALFP>), dim3(cdiv(P, PRE_WG(HALFP))), dim3(PRE_WG(HALFP)), 0, s, cam, P, \
                       means3D, shs, shs_rest, shs_tail, tail_start, colors_precomp, opacities, scales, rotations,         \
                       cov3D_precomp, rec, radii, rect, depth_key, tile_count, clampb, shd)
    if (cam.flags & FLAG_RAW_PARAMS) { if (half) VR_PRE(true, 1); else if (split) VR_PRE(true, 2); else VR_PRE(true, 0); }
    else { if (half) VR_PRE(false, 1); else if (split) VR_PRE(false, 2); else VR_PRE(false, 0); }
#undef VR_PRE
    VR_KERNEL_CHECK("preprocess", s, debug);
    return 0;
}

int launch_mark_visible(const float* xyz, int P, const float* view, uint8_t* present, hipStream_t s)
{
    if (P == 0) return 0;
    hipLaunchKernelGGL(k_mark_visible, dim3(cdiv(P, 256)), dim3(256), 0, s, xyz, P, view, present);
    VR_KERNEL_CHECK("mark_visible", s, false);
    return 0;
}

}  // namespace vr
