// render_bwd.hip -- backward of the tile compositor.
//
// Replaces the native render-backward stage that loss.backward() reaches through the op's
// autograd.Function (reference train.py:196); semantics SURVEY.md A.5 + fork channels: for every
// fragment (pixel p, splat s) with weight w = alpha*T,
//   dL/dattr_s[k] += w * g_p[k]                                    (11 blended channels)
//   dL/dalpha     = T*u - (behind + T_final*bgterm) / (1 - alpha),  u = <attr_s, g_p>,
//                   behind = sum_{j behind s} w_j u_j
// and dL/dalpha fans out to opacity, conic and the 2D mean.
//
// MI355X design -- SPLAT-parallel with wave64 scans (not the pixel-parallel + per-fragment-atomic
// scheme of CUDA rasterizers).  The reduction target of the backward pass is the splat, so the splat
// owns the lane: a wave loads 64 consecutive list entries (one per lane, kept in registers together
// with their 17 gradient accumulators) and walks its 64 pixels; the per-pixel quantities are
// wave-uniform (v_readlane -> SGPR).  The two compositing recurrences become wave scans:
//   T before splat s   = T_after_chunk / prod_{j>=s}(1-alpha_j)    (inclusive scan-product)
//   behind(s)          = carry + sum_{j>s} w_j u_j                 (inclusive scan-sum)
// Lanes hold the chunk back-to-front (lane l <-> entry 63-l) so both are PREFIX scans over lanes.
// Per-splat gradients never leave registers until the chunk is finished; the four waves of a tile
// are then reduced through LDS and one coalesced set of global atomics per (tile, splat) is issued
// -- 256x fewer atomics than one per fragment.
#include "vr_host.h"

namespace vr {

constexpr int NACC = 17;  // conic(3) opacity(1) attr(11) mean2D(2)

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

__device__ __forceinline__ float wave_prefix_mul(float v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float n = __shfl_up(v, d, 64);
        if (lane >= d) v *= n;
    }
    return v;
}
__device__ __forceinline__ float wave_prefix_add(float v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

__global__ void __launch_bounds__(256)
k_render_bwd(Camera cam, const int2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
             const Splat* __restrict__ rec, const float* __restrict__ final_T,
             const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor,
             const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dquat,
             const float* __restrict__ dL_dscale, const float* __restrict__ dL_dalpha, float* __restrict__ gacc,
             float* __restrict__ gmean2D)
{
    __shared__ float red[4][64 * NACC];
    __shared__ uint32_t ids[64];
    __shared__ int wmax[4];
    const int ntiles = cam.gx * cam.gy;
    const int tile = xcd_tile(blockIdx.x, ntiles);
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int2 range = ranges[tile];
    const int nlist = range.y - range.x;

    // ---- pixel state, lane = pixel of this wave's 16x4 strip
    const int px = tx * TILE + (lane & 15), py = ty * TILE + w * 4 + (lane >> 4);
    const bool inside = px < cam.W && py < cam.H;
    const size_t N = (size_t)cam.H * cam.W;
    const size_t pix = (size_t)py * cam.W + px;
    const float v_pxf = (float)px, v_pyf = (float)py;
    float v_g[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) v_g[k] = 0.0f;
    float v_galpha = 0.0f, v_Tf = 1.0f;
    int v_nc = 0;
    if (inside) {
        if (dL_dcolor) { v_g[0] = dL_dcolor[pix]; v_g[1] = dL_dcolor[N + pix]; v_g[2] = dL_dcolor[2 * N + pix]; }
        if (dL_ddepth) v_g[3] = dL_ddepth[pix];
        if (dL_dquat) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v_g[4 + k] = dL_dquat[k * N + pix];
        }
        if (dL_dscale) {
#pragma unroll
            for (int k = 0; k < 3; ++k) v_g[8 + k] = dL_dscale[k * N + pix];
        }
        if (dL_dalpha) v_galpha = dL_dalpha[pix];
        v_Tf = final_T[pix];
        v_nc = (int)n_contrib[pix];
    }
    const float v_bgterm =
        v_Tf * (fmaf(cam.bg[2], v_g[2], fmaf(cam.bg[1], v_g[1], cam.bg[0] * v_g[0])) - v_galpha);
    float v_Tcar = v_Tf;  // transmittance after the last processed (later) chunk
    float v_Scar = 0.0f;  // sum of w*u over all later chunks

    int m = v_nc;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    const int wave_maxc = m;
    if (lane == 0) wmax[w] = m;
    __syncthreads();
    const int tile_maxc = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    const int nchunks = (tile_maxc + 63) >> 6;

    for (int c = nchunks - 1; c >= 0; --c) {
        // ---- lane l owns list entry c*64 + (63-l): back-to-front over lanes
        const int e = c * 64 + (63 - lane);
        const bool has = e < nlist;
        uint32_t id = 0;
        float sx = 0.f, sy = 0.f, cA = 0.f, cB = 0.f, cC = 0.f, op = 0.f;
        float at[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) at[k] = 0.0f;
        if (has) {
            id = point_list[range.x + e];
            const float4* src = reinterpret_cast<const float4*>(rec + id);
            const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3], q4 = src[4];
            sx = q0.x; sy = q0.y; cA = q0.z; cB = q0.w; cC = q1.x; op = q1.y;
            at[0] = q1.w; at[1] = q2.x; at[2] = q2.y; at[3] = q1.z;
            at[4] = q2.z; at[5] = q2.w; at[6] = q3.x; at[7] = q3.y;
            at[8] = q3.z; at[9] = q3.w; at[10] = q4.x;
        }
        float acc[NACC];
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = 0.0f;

        if (c * 64 < wave_maxc) {
            for (int p = 0; p < 64; ++p) {
                const int nc = __builtin_amdgcn_readlane(v_nc, p);
                if (nc <= c * 64) continue;  // pixel p has no contributor in this chunk (wave-uniform)
                const float pxf = readlane_f(v_pxf, p), pyf = readlane_f(v_pyf, p);
                const float Tc = readlane_f(v_Tcar, p), Sc = readlane_f(v_Scar, p);
                const float bgterm = readlane_f(v_bgterm, p);
                float g[NCH];
#pragma unroll
                for (int k = 0; k < NCH; ++k) g[k] = readlane_f(v_g[k], p);

                float dx, dy;
                const float power = splat_power(sx, sy, cA, cB, cC, pxf, pyf, dx, dy);
                const float G = vr_exp(power);
                const float alpha = fminf(ALPHA_MAX, op * G);
                const bool contrib = has && (e < nc) && !(power > 0.0f) && !(alpha < ALPHA_MIN);
                const float a_eff = contrib ? alpha : 0.0f;
                const float om = 1.0f - a_eff;
                const float pprod = wave_prefix_mul(om, lane);           // prod over entries >= mine
                const float Tl = Tc * __builtin_amdgcn_rcpf(pprod);      // T in front of my splat
                const float wgt = a_eff * Tl;
                float u = 0.0f;
#pragma unroll
                for (int k = 0; k < NCH; ++k) u = fmaf(at[k], g[k], u);
                const float wu = wgt * u;
                const float psum = wave_prefix_add(wu, lane);            // sum over entries >= mine
                const float behind = Sc + (psum - wu);
                // carries for the next (nearer) chunk: values at the chunk's first entry = lane 63
                v_Tcar = (lane == p) ? readlane_f(Tl, 63) : v_Tcar;
                v_Scar = (lane == p) ? Sc + readlane_f(psum, 63) : v_Scar;
                if (contrib) {
                    const float dLda = fmaf(Tl, u, -(behind + bgterm) * __builtin_amdgcn_rcpf(om));
                    const float dLdG = op * dLda;
                    const float gdx = G * dx, gdy = G * dy;
                    acc[0] = fmaf(-0.5f * gdx * dx, dLdG, acc[0]);
                    acc[1] = fmaf(-gdx * dy, dLdG, acc[1]);
                    acc[2] = fmaf(-0.5f * gdy * dy, dLdG, acc[2]);
                    acc[3] = fmaf(G, dLda, acc[3]);
#pragma unroll
                    for (int k = 0; k < NCH; ++k) acc[4 + k] = fmaf(wgt, g[k], acc[4 + k]);
                    acc[15] = fmaf(dLdG, -gdx * cA - gdy * cB, acc[15]);
                    acc[16] = fmaf(dLdG, -gdy * cC - gdx * cB, acc[16]);
                }
            }
        }
        // ---- reduce the four waves through LDS, then one coalesced atomic set per (tile, splat)
        __syncthreads();  // previous chunk's readers are done with red/ids
#pragma unroll
        for (int k = 0; k < NACC; ++k) red[w][lane * NACC + k] = acc[k];
        if (w == 0) ids[lane] = has ? id : 0xFFFFFFFFu;
        __syncthreads();
        for (int v = threadIdx.x; v < 64 * NACC; v += 256) {
            const int l = v / NACC, k = v - l * NACC;
            const uint32_t gid = ids[l];
            if (gid == 0xFFFFFFFFu) continue;
            float sum = (red[0][v] + red[1][v]) + (red[2][v] + red[3][v]);
            if (sum == 0.0f) continue;
            if (k < 15) atomicAdd(&gacc[(size_t)gid * 16 + k], sum);
            else atomicAdd(&gmean2D[(size_t)gid * 3 + (k - 15)], sum * (k == 15 ? 0.5f * (float)cam.W : 0.5f * (float)cam.H));
        }
    }
}

int launch_render_bwd(const Camera& cam, const int2* ranges, const uint32_t* point_list, const Splat* rec,
                      const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                      const float* dL_ddepth, const float* dL_dquat, const float* dL_dscale,
                      const float* dL_dalpha, float* gacc, float* gmean2D, hipStream_t s, bool debug)
{
    int ntiles = cam.gx * cam.gy;
    if (ntiles == 0) return 0;
    hipLaunchKernelGGL(k_render_bwd, dim3(ntiles), dim3(256), 0, s, cam, ranges, point_list, rec, final_T,
                       n_contrib, dL_dcolor, dL_ddepth, dL_dquat, dL_dscale, dL_dalpha, gacc, gmean2D);
    VR_KERNEL_CHECK("render_bwd", s, debug);
    return 0;
}

}  // namespace vr
