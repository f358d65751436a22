// instances.hip -- box-instance transform + concatenation feeding the rasterizer.
//
// The reference does this per instance with ~30 ATen launches forward and ~60 backward
// (gaussian_renderer/__init__.py:122-126,140-153 + torch.cat in merge_kwargs :182-186); with 8 instances in
// frame that is several hundred launches -- milliseconds of host time -- for 65 k Gaussians of work.  Here:
// ONE launch writes the transformed means / scales / rotations of ALL instances (and the static model's copy)
// straight into the concatenated arrays the rasterizer reads; ONE launch computes all per-Gaussian gradients
// and per-block partial sums of the 4x4 box2world gradients, which a tiny second kernel reduces in a fixed
// order and carries through decompose_T_to_RS (utils/graphics_utils.py:49-53).
#include "../../include/vegs_instances.h"
#include "vr_host.h"

namespace vr {

constexpr int INST_MAX = 16;
constexpr int NRED = 28;   // dB from the means (16) + dL/dRb (9) + dL/dS (3)

struct InstSeg {
    const float* means; const float* scales; const float* rot; const float* B;
    float* dmeans; float* dscales; float* drot;
    long n, offset;
    int block0;
};
struct InstArgs { InstSeg seg[INST_MAX]; int count; };

__device__ __forceinline__ int inst_of_block(const InstArgs& a)
{
    int t = 0;
#pragma unroll
    for (int i = 1; i < INST_MAX; ++i)
        if (i < a.count && (int)blockIdx.x >= a.seg[i].block0) t = i;
    return t;
}

// column norms S and column-normalised Rb of the upper-left 3x3 of B (decompose_T_to_RS)
__device__ __forceinline__ void decompose(const float* __restrict__ B, float S[3], float Rb[3][3])
{
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float a = B[j], b = B[4 + j], c = B[8 + j];
        S[j] = sqrtf(a * a + b * b + c * c);
        Rb[0][j] = a / S[j]; Rb[1][j] = b / S[j]; Rb[2][j] = c / S[j];
    }
}

__device__ __forceinline__ void quat_to_mat(const float q[4], float R[3][3], float& two_s)
{
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    two_s = 2.0f / (r * r + i * i + j * j + k * k);
    const float t = two_s;
    R[0][0] = 1.0f - t * (j * j + k * k); R[0][1] = t * (i * j - k * r); R[0][2] = t * (i * k + j * r);
    R[1][0] = t * (i * j + k * r); R[1][1] = 1.0f - t * (i * i + k * k); R[1][2] = t * (j * k - i * r);
    R[2][0] = t * (i * k - j * r); R[2][1] = t * (j * k + i * r); R[2][2] = 1.0f - t * (i * i + j * j);
}

// matrix_to_quaternion (utils/graphics_utils.py:140-201): candidate c = argmax_i sqrt(max(0, t_i)) (first maximum),
// q = v_c / (2 max(qa, 0.1)).  Returns c; v[4] = the un-normalised candidate, qa = its q_abs.
__device__ __forceinline__ int mat_to_quat_parts(const float m[3][3], float v[4], float& qa)
{
    const float t0 = 1.0f + m[0][0] + m[1][1] + m[2][2], t1 = 1.0f + m[0][0] - m[1][1] - m[2][2];
    const float t2 = 1.0f - m[0][0] + m[1][1] - m[2][2], t3 = 1.0f - m[0][0] - m[1][1] + m[2][2];
    const float a0 = sqrtf(fmaxf(t0, 0.0f)), a1 = sqrtf(fmaxf(t1, 0.0f)), a2 = sqrtf(fmaxf(t2, 0.0f)), a3 = sqrtf(fmaxf(t3, 0.0f));
    int c = 0; qa = a0;
    if (a1 > qa) { c = 1; qa = a1; }
    if (a2 > qa) { c = 2; qa = a2; }
    if (a3 > qa) { c = 3; qa = a3; }
    const float sq = qa * qa;
    if (c == 0)      { v[0] = sq; v[1] = m[2][1] - m[1][2]; v[2] = m[0][2] - m[2][0]; v[3] = m[1][0] - m[0][1]; }
    else if (c == 1) { v[0] = m[2][1] - m[1][2]; v[1] = sq; v[2] = m[1][0] + m[0][1]; v[3] = m[0][2] + m[2][0]; }
    else if (c == 2) { v[0] = m[0][2] - m[2][0]; v[1] = m[1][0] + m[0][1]; v[2] = sq; v[3] = m[1][2] + m[2][1]; }
    else             { v[0] = m[1][0] - m[0][1]; v[1] = m[2][0] + m[0][2]; v[2] = m[2][1] + m[1][2]; v[3] = sq; }
    return c;
}

__global__ void __launch_bounds__(256)
k_inst_fwd(InstArgs a, float* __restrict__ out_means, float* __restrict__ out_scales, float* __restrict__ out_rot)
{
    const InstSeg s = a.seg[inst_of_block(a)];
    const long i = (long)(blockIdx.x - s.block0) * 256 + threadIdx.x;
    if (i >= s.n) return;
    const long o = s.offset + i;
    const float x = s.means[3 * i], y = s.means[3 * i + 1], z = s.means[3 * i + 2];
    float q[4] = {s.rot[4 * i], s.rot[4 * i + 1], s.rot[4 * i + 2], s.rot[4 * i + 3]};
    float sc[3] = {s.scales[3 * i], s.scales[3 * i + 1], s.scales[3 * i + 2]};
    if (!s.B) {   // the static model: plain copy (torch.cat's first operand)
        out_means[3 * o] = x; out_means[3 * o + 1] = y; out_means[3 * o + 2] = z;
#pragma unroll
        for (int k = 0; k < 3; ++k) out_scales[3 * o + k] = sc[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) out_rot[4 * o + k] = q[k];
        return;
    }
    const float* B = s.B;
    const float w = B[12] * x + B[13] * y + B[14] * z + B[15];
#pragma unroll
    for (int r = 0; r < 3; ++r) out_means[3 * o + r] = (B[4 * r] * x + B[4 * r + 1] * y + B[4 * r + 2] * z + B[4 * r + 3]) / w;
    float S[3], Rb[3][3], Rq[3][3], Rm[3][3], two_s;
    decompose(B, S, Rb);
    quat_to_mat(q, Rq, two_s);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Rm[r][c] = Rb[r][0] * Rq[0][c] + Rb[r][1] * Rq[1][c] + Rb[r][2] * Rq[2][c];
    float v[4], qa;
    mat_to_quat_parts(Rm, v, qa);
    const float den = 2.0f * fmaxf(qa, 0.1f);
#pragma unroll
    for (int k = 0; k < 4; ++k) out_rot[4 * o + k] = v[k] / den;
#pragma unroll
    for (int k = 0; k < 3; ++k) out_scales[3 * o + k] = sc[k] * S[k];
}

__global__ void __launch_bounds__(256)
k_inst_bwd(InstArgs a, const float* __restrict__ g_means, const float* __restrict__ g_scales,
           const float* __restrict__ g_rot, double* __restrict__ partial)
{
    __shared__ double red[4][NRED];
    const InstSeg s = a.seg[inst_of_block(a)];
    const long i = (long)(blockIdx.x - s.block0) * 256 + threadIdx.x;
    float acc[NRED];
#pragma unroll
    for (int k = 0; k < NRED; ++k) acc[k] = 0.0f;
    if (i < s.n) {
        const long o = s.offset + i;
        const float* B = s.B;
        const float x[3] = {s.means[3 * i], s.means[3 * i + 1], s.means[3 * i + 2]};
        const float q[4] = {s.rot[4 * i], s.rot[4 * i + 1], s.rot[4 * i + 2], s.rot[4 * i + 3]};
        const float gm[3] = {g_means[3 * o], g_means[3 * o + 1], g_means[3 * o + 2]};
        // ---- means: y = h / w
        const float w = B[12] * x[0] + B[13] * x[1] + B[14] * x[2] + B[15];
        float yv[3], gh[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            yv[r] = (B[4 * r] * x[0] + B[4 * r + 1] * x[1] + B[4 * r + 2] * x[2] + B[4 * r + 3]) / w;
            gh[r] = gm[r] / w;
        }
        const float gw = -(gm[0] * yv[0] + gm[1] * yv[1] + gm[2] * yv[2]) / w;
#pragma unroll
        for (int c = 0; c < 3; ++c) s.dmeans[3 * i + c] = gh[0] * B[c] + gh[1] * B[4 + c] + gh[2] * B[8 + c] + gw * B[12 + c];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[4 * r + c] = gh[r] * x[c];
            acc[4 * r + 3] = gh[r];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[12 + c] = gw * x[c];
        acc[15] = gw;
        // ---- scales
        float S[3], Rb[3][3];
        decompose(B, S, Rb);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float g = g_scales[3 * o + k];
            s.dscales[3 * i + k] = g * S[k];
            acc[25 + k] = g * s.scales[3 * i + k];
        }
        // ---- rotations
        float Rq[3][3], Rm[3][3], two_s;
        quat_to_mat(q, Rq, two_s);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Rm[r][c] = Rb[r][0] * Rq[0][c] + Rb[r][1] * Rq[1][c] + Rb[r][2] * Rq[2][c];
        float v[4], qa;
        const int cs = mat_to_quat_parts(Rm, v, qa);
        const float den = 2.0f * fmaxf(qa, 0.1f);
        float gv[4], dot = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float g = g_rot[4 * o + k]; gv[k] = g / den; dot += g * v[k]; }
        const float gqa = qa > 0.1f ? -2.0f * dot / (den * den) : 0.0f;       // through the denominator 2 qa
        const float gt = gv[cs] + (qa > 0.0f ? gqa / (2.0f * qa) : 0.0f);       // v_c = t_c, qa = sqrt(t_c)
        float dRm[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        const float sg0 = (cs == 0 || cs == 1) ? 1.0f : -1.0f, sg1 = (cs == 0 || cs == 2) ? 1.0f : -1.0f,
                    sg2 = (cs == 0 || cs == 3) ? 1.0f : -1.0f;
        dRm[0][0] = gt * sg0; dRm[1][1] = gt * sg1; dRm[2][2] = gt * sg2;
        if (cs == 0) {
            dRm[2][1] += gv[1]; dRm[1][2] -= gv[1]; dRm[0][2] += gv[2]; dRm[2][0] -= gv[2]; dRm[1][0] += gv[3]; dRm[0][1] -= gv[3];
        } else if (cs == 1) {
            dRm[2][1] += gv[0]; dRm[1][2] -= gv[0]; dRm[1][0] += gv[2]; dRm[0][1] += gv[2]; dRm[0][2] += gv[3]; dRm[2][0] += gv[3];
        } else if (cs == 2) {
            dRm[0][2] += gv[0]; dRm[2][0] -= gv[0]; dRm[1][0] += gv[1]; dRm[0][1] += gv[1]; dRm[1][2] += gv[3]; dRm[2][1] += gv[3];
        } else {
            dRm[1][0] += gv[0]; dRm[0][1] -= gv[0]; dRm[2][0] += gv[1]; dRm[0][2] += gv[1]; dRm[2][1] += gv[2]; dRm[1][2] += gv[2];
        }
        // dL/dRb = dRm Rq^T (summed over the instance), dL/dRq = Rb^T dRm
        float G[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc[16 + 3 * r + c] = dRm[r][0] * Rq[c][0] + dRm[r][1] * Rq[c][1] + dRm[r][2] * Rq[c][2];
                G[r][c] = Rb[0][r] * dRm[0][c] + Rb[1][r] * dRm[1][c] + Rb[2][r] * dRm[2][c];
            }
        const float r = q[0], ii = q[1], j = q[2], k = q[3], t = two_s;
        const float GM = G[0][0] * -(j * j + k * k) + G[0][1] * (ii * j - k * r) + G[0][2] * (ii * k + j * r) +
                         G[1][0] * (ii * j + k * r) + G[1][1] * -(ii * ii + k * k) + G[1][2] * (j * k - ii * r) +
                         G[2][0] * (ii * k - j * r) + G[2][1] * (j * k + ii * r) + G[2][2] * -(ii * ii + j * j);
        const float dr = -k * G[0][1] + j * G[0][2] + k * G[1][0] - ii * G[1][2] - j * G[2][0] + ii * G[2][1];
        const float di = j * (G[0][1] + G[1][0]) + k * (G[0][2] + G[2][0]) - 2.0f * ii * (G[1][1] + G[2][2]) + r * (G[2][1] - G[1][2]);
        const float dj = -2.0f * j * (G[0][0] + G[2][2]) + ii * (G[0][1] + G[1][0]) + r * (G[0][2] - G[2][0]) + k * (G[1][2] + G[2][1]);
        const float dk = -2.0f * k * (G[0][0] + G[1][1]) + r * (G[1][0] - G[0][1]) + ii * (G[0][2] + G[2][0]) + j * (G[1][2] + G[2][1]);
        const float tt = t * t * GM;
        s.drot[4 * i] = t * dr - tt * r;
        s.drot[4 * i + 1] = t * di - tt * ii;
        s.drot[4 * i + 2] = t * dj - tt * j;
        s.drot[4 * i + 3] = t * dk - tt * k;
    }
    // ---- block partial sums of the 28 reduction values, in double
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NRED; ++k) {
        double v = (double)acc[k];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0) red[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NRED)
        partial[(size_t)blockIdx.x * NRED + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

struct InstFin { const float* B; float* dB; int block0, nblocks; };
struct InstFinArgs { InstFin f[INST_MAX]; };

// one block per instance: fixed-order sum of its blocks' partials, then decompose_T_to_RS backward
__global__ void __launch_bounds__(64) k_inst_finish(InstFinArgs a, const double* __restrict__ partial)
{
    __shared__ double tot[NRED];
    const InstFin f = a.f[blockIdx.x];
    if (threadIdx.x < NRED) {
        double s = 0.0;
        for (int b = 0; b < f.nblocks; ++b) s += partial[(size_t)(f.block0 + b) * NRED + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x >= 16) return;
    const int r = threadIdx.x >> 2, c = threadIdx.x & 3;
    double g = tot[threadIdx.x];
    if (r < 3 && c < 3) {
        // column c of the 3x3 block: Rb[:,c] = B[:,c] / S_c
        double col[3] = {(double)f.B[c], (double)f.B[4 + c], (double)f.B[8 + c]};
        const double S = sqrt(col[0] * col[0] + col[1] * col[1] + col[2] * col[2]);
        const double rb[3] = {col[0] / S, col[1] / S, col[2] / S};
        const double gc[3] = {tot[16 + c], tot[16 + 3 + c], tot[16 + 6 + c]};
        const double proj = gc[0] * rb[0] + gc[1] * rb[1] + gc[2] * rb[2];
        g += gc[r] / S - proj * rb[r] / S + tot[25 + c] * rb[r];
    }
    f.dB[threadIdx.x] = (float)g;
}


// ---- the model's activations (scene/gaussian_model.py:37-45,98-120: get_opacity = sigmoid, get_scaling = exp,
// get_rotation = torch.nn.functional.normalize) for all Gaussians in ONE launch each way.  Through ATen this is
// sigmoid + exp + (norm, clamp_min, expand, div) forward and twice that backward: ~15 launches streaming 2 M rows
// each (0.25 ms of a 2.5 ms iteration).  One Gaussian per thread; 32 bytes in, 32 bytes out.

__global__ void __launch_bounds__(256)
k_activate_fwd(const float* __restrict__ raw_opacity, const float* __restrict__ raw_scaling,
               const float4* __restrict__ raw_rotation, long P, float* __restrict__ opacity, float* __restrict__ scales,
               float4* __restrict__ rotations)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    opacity[i] = act_sigmoid(raw_opacity[i]);
#pragma unroll
    for (int k = 0; k < 3; ++k) scales[3 * i + k] = expf(raw_scaling[3 * i + k]);
    const float4 q4 = raw_rotation[i];
    const float q[4] = {q4.x, q4.y, q4.z, q4.w};
    float y[4];
    act_normalize(q, y);
    rotations[i] = make_float4(y[0], y[1], y[2], y[3]);
}

// y = sigmoid(x): dx = g y (1 - y);  y = exp(x): dx = g y;  y = x / n, n = max(|x|, eps): dx = (g - y <y, g>) / n
// where the norm is not clamped, g / eps where it is (the clamp's gradient is zero there).
__global__ void __launch_bounds__(256)
k_activate_bwd(const float* __restrict__ opacity, const float* __restrict__ scales, const float4* __restrict__ raw_rotation,
               long P, const float* __restrict__ g_opacity, const float* __restrict__ g_scales,
               const float4* __restrict__ g_rotations, float* __restrict__ d_opacity, float* __restrict__ d_scaling,
               float4* __restrict__ d_rotation)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    if (d_opacity) {
        const float y = opacity[i];
        d_opacity[i] = g_opacity ? g_opacity[i] * (1.0f - y) * y : 0.0f;
    }
    if (d_scaling) {
#pragma unroll
        for (int k = 0; k < 3; ++k) d_scaling[3 * i + k] = g_scales ? g_scales[3 * i + k] * scales[3 * i + k] : 0.0f;
    }
    if (d_rotation) {
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g_rotations) {
            const float4 q4 = raw_rotation[i], g4 = g_rotations[i];
            const float q[4] = {q4.x, q4.y, q4.z, q4.w}, g[4] = {g4.x, g4.y, g4.z, g4.w};
            float dd[4];
            act_normalize_bwd(q, g, dd);
            d = make_float4(dd[0], dd[1], dd[2], dd[3]);
        }
        d_rotation[i] = d;
    }
}


// ---- BoxModel (model/boxmodel.py:6-49): the learnable correction of an instance's annotated pose.
//   D = [[diag(delta_s) R(delta_r), delta_t], [0 0 0 1]],   adjusted = box2world @ D       (:30-42)
// with R = quaternion_to_matrix (utils/graphics_utils.py:204-248: normalises by |q|^2).  The reference evaluates this per
// instance with ~15 ATen launches forward and ~30 backward; ten numbers per instance -- one THREAD per instance here, all
// instances of a frame in one launch each way.  The backward also applies the "do not update nan gradients" rule of
// train.py:199-205, and vr_boxmodel_regularizer_grad is the gradient of BoxModel.regularize's loss (:44-49).
constexpr int BOX_MAX = 64;
struct BoxSeg { const float* B; const float* dr; const float* ds; const float* dt; float* g_r; float* g_s; float* g_t; };
struct BoxArgs { BoxSeg seg[BOX_MAX]; int count; };

__global__ void __launch_bounds__(64) k_box_fwd(BoxArgs a, float* __restrict__ out)
{
    const int b = threadIdx.x;
    if (b >= a.count) return;
    const BoxSeg s = a.seg[b];
    const float q[4] = {s.dr[0], s.dr[1], s.dr[2], s.dr[3]};
    float R[3][3], two_s;
    quat_to_mat(q, R, two_s);
    float D[4][4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float sc = s.ds[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) D[i][j] = sc * R[i][j];
        D[i][3] = s.dt[i];
    }
    D[3][0] = D[3][1] = D[3][2] = 0.0f; D[3][3] = 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = s.B[4 * i] * D[0][j];
#pragma unroll
            for (int k = 1; k < 4; ++k) acc = fmaf(s.B[4 * i + k], D[k][j], acc);
            out[16 * b + 4 * i + j] = acc;
        }
}

__global__ void __launch_bounds__(64) k_box_bwd(BoxArgs a, const float* __restrict__ g_out, int nan_guard)
{
    const int b = threadIdx.x;
    if (b >= a.count) return;
    const BoxSeg s = a.seg[b];
    const float* G = g_out + 16 * b;
    // dD = B^T G (rows 0..2 of D are the only ones that depend on the deltas)
    float dD[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = s.B[i] * G[j];
#pragma unroll
            for (int k = 1; k < 4; ++k) acc = fmaf(s.B[4 * k + i], G[4 * k + j], acc);
            dD[i][j] = acc;
        }
    const float q[4] = {s.dr[0], s.dr[1], s.dr[2], s.dr[3]};
    float R[3][3], t;
    quat_to_mat(q, R, t);
    float gs[3], gt[3], dR[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        gt[i] = dD[i][3];
        gs[i] = fmaf(dD[i][2], R[i][2], fmaf(dD[i][1], R[i][1], dD[i][0] * R[i][0]));
        const float sc = s.ds[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) dR[i][j] = sc * dD[i][j];
    }
    // R = I + t A(q), t = 2 / |q|^2:  dL/dq = t dA/dq : dR  +  (A : dR) dt/dq,  dt/dq = -t^2 q
    const float r = q[0], i_ = q[1], j_ = q[2], k_ = q[3];
    const float A00 = -(j_ * j_ + k_ * k_), A01 = i_ * j_ - k_ * r, A02 = i_ * k_ + j_ * r;
    const float A10 = i_ * j_ + k_ * r, A11 = -(i_ * i_ + k_ * k_), A12 = j_ * k_ - i_ * r;
    const float A20 = i_ * k_ - j_ * r, A21 = j_ * k_ + i_ * r, A22 = -(i_ * i_ + j_ * j_);
    const float AdR = A00 * dR[0][0] + A01 * dR[0][1] + A02 * dR[0][2] + A10 * dR[1][0] + A11 * dR[1][1] + A12 * dR[1][2]
                    + A20 * dR[2][0] + A21 * dR[2][1] + A22 * dR[2][2];
    float gq[4];
    gq[0] = t * (-k_ * dR[0][1] + j_ * dR[0][2] + k_ * dR[1][0] - i_ * dR[1][2] - j_ * dR[2][0] + i_ * dR[2][1]);
    gq[1] = t * (j_ * dR[0][1] + k_ * dR[0][2] + j_ * dR[1][0] - 2.0f * i_ * dR[1][1] - r * dR[1][2] + k_ * dR[2][0] + r * dR[2][1]
                 - 2.0f * i_ * dR[2][2]);
    gq[2] = t * (-2.0f * j_ * dR[0][0] + i_ * dR[0][1] + r * dR[0][2] + i_ * dR[1][0] + k_ * dR[1][2] - r * dR[2][0] + k_ * dR[2][1]
                 - 2.0f * j_ * dR[2][2]);
    gq[3] = t * (-2.0f * k_ * dR[0][0] - r * dR[0][1] + i_ * dR[0][2] + r * dR[1][0] - 2.0f * k_ * dR[1][1] + j_ * dR[1][2]
                 + i_ * dR[2][0] + j_ * dR[2][1]);
    const float dt_scale = -(t * t) * AdR;
#pragma unroll
    for (int c = 0; c < 4; ++c) gq[c] = fmaf(dt_scale, q[c], gq[c]);
    if (nan_guard) {      // train.py:199-205: a NaN in delta_r.grad or delta_s.grad zeroes all three gradients
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) bad |= (gq[c] != gq[c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) bad |= (gs[c] != gs[c]);
        if (bad) {
#pragma unroll
            for (int c = 0; c < 4; ++c) gq[c] = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) { gs[c] = 0.0f; gt[c] = 0.0f; }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) s.g_r[c] = gq[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) { s.g_s[c] = gs[c]; s.g_t[c] = gt[c]; }
}

// gradient of  lambda (|delta_r - (1,0,0,0)| + |delta_s - 1| + |delta_t|)  as autograd computes it: x (lambda / |x|), and 0
// where the norm is 0 (torch's norm backward masks that case)
__global__ void __launch_bounds__(64) k_box_reg(BoxArgs a, float lambda)
{
    const int b = threadIdx.x;
    if (b >= a.count) return;
    const BoxSeg s = a.seg[b];
    const float xr[4] = {s.dr[0] - 1.0f, s.dr[1], s.dr[2], s.dr[3]};
    const float xs[3] = {s.ds[0] - 1.0f, s.ds[1] - 1.0f, s.ds[2] - 1.0f};
    const float xt[3] = {s.dt[0], s.dt[1], s.dt[2]};
    const float nr = sqrtf(xr[0] * xr[0] + xr[1] * xr[1] + xr[2] * xr[2] + xr[3] * xr[3]);
    const float ns = sqrtf(xs[0] * xs[0] + xs[1] * xs[1] + xs[2] * xs[2]);
    const float nt = sqrtf(xt[0] * xt[0] + xt[1] * xt[1] + xt[2] * xt[2]);
    const float fr = nr > 0.0f ? lambda / nr : 0.0f, fs = ns > 0.0f ? lambda / ns : 0.0f, ft = nt > 0.0f ? lambda / nt : 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) s.g_r[c] = xr[c] * fr;
#pragma unroll
    for (int c = 0; c < 3; ++c) { s.g_s[c] = xs[c] * fs; s.g_t[c] = xt[c] * ft; }
}

static int fill_box_args(const VrBoxModel* boxes, const VrBoxModelGrads* grads, int first, int n, bool need_base, BoxArgs& a)
{
    a.count = n;
    for (int i = 0; i < n; ++i) {
        const VrBoxModel& b = boxes[first + i];
        if ((need_base && !b.box2world) || !b.delta_r || !b.delta_s || !b.delta_t)
            { set_error("boxmodel: instance %d has a NULL array", first + i); return VR_ERR_INVALID_ARGUMENT; }
        a.seg[i] = BoxSeg{b.box2world, b.delta_r, b.delta_s, b.delta_t, nullptr, nullptr, nullptr};
        if (grads) {
            const VrBoxModelGrads& g = grads[first + i];
            if (!g.d_delta_r || !g.d_delta_s || !g.d_delta_t)
                { set_error("boxmodel: instance %d has a NULL gradient array", first + i); return VR_ERR_INVALID_ARGUMENT; }
            a.seg[i].g_r = g.d_delta_r; a.seg[i].g_s = g.d_delta_s; a.seg[i].g_t = g.d_delta_t;
        }
    }
    return VR_OK;
}

}  // namespace vr

using namespace vr;

static int fill_args(const VrInstance* inst, const VrInstanceGrads* grads, int first, int count, bool backward,
                     InstArgs& a, int& blocks)
{
    a.count = 0;
    blocks = 0;
    for (int i = first; i < count && a.count < INST_MAX; ++i) {
        const VrInstance& t = inst[i];
        if (t.n == 0 || (backward && !t.box2world)) continue;
        InstSeg& s = a.seg[a.count++];
        s.means = t.means; s.scales = t.scales; s.rot = t.rotations; s.B = t.box2world;
        s.dmeans = grads ? grads[i].dL_dmeans : nullptr;
        s.dscales = grads ? grads[i].dL_dscales : nullptr;
        s.drot = grads ? grads[i].dL_drotations : nullptr;
        s.n = (long)t.n; s.offset = (long)t.offset; s.block0 = blocks;
        blocks += cdiv((long)t.n, 256);
    }
    for (int i = a.count; i < INST_MAX; ++i) { a.seg[i] = InstSeg{}; a.seg[i].block0 = 0x7fffffff; }
    return 0;
}

static int check_instances(const VrInstance* inst, int32_t count)
{
    if (count < 0 || (count > 0 && !inst)) { set_error("instances: bad instance list"); return VR_ERR_INVALID_ARGUMENT; }
    for (int i = 0; i < count; ++i)
        if (inst[i].n < 0 || inst[i].offset < 0 || (inst[i].n > 0 && (!inst[i].means || !inst[i].scales || !inst[i].rotations))) {
            set_error("instances: instance %d has a NULL array or a negative size/offset", i);
            return VR_ERR_INVALID_ARGUMENT;
        }
    return VR_OK;
}

extern "C" int vr_instances_forward(const VrInstance* inst, int32_t count, float* out_means, float* out_scales,
                                    float* out_rotations, void* stream)
{
    if (int rc = check_instances(inst, count)) return rc;
    if (count > 0 && (!out_means || !out_scales || !out_rotations)) { set_error("instances: outputs are required"); return VR_ERR_INVALID_ARGUMENT; }
    // INST_MAX instances per launch; the table skips empty ones, so walk by consumed entries
    int i = 0;
    while (i < count) {
        InstArgs a;
        int blocks = 0, used = 0, j = i;
        a.count = 0;
        for (; j < count && used < INST_MAX; ++j)
            if (inst[j].n > 0) ++used;
        fill_args(inst, nullptr, i, j, false, a, blocks);
        if (blocks > 0) {
            hipLaunchKernelGGL(k_inst_fwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, out_means, out_scales, out_rotations);
            if (hipGetLastError() != hipSuccess) { set_error("instances: forward launch failed"); return VR_ERR_HIP; }
        }
        i = j;
    }
    return VR_OK;
}

extern "C" int vr_instances_backward(const VrInstance* inst, const VrInstanceGrads* grads, int32_t count,
                                     const float* g_means, const float* g_scales, const float* g_rotations,
                                     VrAllocFn alloc, void* user, void* stream)
{
    if (int rc = check_instances(inst, count)) return rc;
    if (count > 0 && (!grads || !g_means || !g_scales || !g_rotations || !alloc)) {
        set_error("instances: grads, upstream gradients and alloc are required");
        return VR_ERR_INVALID_ARGUMENT;
    }
    for (int i = 0; i < count; ++i)
        if (inst[i].box2world && inst[i].n > 0 &&
            (!grads[i].dL_dmeans || !grads[i].dL_dscales || !grads[i].dL_drotations || !grads[i].dL_dbox2world)) {
            set_error("instances: instance %d needs all four gradient destinations", i);
            return VR_ERR_INVALID_ARGUMENT;
        }
    int i = 0;
    while (i < count) {
        int used = 0, j = i;
        for (; j < count && used < INST_MAX; ++j)
            if (inst[j].n > 0 && inst[j].box2world) ++used;
        InstArgs a;
        int blocks = 0;
        fill_args(inst, grads, i, j, true, a, blocks);
        if (blocks > 0) {
            double* partial = (double*)alloc(user, VR_BUF_SCRATCH, (size_t)blocks * NRED * sizeof(double));
            if (!partial) { set_error("allocator returned NULL"); return VR_ERR_ALLOC; }
            hipLaunchKernelGGL(k_inst_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, g_means, g_scales, g_rotations, partial);
            if (hipGetLastError() != hipSuccess) { set_error("instances: backward launch failed"); return VR_ERR_HIP; }
            InstFinArgs fa;
            int nf = 0;
            for (int k = i; k < j; ++k) {
                if (inst[k].n == 0 || !inst[k].box2world) continue;
                fa.f[nf] = InstFin{inst[k].box2world, grads[k].dL_dbox2world, a.seg[nf].block0, cdiv((long)inst[k].n, 256)};
                ++nf;
            }
            hipLaunchKernelGGL(k_inst_finish, dim3(nf), dim3(64), 0, (hipStream_t)stream, fa, (const double*)partial);
            if (hipGetLastError() != hipSuccess) { set_error("instances: finish launch failed"); return VR_ERR_HIP; }
        }
        i = j;
    }
    // instances with n == 0 but a box2world: their box2world gradient is zero
    for (int k = 0; k < count; ++k)
        if (inst[k].box2world && inst[k].n == 0 && grads[k].dL_dbox2world)
            if (hipMemsetAsync(grads[k].dL_dbox2world, 0, 16 * sizeof(float), (hipStream_t)stream) != hipSuccess) {
                set_error("instances: memset failed");
                return VR_ERR_HIP;
            }
    return VR_OK;
}

extern "C" int vr_activations_forward(const float* raw_opacity, const float* raw_scaling, const float* raw_rotation,
                                      int64_t P, float* opacity, float* scales, float* rotations, void* stream)
{
    if (P < 0 || (P > 0 && (!raw_opacity || !raw_scaling || !raw_rotation || !opacity || !scales || !rotations))) {
        set_error("activations: six arrays and P >= 0 are required");
        return VR_ERR_INVALID_ARGUMENT;
    }
    if (((uintptr_t)raw_rotation | (uintptr_t)rotations) & 15u) {
        set_error("activations: the rotation arrays must be 16-byte aligned");
        return VR_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return VR_OK;
    hipLaunchKernelGGL(k_activate_fwd, dim3(cdiv((long)P, 256)), dim3(256), 0, (hipStream_t)stream, raw_opacity, raw_scaling,
                       (const float4*)raw_rotation, (long)P, opacity, scales, (float4*)rotations);
    if (hipGetLastError() != hipSuccess) { set_error("activations: forward launch failed"); return VR_ERR_HIP; }
    return VR_OK;
}

extern "C" int vr_activations_backward(const float* opacity, const float* scales, const float* raw_rotation, int64_t P,
                                       const float* g_opacity, const float* g_scales, const float* g_rotations,
                                       float* dL_draw_opacity, float* dL_draw_scaling, float* dL_draw_rotation,
                                       void* stream)
{
    if (P < 0 || (P > 0 && ((dL_draw_opacity && !opacity) || (dL_draw_scaling && !scales) || (dL_draw_rotation && !raw_rotation)))) {
        set_error("activations: every requested gradient needs its forward array");
        return VR_ERR_INVALID_ARGUMENT;
    }
    if (((uintptr_t)raw_rotation | (uintptr_t)g_rotations | (uintptr_t)dL_draw_rotation) & 15u) {
        set_error("activations: the rotation arrays must be 16-byte aligned");
        return VR_ERR_INVALID_ARGUMENT;
    }
    if (P == 0 || (!dL_draw_opacity && !dL_draw_scaling && !dL_draw_rotation)) return VR_OK;
    hipLaunchKernelGGL(k_activate_bwd, dim3(cdiv((long)P, 256)), dim3(256), 0, (hipStream_t)stream, opacity, scales,
                       (const float4*)raw_rotation, (long)P, g_opacity, g_scales, (const float4*)g_rotations, dL_draw_opacity,
                       dL_draw_scaling, (float4*)dL_draw_rotation);
    if (hipGetLastError() != hipSuccess) { set_error("activations: backward launch failed"); return VR_ERR_HIP; }
    return VR_OK;
}


extern "C" int vr_boxmodel_forward(const VrBoxModel* boxes, int32_t count, float* adjusted, void* stream)
{
    if (count < 0 || (count > 0 && (!boxes || !adjusted))) { set_error("boxmodel: bad arguments"); return VR_ERR_INVALID_ARGUMENT; }
    for (int first = 0; first < count; first += BOX_MAX) {
        BoxArgs a;
        const int n = count - first < BOX_MAX ? count - first : BOX_MAX;
        if (int rc = fill_box_args(boxes, nullptr, first, n, true, a)) return rc;
        hipLaunchKernelGGL(k_box_fwd, dim3(1), dim3(64), 0, (hipStream_t)stream, a, adjusted + 16 * (size_t)first);
        if (hipGetLastError() != hipSuccess) { set_error("boxmodel: forward launch failed"); return VR_ERR_HIP; }
    }
    return VR_OK;
}

extern "C" int vr_boxmodel_backward(const VrBoxModel* boxes, const VrBoxModelGrads* grads, int32_t count,
                                    const float* g_adjusted, int32_t nan_guard, void* stream)
{
    if (count < 0 || (count > 0 && (!boxes || !grads || !g_adjusted))) { set_error("boxmodel: bad arguments"); return VR_ERR_INVALID_ARGUMENT; }
    for (int first = 0; first < count; first += BOX_MAX) {
        BoxArgs a;
        const int n = count - first < BOX_MAX ? count - first : BOX_MAX;
        if (int rc = fill_box_args(boxes, grads, first, n, true, a)) return rc;
        hipLaunchKernelGGL(k_box_bwd, dim3(1), dim3(64), 0, (hipStream_t)stream, a, g_adjusted + 16 * (size_t)first, (int)nan_guard);
        if (hipGetLastError() != hipSuccess) { set_error("boxmodel: backward launch failed"); return VR_ERR_HIP; }
    }
    return VR_OK;
}

extern "C" int vr_boxmodel_regularizer_grad(const VrBoxModel* boxes, const VrBoxModelGrads* grads, int32_t count,
                                            float lambda_reg, void* stream)
{
    if (count < 0 || (count > 0 && (!boxes || !grads))) { set_error("boxmodel: bad arguments"); return VR_ERR_INVALID_ARGUMENT; }
    for (int first = 0; first < count; first += BOX_MAX) {
        BoxArgs a;
        const int n = count - first < BOX_MAX ? count - first : BOX_MAX;
        if (int rc = fill_box_args(boxes, grads, first, n, false, a)) return rc;
        hipLaunchKernelGGL(k_box_reg, dim3(1), dim3(64), 0, (hipStream_t)stream, a, lambda_reg);
        if (hipGetLastError() != hipSuccess) { set_error("boxmodel: regularizer launch failed"); return VR_ERR_HIP; }
    }
    return VR_OK;
}
