/*
 * vr_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).
 *
 * Scalar fp32 restatement of the differentiable Gaussian-splatting rasterizer that
 * VEGS calls through `diff_gaussian_rasterization` (reference call sites:
 * gaussian_renderer/__init__.py:38-53,86-94 ; utils/norminit_utils.py:163-179).
 *
 * PARITY UNPINNED: the rasterizer's own source is an un-vendored submodule
 * (reference .gitmodules:7-9, emjay73/diff_gaussian_rasterization_with_depth, gitlink
 * only) and the reference holds no tests or golden vectors for it (SURVEY.md section 4,
 * 8c).  This file therefore restates (i) the in-repo definitions of the same math --
 * SH colour utils/sh_utils.py:57-112 (+0.5 / clamp at gaussian_renderer/__init__.py:79-80),
 * cov3D scene/gaussian_model.py:32-36 + utils/general_utils.py:83-129, quaternion
 * convention utils/graphics_utils.py:204-248, camera matrices scene/cameras.py:76-88 --
 * and (ii) the published 3DGS tile-rasterizer algorithm as recorded in SURVEY.md
 * Appendix A (A.1-A.7) with the fork assumptions A-1..A-6.  It is pinned against
 * golden vectors produced by importing those reference functions
 * (tests/golden/make_golden.py) and against an independent float64 autograd
 * restatement (oracle/torch_ref.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * fp32 operation order is part of the spec: compile with -ffp-contract=off; every
 * fused multiply-add is an explicit fmaf().  The HIP kernels follow the same order
 * so radii, tile rects, sort keys, point lists and the forward images are
 * bit-reproducible between this file and the GPU.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16
#define NEAR_Z 0.2f
#define ALPHA_MIN (1.0f / 255.0f)
#define ALPHA_MAX 0.99f
#define T_EPS 0.0001f

typedef struct {
    int H, W;
    float tanfovx, tanfovy;
    float bg[3];
    float scale_modifier;
    float view[16]; /* row-major [4][4], row-vector convention (scene/cameras.py:76) */
    float proj[16]; /* full_proj_transform (scene/cameras.py:87) */
    float campos[3];
    int sh_degree;
    int M; /* SH coefficients stored per Gaussian (16 for max degree 3) */
    unsigned flags; /* VrFlags of include/vegs_rast.h, bits 0-3: the fork assumptions of SURVEY.md A.8 as switches */
} OrCam;

#define FLAG_SCALE_MODIFIED 1u      /* cov_scale blends scale_modifier * scales                    (A-3 variant) */
#define FLAG_DEPTH_NORMALIZED 2u    /* depth = sum(w z) / (1 - T_final), 0 where nothing contributes (A-1 variant) */
#define FLAG_EXTRA_NO_ALPHA_GRAD 4u /* depth/quat/scale: gradients reach the attributes only, not alpha (A.5 variant) */
#define FLAG_FILL_EMPTY 8u          /* cov_quat += T_final * (1,0,0,0)                              (A-5 variant) */
#define FLAG_FULL_TILE_LISTS 32768u /* tile lists hold the full rectangles (VR_FLAG_FULL_TILE_LISTS); default: tight lists */

/* ------------------------------------------------------------------ math */

/* 2^x for x <= 0 from IEEE basic operations only (bit-reproducible on the GPU): x = n + f with n = rint(x), |f| <= 1/2
 * (the subtraction is exact), 2^f by a degree-5 polynomial (least-squares/minimax fit on [-1/2, 1/2] with p(0) = 1:
 * relative error 1.6e-7 evaluated in fp32), scaled by 2^n.  The compositing evaluates  exp(power) = 2^(power log2 e)
 * with the factor log2 e folded into the conic once per splat (splat_power2 below), as the reference's CUDA kernels do
 * when nvcc turns expf into ex2(x * log2 e). */
#define VR_LOG2E 1.44269504088896341f
#define VR_EXP2_C1 0.6931470036506653f
#define VR_EXP2_C2 0.24022243916988373f
#define VR_EXP2_C3 0.05550731346011162f
#define VR_EXP2_C4 0.009671415202319622f
#define VR_EXP2_C5 0.0013264892622828484f
static inline float vr_exp2(float x)
{
    if (x < -126.0f) return 0.0f;
    float n = rintf(x);
    float f = x - n;
    float p = VR_EXP2_C5;
    p = fmaf(p, f, VR_EXP2_C4);
    p = fmaf(p, f, VR_EXP2_C3);
    p = fmaf(p, f, VR_EXP2_C2);
    p = fmaf(p, f, VR_EXP2_C1);
    p = fmaf(p, f, 1.0f);
    return ldexpf(p, (int)n);
}

static inline void xform43(const float* m, const float* p, float* o)
{
    o[0] = fmaf(m[8], p[2], fmaf(m[4], p[1], fmaf(m[0], p[0], m[12])));
    o[1] = fmaf(m[9], p[2], fmaf(m[5], p[1], fmaf(m[1], p[0], m[13])));
    o[2] = fmaf(m[10], p[2], fmaf(m[6], p[1], fmaf(m[2], p[0], m[14])));
}
static inline void xform44(const float* m, const float* p, float* o)
{
    xform43(m, p, o);
    o[3] = fmaf(m[11], p[2], fmaf(m[7], p[1], fmaf(m[3], p[0], m[15])));
}

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* SH basis b[0..K) at unit direction (x,y,z); polynomials of utils/sh_utils.py:74-100 */
static inline void sh_basis(int deg, float x, float y, float z, float* b)
{
    b[0] = SH_C0;
    if (deg < 1) return;
    b[1] = -SH_C1 * y;
    b[2] = SH_C1 * z;
    b[3] = -SH_C1 * x;
    if (deg < 2) return;
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = SH_C2[0] * xy;
    b[5] = SH_C2[1] * yz;
    b[6] = SH_C2[2] * (2.0f * zz - xx - yy);
    b[7] = SH_C2[3] * xz;
    b[8] = SH_C2[4] * (xx - yy);
    if (deg < 3) return;
    b[9] = SH_C3[0] * y * (3.0f * xx - yy);
    b[10] = SH_C3[1] * xy * z;
    b[11] = SH_C3[2] * y * (4.0f * zz - xx - yy);
    b[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    b[13] = SH_C3[4] * x * (4.0f * zz - xx - yy);
    b[14] = SH_C3[5] * z * (xx - yy);
    b[15] = SH_C3[6] * x * (xx - 3.0f * yy);
}

/* d b[k] / d(x,y,z) for the backward pass */
static inline void sh_basis_grad(int deg, float x, float y, float z, float* bx, float* by, float* bz)
{
    for (int k = 0; k < 16; ++k) bx[k] = by[k] = bz[k] = 0.0f;
    if (deg < 1) return;
    by[1] = -SH_C1;
    bz[2] = SH_C1;
    bx[3] = -SH_C1;
    if (deg < 2) return;
    float xx = x * x, yy = y * y, zz = z * z;
    bx[4] = SH_C2[0] * y;  by[4] = SH_C2[0] * x;
    by[5] = SH_C2[1] * z;  bz[5] = SH_C2[1] * y;
    bx[6] = SH_C2[2] * -2.0f * x; by[6] = SH_C2[2] * -2.0f * y; bz[6] = SH_C2[2] * 4.0f * z;
    bx[7] = SH_C2[3] * z;  bz[7] = SH_C2[3] * x;
    bx[8] = SH_C2[4] * 2.0f * x; by[8] = SH_C2[4] * -2.0f * y;
    if (deg < 3) return;
    bx[9] = SH_C3[0] * 6.0f * x * y;  by[9] = SH_C3[0] * (3.0f * xx - 3.0f * yy);
    bx[10] = SH_C3[1] * y * z; by[10] = SH_C3[1] * x * z; bz[10] = SH_C3[1] * x * y;
    bx[11] = SH_C3[2] * -2.0f * x * y; by[11] = SH_C3[2] * (4.0f * zz - xx - 3.0f * yy); bz[11] = SH_C3[2] * 8.0f * y * z;
    bx[12] = SH_C3[3] * -6.0f * x * z; by[12] = SH_C3[3] * -6.0f * y * z; bz[12] = SH_C3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
    bx[13] = SH_C3[4] * (4.0f * zz - 3.0f * xx - yy); by[13] = SH_C3[4] * -2.0f * x * y; bz[13] = SH_C3[4] * 8.0f * x * z;
    bx[14] = SH_C3[5] * 2.0f * x * z; by[14] = SH_C3[5] * -2.0f * y * z; bz[14] = SH_C3[5] * (xx - yy);
    bx[15] = SH_C3[6] * (3.0f * xx - 3.0f * yy); by[15] = SH_C3[6] * -6.0f * x * y;
}

/* cov3D (6 upper-triangular floats) = R diag(mod*s)^2 R^T; R as utils/general_utils.py:97-118
 * but WITHOUT renormalising q (callers pass normalised, scene/gaussian_model.py:105-106). */
static inline void cov3d_from_scale_rot(const float* s, float mod, const float* q, float* c6)
{
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float R[9];
    R[0] = 1.0f - 2.0f * (y * y + z * z); R[1] = 2.0f * (x * y - r * z); R[2] = 2.0f * (x * z + r * y);
    R[3] = 2.0f * (x * y + r * z); R[4] = 1.0f - 2.0f * (x * x + z * z); R[5] = 2.0f * (y * z - r * x);
    R[6] = 2.0f * (x * z - r * y); R[7] = 2.0f * (y * z + r * x); R[8] = 1.0f - 2.0f * (x * x + y * y);
    float L[9];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) L[3 * i + k] = R[3 * i + k] * (mod * s[k]);
    /* Sigma = L L^T */
    c6[0] = fmaf(L[2], L[2], fmaf(L[1], L[1], L[0] * L[0]));
    c6[1] = fmaf(L[2], L[5], fmaf(L[1], L[4], L[0] * L[3]));
    c6[2] = fmaf(L[2], L[8], fmaf(L[1], L[7], L[0] * L[6]));
    c6[3] = fmaf(L[5], L[5], fmaf(L[4], L[4], L[3] * L[3]));
    c6[4] = fmaf(L[5], L[8], fmaf(L[4], L[7], L[3] * L[6]));
    c6[5] = fmaf(L[8], L[8], fmaf(L[7], L[7], L[6] * L[6]));
}

/* exported for the cov3D pin (tests/golden/ref_cov3d.npz, generated by the reference's own
 * build_scaling_rotation / strip_symmetric, utils/general_utils.py:83-129) */
void or_cov3d(int n, const float* scales, float mod, const float* rotations, float* cov6)
{
    for (int i = 0; i < n; ++i) cov3d_from_scale_rot(scales + 3 * i, mod, rotations + 4 * i, cov6 + 6 * i);
}

/* EWA projection: rows m0,m1 of M2 = J * Wview (2x3) and the 2D covariance (a,b,c), A.2 */
typedef struct { float m0[3], m1[3]; float tx, ty, tz; int clampx, clampy; float a, b, c; } Cov2D;

static inline void cov2d(const OrCam* cam, const float* t_in, const float* c6, Cov2D* o)
{
    float fx = (float)cam->W / (2.0f * cam->tanfovx);
    float fy = (float)cam->H / (2.0f * cam->tanfovy);
    float limx = 1.3f * cam->tanfovx, limy = 1.3f * cam->tanfovy;
    float tz = t_in[2];
    float txtz = t_in[0] / tz, tytz = t_in[1] / tz;
    o->clampx = (txtz < -limx) || (txtz > limx);
    o->clampy = (tytz < -limy) || (tytz > limy);
    float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    o->tx = tx; o->ty = ty; o->tz = tz;
    float j00 = fx / tz, j02 = -(fx * tx) / (tz * tz);
    float j11 = fy / tz, j12 = -(fy * ty) / (tz * tz);
    const float* v = cam->view; /* Wview[j][i] = v[4*i + j] */
    for (int i = 0; i < 3; ++i) {
        o->m0[i] = fmaf(j02, v[4 * i + 2], j00 * v[4 * i + 0]);
        o->m1[i] = fmaf(j12, v[4 * i + 2], j11 * v[4 * i + 1]);
    }
    /* u = Sigma * m0, w = Sigma * m1 */
    const float* m0 = o->m0; const float* m1 = o->m1;
    float u0 = fmaf(c6[2], m0[2], fmaf(c6[1], m0[1], c6[0] * m0[0]));
    float u1 = fmaf(c6[4], m0[2], fmaf(c6[3], m0[1], c6[1] * m0[0]));
    float u2 = fmaf(c6[5], m0[2], fmaf(c6[4], m0[1], c6[2] * m0[0]));
    float w0 = fmaf(c6[2], m1[2], fmaf(c6[1], m1[1], c6[0] * m1[0]));
    float w1 = fmaf(c6[4], m1[2], fmaf(c6[3], m1[1], c6[1] * m1[0]));
    float w2 = fmaf(c6[5], m1[2], fmaf(c6[4], m1[1], c6[2] * m1[0]));
    o->a = fmaf(m0[2], u2, fmaf(m0[1], u1, m0[0] * u0)) + 0.3f;
    o->b = fmaf(m1[2], u2, fmaf(m1[1], u1, m1[0] * u0));
    o->c = fmaf(m1[2], w2, fmaf(m1[1], w1, m1[0] * w0)) + 0.3f;
}

static inline float ndc2pix(float v, int S) { return ((v + 1.0f) * (float)S - 1.0f) * 0.5f; }

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------ preprocess */

/* ---- TIGHT TILE LISTS (vegs_amd/csrc/vr_device.h: the same functions, operation for operation).  A (Gaussian, tile) pair
 * whose footprint ellipse -- alpha >= 1/255 -- cannot reach any pixel centre of the tile is a no-op for every pixel (the
 * per-pixel rule of or_render_fwd skips it), so it is left out of the tile list; images, radii and gradients are what the
 * full rectangles give.  IEEE basic operations only. */
#define TIGHT_MAX_TILES 64
#ifndef TIGHT_BIG_CELLS
#define TIGHT_BIG_CELLS 32
#endif
#define VR_LN2 0.693147180559945309f
static float vr_ln_repro(float v)
{
    uint32_t b;
    memcpy(&b, &v, 4);
    int e = (int)(b >> 23) - 127;
    uint32_t mb = (b & 0x007FFFFFu) | 0x3F800000u;
    float m;
    memcpy(&m, &mb, 4);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float s2 = s * s;
    float p = fmaf(s2, 0.142857149f, 0.2f);
    p = fmaf(s2, p, 0.333333343f);
    p = fmaf(s2, p, 1.0f);
    return fmaf((float)e, VR_LN2, (2.0f * s) * p);
}
typedef struct { int mode; float lim, inv_A, inv_C; } TileTest;
static TileTest tile_test_setup(float A, float B, float C, float opacity)
{
    TileTest t;
    const float k = 2.0f * (vr_ln_repro(255.0f * opacity) + 0.01f);
    t.lim = k * 1.001f + 0.001f;
    t.inv_A = 1.0f / A;
    t.inv_C = 1.0f / C;
    const float det = A * C - B * B;
    t.mode = !(k > 0.0f) ? 0 : ((!(A > 0.0f) || !(C > 0.0f) || !(det > 0.0f)) ? 1 : 2);
    return t;
}
static float tile_edge_min(float a, float inv_a, float b, float c, float fixed, float lo, float hi)
{
    const float t = fminf(hi, fmaxf(lo, -b * fixed * inv_a));
    return fmaf(fmaf(a, t, 2.0f * b * fixed), t, c * fixed * fixed);
}
/* can the splat reach alpha >= 1/255 at a pixel centre of the block of ntx x nty tiles whose first tile is (tx, ty)? */
static int tile_reachable(const TileTest* t, float sx, float sy, float A, float B, float C, int tx, int ty, int ntx, int nty)
{
    if (t->mode != 2) return t->mode == 1;
    const float xl = (float)(tx * TILE) - sx, xh = xl + (float)(ntx * TILE - 1);
    const float yl = (float)(ty * TILE) - sy, yh = yl + (float)(nty * TILE - 1);
    const int in_x = xl <= 0.0f && xh >= 0.0f, in_y = yl <= 0.0f && yh >= 0.0f;
    const float fy = yl > 0.0f ? yl : yh, fx = xl > 0.0f ? xl : xh;
    /* minimum over the rectangle of pixel centres: on the boundary FACING the centre (two edges at most) */
    const float qy = tile_edge_min(A, t->inv_A, B, C, fy, xl, xh);
    const float qx = tile_edge_min(C, t->inv_C, B, A, fx, yl, yh);
    const float q = in_y ? qx : (in_x ? qy : fminf(qx, qy));
    return (in_x && in_y) || q <= t->lim;
}
/* Rectangles of more than 64 tiles are tested in CELLS of k x k tiles, k the smallest size for which the rectangle has at
 * most 32 cells (mask bit j = cell j, row-major over ceil(w / k) x ceil(h / k) cells; the cells of the last column / row may
 * be narrower): a cell none of whose pixel centres can be reached drops all its tiles.  k = 1 up to 64 tiles. */
static int tile_cell_size(int w, int h)
{
    int k = 1;
    const int limit = w * h <= TIGHT_MAX_TILES ? TIGHT_MAX_TILES : TIGHT_BIG_CELLS;
    while (((w + k - 1) / k) * ((h + k - 1) / k) > limit) ++k;
    return k;
}

/* A.2. Per Gaussian outputs (dense [P]); radii==0 marks culled/invisible. rect = xmin,ymin,xmax,ymax */
void or_preprocess(const OrCam* cam, int P, const float* means3D, const float* shs,
                   const float* colors_precomp, const float* opacities, const float* scales,
                   const float* rotations, const float* cov3D_precomp,
                   float* depth, float* xy, float* cov3D, float* conic_op, float* rgb,
                   unsigned char* clamped, int* radii, int* rect, uint32_t* tiles_touched, uint64_t* tile_mask)
{
    int gx = (cam->W + TILE - 1) / TILE, gy = (cam->H + TILE - 1) / TILE;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        radii[i] = 0; tiles_touched[i] = 0; tile_mask[i] = 0;
        depth[i] = 0.f; xy[2 * i] = xy[2 * i + 1] = 0.f;
        for (int k = 0; k < 6; ++k) cov3D[6 * i + k] = 0.f;
        for (int k = 0; k < 4; ++k) { conic_op[4 * i + k] = 0.f; rect[4 * i + k] = 0; }
        for (int k = 0; k < 3; ++k) { rgb[3 * i + k] = 0.f; clamped[3 * i + k] = 0; }
        const float* p = means3D + 3 * i;
        float t[3];
        xform43(cam->view, p, t);
        if (t[2] <= NEAR_Z) continue;
        float ph[4];
        xform44(cam->proj, p, ph);
        float pw = 1.0f / (ph[3] + 0.0000001f);
        float ndcx = ph[0] * pw, ndcy = ph[1] * pw;
        float c6[6];
        if (cov3D_precomp) memcpy(c6, cov3D_precomp + 6 * i, sizeof c6);
        else cov3d_from_scale_rot(scales + 3 * i, cam->scale_modifier, rotations + 4 * i, c6);
        Cov2D cv;
        cov2d(cam, t, c6, &cv);
        float det = cv.a * cv.c - cv.b * cv.b;
        if (det == 0.0f) continue;
        float det_inv = 1.0f / det;
        float mid = 0.5f * (cv.a + cv.c);
        float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        float lam = fmaxf(mid + sq, mid - sq);
        int rad = (int)ceilf(3.0f * sqrtf(lam));
        float px = ndc2pix(ndcx, cam->W), py = ndc2pix(ndcy, cam->H);
        int x0 = clampi((int)((px - (float)rad) / (float)TILE), 0, gx);
        int y0 = clampi((int)((py - (float)rad) / (float)TILE), 0, gy);
        int x1 = clampi((int)((px + (float)rad + (float)(TILE - 1)) / (float)TILE), 0, gx);
        int y1 = clampi((int)((py + (float)rad + (float)(TILE - 1)) / (float)TILE), 0, gy);
        if ((x1 - x0) * (y1 - y0) == 0) continue;
        if (colors_precomp) {
            for (int c = 0; c < 3; ++c) rgb[3 * i + c] = colors_precomp[3 * i + c];
        } else {
            float d[3] = {p[0] - cam->campos[0], p[1] - cam->campos[1], p[2] - cam->campos[2]};
            float len = sqrtf(fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0])));
            float dx = d[0] / len, dy = d[1] / len, dz = d[2] / len;
            float b[16];
            sh_basis(cam->sh_degree, dx, dy, dz, b);
            int K = (cam->sh_degree + 1) * (cam->sh_degree + 1);
            const float* sh = shs + (size_t)i * cam->M * 3;
            for (int c = 0; c < 3; ++c) {
                float acc = b[0] * sh[c];
                for (int k = 1; k < K; ++k) acc = fmaf(b[k], sh[3 * k + c], acc);
                acc += 0.5f;
                clamped[3 * i + c] = acc < 0.0f;
                rgb[3 * i + c] = fmaxf(acc, 0.0f);
            }
        }
        depth[i] = t[2];
        radii[i] = rad;
        xy[2 * i] = px; xy[2 * i + 1] = py;
        memcpy(cov3D + 6 * i, c6, sizeof c6);
        conic_op[4 * i + 0] = cv.c * det_inv;
        conic_op[4 * i + 1] = -cv.b * det_inv;
        conic_op[4 * i + 2] = cv.a * det_inv;
        conic_op[4 * i + 3] = opacities[i];
        rect[4 * i + 0] = x0; rect[4 * i + 1] = y0; rect[4 * i + 2] = x1; rect[4 * i + 3] = y1;
        {   /* tiles the Gaussian's list entries go to: those of the reachable cells of its rectangle */
            const int w = x1 - x0, h = y1 - y0, area = w * h;
            uint64_t mask = 0;
            if (cam->flags & FLAG_FULL_TILE_LISTS) {
                mask = area >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << area) - 1);
                tiles_touched[i] = (uint32_t)area;
            } else {
                const float A = conic_op[4 * i + 0], B = conic_op[4 * i + 1], Cc = conic_op[4 * i + 2];
                const TileTest tt = tile_test_setup(A, B, Cc, opacities[i]);
                const int k = tile_cell_size(w, h);
                int j = 0, cnt = 0;
                for (int y = y0; y < y1; y += k)
                    for (int x = x0; x < x1; x += k, ++j) {
                        const int ntx = x + k < x1 ? k : x1 - x, nty = y + k < y1 ? k : y1 - y;
                        if (tile_reachable(&tt, px, py, A, B, Cc, x, y, ntx, nty)) { mask |= (uint64_t)1 << j; cnt += ntx * nty; }
                    }
                tiles_touched[i] = (uint32_t)cnt;
            }
            tile_mask[i] = mask;
        }
    }
}

/* A.7 */
void or_mark_visible(const OrCam* cam, int P, const float* means3D, unsigned char* present)
{
    for (int i = 0; i < P; ++i) {
        float t[3];
        xform43(cam->view, means3D + 3 * i, t);
        present[i] = t[2] > NEAR_Z;
    }
}

/* --------------------------------------------------------------- binning */

typedef struct { uint64_t key; uint32_t id; } KV;
static int kv_cmp(const void* a, const void* b)
{
    const KV* x = (const KV*)a; const KV* y = (const KV*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0); /* stable: emission order = ascending id */
}

long or_count_rendered(int P, const uint32_t* tiles_touched)
{
    long R = 0;
    for (int i = 0; i < P; ++i) R += tiles_touched[i];
    return R;
}

/* A.3: keys (tile<<32 | depth bits), sorted by (tile, depth, id); ranges[2*t] = start,end */
void or_binning(const OrCam* cam, int P, const float* depth, const int* rect,
                const uint32_t* tiles_touched, const uint64_t* tile_mask, long R, uint64_t* keys, uint32_t* point_list,
                int* ranges)
{
    int gx = (cam->W + TILE - 1) / TILE, gy = (cam->H + TILE - 1) / TILE;
    KV* kv = (KV*)malloc(sizeof(KV) * (size_t)(R > 0 ? R : 1));
    long off = 0;
    for (int i = 0; i < P; ++i) {
        if (!tiles_touched[i]) continue;
        uint32_t dbits;
        memcpy(&dbits, depth + i, 4);
        const int x0 = rect[4 * i + 0], y0 = rect[4 * i + 1], x1 = rect[4 * i + 2], y1 = rect[4 * i + 3];
        const int k = tile_cell_size(x1 - x0, y1 - y0), cw = (x1 - x0 + k - 1) / k;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const int j = ((y - y0) / k) * cw + (x - x0) / k;
                if (!((tile_mask[i] >> j) & 1)) continue;      /* (tight lists: no pixel of this tile's cell can be reached) */
                kv[off].key = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dbits;
                kv[off].id = (uint32_t)i;
                ++off;
            }
    }
    qsort(kv, (size_t)R, sizeof(KV), kv_cmp);
    for (int t = 0; t < gx * gy; ++t) ranges[2 * t] = ranges[2 * t + 1] = 0;
    for (long j = 0; j < R; ++j) {
        keys[j] = kv[j].key;
        point_list[j] = kv[j].id;
        int tile = (int)(kv[j].key >> 32);
        if (j == 0 || (int)(kv[j - 1].key >> 32) != tile) ranges[2 * tile] = (int)j;
        if (j == R - 1 || (int)(kv[j + 1].key >> 32) != tile) ranges[2 * tile + 1] = (int)j + 1;
    }
    free(kv);
}

/* ---------------------------------------------------------------- render */

#define NCH 11 /* rgb(3) depth(1) quat(4) scale(3): every blended channel */

/* The Gaussian exponent at a pixel IN UNITS OF log2 e:  power2 = log2(e) * (-1/2 (A dx^2 + C dy^2) - B dx dy), evaluated
 * as  ((kA dx) dx + (kC dy) dy) + (kB dx) dy  -- the association of the textbook form (the two same-sign terms first,
 * then the cross term) -- with kA = (-1/2 log2 e) A, kB = -(log2 e) B, kC = (-1/2 log2 e) C rounded once per splat;
 * alpha = min(0.99, opacity * 2^power2).  Identical operation order in the HIP kernels (vr_device.h). */
#define VR_K_HALF (-0.5f * VR_LOG2E)
static inline float splat_power2(const float* xy, const float* con, float pxf, float pyf, float* dx, float* dy)
{
    const float kA = VR_K_HALF * con[0], kB = -VR_LOG2E * con[1], kC = VR_K_HALF * con[2];
    *dx = xy[0] - pxf;
    *dy = xy[1] - pyf;
    const float q = fmaf(kC * *dy, *dy, (kA * *dx) * *dx);
    return fmaf(kB * *dx, *dy, q);
}

static inline void splat_attrs(const OrCam* cam, int id, const float* rgb, const float* depth, const float* rotations,
                               const float* scales, float* a)
{
    const float sm = (cam->flags & FLAG_SCALE_MODIFIED) ? cam->scale_modifier : 1.0f;
    a[0] = rgb[3 * id]; a[1] = rgb[3 * id + 1]; a[2] = rgb[3 * id + 2];
    a[3] = depth[id];
    if (rotations) { for (int k = 0; k < 4; ++k) a[4 + k] = rotations[4 * id + k]; }
    else { for (int k = 0; k < 4; ++k) a[4 + k] = 0.f; }      /* A-6 */
    if (scales) { for (int k = 0; k < 3; ++k) a[8 + k] = scales[3 * id + k] * sm; }
    else { for (int k = 0; k < 3; ++k) a[8 + k] = 0.f; }
}

#define SEG 256 /* entries per compositing segment */

/* A.4 + A-1..A-5.  out_* planar [C,H,W].
 *
 * Arithmetic of the blend (part of the spec, chosen so that a massively parallel implementation
 * can reproduce it bit for bit): the tile list is cut into segments of SEG entries.  Inside a
 * segment the pixel keeps a LOCAL transmittance product p (from 1) and LOCAL channel sums Cs (from
 * 0); the transmittance in front of an entry is Tb*p with Tb the transmittance at the segment
 * start; at the segment end  C += Cs  and  Tb *= p.  Mathematically this is the usual
 * front-to-back compositing  C = sum a_i alpha_i T_i,  T_{i+1} = T_i (1 - alpha_i),  stop when
 * T_i (1 - alpha_i) < 1e-4 (that entry is not applied). */
void or_render_fwd(const OrCam* cam, const int* ranges, const uint32_t* point_list,
                   const float* xy, const float* conic_op, const float* rgb, const float* depth,
                   const float* rotations, const float* scales,
                   float* out_color, float* out_depth, float* out_quat, float* out_scale,
                   float* out_alpha, float* final_T, uint32_t* n_contrib)
{
    int H = cam->H, W = cam->W;
    int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; ++tile) {
        int tx = tile % gx, ty = tile / gx;
        int s = ranges[2 * tile], e = ranges[2 * tile + 1];
        for (int ly = 0; ly < TILE; ++ly)
            for (int lx = 0; lx < TILE; ++lx) {
                int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                float pxf = (float)px, pyf = (float)py;
                float Tb = 1.0f, C[NCH];
                for (int k = 0; k < NCH; ++k) C[k] = 0.f;
                uint32_t last = 0;
                int done = 0;
                for (int sb = s; sb < e && !done; sb += SEG) {
                    int se = sb + SEG < e ? sb + SEG : e;
                    float p = 1.0f, Cs[NCH];
                    for (int k = 0; k < NCH; ++k) Cs[k] = 0.f;
                    for (int j = sb; j < se; ++j) {
                        int id = (int)point_list[j];
                        float dx, dy;
                        float power = splat_power2(xy + 2 * id, conic_op + 4 * id, pxf, pyf, &dx, &dy);
                        if (power > 0.0f) continue;
                        float alpha = fminf(ALPHA_MAX, conic_op[4 * id + 3] * vr_exp2(power));
                        if (alpha < ALPHA_MIN) continue;
                        float pn = p * (1.0f - alpha);
                        if (Tb * pn < T_EPS) { done = 1; break; }
                        float w = alpha * (Tb * p), a[NCH];
                        splat_attrs(cam, id, rgb, depth, rotations, scales, a);
                        for (int k = 0; k < NCH; ++k) Cs[k] = fmaf(a[k], w, Cs[k]);
                        p = pn;
                        last = (uint32_t)(j - s + 1);
                    }
                    for (int k = 0; k < NCH; ++k) C[k] += Cs[k];
                    Tb = Tb * p;
                }
                float T = Tb;
                size_t pix = (size_t)py * W + px, N = (size_t)H * W;
                final_T[pix] = T;
                n_contrib[pix] = last;
                for (int c = 0; c < 3; ++c) out_color[c * N + pix] = fmaf(T, cam->bg[c], C[c]);
                float depth_out = C[3];
                if (cam->flags & FLAG_DEPTH_NORMALIZED) {
                    float A = 1.0f - T;
                    depth_out = A > 0.0f ? C[3] / A : 0.0f;
                }
                out_depth[pix] = depth_out;
                if (cam->flags & FLAG_FILL_EMPTY) C[4] += T;
                for (int k = 0; k < 4; ++k) out_quat[k * N + pix] = C[4 + k];
                for (int k = 0; k < 3; ++k) out_scale[k * N + pix] = C[8 + k];
                out_alpha[pix] = 1.0f - T;
            }
    }
}

/* A.5 (+ fork channels).  Per-fragment terms in fp32, per-Gaussian sums accumulated in
 * double so the oracle does not depend on summation order.  Outputs are per-Gaussian:
 * g_mean2D[P,2] (d/d NDC, i.e. pixel gradient * 0.5*W / 0.5*H), g_conic[P,3] (dA,dB,dC of
 * power = -0.5(A dx^2 + C dy^2) - B dx dy), g_opacity[P], g_attr[P,11]. */
void or_render_bwd(const OrCam* cam, int P, const int* ranges, const uint32_t* point_list,
                   const float* xy, const float* conic_op, const float* rgb, const float* depth,
                   const float* rotations, const float* scales,
                   const float* final_T, const uint32_t* n_contrib, const float* out_depth,
                   const float* dL_dcolor, const float* dL_ddepth, const float* dL_dquat,
                   const float* dL_dscale, const float* dL_dalpha,
                   double* g_mean2D, double* g_conic, double* g_opacity, double* g_attr, double* g_abs)
{
    /* out_depth: the forward's depth image (only read with FLAG_DEPTH_NORMALIZED).
     * g_abs (optional, [P][17]: mean2D 2 | conic 3 | opacity 1 | attr 11): sums of the ABSOLUTE values of the same
     * per-fragment terms -- the scale against which an fp32 summation of them rounds; the tests use it to measure
     * which rows of the dense gradients are ill-conditioned (oracle.preprocess_backward, perturb). */
    const int no_extra = (cam->flags & FLAG_EXTRA_NO_ALPHA_GRAD) != 0;
    const int nu = no_extra ? 3 : NCH; /* channels whose gradient also flows through alpha */
    int H = cam->H, W = cam->W;
    int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    size_t N = (size_t)H * W;
    memset(g_mean2D, 0, sizeof(double) * 2 * (size_t)P);
    memset(g_conic, 0, sizeof(double) * 3 * (size_t)P);
    memset(g_opacity, 0, sizeof(double) * (size_t)P);
    memset(g_attr, 0, sizeof(double) * NCH * (size_t)P);
    if (g_abs) memset(g_abs, 0, sizeof(double) * 17 * (size_t)P);
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; ++tile) {
        int tx = tile % gx, ty = tile / gx;
        int s = ranges[2 * tile];
        for (int ly = 0; ly < TILE; ++ly)
            for (int lx = 0; lx < TILE; ++lx) {
                int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                size_t pix = (size_t)py * W + px;
                float pxf = (float)px, pyf = (float)py;
                float g[NCH];
                for (int c = 0; c < 3; ++c) g[c] = dL_dcolor ? dL_dcolor[c * N + pix] : 0.f;
                g[3] = dL_ddepth ? dL_ddepth[pix] : 0.f;
                for (int k = 0; k < 4; ++k) g[4 + k] = dL_dquat ? dL_dquat[k * N + pix] : 0.f;
                for (int k = 0; k < 3; ++k) g[8 + k] = dL_dscale ? dL_dscale[k * N + pix] : 0.f;
                float galpha = dL_dalpha ? dL_dalpha[pix] : 0.f;
                float Tf = final_T[pix];
                float T = Tf;
                if ((cam->flags & FLAG_DEPTH_NORMALIZED) && dL_ddepth) {
                    /* depth = D / A, A = 1 - T_final: dL/dD = g/A and (through alpha) dL/dA -= g D / A^2 */
                    float A = 1.0f - Tf;
                    float inv = A > 0.0f ? 1.0f / A : 0.0f;
                    float D = out_depth[pix] * A;
                    float g3 = g[3];
                    g[3] = g3 * inv;
                    if (!no_extra) galpha -= (g3 * D) * (inv * inv);
                }
                float fill = ((cam->flags & FLAG_FILL_EMPTY) && !no_extra) ? g[4] : 0.0f;
                float bgdot = (fmaf(cam->bg[2], g[2], fmaf(cam->bg[1], g[1], cam->bg[0] * g[0])) + fill) - galpha;
                float behind = 0.f; /* sum_{j>s} w_j u_j */
                for (int j = s + (int)n_contrib[pix] - 1; j >= s; --j) {
                    int id = (int)point_list[j];
                    float dx, dy;
                    float power = splat_power2(xy + 2 * id, conic_op + 4 * id, pxf, pyf, &dx, &dy);
                    if (power > 0.0f) continue;
                    float G = vr_exp2(power);
                    float op = conic_op[4 * id + 3];
                    float alpha = fminf(ALPHA_MAX, op * G);
                    if (alpha < ALPHA_MIN) continue;
                    float oma = 1.0f - alpha;
                    T = T / oma;
                    float w = alpha * T, a[NCH];
                    splat_attrs(cam, id, rgb, depth, rotations, scales, a);
                    float u = 0.f;
                    for (int k = 0; k < NCH; ++k) {
                        if (k < nu) u = fmaf(a[k], g[k], u);
#pragma omp atomic
                        g_attr[(size_t)NCH * id + k] += (double)(w * g[k]);
                    }
                    float dL_dalpha_s = T * u - (behind + Tf * bgdot) / oma;
                    /* magnitude of the OPERANDS of that difference (g_abs below): what an fp32 evaluation of the term rounds
                     * against -- for a near-opaque splat (1 / (1 - alpha) up to 100) in front of others the two operands are
                     * each far larger than their difference */
                    const float dLda_mag = fabsf(T * u) + fabsf(behind + Tf * bgdot) / oma;
                    behind = fmaf(w, u, behind);
                    /* alpha = min(0.99, op*G): gradient passes straight through as upstream does */
                    float dL_dG = op * dL_dalpha_s;
                    float gdx = G * dx, gdy = G * dy;
                    float dG_ddx = -gdx * conic_op[4 * id + 0] - gdy * conic_op[4 * id + 1];
                    float dG_ddy = -gdy * conic_op[4 * id + 2] - gdx * conic_op[4 * id + 1];
#pragma omp atomic
                    g_mean2D[2 * (size_t)id + 0] += (double)(dL_dG * dG_ddx * (0.5f * (float)W));
#pragma omp atomic
                    g_mean2D[2 * (size_t)id + 1] += (double)(dL_dG * dG_ddy * (0.5f * (float)H));
#pragma omp atomic
                    g_conic[3 * (size_t)id + 0] += (double)(-0.5f * gdx * dx * dL_dG);
#pragma omp atomic
                    g_conic[3 * (size_t)id + 1] += (double)(-gdx * dy * dL_dG);
#pragma omp atomic
                    g_conic[3 * (size_t)id + 2] += (double)(-0.5f * gdy * dy * dL_dG);
#pragma omp atomic
                    g_opacity[id] += (double)(G * dL_dalpha_s);
                    if (g_abs) {
                        /* every term is (a coefficient) x dL/dalpha: its scale is |coefficient| x the magnitude of dL/dalpha's
                         * OPERANDS (round 6; until then |term| itself, which hides the cancellation inside dL/dalpha: fuzz seeds
                         * 11136, 17962 -- rows measured "well conditioned" that no fp32 evaluation can hold) */
                        const float mG = op * dLda_mag;
                        const float t[6] = {mG * dG_ddx * (0.5f * (float)W), mG * dG_ddy * (0.5f * (float)H),
                                            -0.5f * gdx * dx * mG, -gdx * dy * mG, -0.5f * gdy * dy * mG,
                                            G * dLda_mag};
                        for (int k = 0; k < 6; ++k) {
#pragma omp atomic
                            g_abs[17 * (size_t)id + k] += (double)fabsf(t[k]);
                        }
                        for (int k = 0; k < NCH; ++k) {
#pragma omp atomic
                            g_abs[17 * (size_t)id + 6 + k] += (double)fabsf(w * g[k]);
                        }
                    }
                }
            }
    }
}

/* A.6.  Inputs: the per-Gaussian sums of or_render_bwd (as float).  Dense outputs. */
void or_preprocess_bwd(const OrCam* cam, int P, const float* means3D, const float* shs,
                       const float* colors_precomp, const float* scales, const float* rotations,
                       const float* cov3D_precomp, const int* radii, const float* cov3D,
                       const unsigned char* clamped,
                       const float* g_mean2D, const float* g_conic, const float* g_attr,
                       float* dL_dmeans3D, float* dL_dshs, float* dL_dcolors, float* dL_dscales,
                       float* dL_drots, float* dL_dcov3D)
{
    float fx = (float)cam->W / (2.0f * cam->tanfovx);
    float fy = (float)cam->H / (2.0f * cam->tanfovy);
    const float* V = cam->view; const float* Pm = cam->proj;
    float mod = cam->scale_modifier;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = 0.f;
        if (dL_dshs) for (int k = 0; k < 3 * cam->M; ++k) dL_dshs[(size_t)i * 3 * cam->M + k] = 0.f;
        if (dL_dcolors) for (int k = 0; k < 3; ++k) dL_dcolors[3 * i + k] = 0.f;
        if (dL_dscales) for (int k = 0; k < 3; ++k) dL_dscales[3 * i + k] = 0.f;
        if (dL_drots) for (int k = 0; k < 4; ++k) dL_drots[4 * i + k] = 0.f;
        if (dL_dcov3D) for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = 0.f;
        if (radii[i] <= 0) continue;
        const float* p = means3D + 3 * i;
        const float* ga = g_attr + (size_t)NCH * i;
        float dmean[3] = {0.f, 0.f, 0.f};

        /* ---- conic -> cov2D -> cov3D, t */
        float t[3];
        xform43(V, p, t);
        const float* c6 = cov3D + 6 * i;
        Cov2D cv;
        cov2d(cam, t, c6, &cv);
        float a = cv.a, b = cv.b, c = cv.c;
        float det = a * c - b * b;
        float gA = g_conic[3 * i], gB = g_conic[3 * i + 1], gC = g_conic[3 * i + 2];
        float da = 0.f, db = 0.f, dc = 0.f;
        if (det != 0.0f) {
            float d2 = 1.0f / (det * det);
            da = d2 * (-c * c * gA + b * c * gB - b * b * gC);
            db = d2 * (2.0f * b * c * gA - (a * c + b * b) * gB + 2.0f * a * b * gC);
            dc = d2 * (-b * b * gA + a * b * gB - a * a * gC);
        }
        const float* m0 = cv.m0; const float* m1 = cv.m1;
        float gS[6]; /* d/d stored cov3D: xx xy xz yy yz zz */
        gS[0] = da * m0[0] * m0[0] + db * m0[0] * m1[0] + dc * m1[0] * m1[0];
        gS[3] = da * m0[1] * m0[1] + db * m0[1] * m1[1] + dc * m1[1] * m1[1];
        gS[5] = da * m0[2] * m0[2] + db * m0[2] * m1[2] + dc * m1[2] * m1[2];
        gS[1] = 2.f * da * m0[0] * m0[1] + db * (m0[0] * m1[1] + m0[1] * m1[0]) + 2.f * dc * m1[0] * m1[1];
        gS[2] = 2.f * da * m0[0] * m0[2] + db * (m0[0] * m1[2] + m0[2] * m1[0]) + 2.f * dc * m1[0] * m1[2];
        gS[4] = 2.f * da * m0[1] * m0[2] + db * (m0[1] * m1[2] + m0[2] * m1[1]) + 2.f * dc * m1[1] * m1[2];
        /* Sigma m0, Sigma m1 */
        float u[3] = {c6[0] * m0[0] + c6[1] * m0[1] + c6[2] * m0[2], c6[1] * m0[0] + c6[3] * m0[1] + c6[4] * m0[2],
                      c6[2] * m0[0] + c6[4] * m0[1] + c6[5] * m0[2]};
        float w[3] = {c6[0] * m1[0] + c6[1] * m1[1] + c6[2] * m1[2], c6[1] * m1[0] + c6[3] * m1[1] + c6[4] * m1[2],
                      c6[2] * m1[0] + c6[4] * m1[1] + c6[5] * m1[2]};
        float dm0[3], dm1[3];
        for (int k = 0; k < 3; ++k) {
            dm0[k] = 2.f * da * u[k] + db * w[k];
            dm1[k] = db * u[k] + 2.f * dc * w[k];
        }
        /* M2[a][i] = sum_j J[a][j] * V[4*i + j] */
        float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
        for (int k = 0; k < 3; ++k) {
            dJ00 += dm0[k] * V[4 * k + 0];
            dJ02 += dm0[k] * V[4 * k + 2];
            dJ11 += dm1[k] * V[4 * k + 1];
            dJ12 += dm1[k] * V[4 * k + 2];
        }
        float tz = cv.tz, tzi = 1.0f / tz, tzi2 = tzi * tzi, tzi3 = tzi2 * tzi;
        float dt[3];
        dt[0] = cv.clampx ? 0.f : -fx * tzi2 * dJ02;
        dt[1] = cv.clampy ? 0.f : -fy * tzi2 * dJ12;
        dt[2] = -fx * tzi2 * dJ00 - fy * tzi2 * dJ11 + 2.f * fx * cv.tx * tzi3 * dJ02 + 2.f * fy * cv.ty * tzi3 * dJ12;
        dt[2] += ga[3]; /* depth_i = t.z (fork channel A-1) */
        for (int k = 0; k < 3; ++k) dmean[k] += V[4 * k + 0] * dt[0] + V[4 * k + 1] * dt[1] + V[4 * k + 2] * dt[2];

        /* ---- mean2D (NDC gradient) -> mean3D through the perspective divide */
        float ph[4];
        xform44(Pm, p, ph);
        float mw = 1.0f / (ph[3] + 0.0000001f);
        float gx2 = g_mean2D[2 * i], gy2 = g_mean2D[2 * i + 1];
        for (int k = 0; k < 3; ++k) {
            float mul1 = (Pm[4 * k + 0] * mw - Pm[4 * k + 3] * ph[0] * mw * mw);
            float mul2 = (Pm[4 * k + 1] * mw - Pm[4 * k + 3] * ph[1] * mw * mw);
            dmean[k] += mul1 * gx2 + mul2 * gy2;
        }

        /* ---- colour */
        if (colors_precomp) {
            for (int ch = 0; ch < 3; ++ch) dL_dcolors[3 * i + ch] = ga[ch];
        } else {
            float d[3] = {p[0] - cam->campos[0], p[1] - cam->campos[1], p[2] - cam->campos[2]};
            float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            float il = 1.0f / len;
            float dir[3] = {d[0] * il, d[1] * il, d[2] * il};
            float bas[16], bx[16], by[16], bz[16];
            sh_basis(cam->sh_degree, dir[0], dir[1], dir[2], bas);
            sh_basis_grad(cam->sh_degree, dir[0], dir[1], dir[2], bx, by, bz);
            int K = (cam->sh_degree + 1) * (cam->sh_degree + 1);
            const float* sh = shs + (size_t)i * cam->M * 3;
            float* gsh = dL_dshs + (size_t)i * cam->M * 3;
            float ddir[3] = {0.f, 0.f, 0.f};
            for (int ch = 0; ch < 3; ++ch) {
                float gc = clamped[3 * i + ch] ? 0.f : ga[ch];
                for (int k = 0; k < K; ++k) {
                    gsh[3 * k + ch] = bas[k] * gc;
                    ddir[0] += bx[k] * sh[3 * k + ch] * gc;
                    ddir[1] += by[k] * sh[3 * k + ch] * gc;
                    ddir[2] += bz[k] * sh[3 * k + ch] * gc;
                }
            }
            float dot = dir[0] * ddir[0] + dir[1] * ddir[1] + dir[2] * ddir[2];
            for (int k = 0; k < 3; ++k) dmean[k] += (ddir[k] - dir[k] * dot) * il;
        }
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = dmean[k];

        /* ---- cov3D -> scale, rotation (+ direct fork channels A-2, A-3) */
        if (cov3D_precomp) {
            for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = gS[k];
        } else {
            const float* s = scales + 3 * i; const float* q = rotations + 4 * i;
            float r = q[0], x = q[1], y = q[2], z = q[3];
            float R[9];
            R[0] = 1.0f - 2.0f * (y * y + z * z); R[1] = 2.0f * (x * y - r * z); R[2] = 2.0f * (x * z + r * y);
            R[3] = 2.0f * (x * y + r * z); R[4] = 1.0f - 2.0f * (x * x + z * z); R[5] = 2.0f * (y * z - r * x);
            R[6] = 2.0f * (x * z - r * y); R[7] = 2.0f * (y * z + r * x); R[8] = 1.0f - 2.0f * (x * x + y * y);
            float Gf[9] = {gS[0], 0.5f * gS[1], 0.5f * gS[2], 0.5f * gS[1], gS[3], 0.5f * gS[4],
                           0.5f * gS[2], 0.5f * gS[4], gS[5]};
            float sp[3] = {mod * s[0], mod * s[1], mod * s[2]};
            float dLm[9]; /* d/dL, L = R*diag(sp):  2 * G * L */
            for (int ii = 0; ii < 3; ++ii)
                for (int k = 0; k < 3; ++k) {
                    float acc = 0.f;
                    for (int j = 0; j < 3; ++j) acc += Gf[3 * ii + j] * (R[3 * j + k] * sp[k]);
                    dLm[3 * ii + k] = 2.f * acc;
                }
            float D[9];
            for (int k = 0; k < 3; ++k) {
                float acc = 0.f;
                for (int ii = 0; ii < 3; ++ii) { acc += dLm[3 * ii + k] * R[3 * ii + k]; D[3 * ii + k] = dLm[3 * ii + k] * sp[k]; }
                dL_dscales[3 * i + k] = mod * acc + ((cam->flags & FLAG_SCALE_MODIFIED) ? mod : 1.0f) * ga[8 + k];
            }
            float gr = 2.f * (z * (D[3] - D[1]) + y * (D[2] - D[6]) + x * (D[7] - D[5]));
            float gxq = 2.f * (y * (D[1] + D[3]) + z * (D[2] + D[6]) + r * (D[7] - D[5])) - 4.f * x * (D[4] + D[8]);
            float gyq = 2.f * (x * (D[1] + D[3]) + r * (D[2] - D[6]) + z * (D[5] + D[7])) - 4.f * y * (D[0] + D[8]);
            float gzq = 2.f * (r * (D[3] - D[1]) + x * (D[2] + D[6]) + y * (D[5] + D[7])) - 4.f * z * (D[0] + D[4]);
            dL_drots[4 * i + 0] = gr + ga[4];
            dL_drots[4 * i + 1] = gxq + ga[5];
            dL_drots[4 * i + 2] = gyq + ga[6];
            dL_drots[4 * i + 3] = gzq + ga[7];
        }
    }
}
