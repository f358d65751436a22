/* pairstats.c -- offline statistics of the backward's (list entry, pixel) contribution matrix, per
 * (tile, 256-entry segment, 8x8 region), computed from the CPU oracle's forward state (test/profiling tool).
 * Build: gcc -O2 -fopenmp -shared -fPIC -o /tmp/libpairstats.so pairstats.c -lm */
#include <math.h>
#include <stdint.h>
#include <string.h>
static inline float vr_exp(float x)
{
    float t = x * 1.44269504088896341f, n = rintf(t);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.0f / 720.0f;
    p = fmaf(p, r, 1.0f / 120.0f); p = fmaf(p, r, 1.0f / 24.0f); p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f); p = fmaf(p, r, 1.0f); p = fmaf(p, r, 1.0f);
    return x < -87.0f ? 0.0f : ldexpf(p, (int)n);
}
/* out[0]=units (seg,region with >=1 pair) out[1]=pairs out[2]=sum max_c out[3]=sum ceil(pairs/64) out[4]=sum nrel(exact)
 * out[5]=units of needed segments (incl. empty) ; hist_max[65] histogram of max_c (capped 64); out[6]=sum over units of max_c with
 * lanes sorted... ; out[7] = sum over units of trips if split in 2 half-waves (32 pixels each, max per half summed/2) */
void pair_stats(int H, int W, const int* ranges, const uint32_t* point_list, const float* xy, const float* conic_op,
                const uint32_t* n_contrib, double* out, double* hist_max, double* hist_nrel)
{
    int gx = (W + 15) / 16, gy = (H + 15) / 16;
    double o[8] = {0};
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; ++tile) {
        int tx = tile % gx, ty = tile / gx;
        int s = ranges[2 * tile], e = ranges[2 * tile + 1];
        double lo[8] = {0}, lh[65] = {0}, ln[257] = {0};
        int maxnc = 0;
        for (int ly = 0; ly < 16; ++ly) for (int lx = 0; lx < 16; ++lx) {
            int px = tx * 16 + lx, py = ty * 16 + ly;
            if (px < W && py < H && (int)n_contrib[(size_t)py * W + px] > maxnc) maxnc = n_contrib[(size_t)py * W + px];
        }
        for (int sb = s; sb < e && sb - s < maxnc; sb += 256) {
            int se = sb + 256 < e ? sb + 256 : e;
            for (int reg = 0; reg < 4; ++reg) {
                int c[64]; memset(c, 0, sizeof c);
                int nrel = 0;
                for (int j = sb; j < se; ++j) {
                    int id = point_list[j];
                    int any = 0;
                    for (int l = 0; l < 64; ++l) {
                        int px = tx * 16 + (reg % 2) * 8 + (l % 8), py = ty * 16 + (reg / 2) * 8 + (l / 8);
                        if (px >= W || py >= H) continue;
                        if (j - s >= (int)n_contrib[(size_t)py * W + px]) continue;
                        float dx = xy[2 * id] - (float)px, dy = xy[2 * id + 1] - (float)py;
                        const float* con = conic_op + 4 * id;
                        float q = fmaf(con[2] * dy, dy, (con[0] * dx) * dx);
                        float power = fmaf(-0.5f, q, -((con[1] * dx) * dy));
                        if (power > 0.0f) continue;
                        float alpha = fminf(0.99f, con[3] * vr_exp(power));
                        if (alpha < 1.0f / 255.0f) continue;
                        c[l]++; any = 1;
                    }
                    nrel += any;
                }
                int mx = 0, sum = 0, mxa = 0, mxb = 0;
                for (int l = 0; l < 64; ++l) { sum += c[l]; if (c[l] > mx) mx = c[l]; if (l < 32) { if (c[l] > mxa) mxa = c[l]; } else if (c[l] > mxb) mxb = c[l]; }
                lo[5] += 1;
                if (sum) { lo[0] += 1; lo[1] += sum; lo[2] += mx; lo[3] += (sum + 63) / 64; lo[4] += nrel; lh[mx > 64 ? 64 : mx] += 1; ln[nrel] += 1;
                           lo[7] += 0.5 * (mxa + mxb); }
            }
        }
#pragma omp critical
        { for (int k = 0; k < 8; ++k) o[k] += lo[k]; for (int k = 0; k < 65; ++k) hist_max[k] += lh[k]; for (int k = 0; k < 257; ++k) hist_nrel[k] += ln[k]; }
    }
    for (int k = 0; k < 8; ++k) out[k] = o[k];
}

/* quadrant statistics.  bwd (needed segments, entries before the pixel's last contributor): per (segment, 8x8 region) the
 * number of contributing entries n and per 4x4 quadrant n_q.  out[0] = sum 32*ceil(n/64) (pixel-pair trips now),
 * out[1] = sum 8*max_q ceil(n_q/16) (rows of 16 entries per quadrant), out[2] = sum n, out[3] = sum_q n_q,
 * fwd (all segments up to the tile's needed count / all segments): out[4] = sum n_fwd (needed segs), out[5] = sum max_q n_q fwd (needed),
 * out[6] = sum n_fwd all segments, out[7] = sum max_q n_q all segments, out[8] = sum_q n_q fwd all, out[9] = units all,
 * out[10], out[11] = the same two sums over the tiles' FIRST segments only */
void quad_stats(int H, int W, const int* ranges, const uint32_t* point_list, const float* xy, const float* conic_op,
                const uint32_t* n_contrib, double* out)
{
    int gx = (W + 15) / 16, gy = (H + 15) / 16;
    double o[12] = {0};
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; ++tile) {
        int tx = tile % gx, ty = tile / gx;
        int s = ranges[2 * tile], e = ranges[2 * tile + 1];
        double lo[12] = {0};
        int maxnc = 0;
        for (int ly = 0; ly < 16; ++ly) for (int lx = 0; lx < 16; ++lx) {
            int px = tx * 16 + lx, py = ty * 16 + ly;
            if (px < W && py < H && (int)n_contrib[(size_t)py * W + px] > maxnc) maxnc = n_contrib[(size_t)py * W + px];
        }
        for (int sb = s; sb < e; sb += 256) {
            int se = sb + 256 < e ? sb + 256 : e;
            int needed = sb - s < maxnc;
            for (int reg = 0; reg < 4; ++reg) {
                int nb = 0, nbq[4] = {0, 0, 0, 0}, nf = 0, nfq[4] = {0, 0, 0, 0};
                for (int j = sb; j < se; ++j) {
                    int id = point_list[j];
                    int anyb[4] = {0, 0, 0, 0}, anyf[4] = {0, 0, 0, 0};
                    for (int l = 0; l < 64; ++l) {
                        int lx = l % 8, ly = l / 8;
                        int px = tx * 16 + (reg % 2) * 8 + lx, py = ty * 16 + (reg / 2) * 8 + ly;
                        if (px >= W || py >= H) continue;
                        float dx = xy[2 * id] - (float)px, dy = xy[2 * id + 1] - (float)py;
                        const float* con = conic_op + 4 * id;
                        float q = fmaf(con[2] * dy, dy, (con[0] * dx) * dx);
                        float power = fmaf(-0.5f, q, -((con[1] * dx) * dy));
                        if (power > 0.0f) continue;
                        float alpha = fminf(0.99f, con[3] * vr_exp(power));
                        if (alpha < 1.0f / 255.0f) continue;
                        int qd = (ly / 4) * 2 + lx / 4;
                        anyf[qd] = 1;
                        if (j - s < (int)n_contrib[(size_t)py * W + px]) anyb[qd] = 1;
                    }
                    int ab = anyb[0] | anyb[1] | anyb[2] | anyb[3], af = anyf[0] | anyf[1] | anyf[2] | anyf[3];
                    nb += ab; nf += af;
                    for (int k = 0; k < 4; ++k) { nbq[k] += anyb[k]; nfq[k] += anyf[k]; }
                }
                int mqb = 0, mqf = 0, mcb = 0;
                for (int k = 0; k < 4; ++k) { if (nbq[k] > mqb) mqb = nbq[k]; if (nfq[k] > mqf) mqf = nfq[k]; int c = (nbq[k] + 15) / 16; if (c > mcb) mcb = c; }
                if (needed) {
                    lo[0] += 32 * ((nb + 63) / 64); lo[1] += 8 * mcb; lo[2] += nb; lo[3] += nbq[0] + nbq[1] + nbq[2] + nbq[3];
                    lo[4] += nf; lo[5] += mqf;
                }
                lo[6] += nf; lo[7] += mqf; lo[8] += nfq[0] + nfq[1] + nfq[2] + nfq[3]; lo[9] += 1;
                if (sb == s) { lo[10] += nf; lo[11] += mqf; }      /* first segments only */
            }
        }
#pragma omp critical
        { for (int k = 0; k < 12; ++k) o[k] += lo[k]; }
    }
    for (int k = 0; k < 12; ++k) out[k] = o[k];
}

/* chunk statistics of k_seg_bwd's pixel loop (needed segments only).  The kernel compacts a region's relevant entries (any
 * pixel of the region reaches alpha >= 1/255), cuts them into chunks of CH entries from the front (the last chunk is the
 * partial one) and, per chunk, walks the region's pixels in groups: a group is skipped outright when none of its pixels has
 * its last contributor behind the chunk's first entry ("cheap"), rejected after the exponent test when no entry of the
 * chunk reaches any of its pixels ("reject"), accepted otherwise.  scheme 0: CH = 64, 32 groups of 2 pixels (lanes 2pp, 2pp+1);
 * scheme 1: CH = 32, 16 groups of 4 pixels (pairs pp and pp + 16); scheme 2: CH = 32, 16 groups of 4 = pairs 2pp', 2pp'+1
 * (a 4 x 1 run).  out[3*s + 0..2] = cheap, reject, accepted trips; out[9 + s] = chunks; out[12] = units */
void chunk_stats(int H, int W, const int* ranges, const uint32_t* point_list, const float* xy, const float* conic_op,
                 const uint32_t* n_contrib, double* out)
{
    int gx = (W + 15) / 16, gy = (H + 15) / 16;
    double o[13] = {0};
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; ++tile) {
        int tx = tile % gx, ty = tile / gx;
        int s = ranges[2 * tile], e = ranges[2 * tile + 1];
        double lo[13] = {0};
        int maxnc = 0;
        for (int ly = 0; ly < 16; ++ly) for (int lx = 0; lx < 16; ++lx) {
            int px = tx * 16 + lx, py = ty * 16 + ly;
            if (px < W && py < H && (int)n_contrib[(size_t)py * W + px] > maxnc) maxnc = n_contrib[(size_t)py * W + px];
        }
        static __thread unsigned long long hit[256];   /* per relevant entry: which of the 64 pixels it contributes to (e < nc) */
        static __thread int relj[256];
        for (int sb = s; sb < e && sb - s < maxnc; sb += 256) {
            int se = sb + 256 < e ? sb + 256 : e;
            for (int reg = 0; reg < 4; ++reg) {
                int nrel = 0, nc[64], wmax = 0;
                for (int l = 0; l < 64; ++l) {
                    int px = tx * 16 + (reg % 2) * 8 + (l % 8), py = ty * 16 + (reg / 2) * 8 + (l / 8);
                    nc[l] = (px < W && py < H) ? (int)n_contrib[(size_t)py * W + px] : 0;
                    if (nc[l] > wmax) wmax = nc[l];
                }
                for (int j = sb; j < se; ++j) {
                    int id = point_list[j];
                    unsigned long long h = 0; int any = 0;
                    for (int l = 0; l < 64; ++l) {
                        int px = tx * 16 + (reg % 2) * 8 + (l % 8), py = ty * 16 + (reg / 2) * 8 + (l / 8);
                        if (px >= W || py >= H) continue;
                        float dx = xy[2 * id] - (float)px, dy = xy[2 * id + 1] - (float)py;
                        const float* con = conic_op + 4 * id;
                        float q = fmaf(con[2] * dy, dy, (con[0] * dx) * dx);
                        float power = fmaf(-0.5f, q, -((con[1] * dx) * dy));
                        if (power > 0.0f) continue;
                        float alpha = fminf(0.99f, con[3] * vr_exp(power));
                        if (alpha < 1.0f / 255.0f) continue;
                        any = 1;
                        if (j - s < nc[l]) h |= 1ull << l;
                    }
                    if (any) { hit[nrel] = h; relj[nrel] = j - s; nrel++; }
                }
                if (!nrel) continue;
                lo[12] += 1;
                for (int sch = 0; sch < 3; ++sch) {
                    int CH = sch == 0 ? 64 : 32, ngroups = sch == 0 ? 32 : 16;
                    for (int c0 = 0; c0 < nrel; c0 += CH) {
                        int c1 = c0 + CH < nrel ? c0 + CH : nrel;
                        int chunk_lo = relj[c0];
                        lo[9 + sch] += 1;
                        if (!(chunk_lo < wmax)) continue;
                        unsigned long long all = 0;
                        for (int r = c0; r < c1; ++r) all |= hit[r];
                        for (int g = 0; g < ngroups; ++g) {
                            unsigned long long pm;
                            if (sch == 0) pm = 3ull << (2 * g);
                            else if (sch == 1) pm = (3ull << (2 * g)) | (3ull << (2 * (g + 16)));
                            else pm = 15ull << (4 * g);
                            int mnc = 0;
                            for (int l = 0; l < 64; ++l) if ((pm >> l) & 1ull) if (nc[l] > mnc) mnc = nc[l];
                            if (mnc <= chunk_lo) lo[3 * sch + 0] += 1;
                            else if (!(all & pm)) lo[3 * sch + 1] += 1;
                            else lo[3 * sch + 2] += 1;
                        }
                    }
                }
            }
        }
#pragma omp critical
        { for (int k = 0; k < 13; ++k) o[k] += lo[k]; }
    }
    for (int k = 0; k < 13; ++k) out[k] = o[k];
}

/* tail statistics (round 6): scheme 0's LAST chunk of a unit holds n = nrel - 64*floor((nrel-1)/64) entries; when n <= 16 the
 * wave could hold the chunk once per 16-lane row and let every row walk another pixel pair (8 trips of 4 pairs), when n <= 32
 * once per half (16 trips of 2 pairs).  out[0..2] cheap/reject/accepted trips of ALL chunks now; out[3..5] the same with tails
 * <= 16 row-packed; out[6..8] with tails <= 16 row-packed and tails <= 32 half-packed; out[9] chunks, out[10] tails <= 16,
 * out[11] tails 17..32, out[12] tails 33..48, out[13] tails 49..64 (incl. full last chunks), out[14] units,
 * out[15] accepted trips now that belong to tails <= 16, out[16] to tails 17..32 */
void tail_stats(int H, int W, const int* ranges, const uint32_t* point_list, const float* xy, const float* conic_op,
                const uint32_t* n_contrib, double* out)
{
    int gx = (W + 15) / 16, gy = (H + 15) / 16;
    double o[27] = {0};
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; ++tile) {
        int tx = tile % gx, ty = tile / gx;
        int s = ranges[2 * tile], e = ranges[2 * tile + 1];
        double lo[27] = {0};
        int maxnc = 0;
        for (int ly = 0; ly < 16; ++ly) for (int lx = 0; lx < 16; ++lx) {
            int px = tx * 16 + lx, py = ty * 16 + ly;
            if (px < W && py < H && (int)n_contrib[(size_t)py * W + px] > maxnc) maxnc = n_contrib[(size_t)py * W + px];
        }
        static __thread unsigned long long hit[256];
        static __thread int relj[256];
        for (int sb = s; sb < e && sb - s < maxnc; sb += 256) {
            int se = sb + 256 < e ? sb + 256 : e;
            for (int reg = 0; reg < 4; ++reg) {
                int nrel = 0, nc[64], wmax = 0;
                for (int l = 0; l < 64; ++l) {
                    int px = tx * 16 + (reg % 2) * 8 + (l % 8), py = ty * 16 + (reg / 2) * 8 + (l / 8);
                    nc[l] = (px < W && py < H) ? (int)n_contrib[(size_t)py * W + px] : 0;
                    if (nc[l] > wmax) wmax = nc[l];
                }
                for (int j = sb; j < se; ++j) {
                    int id = point_list[j];
                    unsigned long long h = 0; int any = 0;
                    for (int l = 0; l < 64; ++l) {
                        int px = tx * 16 + (reg % 2) * 8 + (l % 8), py = ty * 16 + (reg / 2) * 8 + (l / 8);
                        if (px >= W || py >= H) continue;
                        float dx = xy[2 * id] - (float)px, dy = xy[2 * id + 1] - (float)py;
                        const float* con = conic_op + 4 * id;
                        float q = fmaf(con[2] * dy, dy, (con[0] * dx) * dx);
                        float power = fmaf(-0.5f, q, -((con[1] * dx) * dy));
                        if (power > 0.0f) continue;
                        float alpha = fminf(0.99f, con[3] * vr_exp(power));
                        if (alpha < 1.0f / 255.0f) continue;
                        any = 1;
                        if (j - s < nc[l]) h |= 1ull << l;
                    }
                    if (any) { hit[nrel] = h; relj[nrel] = j - s; nrel++; }
                }
                if (!nrel) continue;
                lo[14] += 1;
                /* entries behind EVERY pixel's last contributor dropped before chunking (relj >= wmax), then modes 2's packing */
                int ncut = 0;
                while (ncut < nrel && relj[ncut] < wmax) ++ncut;
                lo[20] += nrel; lo[21] += ncut;
                for (int c0 = 0; c0 < ncut; c0 += 64) {
                    int c1 = c0 + 64 < ncut ? c0 + 64 : ncut;
                    int n = c1 - c0, tail = c1 == ncut;
                    int chunk_lo = relj[c0];
                    lo[22] += 1;
                    unsigned long long all = 0;
                    for (int r = c0; r < c1; ++r) all |= hit[r];
                    int per = tail && n <= 16 ? 4 : tail && n <= 32 ? 2 : 1;
                    for (int g = 0; g < 32 / per; ++g) {
                        unsigned long long pm = 0;
                        for (int k = 0; k < per; ++k) pm |= 3ull << (2 * (g * per + k));
                        int mnc = 0;
                        for (int l = 0; l < 64; ++l) if ((pm >> l) & 1ull) if (nc[l] > mnc) mnc = nc[l];
                        int cls = mnc <= chunk_lo ? 0 : !(all & pm) ? 1 : 2;
                        lo[17 + cls] += 1;
                    }
                    /* ... and a tail of 33 .. 48 entries as a half-packed chunk of 32 + a row-packed rest (out[23..25], out[26] chunks) */
                    if (tail && n > 32 && n <= 48) {
                        for (int part = 0; part < 2; ++part) {
                            int p0 = part == 0 ? c0 : c0 + 32, p1 = part == 0 ? c0 + 32 : c1;
                            int plo = relj[p0], pper = part == 0 ? 2 : 4;
                            unsigned long long pall = 0;
                            for (int r = p0; r < p1; ++r) pall |= hit[r];
                            lo[26] += 1;
                            for (int g = 0; g < 32 / pper; ++g) {
                                unsigned long long pm = 0;
                                for (int k = 0; k < pper; ++k) pm |= 3ull << (2 * (g * pper + k));
                                int mnc = 0;
                                for (int l = 0; l < 64; ++l) if ((pm >> l) & 1ull) if (nc[l] > mnc) mnc = nc[l];
                                int cls = mnc <= plo ? 0 : !(pall & pm) ? 1 : 2;
                                lo[23 + cls] += 1;
                            }
                        }
                    } else {
                        lo[26] += 1;
                        for (int g = 0; g < 32 / per; ++g) {
                            unsigned long long pm = 0;
                            for (int k = 0; k < per; ++k) pm |= 3ull << (2 * (g * per + k));
                            int mnc = 0;
                            for (int l = 0; l < 64; ++l) if ((pm >> l) & 1ull) if (nc[l] > mnc) mnc = nc[l];
                            int cls = mnc <= chunk_lo ? 0 : !(all & pm) ? 1 : 2;
                            lo[23 + cls] += 1;
                        }
                    }
                }
                for (int c0 = 0; c0 < nrel; c0 += 64) {
                    int c1 = c0 + 64 < nrel ? c0 + 64 : nrel;
                    int n = c1 - c0, tail = c1 == nrel;
                    int chunk_lo = relj[c0];
                    lo[9] += 1;
                    if (tail) lo[10 + (n - 1) / 16] += 1;
                    if (!(chunk_lo < wmax)) continue;
                    unsigned long long all = 0;
                    for (int r = c0; r < c1; ++r) all |= hit[r];
                    /* trips of this chunk under a grouping of `per` pairs per trip: pairs g*per .. g*per+per-1 */
                    for (int mode = 0; mode < 3; ++mode) {
                        int per = 1;
                        if (tail && n <= 16 && mode >= 1) per = 4;
                        else if (tail && n <= 32 && mode >= 2) per = 2;
                        for (int g = 0; g < 32 / per; ++g) {
                            unsigned long long pm = 0;
                            for (int k = 0; k < per; ++k) pm |= 3ull << (2 * (g * per + k));
                            int mnc = 0;
                            for (int l = 0; l < 64; ++l) if ((pm >> l) & 1ull) if (nc[l] > mnc) mnc = nc[l];
                            int cls = mnc <= chunk_lo ? 0 : !(all & pm) ? 1 : 2;
                            lo[3 * mode + cls] += 1;
                            if (mode == 0 && cls == 2 && tail && n <= 16) lo[15] += 1;
                            if (mode == 0 && cls == 2 && tail && n > 16 && n <= 32) lo[16] += 1;
                        }
                    }
                }
            }
        }
#pragma omp critical
        { for (int k = 0; k < 27; ++k) o[k] += lo[k]; }
    }
    for (int k = 0; k < 27; ++k) out[k] = o[k];
}
