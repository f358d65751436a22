/* vr_harness.c -- plain-C driver of libvegsrast.so: proof that the drop-in boundary (include/vegs_rast.h)
 * needs nothing but device pointers, sizes and a stream -- no Python, no torch types.  It is what a
 * maintainer of the reference's CUDA extension would call from rasterize_points.cu's counterpart
 * (INTEGRATION.md section 2).
 *
 *   vr_harness <case.bin> <out.bin>
 *
 * case.bin (little endian): int32 header {P, M, H, W, sh_degree, has_gouts}; float {tanfovx, tanfovy,
 * scale_modifier}; float bg[3], view[16], proj[16], campos[3]; float means3D[P*3], shs[P*M*3],
 * opacities[P], scales[P*3], rotations[P*4]; if has_gouts: float dL_dcolor[3HW], dL_dquat[4HW], dL_dscale[3HW].
 * out.bin: float color[3HW], depth[HW], quat[4HW], scale[3HW], alpha[HW]; int32 radii[P]; int64 {R, V};
 * if has_gouts: float dmeans3D[3P], dmeans2D[3P], dshs[3MP], dopac[P], dscales[3P], drot[4P].
 * has_gouts == 2: both backward calls run with VR_FLAG_DETERMINISTIC, and the six arrays follow a SECOND time, after a
 * second vr_backward of the same view with VR_FLAG_ACCUMULATE_GRADS on the same arrays (ABI v9: the view's gradient is
 * added to the rows with radii > 0; dmeans2D is overwritten as always).
 * tests/test_gpu_c_harness.py writes the case, runs this program and compares with the oracle.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vegs_rast.h"

#define CHECK_HIP(e)                                                                         \
    do {                                                                                     \
        hipError_t _e = (e);                                                                 \
        if (_e != hipSuccess) {                                                              \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(_e));       \
            exit(2);                                                                         \
        }                                                                                    \
    } while (0)
#define CHECK_VR(e)                                                                          \
    do {                                                                                     \
        int _rc = (e);                                                                       \
        if (_rc != VR_OK) {                                                                  \
            fprintf(stderr, "%s:%d: libvegsrast error %d: %s\n", __FILE__, __LINE__, _rc, vr_last_error()); \
            exit(3);                                                                         \
        }                                                                                    \
    } while (0)

/* the caller-owned allocator the ABI asks for: here simply hipMalloc, everything freed at exit */
static void* alloc_cb(void* user, int kind, size_t bytes)
{
    void* p = NULL;
    (void)user;
    (void)kind;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return NULL;
    return p;
}

static float* upload(FILE* f, size_t n)
{
    float* h = (float*)malloc(n * sizeof(float) + 1);
    float* d = NULL;
    if (fread(h, sizeof(float), n, f) != n) { fprintf(stderr, "short case file\n"); exit(4); }
    CHECK_HIP(hipMalloc((void**)&d, n * sizeof(float) + 1));
    CHECK_HIP(hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice));
    free(h);
    return d;
}

static void* dev(size_t bytes)
{
    void* d = NULL;
    CHECK_HIP(hipMalloc(&d, bytes + 1));
    return d;
}

static void download(FILE* f, const void* d, size_t bytes)
{
    void* h = malloc(bytes + 1);
    CHECK_HIP(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
    fwrite(h, 1, bytes, f);
    free(h);
}

int main(int argc, char** argv)
{
    int32_t hdr[6];
    float sc[3];
    FILE *fi, *fo;
    if (argc != 3) { fprintf(stderr, "usage: %s case.bin out.bin\n", argv[0]); return 1; }
    if (vr_abi_version() != VR_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    fi = fopen(argv[1], "rb");
    if (!fi || fread(hdr, 4, 6, fi) != 6 || fread(sc, 4, 3, fi) != 3) { fprintf(stderr, "bad case file\n"); return 4; }
    {
        const int32_t P = hdr[0], M = hdr[1], H = hdr[2], W = hdr[3];
        const size_t N = (size_t)H * W;
        hipStream_t stream;
        VrSettings st;
        VrInputs in;
        VrOutputs out;
        VrSaved saved;
        CHECK_HIP(hipStreamCreate(&stream));
        memset(&st, 0, sizeof st); memset(&in, 0, sizeof in); memset(&saved, 0, sizeof saved);
        st.image_height = H; st.image_width = W; st.tanfovx = sc[0]; st.tanfovy = sc[1]; st.scale_modifier = sc[2];
        st.sh_degree = hdr[4];
        st.bg = upload(fi, 3); st.viewmatrix = upload(fi, 16); st.projmatrix = upload(fi, 16); st.campos = upload(fi, 3);
        in.P = P; in.M = M;
        in.means3D = upload(fi, (size_t)P * 3); in.shs = upload(fi, (size_t)P * M * 3); in.opacities = upload(fi, P);
        in.scales = upload(fi, (size_t)P * 3); in.rotations = upload(fi, (size_t)P * 4);
        out.color = (float*)dev(12 * N); out.depth = (float*)dev(4 * N); out.cov_quat = (float*)dev(16 * N);
        out.cov_scale = (float*)dev(12 * N); out.alpha = (float*)dev(4 * N); out.radii = (int32_t*)dev(4 * (size_t)P);
        CHECK_VR(vr_forward(&st, &in, &out, alloc_cb, NULL, stream, &saved));
        CHECK_HIP(hipStreamSynchronize(stream));
        fo = fopen(argv[2], "wb");
        download(fo, out.color, 12 * N); download(fo, out.depth, 4 * N); download(fo, out.cov_quat, 16 * N);
        download(fo, out.cov_scale, 12 * N); download(fo, out.alpha, 4 * N); download(fo, out.radii, 4 * (size_t)P);
        { int64_t rv[2]; rv[0] = saved.num_rendered; rv[1] = saved.num_visible; fwrite(rv, 8, 2, fo); }
        if (hdr[5]) {
            VrOutGrads go;
            VrInGrads gi;
            memset(&go, 0, sizeof go); memset(&gi, 0, sizeof gi);
            go.dL_dcolor = upload(fi, 3 * N); go.dL_dcov_quat = upload(fi, 4 * N); go.dL_dcov_scale = upload(fi, 3 * N);
            gi.dL_dmeans3D = (float*)dev(12 * (size_t)P); gi.dL_dmeans2D = (float*)dev(12 * (size_t)P);
            gi.dL_dshs = (float*)dev(12 * (size_t)P * M); gi.dL_dopacities = (float*)dev(4 * (size_t)P);
            gi.dL_dscales = (float*)dev(12 * (size_t)P); gi.dL_drotations = (float*)dev(16 * (size_t)P);
            {
                int pass, passes = hdr[5] == 2 ? 2 : 1;
                if (passes == 2) st.flags |= VR_FLAG_DETERMINISTIC;
                for (pass = 0; pass < passes; ++pass) {
                    if (pass == 1) st.flags |= VR_FLAG_ACCUMULATE_GRADS;     /* the same view once more, added in place */
                    CHECK_VR(vr_backward(&st, &in, out.radii, &saved, &go, &gi, alloc_cb, NULL, stream));
                    CHECK_HIP(hipStreamSynchronize(stream));
                    download(fo, gi.dL_dmeans3D, 12 * (size_t)P); download(fo, gi.dL_dmeans2D, 12 * (size_t)P);
                    download(fo, gi.dL_dshs, 12 * (size_t)P * M); download(fo, gi.dL_dopacities, 4 * (size_t)P);
                    download(fo, gi.dL_dscales, 12 * (size_t)P); download(fo, gi.dL_drotations, 16 * (size_t)P);
                }
            }
        }
        fclose(fo);
        fclose(fi);
        printf("ok P=%d R=%lld V=%lld\n", P, (long long)saved.num_rendered, (long long)saved.num_visible);
    }
    return 0;
}
