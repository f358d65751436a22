// mfma_outer.hip -- (1) register layout of v_mfma_f32_16x16x1_4b_f32 (four independent 16x16 outer products per
// instruction: block b = lanes 16b..16b+15 of A and B) found by experiment, and (2) whether a wave that is bound by
// VALU issue gets the MFMA for free (the matrix pipe runs beside the vector pipe): a loop of dependent-free
// v_pk_fma_f32 with 0, 1 or 2 MFMAs per 40 packed FMAs, all 4 waves per SIMD resident.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_outer.hip -o mfma_outer ; run: ./mfma_outer
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(64) k_layout(float* __restrict__ out)
{
    const int lane = threadIdx.x;
    f16v d = {0};
    // A[lane] = lane + 1, B[lane] = 128 * (lane + 1): D_b[i][j] = 128 (16b+i+1)(16b+j+1), exact in fp32
    d = __builtin_amdgcn_mfma_f32_16x16x1f32((float)(lane + 1), 128.0f * (float)(lane + 1), d, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[r * 64 + lane] = d[r];
}

template <int NMFMA, int PLAIN>   // PLAIN 1: 80 v_fma_f32 instead of 40 v_pk_fma_f32 (same flops) ; NMFMA 3: two MFMAs into separate accumulators
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_mix(float* __restrict__ out, int iters)
{
    const int lane = threadIdx.x;
    f2 acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = (f2){(float)k, (float)lane};
    f16v d = {0}, d2 = {0};
    f2 x = {1.0f + lane * 1e-6f, 1.0f - lane * 1e-6f};
    float w = lane * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                if (PLAIN) { acc[k].x = __builtin_fmaf(acc[k].x, x.x, x.x); acc[k].y = __builtin_fmaf(acc[k].y, x.y, x.y); }
                else acc[k] = __builtin_elementwise_fma(acc[k], x, x);
            }
            if (rep == 1 && NMFMA >= 1) d = __builtin_amdgcn_mfma_f32_16x16x1f32(w, x.x, d, 0, 0, 0);
            if (rep == 3 && NMFMA == 2) d = __builtin_amdgcn_mfma_f32_16x16x1f32(w, x.y, d, 0, 0, 0);
            if (rep == 3 && NMFMA == 3) d2 = __builtin_amdgcn_mfma_f32_16x16x1f32(w, x.y, d2, 0, 0, 0);
        }
    }
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 10; ++k) r += acc[k].x + acc[k].y;
#pragma unroll
    for (int k = 0; k < 16; ++k) r += d[k] + d2[k];
    out[blockIdx.x * 64 + lane] = r;
}

template <int NMFMA, int PLAIN>
static void run(float* out)
{
    const int wgs = 256 * 16 * 4, iters = 2000;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_mix<NMFMA, PLAIN>), dim3(wgs), dim3(64), 0, 0, out, 8);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_mix<NMFMA, PLAIN>), dim3(wgs), dim3(64), 0, 0, out, iters);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    // per SIMD: wgs/1024 waves, each iters trips
    const double trips_per_simd = (double)wgs / 1024.0 * iters;
    printf("%s + %d mfma_16x16x1_4b%s per trip: %8.3f ms = %6.1f cycles per trip per SIMD (at 2.4 GHz)\n",
           PLAIN ? "80 v_fma_f32   " : "40 v_pk_fma_f32", NMFMA == 3 ? 2 : NMFMA, NMFMA == 3 ? " (separate accumulators)" : "", ms,
           ms * 1e-3 * 2.4e9 / trips_per_simd);
}

int main()
{
    float* out;
    CHECK(hipMalloc(&out, 64 * 65536 * 4));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, out);
    CHECK(hipDeviceSynchronize());
    static float h[16 * 64];
    CHECK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
    // decode: value/128 = (e_a + 1)(e_b + 1) with both entries in the same block of 16
    int ok = 1;
    for (int r = 0; r < 16; ++r) {
        for (int l = 0; l < 64; ++l) {
            const long v = (long)(h[r * 64 + l] / 128.0f + 0.5f);
            int fa = -1, fb = -1;
            for (int b = 0; b < 4 && fa < 0; ++b)
                for (int i = 0; i < 16 && fa < 0; ++i)
                    for (int j = 0; j < 16; ++j)
                        if ((long)(16 * b + i + 1) * (16 * b + j + 1) == v && j == l % 16) { fa = 16 * b + i; fb = 16 * b + j; break; }
            // the claim tested: reg r, lane l holds block r/4, i = 4 (l/16) + r%4, j = l%16
            const int b = r / 4, i = 4 * (l / 16) + r % 4, j = l % 16;
            const long want = (long)(16 * b + i + 1) * (16 * b + j + 1);
            if (want != v) { ok = 0; printf("reg %d lane %d: value %ld decodes to A-lane %d B-lane %d, expected A-lane %d B-lane %d\n", r, l, v, fa, fb, 16 * b + i, 16 * b + j); }
        }
    }
    printf("layout D[reg r][lane l] = A[16 (r/4) + 4 (l/16) + r%%4] * B[16 (r/4) + l%%16]: %s\n", ok ? "CONFIRMED" : "WRONG");
    run<0, 0>(out); run<1, 0>(out); run<2, 0>(out); run<3, 0>(out);
    run<0, 1>(out); run<1, 1>(out); run<2, 1>(out); run<3, 1>(out);
    return 0;
}
