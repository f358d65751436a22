// pk_rate.hip -- microbenchmark: issue rate of v_fma_f32 against v_pk_fma_f32 on gfx950 (is a packed fp32 instruction a
// full-rate instruction, i.e. twice the flops per issue slot, or does it take two slots?), eight independent chains per lane,
// 1 ... 8 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 pk_rate.hip -o pk_rate ; run: ./pk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_exp_f32, 3: v_rcp_f32
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed)
{
    float a[8];
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; p[i] = (f2){a[i], a[i] + 0.5f}; }
    const float m = 0.999f, c = 1e-3f;
    const f2 m2 = {m, m}, c2 = {c, c};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = fmaf(a[i], m, c);
            else if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], m2, c2);
            else if (MODE == 2) a[i] = __builtin_amdgcn_exp2f(a[i] * m);
            else a[i] = __builtin_amdgcn_rcpf(a[i] + c);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int per_iter, float* out)
{
    for (int wgs_per_cu : {1, 2, 4, 8}) {
        const int wgs = 256 * wgs_per_cu, iters = 20000;
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, 8, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0f);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
        const double winstr = (double)wgs * 4 * iters * per_iter;       // wave-level instructions
        printf("%-14s %d waves/SIMD: %7.3f ms  %7.1f G wave-instr/s = %5.2f cycles per wave-instruction per SIMD (1024 SIMDs, 2.4 GHz)\n",
               name, wgs_per_cu, ms, winstr / ms / 1e6, ms * 1e-3 * 2.4e9 * 1024 / winstr);
    }
}

int main()
{
    float* out;
    CHECK(hipMalloc(&out, 256 * 256 * 8 * 4));
    run<0>("v_fma_f32", 8, out);
    run<1>("v_pk_fma_f32", 8, out);
    run<2>("v_exp_f32+mul", 16, out);
    run<3>("v_rcp_f32+add", 16, out);
    return 0;
}
