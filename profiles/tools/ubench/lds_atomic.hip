// lds_atomic.hip -- microbenchmark: throughput of ds_add_f32 (no return) vs ds_write_b32 vs v_fma on gfx950,
// with the access pattern the render backward would use (lane -> acc[entry][17], entries random per lane).
// Build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics lds_atomic.hip -o lds_atomic ; run: ./lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int NE>   // MODE 0: ds_add_f32, 1: ds_write_b32, 2: v_fma only ; NE = distinct entries per wave-instruction pattern
__global__ void __launch_bounds__(64) k(const int* __restrict__ ent, float* __restrict__ out, int iters)
{
    __shared__ float acc[64 * 17];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 17; i += 64) acc[i] = 0.f;
    __syncthreads();
    int e = ent[(blockIdx.x * 64 + lane) % 4096] % NE;
    float v = 1.0f + lane * 1e-3f, s = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k2 = 0; k2 < 17; ++k2) {
            if (MODE == 0) atomicAdd(&acc[e * 17 + k2], v);
            else if (MODE == 1) acc[e * 17 + k2] = v;
            else s = fmaf(s, v, 1.0f);
        }
        e = (e * 5 + 3) % NE;   // next pseudo-random entry (cheap VALU)
        v += 1e-6f;
    }
    __syncthreads();
    float r = s;
    for (int i = lane; i < 64 * 17; i += 64) r += acc[i];
    out[blockIdx.x * 64 + lane] = r;
}

template <int MODE, int NE>
static void run(const char* name, int wgs, int iters, const int* ent, float* out)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE, NE>), dim3(wgs), dim3(64), 0, 0, ent, out, 8);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE, NE>), dim3(wgs), dim3(64), 0, 0, ent, out, iters);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    double instr = (double)wgs * iters * 17;
    // cycles per wave-instruction per CU, assuming 256 CUs at 2.4 GHz all busy
    printf("%-34s wgs %6d: %8.3f ms  %7.2f G wave-instr/s  = %5.2f cycles per wave-instr per CU\n", name, wgs, ms,
           instr / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / instr);
}

int main()
{
    int* ent; float* out;
    int h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = rand();
    CHECK(hipMalloc(&ent, sizeof h)); CHECK(hipMemcpy(ent, h, sizeof h, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, 64 * 65536 * 4));
    for (int wgs : {256 * 4, 256 * 16, 256 * 32}) {
        run<0, 64>("ds_add_f32, 64 distinct entries", wgs, 2000, ent, out);
        run<0, 16>("ds_add_f32, 16 distinct entries", wgs, 2000, ent, out);
        run<0, 4>("ds_add_f32, 4 distinct entries", wgs, 2000, ent, out);
        run<0, 1>("ds_add_f32, 1 entry (same addr)", wgs, 2000, ent, out);
        run<1, 64>("ds_write_b32, 64 entries", wgs, 2000, ent, out);
        run<2, 64>("v_fma_f32 dependent chain", wgs, 2000, ent, out);
    }
    return 0;
}
