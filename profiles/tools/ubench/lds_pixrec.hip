// lds_pixrec.hip -- microbenchmark for VERDICT r04 item 6: where do k_seg_bwd's LDS bank conflicts come from
// (SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = 0.71)?  Candidates, each as its own kernel with the exact access pattern:
//   bcast128   the pixel loop's 8 x ds_read_b128 per trip, every lane the SAME address (pixrec[pp][0..7])
//   fill_pair  the workgroup's prologue as it was: lane = pixel writes its 16 scalars into pixrec[lane >> 1][.] + (lane & 1),
//              i.e. 16 x ds_write_b32 at a lane stride of 32 floats per PAIR -- 32 lanes per bank
//   fill_slot  the round-5 prologue: even lanes write whole float4s {q_a q_b q'_a q'_b} into pix[slot][lane >> 1] (the odd
//              pixel's values come over DPP): 8 x ds_write_b128, 32 consecutive addresses
//   stage17    the flush: ds_write_b32 at lane stride 17 floats, then ds_read_b32 at consecutive addresses
// Build: hipcc -O3 --offload-arch=gfx950 lds_pixrec.hip -o lds_pixrec ; run: ./lds_pixrec
// Counters: rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --stats -d out -- ./lds_pixrec
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(64) k_bcast128(float* __restrict__ out, int iters)
{
    __shared__ float4 pixrec[32][8];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) reinterpret_cast<float4*>(pixrec)[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int pp = it & 31;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 r = pixrec[pp][k];
            s += r.x + r.y + r.z + r.w;
        }
    }
    out[blockIdx.x * 64 + lane] = s;
}

__global__ void __launch_bounds__(64) k_fill_pair(float* __restrict__ out, int iters)
{
    __shared__ float4 pixrec[32][8];
    const int lane = threadIdx.x;
    float v = lane * 0.5f, s = 0.f;
    for (int it = 0; it < iters; ++it) {
        float* rec2 = reinterpret_cast<float*>(&pixrec[lane >> 1][0]) + (lane & 1);
#pragma unroll
        for (int k = 0; k < 16; ++k) rec2[2 * k] = v + k;
        __syncthreads();
        s += pixrec[it & 31][it & 7].x;
        __syncthreads();
        v += 1.f;
    }
    out[blockIdx.x * 64 + lane] = s;
}

__global__ void __launch_bounds__(64) k_fill_slot(float* __restrict__ out, int iters)
{
    __shared__ float4 pix[8][32];
    const int lane = threadIdx.x;
    float v = lane * 0.5f, s = 0.f;
    for (int it = 0; it < iters; ++it) {
        float mine[16], other[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { mine[k] = v + k; other[k] = __shfl_xor(mine[k], 1, 64); }
        if ((lane & 1) == 0) {
#pragma unroll
            for (int sl = 0; sl < 8; ++sl)
                pix[sl][lane >> 1] = make_float4(mine[2 * sl], other[2 * sl], mine[2 * sl + 1], other[2 * sl + 1]);
        }
        __syncthreads();
        s += pix[it & 7][it & 31].x;
        __syncthreads();
        v += 1.f;
    }
    out[blockIdx.x * 64 + lane] = s;
}

__global__ void __launch_bounds__(64) k_stage17(float* __restrict__ out, int iters)
{
    __shared__ float stage[64 * 17];
    const int lane = threadIdx.x;
    float v = lane * 0.5f, s = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 17; ++k) stage[lane * 17 + k] = v + k;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 17; ++k) s += stage[lane + 64 * k];
        __syncthreads();
        v += 1.f;
    }
    out[blockIdx.x * 64 + lane] = s;
}

template <class K>
static void run(const char* name, K kern, int per_iter, float* out)
{
    const int wgs = 256 * 32, iters = 2000;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(64), 0, 0, out, 8);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(64), 0, 0, out, iters);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    const double instr = (double)wgs * iters * per_iter;
    printf("%-10s %8.3f ms  %6.2f cycles per LDS wave-instruction per CU (256 CUs at 2.4 GHz, %d LDS instructions per iteration)\n",
           name, ms, ms * 1e-3 * 2.4e9 * 256 / instr, per_iter);
}

int main()
{
    float* out;
    CHECK(hipMalloc(&out, 64 * 256 * 32 * 4));
    run("bcast128", k_bcast128, 8, out);
    run("fill_pair", k_fill_pair, 17, out);
    run("fill_slot", k_fill_slot, 9, out);
    run("stage17", k_stage17, 34, out);
    return 0;
}
