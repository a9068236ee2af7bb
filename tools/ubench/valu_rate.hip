// Micro-benchmark: cycles per wave64 VALU instruction per SIMD at 1,2,4,8 waves/SIMD (independent v_fma/v_cndmask/int ops).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND>
__global__ void k(float* out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // independent float adds
#pragma unroll
            for (int u = 0; u < 16; ++u) { a0 += 1.0f; a1 += 2.0f; a2 += 3.0f; a3 += 4.0f; a4 += 5.0f; a5 += 6.0f; a6 += 7.0f; a7 += 8.0f; }
        } else if (KIND == 1) {  // dependent chain
#pragma unroll
            for (int u = 0; u < 128; ++u) a0 = a0 * 1.0001f + 0.5f;
        } else {  // int ops + selects
#pragma unroll
            for (int u = 0; u < 32; ++u) { i0 = (i0 < i1) ? i0 + 3 : i1; i1 = (i1 ^ i2) + u; i2 = min(i2, i3) + 1; i3 = (i3 >> 1) | i0; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(i0 + i1 + i2 + i3);
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
    float* out; hipMalloc(&out, sizeof(float) * cus * 32 * 64 * 4);
    const int iters = 20000;
    for (int kind = 0; kind < 3; ++kind)
        for (int wps : {1, 2, 4, 8}) {
            int blocks = cus * 4 * wps;  // 64-thread blocks: wps waves per SIMD
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            auto launch = [&]() { if (kind == 0) k<0><<<blocks, 64>>>(out, iters); else if (kind == 1) k<1><<<blocks, 64>>>(out, iters); else k<2><<<blocks, 64>>>(out, iters); };
            launch(); hipDeviceSynchronize();
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            double instr_per_wave = (double)iters * 128;  // per loop body ~128 VALU
            double cycles = ms * 1e-3 * ghz * 1e9;
            printf("kind %d waves/SIMD %d: %.3f ms, %.2f cycles per VALU instr per SIMD (clock %.2f GHz assumed)\n", kind, wps, ms, cycles / (instr_per_wave * wps), ghz);
        }
    return 0;
}
