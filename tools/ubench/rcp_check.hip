// rcp_check.hip -- exhaustive check (every binary32 of ordinary size) of short reciprocal / square-root sequences against
// the IEEE operators as hipcc compiles them (v_div_scale ... v_div_fixup; the sqrt expansion), on the GPU itself because
// the starting approximations (v_rcp_f32, v_rsq_f32) are the hardware's.
//   rcp1: y0 = v_rcp_f32(d); y1 = y0 + y0 * (1 - d * y0)                         (3 instructions)
//   rcp2: one more Newton step on y1                                            (5 instructions)
//   sqrt: r = v_rsq_f32(x); g = x r; h = r / 2; e = 1/2 - h g; g += g e; h += h e; s = g + (x - g g) h   (8 instructions)
// Output: mismatches per sequence over all operands with 2^-100 <= |d| <= 2^100 (sqrt: 2^-100 <= x <= 2^100).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__device__ __forceinline__ float hw_rcp(float d) { float r; asm("v_rcp_f32 %0, %1" : "=v"(r) : "v"(d)); return r; }
__device__ __forceinline__ float hw_rsq(float d) { float r; asm("v_rsq_f32 %0, %1" : "=v"(r) : "v"(d)); return r; }
__global__ void k(unsigned long long* bad, unsigned int* first)
{
    const unsigned long long n = 1ull << 32;
    unsigned long long b1 = 0, b2 = 0, b3 = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned int bits = (unsigned int)i;
        const unsigned int ex = (bits >> 23) & 255u;
        if (ex < 127u - 100u || ex > 127u + 100u)
            continue;
        const float d = __uint_as_float(bits);
        const float t = 1.0f / d;
        const float y0 = hw_rcp(d);
        const float y1 = fmaf(fmaf(-d, y0, 1.0f), y0, y0);
        const float y2 = fmaf(fmaf(-d, y1, 1.0f), y1, y1);
        if (__float_as_uint(y1) != __float_as_uint(t)) { if (b1 == 0) atomicCAS(&first[0], 0u, bits); b1++; }
        if (__float_as_uint(y2) != __float_as_uint(t)) { if (b2 == 0) atomicCAS(&first[1], 0u, bits); b2++; }
        if (d > 0) {
            const float ts = sqrtf(d);
            const float r = hw_rsq(d);
            float g = d * r, h = 0.5f * r;
            const float e = fmaf(-h, g, 0.5f);
            g = fmaf(g, e, g);
            h = fmaf(h, e, h);
            const float s = fmaf(fmaf(-g, g, d), h, g);
            if (__float_as_uint(s) != __float_as_uint(ts)) { if (b3 == 0) atomicCAS(&first[2], 0u, bits); b3++; }
        }
    }
    atomicAdd(&bad[0], b1);
    atomicAdd(&bad[1], b2);
    atomicAdd(&bad[2], b3);
}
int main()
{
    unsigned long long* d_bad; unsigned int* d_first;
    hipMalloc(&d_bad, 24); hipMalloc(&d_first, 12);
    hipMemset(d_bad, 0, 24); hipMemset(d_first, 0, 12);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d_bad, d_first);
    unsigned long long h[3]; unsigned int f[3];
    if (hipMemcpy(h, d_bad, 24, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    hipMemcpy(f, d_first, 12, hipMemcpyDeviceToHost);
    printf("mismatches over all binary32 with exponent in [-100, 100]: rcp1 %llu (first 0x%08x), rcp2 %llu (first 0x%08x), sqrt %llu (first 0x%08x)\n",
           h[0], f[0], h[1], f[1], h[2], f[2]);
    return 0;
}
