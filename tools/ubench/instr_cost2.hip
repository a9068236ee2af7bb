// instr_cost2.hip -- the broader table behind instr_cost.hip: cycles per wave64 instruction per SIMD for ~85 instruction forms
// (gen_instr_cost.py writes instr_cost2_streams.inc), wall clock x in-kernel clock, waves per SIMD forced by LDS size.
// Build: python3 gen_instr_cost.py > instr_cost2_streams.inc && hipcc -O3 --offload-arch=gfx950 -o instr_cost2 instr_cost2.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#include "instr_cost2_streams.inc"

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define CLOB "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37"

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned long long* stamps, float* sink, int iters, float scal)
{
    extern __shared__ unsigned lds[];
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double p0 = a0, p1 = a1, p2 = a2, p3 = a3, p4 = a4, p5 = a5, p6 = a6, p7 = a7;
    float b = 1.0001f + (float)(threadIdx.x & 1);
    const unsigned la = (threadIdx.x & 63) * 4;
    for (unsigned i = threadIdx.x; i < 1024; i += 256)
        lds[i] = (i * 4) & 255;
    __syncthreads();
    asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 s[22:23], 0\n s_mov_b64 s[24:25], exec\n s_mov_b64 s[26:27], 0\n"
                 "s_mov_b64 s[28:29], exec\n s_mov_b64 s[30:31], 0\n s_mov_b64 s[32:33], exec\n s_mov_b64 s[34:35], 0\n"
                 "s_mov_b64 s[36:37], exec\n s_mov_b64 vcc, exec" ::
                     : CLOB, "vcc");
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int i = 0; i < iters; ++i) {
#define X(N, NAME, CNT, TEXT)                                                                                                     \
    if (KIND == N) {                                                                                                              \
        REP16(asm volatile(TEXT                                                                                                   \
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0), "+v"(p1),  \
                             "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)                                           \
                           : "v"(b), "s"(scal), "v"(la)                                                                           \
                           : CLOB, "vcc", "scc", "memory");)                                                                      \
    }
        STREAMS(X)
#undef X
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
    sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7);
}

template <int KIND>
static void launch(int blocks, size_t lds, unsigned long long* st, float* sink, int iters)
{
    if (hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        printf("# hipFuncSetAttribute(%zu) failed\n", lds);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), lds, 0, st, sink, iters, 1.5f);
}

int main(int argc, char** argv)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess)
        return 1;
    const int cus = p.multiProcessorCount;
    unsigned long long* d_st;
    float* d_sink;
    if (hipMalloc(&d_st, sizeof(unsigned long long) * 2 * cus * 8) != hipSuccess ||
        hipMalloc(&d_sink, sizeof(float) * cus * 8 * 256) != hipSuccess)
        return 1;
    const int iters = 3000;
    printf("# %s, %d CUs; 256-thread workgroups, W per CU forced by LDS\n", p.name, cus);
    printf("%-52s %3s %9s %10s %15s\n", "stream", "W", "ms", "clock GHz", "cyc/instr/SIMD");
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[] = {
#define X(N, NAME, CNT, TEXT) NAME,
        STREAMS(X)
#undef X
    };
    const int counts[] = {
#define X(N, NAME, CNT, TEXT) CNT,
        STREAMS(X)
#undef X
    };
    const int nk = sizeof(names) / sizeof(names[0]);
    for (int kind = 0; kind < nk; ++kind)
        for (int w : {1, 4, 6}) {
            const int blocks = cus * w;
            size_t lds = (size_t)(160 * 1024 / w) & ~(size_t)1023;
            if (w == 1)
                lds = 96 * 1024;
            auto go = [&]() {
                switch (kind) {
#define X(N, NAME, CNT, TEXT) case N: launch<N>(blocks, lds, d_st, d_sink, iters); break;
                    STREAMS(X)
#undef X
                }
            };
            go();
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, 0);
            go();
            (void)hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess)
                return 2;
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(2 * blocks);
            (void)hipMemcpy(h.data(), d_st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
            std::vector<double> ghz(blocks);
            for (int bb = 0; bb < blocks; ++bb)
                ghz[bb] = (double)h[2 * bb] / (double)h[2 * bb + 1] * 0.1;
            std::sort(ghz.begin(), ghz.end());
            const double clk = ghz[blocks / 2];
            const double n = (double)iters * 16 * counts[kind] * w;
            printf("%-52s %3d %9.3f %10.3f %15.2f\n", names[kind], w, ms, clk, ms * 1e-3 * clk * 1e9 / n);
            fflush(stdout);
        }
    return 0;
}
