// issue_rate2.hip -- what ONE SIMD of gfx950 issues, timed by the wall clock (HIP events) and checked against the in-kernel
// clock (s_memtime / s_memrealtime), with the waves per SIMD FORCED: 256-thread workgroups (one wavefront on each of a CU's
// four SIMDs), W workgroups per CU made exact by the LDS each one allocates (160 KiB / W), grid = CUs x W.  issue_rate.hip
// trusted the dispatcher to spread 64-thread workgroups evenly and took a median of per-workgroup cycle counts; its figures
// below 2 cycles per wave64 vector instruction per SIMD contradict the SIMD-32 pipe, so this file re-measures the ceilings.
// Streams (pinned by inline asm):
//   valu        8 independent v_fma_f32 chains
//   salu        8 independent s_add_u32
//   mix52       5 VALU : 2 SALU, independent
//   mix21       2 VALU : 1 SALU, independent (the render kernel's ratio)
//   mask        v_cmp -> s_and/s_or -> v_cndmask (the wave-mask idiom, dependent across units)
//   mask_far    the same instructions, software-pipelined: every consumer sits >= 8 instructions behind its producer
//   cmp_sgpr    v_cmp to SGPR pair + independent v_cndmask reading an OLD SGPR pair (no dependence)
// Output per stream and W: ms, instructions per SIMD, cycles per instruction per SIMD by the wall clock at the clock the
// kernel itself measured.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned long long* stamps, float* sink, int iters)
{
    extern __shared__ unsigned lds[];
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, a8 = a0 + 8;
    const float m = 1.0001f, c = 0.5f;
    unsigned s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // 128 VALU
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\t"
                               "v_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\t"
                               "v_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                               : "v"(m), "v"(c));)
        } else if (KIND == 1) {  // 128 SALU
            REP16(asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 2\n\ts_add_u32 %2, %2, 3\n\ts_add_u32 %3, %3, 4\n\t"
                               "s_add_u32 %4, %4, 5\n\ts_add_u32 %5, %5, 6\n\ts_add_u32 %6, %6, 7\n\ts_add_u32 %7, %7, 8"
                               : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)::"scc");)
        } else if (KIND == 2) {  // 16 x (5 VALU + 2 SALU) = 112
            REP16(asm volatile("v_fma_f32 %0, %0, %7, %8\n\tv_fma_f32 %1, %1, %7, %8\n\ts_add_u32 %5, %5, 1\n\t"
                               "v_fma_f32 %2, %2, %7, %8\n\tv_fma_f32 %3, %3, %7, %8\n\ts_add_u32 %6, %6, 2\n\t"
                               "v_fma_f32 %4, %4, %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+s"(s0), "+s"(s1)
                               : "v"(m), "v"(c)
                               : "scc");)
        } else if (KIND == 3) {  // 16 x (4 VALU + 2 SALU) = 96
            REP16(asm volatile("v_fma_f32 %0, %0, %6, %7\n\tv_fma_f32 %1, %1, %6, %7\n\ts_add_u32 %4, %4, 1\n\t"
                               "v_fma_f32 %2, %2, %6, %7\n\tv_fma_f32 %3, %3, %6, %7\n\ts_add_u32 %5, %5, 2"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1)
                               : "v"(m), "v"(c)
                               : "scc");)
        } else if (KIND == 4) {  // 16 x (3 v_cmp + 3 SALU + 2 v_cndmask) = 128, dependent across units
            REP16(asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cmp_lt_f32 s[22:23], %1, %2\n\tv_cmp_lt_f32 s[24:25], %2, %0\n\t"
                               "s_and_b64 s[26:27], s[20:21], s[22:23]\n\ts_andn2_b64 s[28:29], s[24:25], s[20:21]\n\t"
                               "s_or_b64 s[30:31], s[26:27], s[28:29]\n\t"
                               "v_cndmask_b32 %0, %0, %2, s[26:27]\n\tv_cndmask_b32 %1, %1, %0, s[30:31]"
                               : "+v"(a0), "+v"(a1), "+v"(a2)::"s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28",
                                 "s29", "s30", "s31", "scc");)
        } else if (KIND == 5) {  // the same 128 instructions over three register sets, consumers far behind their producers
            REP16(asm volatile(
                "v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cmp_lt_f32 s[22:23], %1, %2\n\tv_cmp_lt_f32 s[24:25], %2, %0\n\t"
                "s_and_b64 s[46:47], s[40:41], s[42:43]\n\ts_andn2_b64 s[48:49], s[44:45], s[40:41]\n\t"
                "v_cndmask_b32 %6, %6, %8, s[66:67]\n\tv_cndmask_b32 %7, %7, %6, s[70:71]\n\t"
                "s_or_b64 s[50:51], s[46:47], s[48:49]\n\t"
                "v_cmp_lt_f32 s[60:61], %6, %7\n\tv_cmp_lt_f32 s[62:63], %7, %8\n\tv_cmp_lt_f32 s[64:65], %8, %6\n\t"
                "s_and_b64 s[26:27], s[20:21], s[22:23]\n\ts_andn2_b64 s[28:29], s[24:25], s[20:21]\n\t"
                "v_cndmask_b32 %3, %3, %5, s[46:47]\n\tv_cndmask_b32 %4, %4, %3, s[50:51]\n\t"
                "s_or_b64 s[30:31], s[26:27], s[28:29]\n\t"
                "v_cmp_lt_f32 s[40:41], %3, %4\n\tv_cmp_lt_f32 s[42:43], %4, %5\n\tv_cmp_lt_f32 s[44:45], %5, %3\n\t"
                "s_and_b64 s[66:67], s[60:61], s[62:63]\n\ts_andn2_b64 s[68:69], s[64:65], s[60:61]\n\t"
                "v_cndmask_b32 %0, %0, %2, s[26:27]\n\tv_cndmask_b32 %1, %1, %0, s[30:31]\n\t"
                "s_or_b64 s[70:71], s[66:67], s[68:69]"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8)
                :
                : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s40", "s41", "s42",
                  "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s60", "s61", "s62", "s63", "s64", "s65",
                  "s66", "s67", "s68", "s69", "s70", "s71", "scc");)
        } else {  // 16 x (4 v_cmp to SGPR pairs + 4 v_cndmask on a pair written long ago) = 128, no dependence across units
            REP16(asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cndmask_b32 %4, %4, %0, s[30:31]\n\t"
                               "v_cmp_lt_f32 s[22:23], %1, %2\n\tv_cndmask_b32 %5, %5, %1, s[30:31]\n\t"
                               "v_cmp_lt_f32 s[24:25], %2, %3\n\tv_cndmask_b32 %6, %6, %2, s[30:31]\n\t"
                               "v_cmp_lt_f32 s[26:27], %3, %0\n\tv_cndmask_b32 %7, %7, %3, s[30:31]"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)::"s20", "s21",
                                 "s22", "s23", "s24", "s25", "s26", "s27");)
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
        lds[0] = 0;
    }
    sink[blockIdx.x * 256 + threadIdx.x] =
        a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7) + (float)lds[0];
}

template <int KIND>
static void launch(int blocks, size_t lds, unsigned long long* st, float* sink, int iters)
{
    if (hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        printf("# hipFuncSetAttribute(%zu) failed\n", lds);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), lds, 0, st, sink, iters);
}

int main()
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess)
        return 1;
    const int cus = p.multiProcessorCount;
    unsigned long long* d_st;
    float* d_sink;
    hipMalloc(&d_st, sizeof(unsigned long long) * 2 * cus * 8);
    hipMalloc(&d_sink, sizeof(float) * cus * 8 * 256);
    const int iters = 20000;
    const char* names[7] = {"valu", "salu", "mix52", "mix21", "mask", "mask_far", "cmp_sgpr"};
    const int per_iter[7] = {128, 128, 112, 96, 128, 128 * 3, 128};
    printf("# %s, %d CUs, LDS per CU %zu; 256-thread workgroups, W per CU forced by LDS\n", p.name, cus,
           (size_t)p.maxSharedMemoryPerMultiProcessor);
    printf("%-9s %3s %9s %12s %12s %14s %14s\n", "stream", "W", "ms", "clock GHz", "instr/SIMD", "cyc/instr/SIMD", "cyc/instr/wave");
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int kind = 0; kind < 7; ++kind)
        for (int w : {1, 2, 4, 5, 6, 8}) {
            const int blocks = cus * w;
            size_t lds = (size_t)(160 * 1024 / w) & ~(size_t)1023;
            if (w == 1)
                lds = 96 * 1024;  // one fits, two do not
            const int it = kind == 5 ? iters / 3 : iters;
            auto go = [&]() {
                switch (kind) {
                case 0: launch<0>(blocks, lds, d_st, d_sink, it); break;
                case 1: launch<1>(blocks, lds, d_st, d_sink, it); break;
                case 2: launch<2>(blocks, lds, d_st, d_sink, it); break;
                case 3: launch<3>(blocks, lds, d_st, d_sink, it); break;
                case 4: launch<4>(blocks, lds, d_st, d_sink, it); break;
                case 5: launch<5>(blocks, lds, d_st, d_sink, it); break;
                default: launch<6>(blocks, lds, d_st, d_sink, it); break;
                }
            };
            go();
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            go();
            hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess)
                return 2;
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(2 * blocks);
            hipMemcpy(h.data(), d_st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
            std::vector<double> ghz(blocks);
            for (int b = 0; b < blocks; ++b)
                ghz[b] = (double)h[2 * b] / (double)h[2 * b + 1] * 0.1;  // s_memrealtime ticks at 100 MHz
            std::sort(ghz.begin(), ghz.end());
            const double clk = ghz[blocks / 2];
            const double n = (double)it * per_iter[kind] * w;  // instructions per SIMD
            const double cyc = ms * 1e-3 * clk * 1e9;
            printf("%-9s %3d %9.3f %12.3f %12.3e %14.2f %14.2f\n", names[kind], w, ms, clk, n, cyc / n, cyc / n * w);
        }
    return 0;
}
