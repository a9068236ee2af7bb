// issue_rate.hip -- what one SIMD of gfx950 issues per cycle, measured with pinned instruction streams (inline asm, so the
// compiler cannot merge, pack or reorder them) at 1, 2, 4, 5, 6 and 8 waves per SIMD.  In-kernel s_memtime stamps give shader
// cycles directly (no assumed clock).  Streams:
//   valu_indep   8 independent v_fma_f32 chains            -- peak issue rate of plain wave64 VALU
//   valu_dep     one dependent v_fma_f32 chain              -- dependent-issue latency
//   valu_cndmask v_cmp_lt_f32 -> v_cndmask_b32 (VCC)        -- compare + select pairs, as in WaveTracer::step
//   salu_indep   8 independent s_add_u32                    -- scalar issue rate
//   mix_5v2s     5 VALU : 2 SALU, independent               -- the instruction mix of the render kernel's hot path
//   mask_logic   v_cmp -> s_and_b64/s_or_b64 -> v_cndmask   -- the explicit wave-mask idiom (dependent through SGPR pairs)
//   pk_fma       4 independent v_pk_fma_f32 chains          -- packed FP32 (two lanes of math per issue)
// Output: cycles per instruction seen by ONE wave, and cycles per instruction per SIMD (= the former / waves per SIMD).
// Build: hipcc -O3 --offload-arch=gfx950 -o issue_rate issue_rate.hip ; run: ./issue_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned long long* cycles, float* sink, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0001f, c = 0.5f;
    unsigned s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const f2 pm = {m, m}, pc = {c, c};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // 16 x 8 = 128 VALU
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\t"
                               "v_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\t"
                               "v_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                               : "v"(m), "v"(c));)
        } else if (KIND == 1) {  // 128 dependent VALU
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                               "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                               "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2"
                               : "+v"(a0)
                               : "v"(m), "v"(c));)
        } else if (KIND == 2) {  // 16 x (4 cmp + 4 cndmask) = 128 VALU, each select depends on its compare through VCC
            REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\t"
                               "v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %1, %1, %3, vcc\n\t"
                               "v_cmp_lt_f32 vcc, %2, %3\n\tv_cndmask_b32 %2, %2, %0, vcc\n\t"
                               "v_cmp_lt_f32 vcc, %3, %0\n\tv_cndmask_b32 %3, %3, %1, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"vcc");)
        } else if (KIND == 3) {  // 128 SALU
            REP16(asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 2\n\ts_add_u32 %2, %2, 3\n\ts_add_u32 %3, %3, 4\n\t"
                               "s_add_u32 %4, %4, 5\n\ts_add_u32 %5, %5, 6\n\ts_add_u32 %6, %6, 7\n\ts_add_u32 %7, %7, 8"
                               : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)::"scc");)
        } else if (KIND == 4) {  // 16 x (5 VALU + 2 SALU) = 112 instructions
            REP16(asm volatile("v_fma_f32 %0, %0, %7, %8\n\tv_fma_f32 %1, %1, %7, %8\n\ts_add_u32 %5, %5, 1\n\t"
                               "v_fma_f32 %2, %2, %7, %8\n\tv_fma_f32 %3, %3, %7, %8\n\ts_add_u32 %6, %6, 2\n\t"
                               "v_fma_f32 %4, %4, %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+s"(s0), "+s"(s1)
                               : "v"(m), "v"(c)
                               : "scc");)
        } else if (KIND == 5) {  // 16 x (3 v_cmp + 3 SALU + 2 v_cndmask) = 128 instructions, the wave-mask idiom
            REP16(asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cmp_lt_f32 s[22:23], %1, %2\n\tv_cmp_lt_f32 s[24:25], %2, %0\n\t"
                               "s_and_b64 s[26:27], s[20:21], s[22:23]\n\ts_andn2_b64 s[28:29], s[24:25], s[20:21]\n\t"
                               "s_or_b64 s[30:31], s[26:27], s[28:29]\n\t"
                               "v_cndmask_b32 %0, %0, %2, s[26:27]\n\tv_cndmask_b32 %1, %1, %0, s[30:31]"
                               : "+v"(a0), "+v"(a1), "+v"(a2)::"s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28",
                                 "s29", "s30", "s31", "scc");)
        } else {  // 16 x 4 = 64 packed VALU (each = two FP32 fmas per lane)
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\t"
                               "v_pk_fma_f32 %3, %3, %4, %5"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                               : "v"(pm), "v"(pc));)
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0)
        cycles[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] =
        a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7) + p0.x + p0.y + p1.x + p1.y + p2.x +
        p2.y + p3.x + p3.y;
}

int main()
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess)
        return 1;
    const int cus = p.multiProcessorCount;
    const int max_blocks = cus * 4 * 8;
    unsigned long long* d_cycles;
    float* d_sink;
    hipMalloc(&d_cycles, sizeof(unsigned long long) * max_blocks);
    hipMalloc(&d_sink, sizeof(float) * max_blocks * 64);
    const int iters = 4000;
    const char* names[7] = {"valu_indep", "valu_dep", "valu_cndmask", "salu_indep", "mix_5v2s", "mask_logic", "pk_fma"};
    const int per_iter[7] = {128, 128, 128, 128, 112, 128, 64};
    printf("# %s, %d CUs; cycles are s_memtime shader cycles inside the kernel, median over workgroups\n", p.name, cus);
    printf("%-13s %10s %26s %26s\n", "stream", "waves/SIMD", "cycles/instr (one wave)", "cycles/instr per SIMD");
    for (int kind = 0; kind < 7; ++kind)
        for (int wps : {1, 2, 4, 5, 6, 8}) {
            const int blocks = cus * 4 * wps;  // 64-thread workgroups: wps waves on every SIMD
            auto launch = [&]() {
                switch (kind) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d_cycles, d_sink, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d_cycles, d_sink, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d_cycles, d_sink, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, d_cycles, d_sink, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, d_cycles, d_sink, iters); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, d_cycles, d_sink, iters); break;
                default: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(64), 0, 0, d_cycles, d_sink, iters); break;
                }
            };
            launch();
            hipDeviceSynchronize();
            launch();
            if (hipDeviceSynchronize() != hipSuccess)
                return 2;
            std::vector<unsigned long long> h(blocks);
            hipMemcpy(h.data(), d_cycles, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            const double med = (double)h[blocks / 2], n = (double)iters * per_iter[kind];
            printf("%-13s %10d %26.2f %26.2f\n", names[kind], wps, med / n, med / n / wps);
        }
    return 0;
}
