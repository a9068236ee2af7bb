// instr_cost.hip -- cycles per wave64 instruction per SIMD for the instruction CLASSES the render kernels are made of, by the
// wall clock at the clock the kernel itself measures (s_memtime / s_memrealtime), waves per SIMD forced as in issue_rate2.hip
// (256-thread workgroups, W per CU by LDS size).  Each stream is 8 independent instructions of one class, 16 x per loop trip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

// 8 vector registers a0..a7 (in/out), b = another vector, sc = a scalar register; masks live in s[20:35], set up before the loop
#define V8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define CLOB "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35"

#define STREAMS(X)                                                                                                                \
    X(0, "v_fma_f32 v,v,v,v", "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8") \
    X(1, "v_add_f32 v,v,v (VOP2)", "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8") \
    X(2, "v_add_f32 v,s,v (VOP2, SGPR src)", "v_add_f32 %0, %9, %0\n v_add_f32 %1, %9, %1\n v_add_f32 %2, %9, %2\n v_add_f32 %3, %9, %3\n v_add_f32 %4, %9, %4\n v_add_f32 %5, %9, %5\n v_add_f32 %6, %9, %6\n v_add_f32 %7, %9, %7") \
    X(3, "v_add_f32 v,literal,v", "v_add_f32 %0, 0x3f8ccccd, %0\n v_add_f32 %1, 0x3f8ccccd, %1\n v_add_f32 %2, 0x3f8ccccd, %2\n v_add_f32 %3, 0x3f8ccccd, %3\n v_add_f32 %4, 0x3f8ccccd, %4\n v_add_f32 %5, 0x3f8ccccd, %5\n v_add_f32 %6, 0x3f8ccccd, %6\n v_add_f32 %7, 0x3f8ccccd, %7") \
    X(4, "v_add_f32 v,1.0,v (inline const)", "v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3\n v_add_f32 %4, 1.0, %4\n v_add_f32 %5, 1.0, %5\n v_add_f32 %6, 1.0, %6\n v_add_f32 %7, 1.0, %7") \
    X(5, "v_fma_f32 v,v,s,v (VOP3, SGPR src)", "v_fma_f32 %0, %0, %9, %8\n v_fma_f32 %1, %1, %9, %8\n v_fma_f32 %2, %2, %9, %8\n v_fma_f32 %3, %3, %9, %8\n v_fma_f32 %4, %4, %9, %8\n v_fma_f32 %5, %5, %9, %8\n v_fma_f32 %6, %6, %9, %8\n v_fma_f32 %7, %7, %9, %8") \
    X(6, "v_cmp_lt_f32 vcc,v,v (VOPC)", "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8") \
    X(7, "v_cmp_lt_f32 s[n:n+1],v,v (VOP3)", "v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n v_cmp_lt_f32 s[28:29], %4, %8\n v_cmp_lt_f32 s[30:31], %5, %8\n v_cmp_lt_f32 s[32:33], %6, %8\n v_cmp_lt_f32 s[34:35], %7, %8") \
    X(8, "v_cndmask_b32 v,v,v,vcc (VOP2)", "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc") \
    X(9, "v_cndmask_b32 v,v,v,s[n:n+1] (VOP3)", "v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[22:23]\n v_cndmask_b32 %2, %2, %8, s[24:25]\n v_cndmask_b32 %3, %3, %8, s[26:27]\n v_cndmask_b32 %4, %4, %8, s[28:29]\n v_cndmask_b32 %5, %5, %8, s[30:31]\n v_cndmask_b32 %6, %6, %8, s[32:33]\n v_cndmask_b32 %7, %7, %8, s[34:35]") \
    X(10, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %8, %8\n v_mad_u32_u24 %1, %1, %8, %8\n v_mad_u32_u24 %2, %2, %8, %8\n v_mad_u32_u24 %3, %3, %8, %8\n v_mad_u32_u24 %4, %4, %8, %8\n v_mad_u32_u24 %5, %5, %8, %8\n v_mad_u32_u24 %6, %6, %8, %8\n v_mad_u32_u24 %7, %7, %8, %8") \
    X(11, "v_med3_i32", "v_med3_i32 %0, %0, %8, %8\n v_med3_i32 %1, %1, %8, %8\n v_med3_i32 %2, %2, %8, %8\n v_med3_i32 %3, %3, %8, %8\n v_med3_i32 %4, %4, %8, %8\n v_med3_i32 %5, %5, %8, %8\n v_med3_i32 %6, %6, %8, %8\n v_med3_i32 %7, %7, %8, %8") \
    X(12, "v_min3_f32", "v_min3_f32 %0, %0, %8, %8\n v_min3_f32 %1, %1, %8, %8\n v_min3_f32 %2, %2, %8, %8\n v_min3_f32 %3, %3, %8, %8\n v_min3_f32 %4, %4, %8, %8\n v_min3_f32 %5, %5, %8, %8\n v_min3_f32 %6, %6, %8, %8\n v_min3_f32 %7, %7, %8, %8") \
    X(13, "v_cvt_i32_f32", "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7") \
    X(14, "v_lshlrev_b32 v,5,v", "v_lshlrev_b32 %0, 5, %0\n v_lshlrev_b32 %1, 5, %1\n v_lshlrev_b32 %2, 5, %2\n v_lshlrev_b32 %3, 5, %3\n v_lshlrev_b32 %4, 5, %4\n v_lshlrev_b32 %5, 5, %5\n v_lshlrev_b32 %6, 5, %6\n v_lshlrev_b32 %7, 5, %7") \
    X(15, "v_and_b32 v,v,v", "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8") \
    X(16, "v_bfe_u32 v,v,v,5", "v_bfe_u32 %0, %0, %8, 5\n v_bfe_u32 %1, %1, %8, 5\n v_bfe_u32 %2, %2, %8, 5\n v_bfe_u32 %3, %3, %8, 5\n v_bfe_u32 %4, %4, %8, 5\n v_bfe_u32 %5, %5, %8, 5\n v_bfe_u32 %6, %6, %8, 5\n v_bfe_u32 %7, %7, %8, 5") \
    X(17, "v_bfi_b32", "v_bfi_b32 %0, %8, %0, %8\n v_bfi_b32 %1, %8, %1, %8\n v_bfi_b32 %2, %8, %2, %8\n v_bfi_b32 %3, %8, %3, %8\n v_bfi_b32 %4, %8, %4, %8\n v_bfi_b32 %5, %8, %5, %8\n v_bfi_b32 %6, %8, %6, %8\n v_bfi_b32 %7, %8, %7, %8") \
    X(18, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8") \
    X(19, "v_rcp_f32", "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7") \
    X(20, "v_readlane_b32 s,v,3", "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n v_readlane_b32 s24, %4, 3\n v_readlane_b32 s25, %5, 3\n v_readlane_b32 s26, %6, 3\n v_readlane_b32 s27, %7, 3") \
    X(21, "v_mov_b32 v,s", "v_mov_b32 %0, %9\n v_mov_b32 %1, %9\n v_mov_b32 %2, %9\n v_mov_b32 %3, %9\n v_mov_b32 %4, %9\n v_mov_b32 %5, %9\n v_mov_b32 %6, %9\n v_mov_b32 %7, %9") \
    X(22, "s_and_b64", "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[22:23], s[22:23], s[24:25]\n s_and_b64 s[24:25], s[24:25], s[26:27]\n s_and_b64 s[26:27], s[26:27], s[28:29]\n s_and_b64 s[28:29], s[28:29], s[30:31]\n s_and_b64 s[30:31], s[30:31], s[32:33]\n s_and_b64 s[32:33], s[32:33], s[34:35]\n s_and_b64 s[34:35], s[34:35], s[20:21]") \
    X(23, "v_cmp_lt_f32 vcc + v_cndmask vcc pairs (dependent)", "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc") \
    X(24, "v_cmp_lt_u32 s[n:n+1],v,v (VOP3, int)", "v_cmp_lt_u32 s[20:21], %0, %8\n v_cmp_lt_u32 s[22:23], %1, %8\n v_cmp_lt_u32 s[24:25], %2, %8\n v_cmp_lt_u32 s[26:27], %3, %8\n v_cmp_lt_u32 s[28:29], %4, %8\n v_cmp_lt_u32 s[30:31], %5, %8\n v_cmp_lt_u32 s[32:33], %6, %8\n v_cmp_lt_u32 s[34:35], %7, %8") \
    X(25, "v_add_u32 v,v,v", "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8") \
    X(26, "v_mul_f32 v,v,v", "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8") \
    X(27, "v_add_f32 with |abs| modifier (VOP3)", "v_add_f32 %0, %0, |%8|\n v_add_f32 %1, %1, |%8|\n v_add_f32 %2, %2, |%8|\n v_add_f32 %3, %3, |%8|\n v_add_f32 %4, %4, |%8|\n v_add_f32 %5, %5, |%8|\n v_add_f32 %6, %6, |%8|\n v_add_f32 %7, %7, |%8|") \
    X(28, "v_add_co_u32 v,vcc,v,v", "v_add_co_u32 %0, vcc, %0, %8\n v_add_co_u32 %1, vcc, %1, %8\n v_add_co_u32 %2, vcc, %2, %8\n v_add_co_u32 %3, vcc, %3, %8\n v_add_co_u32 %4, vcc, %4, %8\n v_add_co_u32 %5, vcc, %5, %8\n v_add_co_u32 %6, vcc, %6, %8\n v_add_co_u32 %7, vcc, %7, %8") \
    X(29, "ds_read_b32 (8 in flight, then wait)", "ds_read_b32 %0, %10\n ds_read_b32 %1, %10 offset:256\n ds_read_b32 %2, %10 offset:512\n ds_read_b32 %3, %10 offset:768\n ds_read_b32 %4, %10 offset:1024\n ds_read_b32 %5, %10 offset:1280\n ds_read_b32 %6, %10 offset:1536\n ds_read_b32 %7, %10 offset:1792\n s_waitcnt lgkmcnt(0)")

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned long long* stamps, float* sink, int iters, float scal)
{
    extern __shared__ unsigned lds[];
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f + (float)(threadIdx.x & 1);
    const unsigned la = (threadIdx.x & 63) * 4;
    for (unsigned i = threadIdx.x; i < 1024; i += 256)
        lds[i] = i;
    __syncthreads();
    asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 s[22:23], 0\n s_mov_b64 s[24:25], exec\n s_mov_b64 s[26:27], 0\n"
                 "s_mov_b64 s[28:29], exec\n s_mov_b64 s[30:31], 0\n s_mov_b64 s[32:33], exec\n s_mov_b64 s[34:35], 0\n"
                 "s_mov_b64 vcc, exec" ::
                     : CLOB, "vcc");
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int i = 0; i < iters; ++i) {
#define X(N, NAME, TEXT) \
    if (KIND == N) { REP16(asm volatile(TEXT : V8 : "v"(b), "s"(scal), "v"(la) : CLOB, "vcc", "scc");) }
        STREAMS(X)
#undef X
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
    sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
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
    if (hipMalloc(&d_st, sizeof(unsigned long long) * 2 * cus * 8) != hipSuccess || hipMalloc(&d_sink, sizeof(float) * cus * 8 * 256) != hipSuccess)
        return 1;
    const int iters = 6000;
    printf("# %s, %d CUs; 256-thread workgroups, W per CU forced by LDS; 128 instructions per loop trip\n", p.name, cus);
    printf("%-52s %3s %9s %10s %15s\n", "stream", "W", "ms", "clock GHz", "cyc/instr/SIMD");
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[] = {
#define X(N, NAME, TEXT) NAME,
        STREAMS(X)
#undef X
    };
    const int nk = sizeof(names) / sizeof(names[0]);
    for (int kind = 0; kind < nk; ++kind)
        for (int w : {1, 4, 5, 6, 8}) {
            const int blocks = cus * w;
            size_t lds = (size_t)(160 * 1024 / w) & ~(size_t)1023;
            if (w == 1)
                lds = 96 * 1024;
            auto go = [&]() {
                switch (kind) {
#define X(N, NAME, TEXT) case N: launch<N>(blocks, lds, d_st, d_sink, iters); break;
                    STREAMS(X)
#undef X
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
            for (int bb = 0; bb < blocks; ++bb)
                ghz[bb] = (double)h[2 * bb] / (double)h[2 * bb + 1] * 0.1;
            std::sort(ghz.begin(), ghz.end());
            const double clk = ghz[blocks / 2];
            const double n = (double)iters * 128 * w;
            printf("%-52s %3d %9.3f %10.3f %15.2f\n", names[kind], w, ms, clk, ms * 1e-3 * clk * 1e9 / n);
            fflush(stdout);
        }
    return 0;
}
