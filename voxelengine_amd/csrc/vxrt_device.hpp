// vxrt_device.hpp -- gfx950 device code shared by the kernels of libvxrt.so.
//
// HBM layout of a resident world:
//   coarse_bits : u32 words, one bit per brick cell, linear order x, z, y: bit x + cx * (z + cz * y)
//   cell_meta   : one uint2 per brick cell in the same order:
//                 .x = pool slot of the brick (VXRT_EMPTY_SLOT if empty)
//                 .y = tight extents, 6 x 5 bits {min x,y,z, max x,y,z}
//                 (replaces the 24-byte VoxelBuffer3D descriptor + 24-byte Bounds3Df per cell)
//   pool        : u32 words, nslots bricks of f^3 bits, the same linear order inside each brick: bit x + f * (z + f * y)
// The reference's bit order (GetSampleIndex, VoxelRT/VolumeRaytracer.cuh:107-131: 8x8x8 tiles) is the order of the
// tables at the C ABI and in the brickmap file; they are re-ordered on the device when a world comes in or goes out
// (vxrt_worldgen.hip).  In HBM the probe's address is two multiply-adds instead of the twelve bit operations of the
// tiled form: that is 11 of a probe's ~50 vector instructions, +5.3 % frame rate measured, same cache hit rates.
//
// All float arithmetic mirrors the reference expression by expression and is
// compiled with -ffp-contract=off so results are IEEE binary32, bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vxrt {

constexpr int kMaxSteps = 2048;               // MAX_STEPS, VolumeRaytracer.cuh:235
constexpr uint32_t kEmptySlot = 0xFFFFFFFFu;
constexpr float kInf = __builtin_huge_valf();
constexpr float kFltEps = 1.1920928955078125e-7f;  // FLT_EPS, VolumeRaytracer.cuh:22

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// plain comparisons, like the host fminf/fmaxf of helper_math.h:56-64 (no NaN / signed-zero special cases)
__device__ __forceinline__ float lo(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float hi(float a, float b) { return a > b ? a : b; }
// normalize: v * (1/sqrt(dot)) with correctly rounded sqrt and divide (helper_math.h:78-81,1325-1329)
__device__ __forceinline__ f3 unit3(f3 v) { return v * (1.0f / sqrtf(dot3(v, v))); }

// ---- correctly rounded reciprocal, square root and quotient in 3 / 8 / 3 instructions, for operands of ordinary size
// hipcc's IEEE division is v_div_scale x 2, v_rcp, 4 FMAs, v_div_fmas, v_div_fixup (10 instructions) and its sqrtf 15; the
// ray-finished phase of the render kernel runs thirteen divisions and four square roots per execution.  On operands whose
// exponent lies in [-100, 100] the hardware's v_rcp_f32 + one Newton step IS the correctly rounded reciprocal and
// v_rsq_f32 + 7 FMA-class instructions IS the correctly rounded square root -- checked on the GPU against the IEEE
// operators for EVERY such binary32 (tools/ubench/rcp_check.hip) -- and with the correctly rounded reciprocal y of b,
// q' = RN(q + RN(a - b q) y), q = RN(a y), is the correctly rounded quotient (Markstein; checked on the CPU for every operand
// pair of the families the kernel uses it on and 10^9 random pairs, tests/tools/exact_div_check.c).  Callers vote on
// `ordinary` and run the plain operators when any lane's operand is outside (never, on the frames of the tests).
#ifndef VXRT_HOST_CHECK
__device__ __forceinline__ float hw_rcp(float x) { float r; asm("v_rcp_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ float hw_rsq(float x) { float r; asm("v_rsq_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ float rcp_rn(float x)
{
    const float y = hw_rcp(x);
    return fmaf(fmaf(-x, y, 1.0f), y, y);
}
__device__ __forceinline__ float sqrt_rn(float x)
{
    const float r = hw_rsq(x);
    float g = x * r, h = 0.5f * r;
    const float e = fmaf(-h, g, 0.5f);
    g = fmaf(g, e, g);
    h = fmaf(h, e, h);
    return fmaf(fmaf(-g, g, x), h, g);
}
#else
__device__ __forceinline__ float rcp_rn(float x) { return 1.0f / x; }
__device__ __forceinline__ float sqrt_rn(float x) { return sqrtf(x); }
#endif
// a / b given y = rcp_rn(b)
__device__ __forceinline__ float div_rn(float a, float b, float y)
{
    const float q = a * y;
    return fmaf(fmaf(-b, q, a), y, q);
}
// exponent in [-100, 100] (and not zero, infinite or NaN): one subtraction and one compare on the bits
__device__ __forceinline__ bool ordinary(float x)
{
    return ((__float_as_uint(x) & 0x7FFFFFFFu) - 0x0D800000u) < (0x72000000u - 0x0D800000u);
}
// unit3 on a vector whose squared length is ordinary (the caller has voted on that)
__device__ __forceinline__ f3 unit3_ordinary(f3 v, float dd) { return v * rcp_rn(sqrt_rn(dd)); }
// reflect(i, n) = i - 2n*dot(n,i) (helper_math.h:1427-1430)
__device__ __forceinline__ f3 reflect3(f3 i, f3 n) { return i - (n * 2.0f) * dot3(n, i); }

struct WorldView {
    const uint32_t* __restrict__ coarse_bits;
    const uint2* __restrict__ cell_meta;
    const uint32_t* __restrict__ pool;
    int cx, cy, cz;        // coarse cells per axis
    int c_row, c_slice;    // coarse cells per x-row (cx), per x-z plane (cx * cz): what a step along z / y adds to a bit index
    int f;                 // brick edge (8, 16, 32)
    int f_row, f_slice;    // brick voxels per x-row (f), per x-z plane (f * f)
    uint32_t brick_words;  // f^3 / 32
    float ff;              // (float)f
    float inv_f;           // 1/f, exact because f is a power of two: x / f == x * inv_f bit for bit
    float wmax_x, wmax_y, wmax_z;  // (float)((double)c - 1e-6), VolumeRaytracer.cu:375-376
    int X, Y;              // world voxels per axis (hit voxel index)
    int c_wide;            // the coarse grid exceeds WaveTracer2's packed step counters (vxrt_wave2.hpp, "wide grids"): set by the host
    // Load guard (probe-counting instantiations only; vxrt_wave2.hpp, "slack"): the words of each bit table proper
    // ([coarse_bits, coarse_end), [pool, pool_end)) and what is addressable around them ([*_lo, *_hi): the allocation).
    const uint32_t *coarse_end, *coarse_lo, *coarse_hi;
    const uint32_t *pool_end, *pool_lo, *pool_hi;
};

// WaveTracer2 (vxrt_wave2.hpp) packs the steps left to the coarse grid's faces into 11 + 10 + 11 bits; a grid beyond that,
// or one a single walk could cross in MAX_STEPS iterations or more, is "wide" (WorldView::c_wide): the fields then count
// down to virtual faces and are re-armed on the way.
// largest value a field of rem is armed with on a wide grid: the history (rp, rpp) adds up to two steps back, below the guards
#ifndef VXRT_FIELD_CAP_XZ  // (the host harness re-arms every few steps with small caps)
#define VXRT_FIELD_CAP_XZ 1020u
#define VXRT_FIELD_CAP_Y 508u
#endif
constexpr uint32_t kFieldCapXZ = VXRT_FIELD_CAP_XZ, kFieldCapY = VXRT_FIELD_CAP_Y;

__host__ __device__ inline bool grid_is_wide(int cx, int cy, int cz)
{
    return cx > (int)kFieldCapXZ || cz > (int)kFieldCapXZ || cy > (int)kFieldCapY || cx + cy + cz + 4 >= kMaxSteps;
}

// ---- the work queue of the persistent kernels ------------------------------------------------------------------------
// A queue of `total` tickets handed out by atomic counters.  ONE counter serves about 80 M atomics a second whatever the
// launch does between them (same-address atomics execute one after the other at the memory side; measured:
// profiles/r04_work_queue.md) -- 16 views of 1080p primary rays ask for more than twice that, and the bench workload
// already for two thirds of it.  So the queue is cut into kQueueShards interleaved sub-queues, each with a counter on a
// line of its own: shard q holds the tickets q, q + S, q + 2 S, ... (every shard runs through the launch's hand-out order
// at the same pace), a wave draws from shard (workgroup number mod S) -- with S = 8 the waves of one XCD share one counter
// -- and, when that one is dry, looks at all the counters with one load and moves on to the next shard that still has
// tickets.  Tickets stay single tiles: nothing gets coarser at the tail of a launch (chunks of 2-8 tickets per atomic,
// fixed or guided by what is left, with and without shards: the batched launches gain as much, single-view launches lose
// 5-30 %; same note).
#ifndef VXRT_QUEUE_SHARDS
#define VXRT_QUEUE_SHARDS 8
#endif
#ifndef VXRT_QUEUE_STRIDE
#define VXRT_QUEUE_STRIDE 128  // words between two shards' counters
#endif
constexpr uint32_t kQueueShards = VXRT_QUEUE_SHARDS, kQueueStride = VXRT_QUEUE_STRIDE;
constexpr uint32_t kQueueWords = kQueueShards * kQueueStride;  // one launch's queue head (zeroed per launch)
constexpr uint32_t kQueueDry = 0xFFFFFFFFu;
static_assert(kQueueShards >= 1 && kQueueShards <= 32 && (kQueueShards & (kQueueShards - 1)) == 0, "shards: a power of two up to 32");

// Shard `shard` holds the tickets whose group of `granule` consecutive tickets has a number == shard (mod S).
__host__ __device__ inline uint32_t queue_ticket(uint32_t shard, uint32_t local, uint32_t granule)
{
    return ((local / granule) * kQueueShards + shard) * granule + local % granule;
}
// tickets of shard `shard` in a queue of `total`
__host__ __device__ inline uint32_t queue_holds(uint32_t shard, uint32_t total, uint32_t granule)
{
    const uint32_t round = granule * kQueueShards, full = total / round, rest = total - full * round;
    const uint32_t part = rest > shard * granule ? rest - shard * granule : 0u;
    return full * granule + (part < granule ? part : granule);
}

// the next shard with tickets after `shard`, cyclically (`open`: bit k = shard k still has tickets; not zero)
__host__ __device__ inline uint32_t queue_next_shard(uint32_t open, uint32_t shard)
{
    const uint32_t above = shard + 1u < 32u ? open >> (shard + 1u) : 0u;
    return above != 0u ? shard + 1u + (uint32_t)__builtin_ctz(above) : (uint32_t)__builtin_ctz(open);
}

#ifndef VXRT_HOST_CHECK
// this workgroup's row of the statistics counters (vxrt_kernels.hpp: kStatRows rows of kStatRowStride words)
template <unsigned ROWS, unsigned STRIDE>
__device__ __forceinline__ unsigned long long* stats_row_of(unsigned long long* stats, unsigned workgroup)
{
    return stats + (size_t)(workgroup % ROWS) * STRIDE;
}

// The next ticket of the queue, or kQueueDry.  Called by the whole (converged) wave; `shard` is the wave's current shard
// (wave-uniform, kept by the caller; start: workgroup number mod kQueueShards).  Every wave leaves through `open == 0`:
// a counter only grows, a shard found dry stays dry, and the loop moves on only to a shard whose counter was below its end.
// (The ticket is tested against the queue's end directly: queue_ticket grows with `local`, so that is `local <
// queue_holds(shard)`.)
__device__ __forceinline__ uint32_t queue_take(unsigned int* heads, uint32_t total, uint32_t granule, uint32_t& shard, uint32_t lane)
{
    for (;;) {
        uint32_t t = 0;
        if (lane == 0)
            t = atomicAdd(&heads[shard * kQueueStride], 1u);
        // (readfirstlane: the ticket is a scalar, so the queue state stays in scalar registers and its branches are scalar)
        const uint32_t local = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        const uint32_t ticket = ((local / granule) * kQueueShards + shard) * granule + local % granule;
        if (ticket < total)
            return ticket;
        if (kQueueShards == 1u)
            return kQueueDry;
        // this shard is dry: which ones are not?  Lane k looks at shard k.
        const uint32_t mine = lane < kQueueShards ? lane : 0u;
        const uint32_t head = __hip_atomic_load(&heads[mine * kQueueStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t open = (uint32_t)__ballot(lane < kQueueShards && head < queue_holds(mine, total, granule));
        if (open == 0u)
            return kQueueDry;
        const uint32_t above = shard + 1u < 32u ? open >> (shard + 1u) : 0u;  // the next open shard, cyclically (queue_next_shard)
        shard = above != 0u ? shard + 1u + (uint32_t)__builtin_ctz(above) : (uint32_t)__builtin_ctz(open);
    }
}

// The same for a queue of single tickets per shard round (granule 1: the batch queue), written out.  (Not a matter of taste:
// k_trace_batch_persist calling the function above with granule = 1 compiles to all but the same instructions and runs the
// 4 M incoherent rays of tools/batch_probe.py 14 % slower -- 1.71 against 1.97-1.99 Grays/s, four sessions; unexplained,
// profiles/r04_work_queue.md section 7.)
__device__ __forceinline__ uint32_t queue_take(unsigned int* heads, uint32_t total, uint32_t& shard, uint32_t lane)
{
    for (;;) {
        uint32_t t = 0;
        if (lane == 0)
            t = atomicAdd(&heads[shard * kQueueStride], 1u);
        const uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)t) * kQueueShards + shard;
        if (ticket < total)
            return ticket;
        if (kQueueShards == 1u)
            return kQueueDry;
        const uint32_t mine = lane < kQueueShards ? lane : 0u;
        const uint32_t head = __hip_atomic_load(&heads[mine * kQueueStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t open = (uint32_t)__ballot(lane < kQueueShards && head < (total > mine ? (total - mine + kQueueShards - 1u) / kQueueShards : 0u));
        if (open == 0u)
            return kQueueDry;
        const uint32_t above = shard + 1u < 32u ? open >> (shard + 1u) : 0u;
        shard = above != 0u ? shard + 1u + (uint32_t)__builtin_ctz(above) : (uint32_t)__builtin_ctz(open);
    }
}
#endif

struct RayCounters {
    uint32_t coarse_probes, brick_entries, fine_probes;
    uint32_t slack_loads = 0, stray_loads = 0;  // load guard: occupancy loads beyond a table but inside its slack / outside everything addressable
};

// float -> int as the hardware does it (v_cvt_i32_f32): truncation, saturating, NaN -> 0.  Spelled with the
// intrinsic because a C cast is undefined out of range, and such values do occur (1/d overflows for denormal d).
#ifndef VXRT_HOST_CHECK
__device__ __forceinline__ int f2i(float v)
{
    // (spelled as the instruction: __float2int_rz compiles to v_trunc_f32 + v_cvt_i32_f32, and the conversion already
    // truncates)
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
#else
__device__ __forceinline__ int f2i(float v) { return __float2int_rz(v); }
#endif
// a cell coordinate clamped into [0, top] (top >= 0): the median of (v, 0, top), one instruction.  The probes read the
// occupancy word of the clamped cell unconditionally -- in range it is the cell (or, under the edge rule, its clamped
// neighbour, :242-244), out of range any valid word will do, its bit is ignored
#ifndef VXRT_HOST_CHECK
__device__ __forceinline__ int clamp_cell(int v, int top)
{
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "v"(top));
    return r;
}
#else
__device__ __forceinline__ int clamp_cell(int v, int top) { return v < 0 ? 0 : (v > top ? top : v); }
#endif
// (float)f2i(v) == c for an integer-valued c in int range, as ONE instruction + the compare: truncf(v) is that float for
// every finite v of int range, and beyond it (or for NaN) neither side of the original comparison can equal such a c --
// except f2i(NaN) = 0 against c = 0, where the reference's own cast (cvttss2si: INT_MIN) says "different", as truncf does
__device__ __forceinline__ bool trunc_equals(float v, float c) { return truncf(v) == c; }

// Lane conditions as explicit wave masks: lane_mask(p) is the 64-bit mask of p over the wave (a v_cmp result as it
// is), masks combine with & | ~ on the scalar unit, lane_test(m) reads this lane's bit back as a condition for a
// select.  Only for code the whole wave executes together (all lanes active).
#ifndef VXRT_HOST_CHECK
typedef unsigned long long lanemask_t;
__device__ __forceinline__ lanemask_t lane_mask(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool lane_test(lanemask_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
#endif

// bit address of a cell in HBM: x-fastest linear, for both levels (row, slice = the level's strides)
#ifndef VXRT_HOST_CHECK  // tools/host_wave_check.cpp runs this header on the host and brings its own mad24
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#endif

// HBM order: x fastest, then z, then y -- bit x + row * (z + dz * y), row = cells along x, dz = cells along z.  y last
// because it is the world's short, vertical axis: rays move mostly in x and z, so consecutive probes of a ray stay in one
// word (x steps) or one 64-byte line (z steps) more often than with y in the middle.
__device__ __forceinline__ uint32_t cell_index(int x, int y, int z, int row, int dz)
{
    // 24-bit multiply-adds (full rate; v_mul_lo_u32 / v_mad_u64_u32 are quarter rate): coordinates < 2^16 and dy * dz < 2^24
    // (checked when a world is uploaded or built), so the inner sum stays below 2^24 whatever the grid's x-z plane holds --
    // 4096 x 16 x 4096 bricks, the reference's stated goal (VoxelApp/main.cu:20), included.  Spelled as the instruction:
    // given __umul24 the compiler still selects the 32-bit multiply when it cannot prove the operands' width.
    return mad24(mad24((uint32_t)y, (uint32_t)dz, (uint32_t)z), (uint32_t)row, (uint32_t)x);
}
// the same order for the cold code (re-ordering kernels, builders, host): index of a cell of a dx x dy x dz grid, and back
__host__ __device__ inline uint64_t hbm_index(int x, int y, int z, int dx, int dz)
{
    return (uint64_t)x + (uint64_t)dx * ((uint64_t)z + (uint64_t)dz * (uint64_t)y);
}
__host__ __device__ inline void hbm_cell(uint64_t i, int dx, int dz, int& x, int& y, int& z)
{
    x = (int)(i % (uint64_t)dx);
    z = (int)((i / (uint64_t)dx) % (uint64_t)dz);
    y = (int)(i / ((uint64_t)dx * (uint64_t)dz));
}

// The reference's bit order (GetSampleIndex / GetPositionFromSampleIndex, VolumeRaytracer.cuh:107-171): 8x8x8 tiles,
// tiles x-fastest, cells x-fastest inside a tile.  Only the re-ordering kernels and the builders use it (cold code).
__host__ __device__ inline uint32_t ref_tiled_index(int x, int y, int z, int tiles_x, int tiles_y)
{
    const uint32_t tile = ((uint32_t)(z >> 3) * (uint32_t)tiles_y + (uint32_t)(y >> 3)) * (uint32_t)tiles_x + (uint32_t)(x >> 3);
    return tile * 512u + (uint32_t)((x & 7) | ((y & 7) << 3) | ((z & 7) << 6));
}
__host__ __device__ inline void ref_tiled_cell(uint32_t t, int tiles_x, int tiles_y, int& x, int& y, int& z)
{
    const uint32_t tile = t >> 9, in = t & 511u;
    x = (int)((tile % (uint32_t)tiles_x) * 8u + (in & 7u));
    y = (int)(((tile / (uint32_t)tiles_x) % (uint32_t)tiles_y) * 8u + ((in >> 3) & 7u));
    z = (int)((tile / ((uint32_t)tiles_x * (uint32_t)tiles_y)) * 8u + (in >> 6));
}

// nextafterf(v, neg ? -inf : +inf) by bit manipulation (VolumeRaytracer.cu:452-460)
__device__ __forceinline__ float ulp_step(float v, bool neg)
{
    // straight-line selects (it runs inside the wave tracer's end-of-walk phase, where every branch costs the
    // whole wave an exec-mask round trip)
    const uint32_t b = __float_as_uint(v);
    const uint32_t mag = b & 0x7FFFFFFFu;
    const bool negative = (b >> 31) != 0u;
    const bool away = negative == neg;  // stepping away from zero
    const uint32_t stepped = away ? b + 1u : b - 1u;
    const uint32_t from_zero = neg ? 0x80000001u : 0x00000001u;
    const uint32_t moved = mag == 0u ? from_zero : stepped;
    // NaN stays; so does the infinity we walk toward
    const bool keep = mag > 0x7F800000u || (mag == 0x7F800000u && away);
    return __uint_as_float(keep ? b : moved);
}

// RayIntersectsAABB (VolumeRaytracer.cu:124-174)
__device__ __forceinline__ bool ray_box(f3 s, f3 d, f3 bmin, f3 bmax, f3& p, f3& n)
{
    float ix = 1.0f / (d.x == 0 ? kFltEps : d.x);
    float iy = 1.0f / (d.y == 0 ? kFltEps : d.y);
    float iz = 1.0f / (d.z == 0 ? kFltEps : d.z);
    float ax = (bmin.x - s.x) * ix, bx = (bmax.x - s.x) * ix;
    float ay = (bmin.y - s.y) * iy, by = (bmax.y - s.y) * iy;
    float az = (bmin.z - s.z) * iz, bz = (bmax.z - s.z) * iz;
    float nx = lo(ax, bx), fx = hi(ax, bx);
    float ny = lo(ay, by), fy = hi(ay, by);
    float nz = lo(az, bz), fz = hi(az, bz);
    float t_in = hi(hi(nx, ny), nz);
    float t_out = lo(lo(fx, fy), fz);
    if (t_out < hi(t_in, 0.0f))
        return false;
    p = mk3(s.x + t_in * d.x, s.y + t_in * d.y, s.z + t_in * d.z);
    if (t_in == nx)
        n = mk3(ix < 0.0f ? -1.0f : 1.0f, 0.0f, 0.0f);
    else if (t_in == ny)
        n = mk3(0.0f, iy < 0.0f ? -1.0f : 1.0f, 0.0f);
    else
        n = mk3(0.0f, 0.0f, iz < 0.0f ? -1.0f : 1.0f);
    return true;
}

// Ray validity at the C ABI (include/vxrt.h, "ray validity"): the origin's components are finite (their absolute sum is
// a finite binary32) and the direction's squared length, evaluated in binary32 as normalize does (helper_math.h:1325-1329),
// is positive and finite -- i.e. normalize(ray) is a vector of finite numbers.  Anything else (NaN, infinities, the zero
// vector, a direction so short or so long that its squared length leaves the binary32 range) makes Raytrace's prologue
// produce NaN or infinite directions in the reference (VolumeRaytracer.cu:359-367), from where its behaviour is
// undefined (float -> int casts of NaN); this build defines the result of such a ray instead: a miss with 0 steps.
__device__ __forceinline__ bool ray_valid(f3 o, f3 d)
{
    const float dd = dot3(d, d);
    const float mag = fabsf(o.x) + fabsf(o.y) + fabsf(o.z);
    return mag < kInf && dd > 0.0f && dd < kInf;  // (every comparison is false for NaN)
}

struct WalkResult {
    bool hit, oob;
    int hx, hy, hz;     // HitCell (clamped cell of the last in-range probe)
    int ncx, ncy, ncz;  // NextCell
    f3 point;           // HitIntersectedPoint
    f3 normal;          // HitNormal
    int steps;          // stepsTaken
};

// DDARayTraversal (VolumeRaytracer.cu:176-352), straightforward form: one call walks one
// level from `s` until a hit or the exit.  COARSE: per-cell tight boxes from cell_meta are
// tested from the walk's start (:248-273); otherwise the region check [0,f]^3 on the crossing
// point applies (:325-341).
template <bool COARSE>
__device__ void walk_level(const WorldView& W, const uint32_t* __restrict__ bits, int dim_x, int dim_y, int dim_z,
                           f3 s, f3 d, WalkResult& R, uint32_t& probes)
{
    int cell_x = f2i(s.x), cell_y = f2i(s.y), cell_z = f2i(s.z);
    const int sgn_x = d.x > 0 ? 1 : -1, sgn_y = d.y > 0 ? 1 : -1, sgn_z = d.z > 0 ? 1 : -1;
    const float td_x = d.x != 0 ? fabsf(1.0f / d.x) : kInf;
    const float td_y = d.y != 0 ? fabsf(1.0f / d.y) : kInf;
    const float td_z = d.z != 0 ? fabsf(1.0f / d.z) : kInf;
    float tn_x = d.x != 0 ? ((float)(cell_x + (sgn_x > 0)) - s.x) / d.x : kInf;
    float tn_y = d.y != 0 ? ((float)(cell_y + (sgn_y > 0)) - s.y) / d.y : kInf;
    float tn_z = d.z != 0 ? ((float)(cell_z + (sgn_z > 0)) - s.z) / d.z : kInf;

    R.hit = false;
    R.oob = false;
    R.hx = R.hy = R.hz = 0;
    R.ncx = R.ncy = R.ncz = 0;
    R.point = s;
    R.normal = mk3(0, 0, 0);
    R.steps = 0;

    int pad_x = 0, pad_y = 0, pad_z = 0;  // edge rule, :216-232
    if (cell_x == dim_x || cell_y == dim_y || cell_z == dim_z) {
        pad_x = d.x < 0;
        pad_y = d.y < 0;
        pad_z = d.z < 0;
    }

    bool leaving = false;
    for (int it = 0; it < kMaxSteps; ++it) {
        bool inside = 0 <= cell_x && cell_x < dim_x + pad_x && 0 <= cell_y && cell_y < dim_y + pad_y &&
                      0 <= cell_z && cell_z < dim_z + pad_z;
        if (inside) {
            int qx = min(max(cell_x, 0), dim_x - 1), qy = min(max(cell_y, 0), dim_y - 1),
                qz = min(max(cell_z, 0), dim_z - 1);
            R.hx = qx;
            R.hy = qy;
            R.hz = qz;
            probes += 1;
            uint32_t idx = cell_index(qx, qy, qz, dim_x, dim_z);
            bool solid = bits ? ((bits[idx >> 5] >> (idx & 31u)) & 1u) != 0u : false;
            if (COARSE) {
                if (solid) {
                    uint32_t e = W.cell_meta[idx].y;
                    f3 bmin = mk3(((float)(e & 31u) + 0) * W.inv_f + (float)qx,
                                  ((float)((e >> 5) & 31u) + 0) * W.inv_f + (float)qy,
                                  ((float)((e >> 10) & 31u) + 0) * W.inv_f + (float)qz);
                    f3 bmax = mk3(((float)((e >> 15) & 31u) + 1) * W.inv_f + (float)qx,
                                  ((float)((e >> 20) & 31u) + 1) * W.inv_f + (float)qy,
                                  ((float)((e >> 25) & 31u) + 1) * W.inv_f + (float)qz);
                    f3 bp, bn;
                    if (bmin.x <= bmax.x && ray_box(s, d, bmin, bmax, bp, bn)) {
                        R.hit = true;
                        R.normal = bn;
                        if (it != 0)
                            R.point = bp;
                        leaving = true;
                    }
                }
            } else if (solid) {
                R.hit = true;
                leaving = true;
            }
        } else {
            R.oob = true;
            leaving = true;
        }

        // advance one cell, also on the exit iteration (:290-322)
        f3 cross;
        f3 step_n;
        if (tn_x < tn_y && tn_x < tn_z) {
            cross = mk3((float)(cell_x + (sgn_x > 0)), s.y + (tn_x * d.y), s.z + (tn_x * d.z));
            cell_x += sgn_x;
            tn_x += td_x;
            step_n = mk3((float)sgn_x, 0, 0);
        } else if (tn_y <= tn_x && tn_y < tn_z) {
            cross = mk3(s.x + (tn_y * d.x), (float)(cell_y + (sgn_y > 0)), s.z + (tn_y * d.z));
            cell_y += sgn_y;
            tn_y += td_y;
            step_n = mk3(0, (float)sgn_y, 0);
        } else {
            cross = mk3(s.x + (tn_z * d.x), s.y + (tn_z * d.y), (float)(cell_z + (sgn_z > 0)));
            cell_z += sgn_z;
            tn_z += td_z;
            step_n = mk3(0, 0, (float)sgn_z);
        }
        if (leaving) {
            R.ncx = cell_x;
            R.ncy = cell_y;
            R.ncz = cell_z;
            break;
        }
        R.normal = step_n;
        if (!COARSE) {
            // region [0,f]^3, int-truncated and inclusive, tested on the crossing point (:325-341)
            const float fmax = W.ff;
            if (cross.x < 0.0f || cross.x > fmax || cross.y < 0.0f || cross.y > fmax || cross.z < 0.0f ||
                cross.z > fmax) {
                R.oob = true;
                break;
            }
        }
        R.steps += 1;
        R.point = cross;
    }
}

struct TraceResult {
    bool hit;
    int steps;
    f3 pos;     // valid on hit
    f3 normal;  // step-direction convention; zero on a miss
    int vx, vy, vz;  // global voxel that ended the ray (valid on hit)
    uint32_t ncode;  // wave tracer only: `normal` as its small code (normal_decode, vxrt_wave2.hpp)
};

// Raytrace (VolumeRaytracer.cu:354-525), straightforward form.
__device__ void trace_direct(const WorldView& W, int max_steps, f3 origin, f3 ray, TraceResult& out, RayCounters& cnt)
{
    float last_x = -1, last_y = -1, last_z = -1;
    int total = 0;
    f3 start = mk3(origin.x * W.inv_f, origin.y * W.inv_f, origin.z * W.inv_f);  // origin / factor
    const f3 dir = unit3(ray);
    f3 entry_n = mk3(0, 0, 0);
    if (!(start.x >= 0 && start.y >= 0 && start.z >= 0 && start.x < (float)W.cx && start.y < (float)W.cy &&
          start.z < (float)W.cz)) {
        const float e = (float)1e-6;
        f3 p, n;
        if (ray_box(start, dir, mk3(e, e, e), mk3(W.wmax_x, W.wmax_y, W.wmax_z), p, n)) {
            start = p;
            entry_n = n;
        }
    }
    out.normal = mk3(0, 0, 0);
    out.hit = false;
    out.vx = out.vy = out.vz = 0;
    f3 hit_pos = mk3(0, 0, 0);

    while (total < max_steps) {
        WalkResult c;
        walk_level<true>(W, W.coarse_bits, W.cx, W.cy, W.cz, start, dir, c, cnt.coarse_probes);
        total += c.steps;
        f3 local = mk3(c.point.x * W.ff, c.point.y * W.ff, c.point.z * W.ff);
        hit_pos = local;
        if (!(c.hit && !c.oob))
            break;
        const float hx = (float)c.hx, hy = (float)c.hy, hz = (float)c.hz;
        if (last_x == hx && last_y == hy && last_z == hz)
            break;  // :402-407
        last_x = hx;
        last_y = hy;
        last_z = hz;
        local = mk3(local.x - hx * W.ff, local.y - hy * W.ff, local.z - hz * W.ff);

        uint32_t ci = cell_index(c.hx, c.hy, c.hz, W.cx, W.cz);
        uint32_t slot = W.cell_meta[ci].x;
        const uint32_t* bits = nullptr;
        int bd = 0;
        if (slot != kEmptySlot) {
            bits = W.pool + (size_t)slot * W.brick_words;
            bd = W.f;
        }
        cnt.brick_entries += 1;
        WalkResult b;
        walk_level<false>(W, bits, bd, bd, bd, local, dir, b, cnt.fine_probes);
        total += b.steps;
        hit_pos = mk3(b.point.x + hx * W.ff, b.point.y + hy * W.ff, b.point.z + hz * W.ff);
        if (b.hit) {
            out.normal = (b.steps == 0) ? c.normal : b.normal;
            out.vx = c.hx * W.f + b.hx;
            out.vy = c.hy * W.f + b.hy;
            out.vz = c.hz * W.f + b.hz;
            out.hit = true;
            break;
        }
        start = mk3(hit_pos.x * W.inv_f, hit_pos.y * W.inv_f, hit_pos.z * W.inv_f);
        if (b.oob) {
            bool same = hx == (float)f2i(start.x) && hy == (float)f2i(start.y) && hz == (float)f2i(start.z);
            if (same) {
                start.x = ulp_step(start.x, dir.x < 0);
                start.y = ulp_step(start.y, dir.y < 0);
                start.z = ulp_step(start.z, dir.z < 0);
                same = hx == (float)f2i(start.x) && hy == (float)f2i(start.y) && hz == (float)f2i(start.z);
                if (same) {
                    float gx = (float)c.ncx - start.x, gy = (float)c.ncy - start.y, gz = (float)c.ncz - start.z;
                    float mx = fabsf(gx), my = fabsf(gy), mz = fabsf(gz);
                    if (mx < my && mx < mz)
                        start.x += gx;
                    else if (my < mx && my < mz)
                        start.y += gy;
                    else
                        start.z += gz;
                }
            }
        }
    }
    out.steps = total;
    if (out.hit) {
        out.pos = hit_pos;
        if (total == 0) {
            out.pos = mk3(start.x * W.ff, start.y * W.ff, start.z * W.ff);
            out.normal = entry_n;
        }
    }
}

}  // namespace vxrt
