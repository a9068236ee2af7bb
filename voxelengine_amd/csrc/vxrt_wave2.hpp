// vxrt_wave2.hpp -- the two-level brickmap DDA (Raytrace / DDARayTraversal, VoxelRT/VolumeRaytracer.cu:176-525) as a flat,
// wave-level state machine made for what a gfx950 SIMD actually issues.  The one tracer of the product kernels
// (k_render_persist2, k_trace_batch_persist, k_trace_batch_wave2); trace_direct (vxrt_device.hpp) is the straightforward
// cross-check.
//
// Organisation (rounds 1-2, profiles/r01_*, r02_*): ONE loop for 64 rays; every iteration probes one cell and advances
// one step for every walking lane, on the coarse grid or inside a brick alike.  The rare, expensive events PARK the lane
// -- the tight-box test of an occupied coarse cell with the entry into its brick (ST_BOX), the end of a walk with the brick
// exit / ulp nudge / ray end (ST_END, ST_ENDHIT) -- and a __ballot vote runs a parked phase only when enough lanes wait for
// it.  Per-ray invariants are hoisted (1/(d or eps) of the slab test, |1/d| of the DDA); the 8-byte cell_meta record read
// for the slab test also carries the brick's pool slot.  The state only the parked phases touch lives in the wave's LDS
// block, one 64-lane column per field (`cold`), so the probe loop fits 96 VGPRs = 5 waves per SIMD.
//
// The probe (round 3, profiles/r03_instr_cost.md): the render kernels are bound by the vector ALU pipe, and its
// instructions come in two classes -- ~2.3 cycles per wave64 instruction (add / sub / mul f32, and / or / xor, add / sub
// u32, right shifts, moves, with vector or constant operands) and ~4.15 cycles (every compare, v_cndmask, min / max,
// conversions, every three-operand form, anything with a scalar-register operand) -- beside a scalar pipe with one
// instruction per ~4.15 cycles.  Hence:
//
//  * The DDA advance is SPECULATIVE and exec-masked.  A DDA's path does not depend on the voxels, only where it stops
//    does; so every walking lane advances in every probe, before its occupancy word has arrived, and the word only
//    decides who stops.  The advance is committed in place under the wave masks of the three axes
//    (`s_mov_b64 exec, mask` on the scalar pipe + fast vector instructions) instead of add + v_cndmask per quantity.
//    A lane that stops has the state before its last advance in a two-deep history (rp / rpp, tp), so nothing is undone.
//  * The cell is not kept as three integers that are clamped (3 v_med3), range-checked (3 v_cmp) and multiplied into a
//    bit index (2 v_mad) per probe.  The probe carries the BIT INDEX itself, advanced by the axis' stride, and the
//    remaining steps to the grid's faces as three counters packed in one word with a guard bit each (`rem`): one
//    subtraction per advance, one and + one compare per probe to see any of the three run out.  The cell coordinates,
//    the step count (stepsTaken) and the last step's axis are decoded from rem and its history when a walk ends.
//  * The crossing point, the [0,f]^3 region check on it (:325-341) and HitIntersectedPoint are not evaluated per probe.
//    cross(t) = start + t * d is monotone in t, so the check can only fail before a time t_lo or after a time t_hi that
//    begin_walk computes with a relative margin; the probe compares t with t_hi (one compare), a lane beyond it -- or
//    leaving the grid, which is later than t_hi by construction -- ends its walk, and the end-of-walk phase evaluates the
//    reference's expressions on that one step exactly (and puts a lane that passes back on its walk).  The point is
//    recomputed there from the step's t and axis.
//  * Lane state that the probe tests lives in wave masks (scalar registers): brick / coarse level.  Everything rare is
//    pushed to "single-step mode" (t_hi = -inf: every step is validated by the end-of-walk phase, which then also
//    recomputes the bit index from the cell): edge-rule starts on several far faces at once, starts whose first crossing
//    may fail the region check.  The common edge-rule start (one far face, stepped first) costs one subtraction per pair.
//
// Wide grids (WIDE, set by the host: a coarse dimension beyond rem's fields -- x, z > 1020 or y > 508 cells -- or a
// grid a single walk could cross in MAX_STEPS iterations or more, cx + cy + cz + 4 >= 2048).  The probes are the same; a
// field of rem then holds the steps left to a VIRTUAL face, min(steps to the real face, a cap), and the rest sits in two
// cold words (CF_OFF_XZ, CF_OFF_Y).  A field that runs out raises its guard as a real face does; the end-of-walk phase
// sees the offset, counts the step and re-arms the fields (a lane crosses at least cap + 1 cells between two such trips).
// The caps also keep the sum of the fields below the iterations the walk has left of DDARayTraversal's MAX_STEPS
// (:234), so the phase that re-arms is also the place where a walk of 2048 counted steps ends as the reference's loop
// does: no hit, not out of bounds, Raytrace breaks (:508-511).  Ordinary grids run the instantiation without any of it.
// The world's tables need addressable slack of one x-z slice before and behind the coarse bits and of one
// brick around the pool (vxrt_api.hip allocates it): a lane that has just left the grid issues one more load.
#pragma once

#include "vxrt_device.hpp"
#ifdef VXRT_HOST_DEBUG
#include <cstdio>
#include <cstdlib>
#endif

namespace vxrt {

// lane states: walking; parked for the tight-box phase; parked for the end-of-walk phase (ST_END, and ST_ENDHIT after a brick
// probe that found an occupied voxel: the two codes differ in bit 0 only, so "waits for the end-of-walk phase" is one OR and
// one compare -- written as `a || b` the vote's ballot compiled to two compares, a select and a third compare); ray finished;
// no work left
enum : uint32_t { ST_WALK = 0u, ST_BOX = 1u, ST_END = 2u, ST_ENDHIT = 3u, ST_DONE = 4u, ST_IDLE = 5u };
__device__ __forceinline__ bool waits_for_end(uint32_t st) { return (st | 1u) == 3u; }
// (Measured in round 4 and not kept, profiles/r04_finish_walks.md: the two common ends of a ray -- a brick probe that finds a
// voxel, a coarse walk that steps out of the world -- settled by the caller's ray-finished phase instead of an end-of-walk
// phase of their own: 36 % fewer end-of-walk executions, but the finished lanes then idle until the rarer ray-finished phase
// runs; 16 views per launch 7127 against 7164 Mrays/s.)
// A/B knob: 1 = the probes of ordinary grids test rem's guard bits too (as in round 3)
#ifndef VXRT_PROBE_GD
#define VXRT_PROBE_GD 0
#endif
#ifdef VXRT_HOST_CHECK
// host harness: advances that left a grid without being beyond the walk's time threshold (must stay 0 on ordinary grids)
inline unsigned long long& host_unsuspected_exits() { static unsigned long long n = 0; return n; }
#endif

// unit normals as small codes: 0 = zero vector, (axis+1) | 4*negative
__device__ __forceinline__ f3 normal_decode(uint32_t c)
{
    float v = (c & 4u) ? -1.0f : 1.0f;
    uint32_t a = c & 3u;
    return mk3(a == 1u ? v : 0.0f, a == 2u ? v : 0.0f, a == 3u ? v : 0.0f);
}

// a parked phase runs when its lanes, times `num`, are at least the other live lanes (or nobody else can move): parked > 0
// and parked * num >= others, as ONE scalar comparison (the two-condition form compiled to two compare + select pairs and an
// and of masks per vote, and a round has five votes)
__device__ __forceinline__ bool vote_run(int parked, int others, int num) { return parked * num >= max(others, 1); }
// thresholds of trace_wave2 (one ray per lane: the batch kernel for small batches, the host harness)
#ifndef VXRT_VOTE_END
#define VXRT_VOTE_END 2
#endif
#ifndef VXRT_VOTE_BOX
#define VXRT_VOTE_BOX 4
#endif

// The "cold" part of a lane's ray -- everything only the parked phases, begin_ray and result touch (the probes never do):
// one 64-lane column per field in the wave's LDS block (conflict-free ds_read / ds_write), no registers between phases.
// The small ones share a word: CF_RAY_CODES = entry_code (3 bits) | out_code << 3 | ray_hit << 6 | max_steps << 7,
// CF_BOX_CODES = c_code (3 bits) | nc_axis << 3 (both written by the tight-box phase only).  CF_OFF_XZ / CF_OFF_Y: wide
// grids only, the steps to the real faces beyond rem's fields (x | z << 16; y).
enum : int {
    CF_RAY_CODES = 0, CF_START_X, CF_START_Y, CF_START_Z, CF_LAST_CI, CF_TOTAL,
    CF_CHX, CF_CHY, CF_CHZ, CF_BOX_CODES, CF_OFF_XZ, CF_OFF_Y, CF_TRACER_FIELDS
};

constexpr uint32_t kRemDecX = 1u, kRemDecY = 1u << 11, kRemDecZ = 1u << 21;
constexpr uint32_t kRemGuards = (1u << 10) | (1u << 20) | (1u << 31);
// Relative margins of the walk's time threshold t_hi.  Since round 4 the threshold is also what stops a lane that leaves its
// grid on ordinary grids (the probe no longer tests rem's guard bits there), so the margin must cover the worst drift between
// the DDA's accumulated t of the exit step -- tMax_k after n_k additions of |1/d_k|, each rounded: at most (n_k + 5) * 2^-24
// relative, n_k <= 32 steps along an axis in a brick, <= 1020 on an ordinary coarse grid -- and the directly computed exit
// time (bound_k - s_k) * (1/d_k): 2^-17 for brick walks (37 * 2^-24 < 2^-18.7), 2^-13 for coarse ones (1025 * 2^-24 < 2^-13.9).
constexpr float kThrEpsFine = 7.62939453125e-06f;     // 2^-17
constexpr float kThrEpsCoarse = 1.220703125e-04f;     // 2^-13
constexpr float kMinFastDir = 9.094947017729282e-13f;  // 2^-40: smallest direction component begin_walk_fast divides by

// The machine's min / max as single instructions (fminf / fmaxf compile to the instruction plus one canonicalising
// v_max_f32 per operand under IEEE rules).  Callers guarantee ordinary operands (no NaN; see slab_fast).
#ifndef VXRT_HOST_CHECK
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// A value computed HERE, on every lane that is active here: keeps the compiler from sinking a few cheap instructions into an
// exec-mask branch of their own (four scalar instructions and a branch to skip two vector ones)
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }
#else
// (host build: IEEE minNum / maxNum as the instructions are -- a NaN operand loses)
__device__ __forceinline__ float vmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float vmin3(float a, float b, float c) { return vmin(vmin(a, b), c); }
__device__ __forceinline__ float vmax3(float a, float b, float c) { return vmax(vmax(a, b), c); }
__device__ __forceinline__ void pin(float&) {}
#endif

// bit (index mod 32) of a word: v_bfe_u32 takes the offset from the low five bits of its operand, so the index needs no mask
#ifndef VXRT_HOST_CHECK
__device__ __forceinline__ uint32_t bit_of(uint32_t word, uint32_t index)
{
    uint32_t r;
    asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(r) : "v"(word), "v"(index));
    return r;
}
#else
__device__ __forceinline__ uint32_t bit_of(uint32_t word, uint32_t index) { return (word >> (index & 31u)) & 1u; }
#endif
__device__ __forceinline__ uint32_t rem_fx(uint32_t r) { return r & 0x7FFu; }
__device__ __forceinline__ uint32_t rem_fy(uint32_t r) { return (r >> 11) & 0x3FFu; }
__device__ __forceinline__ uint32_t rem_fz(uint32_t r) { return r >> 21; }
__device__ __forceinline__ uint32_t rem_sum(uint32_t r) { return rem_fx(r) + rem_fy(r) + rem_fz(r); }

// WIDE: the instantiation for wide grids (WorldView::c_wide; the launchers pick it): as a run-time branch on every grid its
// scalar registers and branches cost ordinary grids 1.5 % (profiles/r04_finish_walks.md, `nowide`).
template <bool WIDE>
struct WaveTracerT {
    // per ray
    f3 d;
    float ivx, ivy, ivz;  // 1/(d or eps) (:127-129); |iv| is the DDA's tDelta (:199-201)
    // current walk, read by the probes
    float tn_x, tn_y, tn_z;
    uint32_t idx;             // bit index of the (clamped) current cell + the level's bias (one x-z slice)
    uint32_t di_x, di_y, di_z;  // what one step along the axis adds to idx
    uint32_t rem, rp, rpp;    // steps left to the faces, packed (x | y << 11 | z << 21, one guard bit per field); history
    float tl, tp;             // t of the last advance and of the one before
    float t_hi;               // the probe ends the walk of a lane whose t exceeds it (-inf: single-step mode)
    uint32_t fix;             // edge rule: the first step along the axis that starts on its far face does not move idx
    const uint32_t* bits;     // the level's words - bias
    uint32_t st;
    // current walk, read by the phases only
    f3 ws;
    uint32_t rem0;            // sum of rem's fields at the start of the walk
    float t_hi_real;
    bool special;             // per ray: a direction component is zero or below 2^-40, or a start component is -0.0
    uint32_t dn;              // per ray: bit k set where the ray does not move up axis k (d_k <= 0)
    f3 point;                 // HitIntersectedPoint of the walk that ended (tight-box phase / end-of-walk phase)
    lanemask_t fine_m;        // wave mask: lanes walking inside a brick
    uint32_t* cold;           // &block[lane]; field F of this lane is cold[F * 64]
    // Probe counters of SURVEY 8(d)'s algorithmic bytes (the STATS instantiations of the phases only; dead otherwise):
    // in-range coarse probes (:247-256), brick entries (:420), in-range brick probes (:276).  They fall out of the walk's
    // packed step counters when the walk ends -- a walk that ends after k advances has probed k + 1 cells -- so the probes
    // themselves count nothing.
    RayCounters cnt;

    __device__ __forceinline__ bool lane_fine() const
    {
#ifdef VXRT_HOST_CHECK
        return (fine_m & 1ull) != 0ull;
#else
        return lane_test(fine_m);
#endif
    }

    __device__ __forceinline__ void init(const WorldView& W, uint32_t* cold_column)
    {
        cold = cold_column;
        st = ST_DONE;
        d = mk3(1.0f, 0.0f, 0.0f);
        ivx = ivy = ivz = 1.0f;
        tn_x = tn_y = tn_z = 0.0f;
        idx = (uint32_t)W.c_slice;
        di_x = di_y = di_z = 0u;
        rem = rp = rpp = 0u;
        tl = tp = 0.0f;
        t_hi = t_hi_real = kInf;
        special = false;
        dn = 6u;
        fix = 0u;
        bits = W.coarse_bits - (W.c_slice >> 5);
        ws = point = mk3(0, 0, 0);
        rem0 = 0u;
        fine_m = 0ull;
        cnt = RayCounters{0u, 0u, 0u, 0u, 0u};
    }

    // slab test against [bmin,bmax] from point s with the hoisted reciprocals (RayIntersectsAABB, :124-174)
    __device__ __forceinline__ bool slab(f3 s, f3 bmin, f3 bmax, f3& p, uint32_t& code) const
    {
        float ax = (bmin.x - s.x) * ivx, bx = (bmax.x - s.x) * ivx;
        float ay = (bmin.y - s.y) * ivy, by = (bmax.y - s.y) * ivy;
        float az = (bmin.z - s.z) * ivz, bz = (bmax.z - s.z) * ivz;
        float nx = lo(ax, bx), fx = hi(ax, bx);
        float ny = lo(ay, by), fy = hi(ay, by);
        float nz = lo(az, bz), fz = hi(az, bz);
        float t_in = hi(hi(nx, ny), nz);
        float t_out = lo(lo(fx, fy), fz);
        p = mk3(s.x + t_in * d.x, s.y + t_in * d.y, s.z + t_in * d.z);
        code = (t_in == nx) ? (1u | (ivx < 0.0f ? 4u : 0u))
                            : (t_in == ny) ? (2u | (ivy < 0.0f ? 4u : 0u)) : (3u | (ivz < 0.0f ? 4u : 0u));
        return !(t_out < hi(t_in, 0.0f));
    }

    // The same test with the machine's min / max (v_min_f32, v_max3_f32 ...: 8 instructions instead of 16 compare + select
    // pairs).  They differ from lo() / hi() only when an operand is a NaN (0 * inf: a direction component so small that
    // its reciprocal overflows) or when a zero of the other sign wins (visible only in the sign of a zero of p, and only
    // if a start component is -0.0): both are `special` rays (begin_ray), which keep slab().
    __device__ __forceinline__ bool slab_fast(f3 s, f3 bmin, f3 bmax, f3& p, uint32_t& code) const
    {
        float ax = (bmin.x - s.x) * ivx, bx = (bmax.x - s.x) * ivx;
        float ay = (bmin.y - s.y) * ivy, by = (bmax.y - s.y) * ivy;
        float az = (bmin.z - s.z) * ivz, bz = (bmax.z - s.z) * ivz;
        float nx = vmin(ax, bx), fx = vmax(ax, bx);
        float ny = vmin(ay, by), fy = vmax(ay, by);
        float nz = vmin(az, bz), fz = vmax(az, bz);
        float t_in = vmax3(nx, ny, nz);
        float t_out = vmin3(fx, fy, fz);
        p = mk3(s.x + t_in * d.x, s.y + t_in * d.y, s.z + t_in * d.z);
        // the axis code without branches: (axis + 1) | 4 * sign bit of the reciprocal (a reciprocal is never -0 or NaN), then
        // two selects in the reference's order (:157-171)
        const uint32_t c1 = ((__float_as_uint(ivx) >> 29) & 4u) | 1u, c2 = ((__float_as_uint(ivy) >> 29) & 4u) | 2u,
                       c3 = ((__float_as_uint(ivz) >> 29) & 4u) | 3u;
        const uint32_t c23 = t_in == ny ? c2 : c3;
        code = t_in == nx ? c1 : c23;
        return !(t_out < vmax(t_in, 0.0f));
    }

    // cell coordinates (unclamped) of a rem word of the current walk
    __device__ __forceinline__ void cells_of(const WorldView& W, bool fine, uint32_t r, int& x, int& y, int& z) const
    {
        const int dmx = fine ? W.f : W.cx, dmy = fine ? W.f : W.cy, dmz = fine ? W.f : W.cz;
        int fx = (int)rem_fx(r), fy = (int)rem_fy(r), fz = (int)rem_fz(r);
        if (WIDE) {  // (wave-uniform) the fields count down to virtual faces: the real ones are further by the offsets
            const uint32_t oxz = cold[CF_OFF_XZ * 64], oy = cold[CF_OFF_Y * 64];
            fx += fine ? 0 : (int)(oxz & 0xFFFFu);
            fy += fine ? 0 : (int)oy;
            fz += fine ? 0 : (int)(oxz >> 16);
        }
        x = d.x > 0 ? dmx - 1 - fx : fx;  // (an axis that moves up never starts on its far face: no pad)
        y = d.y > 0 ? dmy - 1 - fy : fy;
        z = d.z > 0 ? dmz - 1 - fz : fz;
    }

    // x / dv for a direction component of ordinary size, with iv = RN(1 / dv): q = RN(x * iv), then one correction with
    // the exact residual, q' = RN(q + (x - q * dv) * iv).  With a correctly rounded reciprocal that is the correctly
    // rounded quotient (Markstein); checked against the division on 7e8 operand pairs of this code's ranges, the all-ones
    // significands included.  (x is a difference of two floats here: never -0, the one operand the form gets wrong.)
    static __device__ __forceinline__ float fast_div(float x, float dv, float iv)
    {
        const float q = x * iv;
        return fmaf(fmaf(-q, dv, x), iv, q);
    }

    // DDARayTraversal's set-up (:178-232) for a walk from s on the level FINE, written for the vector pipe's two
    // instruction classes: sign masks and integer arithmetic instead of compare + select, the three quotients by fast_div,
    // no entry-side threshold (a start outside [0, f] begins in single-step mode and leaves it with the first validated
    // step whose three components are inside: walk_on).  Sets st: ST_WALK, or ST_END for a start outside the grid (the
    // first iteration of the reference leaves at once: no step, isOutOfBounds).  The caller updates fine_m.
    // Rays with a direction component that is zero or below 2^-40 (`special`: the centre column of an axis-aligned
    // camera, adversarial tests) get their quotients from the division itself and validate every step of a brick walk.
    template <bool FINE>
    __device__ __forceinline__ void start_walk(const WorldView& W, const f3 s)
    {
        ws = s;
        const int dmx = FINE ? W.f : W.cx, dmy = FINE ? W.f : W.cy, dmz = FINE ? W.f : W.cz;
        const uint32_t row = (uint32_t)(FINE ? W.f_row : W.c_row), slice = (uint32_t)(FINE ? W.f_slice : W.c_slice);
        // n = all ones where the ray does not move up the axis (d <= 0), m = all ones where it moves up (:195-197)
        const uint32_t nx = 0u - (dn & 1u), ny = 0u - ((dn >> 1) & 1u), nz = 0u - (dn >> 2);
        const uint32_t mx = ~nx, my = ~ny, mz = ~nz;
        const int c_x = f2i(s.x), c_y = f2i(s.y), c_z = f2i(s.z);
        const float xx = (float)(c_x + (int)(nx + 1u)) - s.x, xy = (float)(c_y + (int)(ny + 1u)) - s.y,
                    xz = (float)(c_z + (int)(nz + 1u)) - s.z;
        tn_x = fast_div(xx, d.x, ivx);
        tn_y = fast_div(xy, d.y, ivy);
        tn_z = fast_div(xz, d.z, ivz);
        bool zero_on_face = false;  // an axis the ray does not move along starts on its far face: no pad there (d < 0 only)
        if (__ballot(special) != 0ull) {
            if (special) {
                tn_x = d.x != 0 ? xx / d.x : kInf;
                tn_y = d.y != 0 ? xy / d.y : kInf;
                tn_z = d.z != 0 ? xz / d.z : kInf;
                zero_on_face = (d.x == 0 && c_x == dmx) || (d.y == 0 && c_y == dmy) || (d.z == 0 && c_z == dmz);
            }
        }
        // steps left to the face: up: dim - 1 - cell = (cell ^ ~0) + dim; down: cell
        const uint32_t fx = ((uint32_t)c_x ^ mx) + (mx & (uint32_t)dmx), fy = ((uint32_t)c_y ^ my) + (my & (uint32_t)dmy),
                       fz = ((uint32_t)c_z ^ mz) + (mz & (uint32_t)dmz);
        // inside: up: 0 <= f <= dim - 1; down: 0 <= f <= dim (f == dim only under the edge rule, which it then switches on,
        // :216-232)
        const bool inside = fx < (uint32_t)dmx - nx && fy < (uint32_t)dmy - ny && fz < (uint32_t)dmz - nz && !zero_on_face;
        uint32_t px = fx, py = fy, pz = fz;  // what rem's fields are armed with
        if (!FINE && WIDE) {  // (wave-uniform) wide grid: virtual faces, the whole MAX_STEPS budget ahead
            constexpr uint32_t cap = ((uint32_t)kMaxSteps - 1u) / 3u;
            px = min(fx, min(cap, kFieldCapXZ));
            py = min(fy, min(cap, kFieldCapY));
            pz = min(fz, min(cap, kFieldCapXZ));
            cold[CF_OFF_XZ * 64] = (fx - px) | ((fz - pz) << 16);
            cold[CF_OFF_Y * 64] = fy - py;
        }
        rem = rp = rpp = inside ? (px | (py << 11) | (pz << 21)) : 0u;
        rem0 = inside ? px + py + pz : 0u;
        // lookups use the cell clamped to dim-1 (:242-244; differs from the cell only on a far face under the edge rule)
        const int q_x = clamp_cell(c_x, dmx - 1), q_y = clamp_cell(c_y, dmy - 1), q_z = clamp_cell(c_z, dmz - 1);
        idx = cell_index(q_x, q_y, q_z, (int)row, dmz) + slice;
        di_x = nx | 1u;
        di_z = (row ^ nz) - nz;
        di_y = (slice ^ ny) - ny;
        // Edge rule: an axis that starts ON its far face (cell == dim; inside: such an axis moves down) takes one step that
        // does not change the clamped cell.  If there is one such axis and it is the DDA's first choice, the probes handle
        // it (idx gets `fix` subtracted after the first advance); anything else goes to single-step mode.
        const uint32_t ex = (uint32_t)(c_x - q_x), ey = (uint32_t)(c_y - q_y), ez = (uint32_t)(c_z - q_z);
        bool single = special;
        fix = 0u;
        if (__ballot(inside && (ex | ey | ez) != 0u) != 0ull) {
            const bool a0 = tn_x < tn_y && tn_x < tn_z;
            const bool a1 = !(tn_x < tn_y) && tn_y < tn_z;
            const bool a2 = !a0 && !a1;
            const uint32_t npend = ex + ey + ez;
            const bool simple = npend == 1u && ((ex != 0u && a0) || (ey != 0u && a1) || (ez != 0u && a2));
            fix = simple ? (((0u - ex) & di_x) | ((0u - ey) & di_y) | ((0u - ez) & di_z)) : 0u;
            single = single || (npend != 0u && !simple);
        }
        // The walk's time threshold.  Component k of the crossing point, start_k + (t * d_k), is monotone in t; on its exit
        // side it stays inside the level's grid until t = (bound_k - s_k) / d_k, bound = the dimension (up) or 0.  A brick
        // walk's region check (:325-341) can only fail beyond the minimum over the axes, and NO walk can leave its grid before
        // it; with the margin of kThrEps* towards the inside the probe's one compare of t against it catches both.  (A step
        // along k itself is not checked against k's own bound by the reference, but a step that is later than this along
        // another axis is rare enough -- the lane is about to leave through k -- to be validated one by one.)
        // (coarse: the world box's far corner, dims - 1e-6 rounded to binary32 (:375-376), is at or just inside the grid's far
        // faces -- a threshold a hair earlier is as good, and the value is in a scalar register already)
        const uint32_t bxb = __float_as_uint(FINE ? W.ff : W.wmax_x), byb = __float_as_uint(FINE ? W.ff : W.wmax_y),
                       bzb = __float_as_uint(FINE ? W.ff : W.wmax_z);
        const float hx = (__uint_as_float(bxb & mx) - s.x) * ivx, hy = (__uint_as_float(byb & my) - s.y) * ivy,
                    hz = (__uint_as_float(bzb & mz) - s.z) * ivz;
        float hi_t = vmin3(hx, hy, hz);
        hi_t = hi_t - fabsf(hi_t) * (FINE ? kThrEpsFine : kThrEpsCoarse);
        // (a direction component below 2^-40 makes its quotient overflow, inf - inf above: such rays validate every step)
        hi_t = special ? -kInf : hi_t;
        if (FINE) {
            // entry side: a start outside [0, f] on some axis (sign bit of s_k or of f - s_k)
            const uint32_t out = __float_as_uint(s.x) | __float_as_uint(s.y) | __float_as_uint(s.z) | __float_as_uint(W.ff - s.x) |
                                 __float_as_uint(W.ff - s.y) | __float_as_uint(W.ff - s.z);
            single = single || (int32_t)out < 0;
        }
        t_hi_real = hi_t;
        t_hi = single ? -kInf : hi_t;
        if (!FINE)
            bits = W.coarse_bits - (slice >> 5);  // (brick walks: the caller has set the brick's words)
        st = inside ? (uint32_t)ST_WALK : (uint32_t)ST_END;
    }

    // Raytrace's prologue (:359-384): per-ray constants, world entry, first coarse walk.  Per lane; the caller clears the
    // lanes' bits in fine_m afterwards (after_begin_ray), where the wave is converged again.
    __device__ __forceinline__ void begin_ray(const WorldView& W, f3 origin, f3 ray, int max_steps_)
    {
        // normalize and the three reciprocals of the slab test (:127-129, :366): the short exact forms of vxrt_device.hpp where
        // every operand is of ordinary size, the plain operators for the wave if any lane's is not
        const float dd = dot3(ray, ray);
        d = unit3_ordinary(ray, dd);
        const float ex = d.x == 0 ? kFltEps : d.x, ey = d.y == 0 ? kFltEps : d.y, ez = d.z == 0 ? kFltEps : d.z;
        ivx = rcp_rn(ex);
        ivy = rcp_rn(ey);
        ivz = rcp_rn(ez);
        if (__ballot(!(ordinary(dd) & ordinary(ex) & ordinary(ey) & ordinary(ez))) != 0ull) {
            d = unit3(ray);
            ivx = 1.0f / (d.x == 0 ? kFltEps : d.x);
            ivy = 1.0f / (d.y == 0 ? kFltEps : d.y);
            ivz = 1.0f / (d.z == 0 ? kFltEps : d.z);
        }
        f3 s0 = mk3(origin.x * W.inv_f, origin.y * W.inv_f, origin.z * W.inv_f);
        uint32_t ec = 0u;
        if (!(s0.x >= 0 && s0.y >= 0 && s0.z >= 0 && s0.x < (float)W.cx && s0.y < (float)W.cy && s0.z < (float)W.cz)) {
            const float e = (float)1e-6;
            f3 p;
            uint32_t c;
            if (slab(s0, mk3(e, e, e), mk3(W.wmax_x, W.wmax_y, W.wmax_z), p, c)) {
                s0 = p;
                ec = c;
            }
        }
        cold[CF_RAY_CODES * 64] = ec | ((uint32_t)max_steps_ << 7);  // out_code = 0, ray_hit = false
        cold[CF_START_X * 64] = __float_as_uint(s0.x);
        cold[CF_START_Y * 64] = __float_as_uint(s0.y);
        cold[CF_START_Z * 64] = __float_as_uint(s0.z);
        cold[CF_LAST_CI * 64] = 0xFFFFFFFFu;  // previous_cell as its cell index (unique per cell); none yet
        cold[CF_TOTAL * 64] = 0u;
        special = !(fabsf(d.x) >= kMinFastDir && fabsf(d.y) >= kMinFastDir && fabsf(d.z) >= kMinFastDir) ||
                  __float_as_uint(s0.x) == 0x80000000u || __float_as_uint(s0.y) == 0x80000000u || __float_as_uint(s0.z) == 0x80000000u;
        dn = (d.x > 0 ? 0u : 1u) | (d.y > 0 ? 0u : 2u) | (d.z > 0 ? 0u : 4u);
        start_walk<false>(W, s0);
    }
    __device__ __forceinline__ void after_begin_ray(bool launched) { fine_m &= ~__ballot(launched); }

    // The crossing point of the step that started from rem word `before` (the cell before the step) along the axis
    // whose field the step decremented by `dec`, at time t -- the reference's `cross` (:293-313).
    __device__ __forceinline__ f3 cross_of(const WorldView& W, bool fine, uint32_t before, uint32_t dec, float t) const
    {
        int bx, by, bz;
        cells_of(W, fine, before, bx, by, bz);
        const bool on_x = dec == kRemDecX, on_y = dec == kRemDecY, on_z = dec == kRemDecZ;
        float face_x = (float)(bx + (d.x > 0 ? 1 : 0)), face_y = (float)(by + (d.y > 0 ? 1 : 0)), face_z = (float)(bz + (d.z > 0 ? 1 : 0));
        float lin_x = ws.x + (t * d.x), lin_y = ws.y + (t * d.y), lin_z = ws.z + (t * d.z);
        pin(face_x), pin(face_y), pin(face_z), pin(lin_x), pin(lin_y), pin(lin_z);  // (both arms, then a select)
        return mk3(on_x ? face_x : lin_x, on_y ? face_y : lin_y, on_z ? face_z : lin_z);
    }

    // A lane goes back to its walk after a phase (a suspected step that passed its checks, a tight box that was missed).
    // Single-step mode recomputes idx from the cell (edge rule: the clamped cell), and ends once no axis is on its far
    // face any more and the crossing point just validated has all three components inside [0, f] (from there on only
    // the exit side of the region check can fail, which t_hi watches).
    __device__ __forceinline__ void walk_on(const WorldView& W, bool is_fine, bool lin_inside)
    {
        const bool single = t_hi == -kInf;
        if (__ballot(single) == 0ull)  // (nothing to do for a lane on an ordinary walk: it keeps its idx and its threshold)
            return;
        int x, y, z;
        cells_of(W, is_fine, rem, x, y, z);
        const int dmx = is_fine ? W.f : W.cx, dmy = is_fine ? W.f : W.cy, dmz = is_fine ? W.f : W.cz;
        const int row = is_fine ? W.f_row : W.c_row, slice = is_fine ? W.f_slice : W.c_slice;
        const uint32_t exact = cell_index(min(x, dmx - 1), min(y, dmy - 1), min(z, dmz - 1), row, dmz) + (uint32_t)slice;
        idx = single ? exact : idx;
        const bool pending = x == dmx || y == dmy || z == dmz;
        t_hi = (single && !pending && lin_inside) ? t_hi_real : t_hi;
    }

    // parked phase: end of a walk (:395-511).  Called by the whole wave (converged); works on the lanes with st == ST_END
    // (the last advance left the grid and / or was later than t_hi, or the walk never started) and ST_ENDHIT (a brick walk
    // whose probe found an occupied voxel).  Coarse walks that end on a hit tight box never come here: the tight-box
    // phase enters the brick itself.
    template <bool STATS = false>
    __device__ __forceinline__ void phase_end(const WorldView& W)
    {
        const bool me = waits_for_end(st);
        const bool is_fine = lane_fine();
        bool go_coarse = false;  // this lane restarts the coarse walk
        if (me) {
            // ---- what the walk that just ended did.  ST_ENDHIT: the probe of the cell `rp` found it occupied; the advance
            // after it (rp -> rem) is the reference's exit advance.  ST_END: validate the last advance (rp -> rem, at time
            // tl) with the reference's expressions.
            const bool hit = st == ST_ENDHIT;
            const uint32_t dec_last = rp - rem, dec_prev = rpp - rp;
            const bool stepped = dec_last != 0u;
            const bool guard = (rem & kRemGuards) != 0u;
            bool exiting = guard || !stepped;
            // wide grids: a guard raised by a field that ran out before its real face (`virt`) is a step to validate and
            // count like a suspected one, and the walk's MAX_STEPS-th counted step ends it (`exhausted`; :234)
            bool virt = false, exhausted = false;
            uint32_t offx = 0u, offy = 0u, offz = 0u;
            if (WIDE) {
                const uint32_t oxz = cold[CF_OFF_XZ * 64], oy = cold[CF_OFF_Y * 64];
                offx = oxz & 0xFFFFu;
                offy = oy;
                offz = oxz >> 16;
                const uint32_t off_k = dec_last == kRemDecX ? offx : (dec_last == kRemDecY ? offy : offz);
                virt = !is_fine && guard && stepped && off_k != 0u;
                exiting = exiting && !virt;
                exhausted = virt && !hit && rem0 - rem_sum(rp) + 1u >= (uint32_t)kMaxSteps;
            }
            const float F = W.ff;
            const f3 lin = mk3(ws.x + (tl * d.x), ws.y + (tl * d.y), ws.z + (tl * d.z));
            int bx, by, bz;
            cells_of(W, is_fine, rp, bx, by, bz);
            const f3 cr = mk3(dec_last == kRemDecX ? (float)(bx + (d.x > 0 ? 1 : 0)) : lin.x,
                              dec_last == kRemDecY ? (float)(by + (d.y > 0 ? 1 : 0)) : lin.y,
                              dec_last == kRemDecZ ? (float)(bz + (d.z > 0 ? 1 : 0)) : lin.z);
            // (a component outside [0, F] on either side; min3 / max3 pass over a NaN component as the comparisons do)
            float cr_lo = vmin3(cr.x, cr.y, cr.z), cr_hi = vmax3(cr.x, cr.y, cr.z);
            float lin_lo = vmin3(lin.x, lin.y, lin.z), lin_hi = vmax3(lin.x, lin.y, lin.z);
            f3 pc_prev = cross_of(W, is_fine, rpp, dec_prev, tp);  // the crossing point of the step before, if the last one does not count
            pin(cr_lo), pin(cr_hi), pin(lin_lo), pin(lin_hi);
            const bool region_fail = is_fine & stepped & ((cr_lo < 0.0f) | (cr_hi > F));
            const bool resume = !hit && !exiting && !region_fail && !exhausted;  // a step that was only suspected: walk on
            if (resume) {
                if (WIDE) {
                    // re-arm the fields of a lane whose virtual face was reached: the steps left to the real faces after this
                    // step, capped so that their sum stays below the iterations the walk has left; the history in the new frame
                    const uint32_t so_far = rem0 - rem_sum(rp) + 1u;  // counted steps of this walk, this one included
                    const uint32_t cap = ((uint32_t)kMaxSteps - so_far - 1u) / 3u;
                    const uint32_t tx = rem_fx(rp) + offx - (dec_last == kRemDecX ? 1u : 0u), ty = rem_fy(rp) + offy - (dec_last == kRemDecY ? 1u : 0u),
                                   tz = rem_fz(rp) + offz - (dec_last == kRemDecZ ? 1u : 0u);
                    const uint32_t px = min(tx, min(cap, kFieldCapXZ)), py = min(ty, min(cap, kFieldCapY)), pz = min(tz, min(cap, kFieldCapXZ));
                    const uint32_t armed = px | (py << 11) | (pz << 21);
                    if (virt) {
                        cold[CF_OFF_XZ * 64] = (tx - px) | ((tz - pz) << 16);
                        cold[CF_OFF_Y * 64] = ty - py;
                    }
                    rem = virt ? armed : rem;
                    rp = virt ? armed + dec_last : rp;
                    rpp = virt ? armed + dec_last + dec_prev : rpp;
                    rem0 = virt ? so_far + px + py + pz : rem0;
                }
                const bool lin_inside = !((lin_lo < 0.0f) | (lin_hi > F));
                walk_on(W, is_fine, lin_inside);
                st = ST_WALK;
            } else {
                // steps counted by this walk, and its last counted step
                const bool last_counts = !hit && stepped && !region_fail;
                const uint32_t steps = rem0 - rem_sum(rp) + (last_counts ? 1u : 0u);
                const uint32_t dec_c = last_counts ? dec_last : dec_prev;  // the last counted step's axis
                if (STATS) {
                    // every iteration of the walk up to the cell `rp` was in range and probed its cell (:240-280): the start
                    // cell plus one per advance before rp; a walk that never started (start outside the grid) probed nothing
                    const uint32_t probed = (hit || stepped) ? rem0 - rem_sum(rp) + 1u : 0u;
                    cnt.fine_probes += is_fine ? probed : 0u;
                    cnt.coarse_probes += is_fine ? 0u : probed;
                }
                const f3 pc = mk3(last_counts ? cr.x : pc_prev.x, last_counts ? cr.y : pc_prev.y, last_counts ? cr.z : pc_prev.z);
                point.x = steps != 0u ? pc.x : ws.x;
                point.y = steps != 0u ? pc.y : ws.y;
                point.z = steps != 0u ? pc.z : ws.z;
                // ---- Raytrace's loop body after the walk (:395-511)
                const int total_ = (int)cold[CF_TOTAL * 64] + (int)steps;
                cold[CF_TOTAL * 64] = (uint32_t)total_;
                if (hit) {  // :493-506 (brick walks only)
                    const uint32_t box_codes = cold[CF_BOX_CODES * 64];
                    const uint32_t ray_codes = cold[CF_RAY_CODES * 64];
                    // normal code of the hit = (axis + 1) | 4 * negative of the last counted step; the coarse hit's if none
                    const uint32_t axis1 = dec_c == kRemDecX ? 1u : (dec_c == kRemDecY ? 2u : 3u);
                    const bool up_last = dec_c == kRemDecX ? d.x > 0 : (dec_c == kRemDecY ? d.y > 0 : d.z > 0);
                    const uint32_t hit_code = steps == 0u ? (box_codes & 7u) : (axis1 + (up_last ? 0u : 4u));
                    cold[CF_RAY_CODES * 64] = (ray_codes & ~0x78u) | (hit_code << 3) | 0x40u;
                    st = ST_DONE;
                } else if (!is_fine) {
                    st = ST_DONE;  // the coarse walk left the world (:399-401)
                } else {
                    // brick miss (:431-491): start = hitPosition / f; if start is still inside HitCell, nudge all three
                    // components one ulp along the ray, and if that is not enough snap one axis to NextCell
                    const int hx = (int)cold[CF_CHX * 64], hy = (int)cold[CF_CHY * 64], hz = (int)cold[CF_CHZ * 64];
                    const float fx = (float)hx, fy = (float)hy, fz = (float)hz;
                    const f3 hp = mk3(point.x + fx * W.ff, point.y + fy * W.ff, point.z + fz * W.ff);  // :426
                    float sx = hp.x * W.inv_f, sy = hp.y * W.inv_f, sz = hp.z * W.inv_f;
                    const bool nudge = trunc_equals(sx, fx) & trunc_equals(sy, fy) & trunc_equals(sz, fz);  // :441-444
                    if (__ballot(nudge) != 0ull) {
                        // nextafterf of an ordinary value (finite, not zero): the bits +- 1, by the sign of value and direction; the
                        // general form (zero, infinities, NaN) behind a wave vote
                        // (direction's sign bit = "towards negative" except for a -0 component: such rays are `special`)
                        auto ulp_fast = [](float v, float dir) {
                            const uint32_t b = __float_as_uint(v);
                            const uint32_t toward_zero = (b ^ __float_as_uint(dir)) >> 31;  // 1: the magnitude shrinks
                            return __uint_as_float((b + 1u) - (toward_zero + toward_zero));
                        };
                        float ux = ulp_fast(sx, d.x), uy = ulp_fast(sy, d.y), uz = ulp_fast(sz, d.z);
                        const uint32_t m_or = (__float_as_uint(sx) & 0x7FFFFFFFu) - 1u | (__float_as_uint(sy) & 0x7FFFFFFFu) - 1u |
                                              (__float_as_uint(sz) & 0x7FFFFFFFu) - 1u;  // >= 0x7F7FFFFF: a zero (wraps), an infinity or a NaN
                        if (__ballot(m_or >= 0x7F7FFFFFu || special) != 0ull) {
                            ux = ulp_step(sx, d.x < 0);
                            uy = ulp_step(sy, d.y < 0);
                            uz = ulp_step(sz, d.z < 0);
                        }
                        sx = nudge ? ux : sx;
                        sy = nudge ? uy : sy;
                        sz = nudge ? uz : sz;
                        const bool snap = nudge & trunc_equals(sx, fx) & trunc_equals(sy, fy) & trunc_equals(sz, fz);
                        if (snap) {
                            // NextCell (:347) = the UNCLAMPED coarse cell after the exit advance; it differs from the clamped
                            // HitCell by one when the walk started on a far face (edge rule)
                            const int nca = (int)(cold[CF_BOX_CODES * 64] >> 3);
                            const int axis = nca & 3;
                            const int ncx = hx + ((nca >> 2) & 1) + (axis == 0 ? (d.x > 0 ? 1 : -1) : 0);
                            const int ncy = hy + ((nca >> 3) & 1) + (axis == 1 ? (d.y > 0 ? 1 : -1) : 0);
                            const int ncz = hz + ((nca >> 4) & 1) + (axis == 2 ? (d.z > 0 ? 1 : -1) : 0);
                            const float gx = (float)ncx - sx, gy = (float)ncy - sy, gz = (float)ncz - sz;
                            const float mx = fabsf(gx), my = fabsf(gy), mz = fabsf(gz);
                            const bool snap_x = (mx < my) & (mx < mz);               // :475-486, in the reference's order
                            const bool snap_y = !snap_x & (my < mx) & (my < mz);
                            const bool snap_z = !snap_x & !snap_y;
                            sx = snap_x ? sx + gx : sx;
                            sy = snap_y ? sy + gy : sy;
                            sz = snap_z ? sz + gz : sz;
                        }
                    }
                    cold[CF_START_X * 64] = __float_as_uint(sx);
                    cold[CF_START_Y * 64] = __float_as_uint(sy);
                    cold[CF_START_Z * 64] = __float_as_uint(sz);
                    const bool restart = total_ < (int)(cold[CF_RAY_CODES * 64] >> 7);  // the while condition (:386)
                    if (restart) {
                        start_walk<false>(W, mk3(sx, sy, sz));
                        go_coarse = true;
                    } else {
                        st = ST_DONE;
                    }
                }
            }
        }
        // the level mask, where the wave is converged again
        fine_m &= ~__ballot(go_coarse);
    }

    // parked phase: tight-box test of an occupied coarse cell (:248-273) and, on a hit, the end of the coarse walk with
    // the entry into the cell's brick (:395-429).  Called by the whole wave; works on st == ST_BOX.
    template <bool STATS = false>
    __device__ __forceinline__ void phase_box(const WorldView& W)
    {
        bool go_fine = false;
        if (st == ST_BOX) {
            int x, y, z;
            cells_of(W, false, rp, x, y, z);  // the cell the probe found occupied
            const int qx = min(x, W.cx - 1), qy = min(y, W.cy - 1), qz = min(z, W.cz - 1);
            const uint32_t ci = cell_index(qx, qy, qz, W.c_row, W.cz);
            const uint2 meta = W.cell_meta[ci];
            const uint32_t e = meta.y;
            const float fqx = (float)qx, fqy = (float)qy, fqz = (float)qz;
            f3 bmin = mk3(((float)(e & 31u) + 0) * W.inv_f + fqx, ((float)((e >> 5) & 31u) + 0) * W.inv_f + fqy,
                          ((float)((e >> 10) & 31u) + 0) * W.inv_f + fqz);
            f3 bmax = mk3(((float)((e >> 15) & 31u) + 1) * W.inv_f + fqx, ((float)((e >> 20) & 31u) + 1) * W.inv_f + fqy,
                          ((float)((e >> 25) & 31u) + 1) * W.inv_f + fqz);
            f3 bp;
            uint32_t bc;
            bool box_hit = slab_fast(ws, bmin, bmax, bp, bc) && bmin.x <= bmax.x;
            if (__ballot(special) != 0ull) {
                if (special)
                    box_hit = slab(ws, bmin, bmax, bp, bc) && bmin.x <= bmax.x;
            }
            if (box_hit) {
                // the coarse walk ends here (:395-407): its steps, its HitIntersectedPoint (`step != 0`, :266)
                const uint32_t steps = rem0 - rem_sum(rp);
                cold[CF_TOTAL * 64] += steps;
                if (STATS)
                    cnt.coarse_probes += steps + 1u;  // the start cell and one cell per step, the hit cell included
                point.x = steps != 0u ? bp.x : ws.x;
                point.y = steps != 0u ? bp.y : ws.y;
                point.z = steps != 0u ? bp.z : ws.z;
                const bool enter = ci != cold[CF_LAST_CI * 64];  // not the previous_cell (:402-407)
                if (enter) {
                    // the exit iteration's extra advance (:290-322) only matters through NextCell: its axis (bits 0-1) and,
                    // per axis, whether the unclamped cell sits one past the clamped HitCell (bits 2-4; edge rule only)
                    const uint32_t dec = rp - rem;
                    const int axis = dec == kRemDecX ? 0 : (dec == kRemDecY ? 1 : 2);
                    const int packed = axis | ((x - qx) << 2) | ((y - qy) << 3) | ((z - qz) << 4);
                    if (STATS)
                        cnt.brick_entries += 1u;  // the descriptor load of :419-420
                    cold[CF_LAST_CI * 64] = ci;
                    cold[CF_CHX * 64] = (uint32_t)qx;
                    cold[CF_CHY * 64] = (uint32_t)qy;
                    cold[CF_CHZ * 64] = (uint32_t)qz;
                    cold[CF_BOX_CODES * 64] = bc | ((uint32_t)packed << 3);
                    // hitPosition = point * f, the brick walk starts at hitPosition - HitCell * f (:397, :409-411)
                    const f3 hp = mk3(point.x * W.ff, point.y * W.ff, point.z * W.ff);
                    const f3 ns = mk3(hp.x - fqx * W.ff, hp.y - fqy * W.ff, hp.z - fqz * W.ff);
                    bits = W.pool + (size_t)meta.x * W.brick_words - (W.f_slice >> 5);
                    start_walk<true>(W, ns);
                    go_fine = true;
                } else {
                    st = ST_DONE;
                }
            } else {
                // not a hit: the advance the probe made stands; it may have left the world
                const bool left = (rem & kRemGuards) != 0u;
                if (!left)
                    walk_on(W, false, true);
                st = left ? (uint32_t)ST_END : (uint32_t)ST_WALK;
            }
        }
        fine_m |= __ballot(go_fine);
    }

    // PAIRS x two probes without a vote between them.  Per probe: load the occupancy word of idx; advance every walking
    // lane (history, t, then tn / idx / rem of the chosen axis under that axis' mask); see who left the grid or passed
    // t_hi; when the word arrives, see who stood on an occupied cell.  The second probe's address of a pair does not depend
    // on the first word.  The walking mask is carried from probe to probe in scalar registers and `st` is written once,
    // after the last pair; `fix` can only be set by a phase, i.e. before the first pair.
    // GUARD (the probe-counting instantiations): classify every load address -- inside a table, in the slack the allocator
    // left around it (a lane that has just left its grid), or outside everything addressable (`stray`: must never happen; it
    // is what a world path that forgets the slack would produce, and tests/test_gpu_parity.py holds it at zero).
    // MASKED (the batch kernels): only walking lanes load their cell's word; the others load one fixed, hot word (the first
    // word of the coarse bits: one request per wave, always in cache) whose value is ignored -- an address select, not a branch
    // (a branch here would sit between the probes' wave-mask arithmetic).  The render kernels' parked lanes re-load the word
    // of their last probe, which their wave's coherent neighbours keep in cache; a batch has no coherence to rely on, and a
    // parked lane's load is a miss that competes with the walking lanes' (4 M incoherent rays: profiles/r04_batch_api.md).
    template <int PAIRS, bool GUARD = false, bool MASKED = false>
    __device__ __forceinline__ void probe_pairs(const WorldView& W)
    {
#ifdef VXRT_HOST_CHECK
        for (int p = 0; p < 2 * PAIRS; ++p) {
            if (st != ST_WALK)
                return;
            const uint32_t i1 = idx;
            const uint32_t word = bits[i1 >> 5];
            if (GUARD)
                guard_load(W, bits + (i1 >> 5));
            const bool a0 = tn_x < tn_y && tn_x < tn_z;
            const bool a1 = !(tn_x < tn_y) && tn_y < tn_z;
            tp = tl;
            rpp = rp;
            rp = rem;
            tl = lo(lo(tn_x, tn_y), tn_z);
            if (a0) {
                tn_x += fabsf(ivx);
                idx += di_x;
                rem -= kRemDecX;
            } else if (a1) {
                tn_y += fabsf(ivy);
                idx += di_y;
                rem -= kRemDecY;
            } else {
                tn_z += fabsf(ivz);
                idx += di_z;
                rem -= kRemDecZ;
            }
            if (p == 0) {
                idx -= fix;
                fix = 0u;
            }
            const bool sus = !(tl < t_hi), gd = WIDE && (rem & kRemGuards) != 0u;  // (ordinary grids: t_hi stops a lane that leaves)
            if (!WIDE && (rem & kRemGuards) != 0u && !sus)  // the invariant the GPU probe relies on, counted by the host harness
                host_unsuspected_exits() += 1;
            const bool solid = ((word >> (i1 & 31u)) & 1u) != 0u;
            if (solid)
                st = lane_fine() ? (uint32_t)ST_ENDHIT : (uint32_t)ST_BOX;
            else if (sus || gd)
                st = ST_END;
        }
#else
        const lanemask_t w0 = lane_mask(st == ST_WALK);
        lanemask_t w = w0, hits = 0ull;
#pragma unroll
        for (int k = 0; k < PAIRS; ++k) {
            lanemask_t sus1, gd1, sus2, gd2;
            // ---- probe 1
            const uint32_t i1 = idx;
            const uint32_t* a1 = bits + (i1 >> 5);
            if (MASKED)
                a1 = lane_test(w) ? a1 : W.coarse_bits;
            const uint32_t word1 = *a1;
            if (GUARD)
                guard_load(W, a1);
            advance<WIDE || VXRT_PROBE_GD>(w, sus1, gd1);
            if (k == 0) {
                idx -= fix;
                fix = 0u;
            }
            // ---- probe 2's load
            // (probe 2's load is issued before probe 1's word is known: MASKED limits it to the lanes that were walking into
            // the pair; a lane that stops in probe 1 still loads once more, inside the tables' slack at worst)
            const uint32_t i2 = idx;
            const uint32_t* a2 = bits + (i2 >> 5);
            if (MASKED)
                a2 = lane_test(w) ? a2 : W.coarse_bits;
            const uint32_t word2 = *a2;
            if (GUARD)
                guard_load(W, a2);
            // ---- probe 1: who stood on an occupied cell
            const lanemask_t h1 = lane_mask(bit_of(word1, i1) != 0u) & w;
            const lanemask_t w2 = w & ~(h1 | sus1 | gd1);
            advance<WIDE || VXRT_PROBE_GD>(w2, sus2, gd2);
            const lanemask_t h2 = lane_mask(bit_of(word2, i2) != 0u) & w2;
            hits |= h1 | h2;
            w = w2 & ~(h2 | sus2 | gd2);
        }
        // (who stopped without standing on an occupied cell: everyone who walked in and does not walk out, minus the hits -- a lane
        // that is beyond t_hi on an occupied cell counts as a hit)
        const lanemask_t other = (w0 & ~w) & ~hits;
        const lanemask_t park = hits & ~fine_m, lhit = hits & fine_m;
        // (exec is all ones here and in `advance`: the probes run where the wave is converged and every kernel that uses this
        // tracer launches whole wavefronts, so the asm restores -1 instead of saving and restoring the mask)
        asm volatile("s_mov_b64 exec, %[park]\n\t"
                     "v_mov_b32 %[st], 1\n\t"
                     "s_mov_b64 exec, %[lhit]\n\t"
                     "v_mov_b32 %[st], 3\n\t"
                     "s_mov_b64 exec, %[other]\n\t"
                     "v_mov_b32 %[st], 2\n\t"
                     "s_mov_b64 exec, -1"
                     : [st] "+v"(st)
                     : [park] "s"(park), [lhit] "s"(lhit), [other] "s"(other));
        static_assert(ST_BOX == 1u && ST_ENDHIT == 3u && ST_END == 2u, "state codes of the asm above");
#endif
    }

    __device__ __forceinline__ void guard_load(const WorldView& W, const uint32_t* a)
    {
        // (straight-line: a branch here would sit between the probes' wave-mask arithmetic)
        const uint32_t in_table = (uint32_t)(a >= W.coarse_bits) & (uint32_t)(a < W.coarse_end) | (uint32_t)(a >= W.pool) & (uint32_t)(a < W.pool_end);
        const uint32_t addressable = (uint32_t)(a >= W.coarse_lo) & (uint32_t)(a < W.coarse_hi) | (uint32_t)(a >= W.pool_lo) & (uint32_t)(a < W.pool_hi);
        cnt.slack_loads += (in_table ^ 1u) & addressable;
        cnt.stray_loads += addressable ^ 1u;
    }

#ifndef VXRT_HOST_CHECK
    // One speculative DDA advance (:293-322) of the lanes in `w`, in place.  Outputs (limited to w): sus = lanes whose t
    // is beyond t_hi, gd = lanes with a guard bit in rem (the advance left the grid).
    //
    // GD: also test rem's guard bits.  Only the wide-grid instantiation needs it (a virtual face is not where t_hi expects the
    // grid to end); on ordinary grids a lane that leaves its grid is always beyond t_hi (kThrEps*), which saves the probe one
    // fast and one slow vector instruction and the mask arithmetic behind them.
#define VXRT_ADVANCE_ASM(GUARD_PART)                                                                                     \
            "s_mov_b64 exec, %[w]\n\t"                                                                                   \
            /* compares of the three tMax under w: their masks are limited to w */                                        \
            "v_cmp_lt_f32 s[84:85], %[tx], %[ty]\n\t"                                                                    \
            "v_cmp_lt_f32 s[86:87], %[tx], %[tz]\n\t"                                                                    \
            "v_cmp_lt_f32 s[88:89], %[ty], %[tz]\n\t"                                                                    \
            /* history, then t of this advance and who is beyond the threshold (not less than: a start ON its exit   */   \
            /* face has t_hi = 0 and leaves with t = 0)                                                               */   \
            "v_mov_b32 %[tp], %[tl]\n\t"                                                                                 \
            "v_mov_b32 %[rpp], %[rp]\n\t"                                                                                \
            "v_mov_b32 %[rp], %[rem]\n\t"                                                                                \
            "v_min3_f32 %[tl], %[tx], %[ty], %[tz]\n\t"                                                                  \
            "v_cmp_nlt_f32 %[sus], %[tl], %[thi]\n\t"                                                                    \
            /* x: tx < ty && tx < tz -- the mask goes straight into exec */                                               \
            "s_and_b64 exec, s[84:85], s[86:87]\n\t"                                                                     \
            "v_add_f32 %[tx], %[tx], |%[ivx]|\n\t"                                                                       \
            "v_add_u32 %[idx], %[idx], %[dix]\n\t"                                                                       \
            "v_add_u32 %[rem], -1, %[rem]\n\t"                                                                           \
            /* the rest of w; y: !(tx < ty) && ty < tz */                                                                 \
            "s_andn2_b64 s[86:87], %[w], exec\n\t"                                                                       \
            "s_andn2_b64 exec, s[88:89], s[84:85]\n\t"                                                                   \
            "v_add_f32 %[ty], %[ty], |%[ivy]|\n\t"                                                                       \
            "v_add_u32 %[idx], %[idx], %[diy]\n\t"                                                                       \
            "v_add_u32 %[rem], 0xfffff800, %[rem]\n\t"                                                                   \
            /* z: the rest of w that is not y */                                                                          \
            "s_andn2_b64 exec, s[86:87], exec\n\t"                                                                       \
            "v_add_f32 %[tz], %[tz], |%[ivz]|\n\t"                                                                       \
            "v_add_u32 %[idx], %[idx], %[diz]\n\t"                                                                       \
            "v_add_u32 %[rem], 0xffe00000, %[rem]\n\t"                                                                   \
            GUARD_PART                                                                                                    \
            "s_mov_b64 exec, -1"
    template <bool GD>
    __device__ __forceinline__ void advance(const lanemask_t w, lanemask_t& sus, lanemask_t& gd)
    {
        if (!GD) {
            asm volatile(VXRT_ADVANCE_ASM("")
                         : [tx] "+v"(tn_x), [ty] "+v"(tn_y), [tz] "+v"(tn_z), [idx] "+v"(idx), [rem] "+v"(rem), [rp] "+v"(rp), [rpp] "+v"(rpp),
                           [tl] "+v"(tl), [tp] "+v"(tp), [sus] "=&s"(sus)
                         : [w] "s"(w), [ivx] "v"(ivx), [ivy] "v"(ivy), [ivz] "v"(ivz), [dix] "v"(di_x), [diy] "v"(di_y), [diz] "v"(di_z),
                           [thi] "v"(t_hi)
                         : "s84", "s85", "s86", "s87", "s88", "s89", "scc");
            gd = 0ull;
            return;
        }
        uint32_t tmp;
        asm volatile(VXRT_ADVANCE_ASM("s_mov_b64 exec, %[w]\n\t"
                                      "v_and_b32 %[tmp], 0x80100400, %[rem]\n\t"
                                      "v_cmp_ne_u32 %[gd], 0, %[tmp]\n\t")
                     : [tx] "+v"(tn_x), [ty] "+v"(tn_y), [tz] "+v"(tn_z), [idx] "+v"(idx), [rem] "+v"(rem), [rp] "+v"(rp), [rpp] "+v"(rpp),
                       [tl] "+v"(tl), [tp] "+v"(tp), [sus] "=&s"(sus), [gd] "=&s"(gd), [tmp] "=&v"(tmp)
                     : [w] "s"(w), [ivx] "v"(ivx), [ivy] "v"(ivy), [ivz] "v"(ivz), [dix] "v"(di_x), [diy] "v"(di_y), [diz] "v"(di_z),
                       [thi] "v"(t_hi)
                     : "s84", "s85", "s86", "s87", "s88", "s89", "scc");
    }
#undef VXRT_ADVANCE_ASM
#endif

    // Raytrace's epilogue (:514-523); the ray has ended (st == ST_DONE)
    __device__ __forceinline__ void result(const WorldView& W, TraceResult& out) const
    {
        const uint32_t ray_codes = cold[CF_RAY_CODES * 64];
        const bool hit = (ray_codes & 0x40u) != 0u;
        const int total_ = (int)cold[CF_TOTAL * 64];
        const int hx = (int)cold[CF_CHX * 64], hy = (int)cold[CF_CHY * 64], hz = (int)cold[CF_CHZ * 64];
        out.hit = hit;
        out.steps = total_;
        // hitPosition of the walk that ended the ray, as phase_end computed it (the same operands, the same operations)
        const bool is_fine = lane_fine();
        const float ox = (float)hx * W.ff, oy = (float)hy * W.ff, oz = (float)hz * W.ff;
        out.pos = mk3(is_fine ? point.x + ox : point.x * W.ff, is_fine ? point.y + oy : point.y * W.ff,
                      is_fine ? point.z + oz : point.z * W.ff);
        // brick HitCell = the clamped cell that was probed last (rp: the walk does not advance past a hit)
        int x, y, z;
        cells_of(W, true, rp, x, y, z);
        out.vx = hx * W.f + min(x, W.f - 1);
        out.vy = hy * W.f + min(y, W.f - 1);
        out.vz = hz * W.f + min(z, W.f - 1);
        const bool at_entry = hit && total_ == 0;
        out.ncode = at_entry ? (ray_codes & 7u) : ((ray_codes >> 3) & 7u);
        out.normal = normal_decode(out.ncode);
        if (at_entry)
            out.pos = mk3(__uint_as_float(cold[CF_START_X * 64]) * W.ff, __uint_as_float(cold[CF_START_Y * 64]) * W.ff,
                          __uint_as_float(cold[CF_START_Z * 64]) * W.ff);
    }
};

// one ray per lane, entered by the whole wave at a converged point (host check and the batch test kernel)
template <int PAIRS = 1, bool STATS = false, bool WIDE = false, bool MASKED = false>
__device__ inline void trace_wave2(const WorldView& W, const int max_steps, const bool active, const f3 origin, const f3 ray,
                                   TraceResult& out, uint32_t* cold_column, RayCounters* counters = nullptr)
{
    WaveTracerT<WIDE> T;
    T.init(W, cold_column);
    if (active)
        T.begin_ray(W, origin, ray, max_steps);
    T.after_begin_ray(active);
    for (;;) {
        const unsigned long long m_walk = __ballot(T.st == ST_WALK);
        const unsigned long long m_box = __ballot(T.st == ST_BOX);
        const unsigned long long m_end = __ballot(waits_for_end(T.st));
        if ((m_walk | m_box | m_end) == 0ull)
            break;
        const int n_walk = __popcll(m_walk), n_box = __popcll(m_box), n_end = __popcll(m_end);
        if (vote_run(n_end, n_walk + n_box, VXRT_VOTE_END))
            T.template phase_end<STATS>(W);
        if (vote_run(n_box, n_walk, VXRT_VOTE_BOX))
            T.template phase_box<STATS>(W);
        T.template probe_pairs<PAIRS, STATS, MASKED>(W);
    }
    if (active)
        T.result(W, out);
    if (STATS && counters) {
        counters->coarse_probes += T.cnt.coarse_probes;
        counters->brick_entries += T.cnt.brick_entries;
        counters->fine_probes += T.cnt.fine_probes;
        counters->slack_loads += T.cnt.slack_loads;
        counters->stray_loads += T.cnt.stray_loads;
    }
}

}  // namespace vxrt
