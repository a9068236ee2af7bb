// vxrt_wave.hpp -- the two-level brickmap DDA as a flat, wave-level state machine for gfx950.
//
// Same results as the straightforward form in vxrt_device.hpp (and therefore as Raytrace/DDARayTraversal,
// VoxelRT/VolumeRaytracer.cu:176-525), but organised for 64-wide wavefronts.  Profiling the straightforward
// kernel showed it is instruction-issue bound with the CU's single scalar unit as the hot spot (about one
// s_and_saveexec/s_cbranch/s_or per vector instruction from nested divergent loops), so:
//
//  * ONE loop.  Every iteration performs one cell probe + one DDA advance for every walking lane, whether it is
//    on the coarse grid or inside a brick; lanes on different levels share the same vector instructions.
//  * The hot path (WaveTracer::step) is straight-line and predicated: lane conditions are explicit wave masks
//    (ballot of a compare = the v_cmp result itself), combined on the scalar unit and handed back to v_cndmask by
//    an inverse ballot; values and the small state code are committed with v_cndmask.  No exec-mask branches; the
//    occupancy word of the (clamped) current cell is loaded unconditionally, one global_load_dword per lane per probe.
//  * The rare, expensive events -- tight-box slab test on an occupied coarse cell (ST_BOX) and the end-of-walk
//    transitions: brick entry, brick exit + re-seed with the ulp nudge, ray end (ST_END) -- PARK the lane.
//    A __ballot vote runs a parked phase only when enough lanes wait for it (or nobody can walk), so its
//    cost is amortised over many lanes; an all-zero __ballot is the wave's early-out.
//  * Per-ray invariants are hoisted: 1/(d or eps) of the slab test (VolumeRaytracer.cu:127-129) and |1/d| of
//    the DDA (:199-201) are the same IEEE quotients for every walk of a ray, so they are divided once per ray.
//  * The reference advances the DDA once more on the exit iteration only to produce NextCell, which is read
//    only after a coarse hit (:473); here that extra advance is reduced to picking its axis in the box phase.
//  * The 8-byte cell_meta record read for the slab test also carries the brick's pool slot, so brick entry
//    needs no second dependent load (the reference chases a 24-byte descriptor, then the brick pointer).
#pragma once

#include "vxrt_device.hpp"

namespace vxrt {

enum : uint32_t { ST_WALK = 0u, ST_BOX = 1u, ST_END = 2u, ST_DONE = 3u, ST_IDLE = 4u };
enum : uint32_t { WF_HIT = 1u, WF_OOB = 2u };

// unit normals as small codes: 0 = zero vector, (axis+1) | 4*negative
__device__ __forceinline__ f3 normal_decode(uint32_t c)
{
    float v = (c & 4u) ? -1.0f : 1.0f;
    uint32_t a = c & 3u;
    return mk3(a == 1u ? v : 0.0f, a == 2u ? v : 0.0f, a == 3u ? v : 0.0f);
}

// run a parked phase when its lanes are at least a quarter of the other live lanes (or nobody else can move)
#ifndef VXRT_VOTE_NUM
#define VXRT_VOTE_NUM 4
#endif
#ifndef VXRT_VOTE_DEN
#define VXRT_VOTE_DEN 1  // A/B knob: parked * num >= others * den
#endif
__device__ __forceinline__ bool vote_run(int parked, int others, int num = VXRT_VOTE_NUM)
{
    return parked > 0 && parked * num >= others * VXRT_VOTE_DEN;
}
// Per-phase thresholds, from a sweep on the bench workload: the expensive phases (ray finished ~450 VALU, end of
// walk ~130) wait until their lanes are a third of the live ones, the cheap tight-box test (~60) runs at a fifth.
// (2,2,4) against (4,4,4): +2 % at 1080p, +5 % at 4K; waiting longer (1) or running sooner (6..12) both lose.
#ifndef VXRT_STEPS_PER_ROUND
#define VXRT_STEPS_PER_ROUND 2  // probes per group between votes (vxrt_persist.hpp runs VXRT_SUBROUNDS groups per round)
#endif
#ifndef VXRT_VOTE_NEXT
#define VXRT_VOTE_NEXT 2
#endif
#ifndef VXRT_VOTE_END
#define VXRT_VOTE_END 2
#endif
#ifndef VXRT_VOTE_BOX
#define VXRT_VOTE_BOX 4
#endif

// The "cold" part of a lane's ray -- everything only the parked phases, begin_ray and result touch (the probes never
// do).  With LDS_COLD it lives in the wave's LDS block, one 64-lane column per field (conflict-free ds_read/ds_write,
// pairs of fields per ds_read2st64), and costs no registers between phases: that is what lets the kernel that uses it
// run 5 waves per SIMD (96 VGPRs) instead of 4.
// In LDS the small ones share a word: CF_RAY_CODES = entry_code (3 bits) | out_code << 3 | ray_hit << 6 | max_steps << 7,
// CF_BOX_CODES = c_code (3 bits) | nc_axis << 3 (both written by the tight-box phase only).
enum : int {
    CF_RAY_CODES = 0, CF_START_X, CF_START_Y, CF_START_Z, CF_LAST_CI, CF_TOTAL,
    CF_CHX, CF_CHY, CF_CHZ, CF_BOX_CODES, CF_SLOT, CF_C_CI, CF_TRACER_FIELDS
};

// One lane's ray: Raytrace-level state (:359-384), the current DDARayTraversal walk (:178-232) and the coarse
// results that outlive the coarse walk (:399-429,:438-488).  All members live in registers, except that with LDS_COLD
// the cold ones (marked) are never touched and their values live in LDS (`cold` = this lane's column).
template <bool STATS, bool MASKED_LOAD = false, bool LDS_COLD = false>
struct WaveTracer {
    // per-ray constants
    f3 d;                 // normalised direction
    float ivx, ivy, ivz;  // 1/(d or eps): the slab test's reciprocal (:127-129), and |iv| is the DDA's tDelta (:199-201)
    int max_steps;  // (cold)
    // Raytrace level (cold).  With LDS_COLD hitPosition (:397,:426) is not kept: when the ray has ended it is `point` in
    // voxel units, recomputed by result() with the same float operations on the same operands
    f3 start, hit_pos;
    uint32_t entry_code, last_ci, out_code;
    int total;
    bool ray_hit;
    // current walk
    uint32_t st, fine, wf, w_code, skip;
    f3 ws, point;
    int cell_x, cell_y, cell_z, lim_x, lim_y, lim_z;
    int dm1_x, dm1_y, dm1_z, row, slice;  // per-walk level constants kept in registers: dimension-1, cells per row / slice
    int up_x, up_y, up_z;              // per-ray: 1 where the direction component is positive (:195-197)
    float tn_x, tn_y, tn_z;
    int steps;  // stepsTaken; also the reference's loop index: an iteration continues exactly when a step is counted
    // coarse results kept across the brick walk (cold)
    int chx, chy, chz, nc_axis;
    uint32_t c_code, slot, c_ci;  // c_ci: cell index of the coarse HitCell (previous_cell compares it, :402-407)
    const uint32_t* bits;
    RayCounters cnt;
    uint32_t* cold;  // LDS_COLD: &block[lane]; field F of this lane is cold[F * 64]

    // a cold field: the register member, or its LDS cell
    __device__ __forceinline__ uint32_t cget(int f, uint32_t reg) const { return LDS_COLD ? cold[f * 64] : reg; }
    __device__ __forceinline__ int cget(int f, int reg) const { return LDS_COLD ? (int)cold[f * 64] : reg; }
    __device__ __forceinline__ float cget(int f, float reg) const { return LDS_COLD ? __uint_as_float(cold[f * 64]) : reg; }
    __device__ __forceinline__ bool cget(int f, bool reg) const { return LDS_COLD ? cold[f * 64] != 0u : reg; }
    __device__ __forceinline__ void cput(int f, uint32_t& reg, uint32_t v) { if (LDS_COLD) cold[f * 64] = v; else reg = v; }
    __device__ __forceinline__ void cput(int f, int& reg, int v) { if (LDS_COLD) cold[f * 64] = (uint32_t)v; else reg = v; }
    __device__ __forceinline__ void cput(int f, float& reg, float v) { if (LDS_COLD) cold[f * 64] = __float_as_uint(v); else reg = v; }
    __device__ __forceinline__ void cput(int f, bool& reg, bool v) { if (LDS_COLD) cold[f * 64] = v ? 1u : 0u; else reg = v; }

    __device__ __forceinline__ void init(const WorldView& W, uint32_t* cold_column = nullptr)
    {
        cold = cold_column;
        st = ST_DONE;
        fine = wf = w_code = skip = 0u;
        d = mk3(1.0f, 0.0f, 0.0f);
        ivx = ivy = ivz = 1.0f;
        max_steps = 0;
        start = ws = point = hit_pos = mk3(0, 0, 0);
        entry_code = out_code = 0u;
        last_ci = 0xFFFFFFFFu;
        total = 0;
        ray_hit = false;
        cell_x = cell_y = cell_z = 0;
        lim_x = lim_y = lim_z = 0;
        dm1_x = dm1_y = dm1_z = 0;
        row = W.c_row;
        slice = W.c_slice;
        up_x = 1;
        up_y = up_z = 0;
        tn_x = tn_y = tn_z = 0.0f;
        steps = 0;
        chx = chy = chz = nc_axis = 0;
        c_code = slot = c_ci = 0u;
        bits = W.coarse_bits;
        cnt = RayCounters{0, 0, 0};
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

    __device__ __forceinline__ void begin_walk(const WorldView& W, f3 s, uint32_t to_fine)
    {
        fine = to_fine;
        ws = s;
        cell_x = f2i(s.x);
        cell_y = f2i(s.y);
        cell_z = f2i(s.z);
        // (each division compiles into its own exec-mask branch; dividing unconditionally and selecting afterwards
        // was measured: no faster, more spills)
        tn_x = d.x != 0 ? ((float)(cell_x + up_x) - s.x) / d.x : kInf;
        tn_y = d.y != 0 ? ((float)(cell_y + up_y) - s.y) / d.y : kInf;
        tn_z = d.z != 0 ? ((float)(cell_z + up_z) - s.z) / d.z : kInf;
        point = s;
        steps = 0;
        wf = 0u;
        w_code = 0u;
        skip = 0u;
        // an occupied coarse cell always owns a brick (checked at upload, guaranteed by the device builder),
        // so a brick walk's dimensions are always f
        const int dmx = to_fine ? W.f : W.cx, dmy = to_fine ? W.f : W.cy, dmz = to_fine ? W.f : W.cz;
        const bool edge = cell_x == dmx || cell_y == dmy || cell_z == dmz;  // :216-232
        lim_x = dmx + ((edge && d.x < 0) ? 1 : 0);
        lim_y = dmy + ((edge && d.y < 0) ? 1 : 0);
        lim_z = dmz + ((edge && d.z < 0) ? 1 : 0);
        dm1_x = dmx - 1;
        dm1_y = dmy - 1;
        dm1_z = dmz - 1;
        row = to_fine ? W.f_row : W.c_row;
        slice = to_fine ? W.f_slice : W.c_slice;
    }

    // Raytrace's prologue (:359-384): per-ray constants, world entry, first coarse walk
    __device__ __forceinline__ void begin_ray(const WorldView& W, f3 origin, f3 ray, int max_steps_)
    {
        d = unit3(ray);
        ivx = 1.0f / (d.x == 0 ? kFltEps : d.x);  // :127-129
        ivy = 1.0f / (d.y == 0 ? kFltEps : d.y);
        ivz = 1.0f / (d.z == 0 ? kFltEps : d.z);
        // tDelta = |1/d|, or inf for d == 0 (:199-201), is not kept: where d != 0 it is |iv| (the same quotient), and an
        // axis with d == 0 has tMax = inf, is never the smallest, and inf + anything stays inf -- so `tn + |iv|` is
        // the reference's `tMax + tDelta` in every case
        up_x = d.x > 0 ? 1 : 0;
        up_y = d.y > 0 ? 1 : 0;
        up_z = d.z > 0 ? 1 : 0;
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
        if (LDS_COLD)
            cold[CF_RAY_CODES * 64] = ec | ((uint32_t)max_steps_ << 7);  // out_code = 0, ray_hit = false
        else
            max_steps = max_steps_;
        cput(CF_START_X, start.x, s0.x);
        cput(CF_START_Y, start.y, s0.y);
        cput(CF_START_Z, start.z, s0.z);
        if (!LDS_COLD)
            entry_code = ec;
        cput(CF_LAST_CI, last_ci, 0xFFFFFFFFu);  // previous_cell as its cell index (unique per cell); none yet
        cput(CF_TOTAL, total, 0);
        if (!LDS_COLD) {
            ray_hit = false;
            out_code = 0u;
        }
        bits = W.coarse_bits;
        begin_walk(W, s0, 0u);
        st = ST_WALK;
    }

    // The same start from a PREPARED ray (prepare_ray below, evaluated by the shading kernels of vxrt_ts.hpp at full lane
    // occupancy): Raytrace's prologue and the first walk's three quotients arrive as 13 words, so starting a ray costs
    // this kernel no division, no square root and no world-entry test.  LDS_COLD builds only.
    //   a = {d.x, d.y, d.z, iv.x}  b = {iv.y, iv.z, start.x, start.y}  c = {start.z, tMax.x, tMax.y, tMax.z}
    //   codes = entry normal code | maxSteps << 7 (the CF_RAY_CODES word)
    __device__ __forceinline__ void begin_prepared(const WorldView& W, const uint4 a, const uint4 b, const uint4 c, const uint32_t codes)
    {
        static_assert(LDS_COLD, "prepared rays start LDS_COLD tracers");
        d = mk3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
        ivx = __uint_as_float(a.w);
        ivy = __uint_as_float(b.x);
        ivz = __uint_as_float(b.y);
        up_x = d.x > 0 ? 1 : 0;
        up_y = d.y > 0 ? 1 : 0;
        up_z = d.z > 0 ? 1 : 0;
        const f3 s0 = mk3(__uint_as_float(b.z), __uint_as_float(b.w), __uint_as_float(c.x));
        cold[CF_RAY_CODES * 64] = codes;
        cold[CF_START_X * 64] = b.z;
        cold[CF_START_Y * 64] = b.w;
        cold[CF_START_Z * 64] = c.x;
        cold[CF_LAST_CI * 64] = 0xFFFFFFFFu;
        cold[CF_TOTAL * 64] = 0u;
        bits = W.coarse_bits;
        // begin_walk(W, s0, 0) with its quotients from the record
        fine = 0u;
        ws = s0;
        cell_x = f2i(s0.x);
        cell_y = f2i(s0.y);
        cell_z = f2i(s0.z);
        tn_x = __uint_as_float(c.y);
        tn_y = __uint_as_float(c.z);
        tn_z = __uint_as_float(c.w);
        point = s0;
        steps = 0;
        wf = 0u;
        w_code = 0u;
        skip = 0u;
        const bool edge = cell_x == W.cx || cell_y == W.cy || cell_z == W.cz;  // :216-232
        lim_x = W.cx + ((edge && d.x < 0) ? 1 : 0);
        lim_y = W.cy + ((edge && d.y < 0) ? 1 : 0);
        lim_z = W.cz + ((edge && d.z < 0) ? 1 : 0);
        dm1_x = W.cx - 1;
        dm1_y = W.cy - 1;
        dm1_z = W.cz - 1;
        row = W.c_row;
        slice = W.c_slice;
        st = ST_WALK;
    }

    // parked phase: end of a walk (:395-511); call for lanes with st == ST_END
    __device__ __forceinline__ void phase_end(const WorldView& W)
    {
        // Straight-line selects up to the shared begin_walk: the lanes of one execution are a mix of all four cases
        // (brick entry, coarse walk over, brick hit, brick miss), so every branch would be taken by somebody, and
        // the branchy form (three nested short-circuit compares, three ulp_steps with early returns, ...) paid
        // ~40 exec-mask round trips per execution for ~60 instructions of arithmetic.
        const int total_ = cget(CF_TOTAL, total) + steps;
        cput(CF_TOTAL, total, total_);
        const bool is_fine = fine != 0u;
        const int hx = cget(CF_CHX, chx), hy = cget(CF_CHY, chy), hz = cget(CF_CHZ, chz);
        const float fx = (float)hx, fy = (float)hy, fz = (float)hz;
        const float ox = fx * W.ff, oy = fy * W.ff, oz = fz * W.ff;  // the brick's origin in voxels
        // coarse walk: hitPosition = point * f (:397); brick walk: point + HitCell * f (:426)
        const f3 hp = mk3(is_fine ? point.x + ox : point.x * W.ff, is_fine ? point.y + oy : point.y * W.ff,
                          is_fine ? point.z + oz : point.z * W.ff);
        if (!LDS_COLD) {
            hit_pos.x = hp.x;
            hit_pos.y = hp.y;
            hit_pos.z = hp.z;
        }
        const uint32_t ci = cget(CF_C_CI, c_ci);  // (computed by the tight-box phase that produced this coarse hit)
        const uint32_t last_ = cget(CF_LAST_CI, last_ci);
        // coarse walk ended on an occupied cell that is not the previous_cell (:399-407): enter its brick
        const bool enter = !is_fine && wf == WF_HIT && ci != last_;
        const bool fine_hit = is_fine && (wf & WF_HIT) != 0u;  // :493-506
        const bool fine_miss = is_fine && (wf & WF_HIT) == 0u;  // restart the coarse walk just past the brick (:431-491)
        cput(CF_LAST_CI, last_ci, enter ? ci : last_);
        if (STATS)
            cnt.brick_entries += enter ? 1u : 0u;
        // w_code = axis + 1 of the walk's last counted step; normal code = (axis+1) | 4*negative
        // (selects against 0, OR-ed: a chained `?:` over the three members is the selected-address trap, which
        // demoted the whole tracer to scratch memory and cost 60 % of the frame rate)
        const int up_last = (w_code == 1u ? up_x : 0) | (w_code == 2u ? up_y : 0) | (w_code == 3u ? up_z : 0);
        const uint32_t box_codes = LDS_COLD ? cold[CF_BOX_CODES * 64] : 0u;
        const uint32_t ray_codes = LDS_COLD ? cold[CF_RAY_CODES * 64] : 0u;
        const uint32_t hit_code = (steps == 0) ? (LDS_COLD ? (box_codes & 7u) : c_code) : (w_code + 4u - 4u * (uint32_t)up_last);
        if (LDS_COLD) {
            cold[CF_RAY_CODES * 64] = fine_hit ? ((ray_codes & ~0x78u) | (hit_code << 3) | 0x40u) : ray_codes;
        } else {
            out_code = fine_hit ? hit_code : out_code;
            ray_hit = fine_hit ? true : ray_hit;
        }
        // brick miss: start = hitPosition / f; if the brick walk left the brick and start is still inside HitCell,
        // nudge all three components one ulp along the ray, and if that is not enough snap one axis to NextCell
        float sx = hp.x * W.inv_f, sy = hp.y * W.inv_f, sz = hp.z * W.inv_f;
        const bool same1 = trunc_equals(sx, fx) & trunc_equals(sy, fy) & trunc_equals(sz, fz);  // HitCell.x == (int)start.x ..., :441-444
        const bool nudge = fine_miss & ((wf & WF_OOB) != 0u) & same1;
        const float ux = ulp_step(sx, d.x < 0), uy = ulp_step(sy, d.y < 0), uz = ulp_step(sz, d.z < 0);
        sx = nudge ? ux : sx;
        sy = nudge ? uy : sy;
        sz = nudge ? uz : sz;
        const bool snap = nudge & trunc_equals(sx, fx) & trunc_equals(sy, fy) & trunc_equals(sz, fz);
        // The snap is rare (the exit point must lie strictly inside HitCell and survive the one-ulp nudge there): one vote
        // lets the whole wave skip NextCell and the three-way comparison (~35 instructions of this phase) when no lane
        // of the execution needs them
        if (__ballot(snap) != 0ull) {
            // NextCell (:347) = the UNCLAMPED coarse cell after the exit advance; it differs from the clamped HitCell by
            // one when the walk started on a far face (edge rule)
            const int nca = LDS_COLD ? (int)(box_codes >> 3) : nc_axis;
            const int axis = nca & 3;
            const int ncx = hx + ((nca >> 2) & 1) + (axis == 0 ? 2 * up_x - 1 : 0);
            const int ncy = hy + ((nca >> 3) & 1) + (axis == 1 ? 2 * up_y - 1 : 0);
            const int ncz = hz + ((nca >> 4) & 1) + (axis == 2 ? 2 * up_z - 1 : 0);
            const float gx = (float)ncx - sx, gy = (float)ncy - sy, gz = (float)ncz - sz;
            const float mx = fabsf(gx), my = fabsf(gy), mz = fabsf(gz);
            const bool snap_x = (mx < my) & (mx < mz);               // :475-486, in the reference's order
            const bool snap_y = !snap_x & (my < mx) & (my < mz);
            const bool snap_z = !snap_x & !snap_y;
            sx = (snap & snap_x) ? sx + gx : sx;
            sy = (snap & snap_y) ? sy + gy : sy;
            sz = (snap & snap_z) ? sz + gz : sz;
        }
        if (LDS_COLD) {
            if (fine_miss) {  // (plain stores under one exec mask: no read-modify-write of the three cells)
                cold[CF_START_X * 64] = __float_as_uint(sx);
                cold[CF_START_Y * 64] = __float_as_uint(sy);
                cold[CF_START_Z * 64] = __float_as_uint(sz);
            }
        } else {
            start.x = fine_miss ? sx : start.x;
            start.y = fine_miss ? sy : start.y;
            start.z = fine_miss ? sz : start.z;
        }
        const bool restart = fine_miss && total_ < (LDS_COLD ? (int)(ray_codes >> 7) : max_steps);  // the while condition, checked only here (:386)
        const bool go = enter | restart;
        // both continuations (enter the brick / restart the coarse walk) share ONE begin_walk: its three IEEE
        // divisions are the bulk of this phase
        const uint32_t* const brick = W.pool + (size_t)cget(CF_SLOT, slot) * W.brick_words;
        bits = enter ? brick : (restart ? W.coarse_bits : bits);
        st = go ? (uint32_t)ST_WALK : (uint32_t)ST_DONE;
        if (go) {
            const f3 ns = mk3(enter ? hp.x - ox : sx, enter ? hp.y - oy : sy, enter ? hp.z - oz : sz);
            begin_walk(W, ns, enter ? 1u : 0u);
        }
    }

    // parked phase: tight-box test of an occupied coarse cell (:248-273); call for lanes with st == ST_BOX
    __device__ __forceinline__ void phase_box(const WorldView& W)
    {
        const int qx = min(cell_x, W.cx - 1), qy = min(cell_y, W.cy - 1), qz = min(cell_z, W.cz - 1);
        const uint32_t idx = cell_index(qx, qy, qz, W.c_row, W.c_slice);
        const uint2 meta = W.cell_meta[idx];
        const uint32_t e = meta.y;
        f3 bmin = mk3(((float)(e & 31u) + 0) * W.inv_f + (float)qx, ((float)((e >> 5) & 31u) + 0) * W.inv_f + (float)qy,
                      ((float)((e >> 10) & 31u) + 0) * W.inv_f + (float)qz);
        f3 bmax = mk3(((float)((e >> 15) & 31u) + 1) * W.inv_f + (float)qx, ((float)((e >> 20) & 31u) + 1) * W.inv_f + (float)qy,
                      ((float)((e >> 25) & 31u) + 1) * W.inv_f + (float)qz);
        f3 bp;
        uint32_t bc;
        const bool box_hit = slab(ws, bmin, bmax, bp, bc) && bmin.x <= bmax.x;
        // All updates as selects.  (Written as `if (hit) { wf = 1; ... } else { skip = 1; }` the optimiser merges
        // the two stores of the constant 1 into ONE store through a selected address, which demotes these members
        // from registers to scratch memory -- seen as scratch_store/scratch_load inside the hot loop.)
        wf = box_hit ? (uint32_t)WF_HIT : wf;
        skip = box_hit ? 0u : 1u;  // not a hit: walk on without probing this cell again
        st = box_hit ? (uint32_t)ST_END : (uint32_t)ST_WALK;
        const bool move_point = box_hit && steps != 0;  // `step != 0`, :266
        point.x = move_point ? bp.x : point.x;  // per component: a struct-valued ?: selects an ADDRESS (same trap)
        point.y = move_point ? bp.y : point.y;
        point.z = move_point ? bp.z : point.z;
        if (!LDS_COLD) {
            chx = box_hit ? qx : chx;
            chy = box_hit ? qy : chy;
            chz = box_hit ? qz : chz;
            c_code = box_hit ? bc : c_code;
            slot = box_hit ? meta.x : slot;
            c_ci = box_hit ? idx : c_ci;
        }
        // the exit iteration's extra advance (:290-322) only matters through NextCell: keep its axis (bits 0-1)
        // and, per axis, whether the unclamped cell sits one past the clamped HitCell (bits 2-4; edge rule only)
        const int axis = (tn_x < tn_y && tn_x < tn_z) ? 0 : ((tn_y <= tn_x && tn_y < tn_z) ? 1 : 2);
        const int packed = axis | ((cell_x - qx) << 2) | ((cell_y - qy) << 3) | ((cell_z - qz) << 4);
        if (!LDS_COLD)
            nc_axis = box_hit ? packed : nc_axis;
        if (LDS_COLD && box_hit) {  // (the phase runs under the ST_BOX lanes' exec mask anyway: plain stores)
            cold[CF_CHX * 64] = (uint32_t)qx;
            cold[CF_CHY * 64] = (uint32_t)qy;
            cold[CF_CHZ * 64] = (uint32_t)qz;
            cold[CF_BOX_CODES * 64] = bc | ((uint32_t)packed << 3);
            cold[CF_SLOT * 64] = meta.x;
            cold[CF_C_CI * 64] = idx;
        }
    }

    // hot path: probe the current cell and advance, predicated on st == ST_WALK; executed by every lane.  Lane
    // conditions are explicit wave masks (lane_mask / lane_test, vxrt_device.hpp): each compare is one v_cmp whose
    // result IS the mask, all the logic between them runs on the scalar unit, and values are committed with
    // v_cndmask -- no exec-mask branches and no 0/1 integers in vector registers.  (Written with bool & | the
    // compiler still materialised several conditions as 0/1 vector integers; with && || it built branches.)
    __device__ __forceinline__ void step(const WorldView& W)
    {
        const lanemask_t w = lane_mask(st == ST_WALK);
        // 0 <= cell < dim + pad on all three axes: unsigned compares (a negative cell is a huge unsigned)
        const lanemask_t in = lane_mask((uint32_t)cell_x < (uint32_t)lim_x) & lane_mask((uint32_t)cell_y < (uint32_t)lim_y) &
                              lane_mask((uint32_t)cell_z < (uint32_t)lim_z);
        // lookups use the cell clamped to dim-1 (:242-244; matters only under the edge rule); clamped at 0 as well, so
        // that the unconditional load below has a valid address for an out-of-range lane too (its bit is ignored)
        const uint32_t idx = cell_index(clamp_cell(cell_x, dm1_x), clamp_cell(cell_y, dm1_y), clamp_cell(cell_z, dm1_z), row, slice);
        // MASKED_LOAD: only walking lanes load.  In the render kernels the load is unconditional (parked lanes re-read
        // their last word from L1/L2; the exec-mask branch around the load costs 3.6 % of the frame rate); for a batch
        // of incoherent rays, where every request is an HBM miss, masking it is worth +23 % (tools/batch_probe.py)
        uint32_t word = 0u;
        if (!MASKED_LOAD || st == ST_WALK)
            word = bits[idx >> 5];
        const lanemask_t is_fine = lane_mask(fine != 0u), skipping = lane_mask(skip != 0u);
        const lanemask_t solid = lane_mask(((word >> (idx & 31u)) & 1u) != 0u) & ~skipping;
        if (STATS) {
            const lanemask_t probed = w & in & ~skipping;
            cnt.fine_probes += lane_test(probed & is_fine) ? 1u : 0u;
            cnt.coarse_probes += lane_test(probed & ~is_fine) ? 1u : 0u;
        }
        const lanemask_t leave_oob = w & ~in;                       // left the grid / brick: isOutOfBounds (:283-287)
        const lanemask_t leave_hit = w & in & solid & is_fine;      // solid voxel inside a brick (:276-280)
        const lanemask_t park = w & in & solid & ~is_fine;          // occupied coarse cell: tight-box test pending
        const lanemask_t adv = w & in & ~solid;
        skip = lane_test(w) ? 0u : skip;

        // DDA advance (:293-322), computed for every lane, committed where adv
        const lanemask_t lt_xy = lane_mask(tn_x < tn_y), lt_xz = lane_mask(tn_x < tn_z), lt_yz = lane_mask(tn_y < tn_z);
        const lanemask_t ax0 = lt_xy & lt_xz;
        const lanemask_t ax1 = ~lt_xy & lt_yz;  // tn_y <= tn_x && tn_y < tn_z; implies !ax0
        const lanemask_t ax2 = ~(ax0 | ax1);
        const bool on0 = lane_test(ax0), on1 = lane_test(ax1), on2 = lane_test(ax2);
        const float t = on0 ? tn_x : (on1 ? tn_y : tn_z);
        const float crx = on0 ? (float)(cell_x + up_x) : ws.x + (t * d.x);
        const float cry = on1 ? (float)(cell_y + up_y) : ws.y + (t * d.y);
        const float crz = on2 ? (float)(cell_z + up_z) : ws.z + (t * d.z);
        // region check on the crossing point, brick walks only (:325-341): [0,f]^3, step not counted
        const float cmin = fminf(fminf(crx, cry), crz), cmax = fmaxf(fmaxf(crx, cry), crz);
        const lanemask_t region_oob = (lane_mask(cmin < 0.0f) | lane_mask(cmax > W.ff)) & is_fine & adv;
        const lanemask_t ok = adv & ~region_oob;
        const bool c0 = lane_test(adv & ax0), c1 = lane_test(adv & ax1), c2 = lane_test(adv & ax2), counted = lane_test(ok);
        cell_x += c0 ? 2 * up_x - 1 : 0;
        cell_y += c1 ? 2 * up_y - 1 : 0;
        cell_z += c2 ? 2 * up_z - 1 : 0;
        tn_x = c0 ? tn_x + fabsf(ivx) : tn_x;
        tn_y = c1 ? tn_y + fabsf(ivy) : tn_y;
        tn_z = c2 ? tn_z + fabsf(ivz) : tn_z;
        w_code = counted ? (on0 ? 1u : (on1 ? 2u : 3u)) : w_code;  // the axis; its sign is applied at the end of the walk
        point.x = counted ? crx : point.x;
        point.y = counted ? cry : point.y;
        point.z = counted ? crz : point.z;
        steps += counted ? 1 : 0;
        const lanemask_t exhausted = ok & lane_mask(steps >= kMaxSteps);  // walk ran out of iterations (:234)
        wf = lane_test(leave_hit) ? (wf | (uint32_t)WF_HIT) : wf;
        wf = lane_test(leave_oob | region_oob) ? (wf | (uint32_t)WF_OOB) : wf;
        const lanemask_t ending = leave_oob | leave_hit | region_oob | exhausted;
        st = lane_test(park) ? (uint32_t)ST_BOX : (lane_test(ending) ? (uint32_t)ST_END : st);
    }

    // Two probes with both occupancy loads in flight together.  The cell of the second probe is known before the first
    // word arrives -- a DDA's path does not depend on the voxels, only where it stops does -- so its load is issued ahead of
    // the first probe's `s_waitcnt` instead of behind it (the straightforward pair exposed the whole latency of the second
    // load: the compiler had ~8 independent instructions to put between that load and its use).  A lane that does not
    // advance in the first probe is not walking in the second, so the speculative word is simply unused.  Same results as
    // step(); step(): 3 more vector instructions per pair (the next cell is computed, then committed by select).
    __device__ __forceinline__ void step2(const WorldView& W)
    {
#if !defined(VXRT_HOST_CHECK) && !defined(VXRT_NO_ENTRY_WAIT)
        // Start from a KNOWN memory scoreboard.  The pair is entered from every phase's exit, some with stores or loads still
        // outstanding; at such a join the compiler's wait insertion no longer knows how many, and the first use of word1
        // got `s_waitcnt vmcnt(0)` -- waiting for word2 as well, the very latency the pair is built to hide.  With
        // everything older drained here (it would be waited for a few instructions later anyway), word1 waits with vmcnt(1).
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
#endif
        // ---- probe 1: address, load
        const lanemask_t w1 = lane_mask(st == ST_WALK);
        const lanemask_t in1 = lane_mask((uint32_t)cell_x < (uint32_t)lim_x) & lane_mask((uint32_t)cell_y < (uint32_t)lim_y) &
                               lane_mask((uint32_t)cell_z < (uint32_t)lim_z);
        const uint32_t idx1 = cell_index(clamp_cell(cell_x, dm1_x), clamp_cell(cell_y, dm1_y), clamp_cell(cell_z, dm1_z), row, slice);
        uint32_t word1 = 0u;
        if (!MASKED_LOAD || st == ST_WALK)
            word1 = bits[idx1 >> 5];
        const lanemask_t is_fine = lane_mask(fine != 0u), skipping = lane_mask(skip != 0u);
        // ---- probe 1: the DDA advance, as far as it does not need the word (:293-322)
        const lanemask_t lt_xy = lane_mask(tn_x < tn_y), lt_xz = lane_mask(tn_x < tn_z), lt_yz = lane_mask(tn_y < tn_z);
        const lanemask_t ax0 = lt_xy & lt_xz, ax1 = ~lt_xy & lt_yz, ax2 = ~(ax0 | ax1);
        const bool on0 = lane_test(ax0), on1 = lane_test(ax1), on2 = lane_test(ax2);
        const float t = on0 ? tn_x : (on1 ? tn_y : tn_z);
        const float crx = on0 ? (float)(cell_x + up_x) : ws.x + (t * d.x);
        const float cry = on1 ? (float)(cell_y + up_y) : ws.y + (t * d.y);
        const float crz = on2 ? (float)(cell_z + up_z) : ws.z + (t * d.z);
        const float cmin = fminf(fminf(crx, cry), crz), cmax = fmaxf(fmaxf(crx, cry), crz);
        const lanemask_t outside1 = (lane_mask(cmin < 0.0f) | lane_mask(cmax > W.ff)) & is_fine;
        // the cell the lane is in next if it advances, and ITS load
        const int nx = cell_x + (on0 ? 2 * up_x - 1 : 0), ny = cell_y + (on1 ? 2 * up_y - 1 : 0), nz = cell_z + (on2 ? 2 * up_z - 1 : 0);
        const lanemask_t in2 = lane_mask((uint32_t)nx < (uint32_t)lim_x) & lane_mask((uint32_t)ny < (uint32_t)lim_y) &
                               lane_mask((uint32_t)nz < (uint32_t)lim_z);
        const uint32_t idx2 = cell_index(clamp_cell(nx, dm1_x), clamp_cell(ny, dm1_y), clamp_cell(nz, dm1_z), row, slice);
        uint32_t word2 = 0u;
        if (!MASKED_LOAD || st == ST_WALK)
            word2 = bits[idx2 >> 5];
        // ---- probe 1: decision and commit
        {
            const lanemask_t solid = lane_mask(((word1 >> (idx1 & 31u)) & 1u) != 0u) & ~skipping;
            if (STATS) {
                const lanemask_t probed = w1 & in1 & ~skipping;
                cnt.fine_probes += lane_test(probed & is_fine) ? 1u : 0u;
                cnt.coarse_probes += lane_test(probed & ~is_fine) ? 1u : 0u;
            }
            const lanemask_t leave_oob = w1 & ~in1, leave_hit = w1 & in1 & solid & is_fine, park = w1 & in1 & solid & ~is_fine;
            const lanemask_t adv = w1 & in1 & ~solid;
            skip = lane_test(w1) ? 0u : skip;
            const lanemask_t region_oob = outside1 & adv, ok = adv & ~region_oob;
            const bool moved = lane_test(adv), counted = lane_test(ok);
            cell_x = moved ? nx : cell_x;
            cell_y = moved ? ny : cell_y;
            cell_z = moved ? nz : cell_z;
            tn_x = lane_test(adv & ax0) ? tn_x + fabsf(ivx) : tn_x;
            tn_y = lane_test(adv & ax1) ? tn_y + fabsf(ivy) : tn_y;
            tn_z = lane_test(adv & ax2) ? tn_z + fabsf(ivz) : tn_z;
            w_code = counted ? (on0 ? 1u : (on1 ? 2u : 3u)) : w_code;
            point.x = counted ? crx : point.x;
            point.y = counted ? cry : point.y;
            point.z = counted ? crz : point.z;
            steps += counted ? 1 : 0;
            const lanemask_t exhausted = ok & lane_mask(steps >= kMaxSteps);
            wf = lane_test(leave_hit) ? (wf | (uint32_t)WF_HIT) : wf;
            wf = lane_test(leave_oob | region_oob) ? (wf | (uint32_t)WF_OOB) : wf;
            const lanemask_t ending = leave_oob | leave_hit | region_oob | exhausted;
            st = lane_test(park) ? (uint32_t)ST_BOX : (lane_test(ending) ? (uint32_t)ST_END : st);
        }
        // ---- probe 2: a lane still walking advanced in probe 1, so (nx,ny,nz) is its cell and word2 its word
        {
            const lanemask_t w = lane_mask(st == ST_WALK);
            const lanemask_t solid = lane_mask(((word2 >> (idx2 & 31u)) & 1u) != 0u);  // (skip is clear for every walking lane)
            if (STATS) {
                const lanemask_t probed = w & in2;
                cnt.fine_probes += lane_test(probed & is_fine) ? 1u : 0u;
                cnt.coarse_probes += lane_test(probed & ~is_fine) ? 1u : 0u;
            }
            const lanemask_t leave_oob = w & ~in2, leave_hit = w & in2 & solid & is_fine, park = w & in2 & solid & ~is_fine;
            const lanemask_t adv = w & in2 & ~solid;
            const lanemask_t l_xy = lane_mask(tn_x < tn_y), l_xz = lane_mask(tn_x < tn_z), l_yz = lane_mask(tn_y < tn_z);
            const lanemask_t b0 = l_xy & l_xz, b1 = ~l_xy & l_yz, b2 = ~(b0 | b1);
            const bool o0 = lane_test(b0), o1 = lane_test(b1), o2 = lane_test(b2);
            const float t2 = o0 ? tn_x : (o1 ? tn_y : tn_z);
            const float c2x = o0 ? (float)(cell_x + up_x) : ws.x + (t2 * d.x);
            const float c2y = o1 ? (float)(cell_y + up_y) : ws.y + (t2 * d.y);
            const float c2z = o2 ? (float)(cell_z + up_z) : ws.z + (t2 * d.z);
            const float mn = fminf(fminf(c2x, c2y), c2z), mx = fmaxf(fmaxf(c2x, c2y), c2z);
            const lanemask_t region_oob = (lane_mask(mn < 0.0f) | lane_mask(mx > W.ff)) & is_fine & adv;
            const lanemask_t ok = adv & ~region_oob;
            const bool c0 = lane_test(adv & b0), c1 = lane_test(adv & b1), c2 = lane_test(adv & b2), counted = lane_test(ok);
            cell_x += c0 ? 2 * up_x - 1 : 0;
            cell_y += c1 ? 2 * up_y - 1 : 0;
            cell_z += c2 ? 2 * up_z - 1 : 0;
            tn_x = c0 ? tn_x + fabsf(ivx) : tn_x;
            tn_y = c1 ? tn_y + fabsf(ivy) : tn_y;
            tn_z = c2 ? tn_z + fabsf(ivz) : tn_z;
            w_code = counted ? (o0 ? 1u : (o1 ? 2u : 3u)) : w_code;
            point.x = counted ? c2x : point.x;
            point.y = counted ? c2y : point.y;
            point.z = counted ? c2z : point.z;
            steps += counted ? 1 : 0;
            const lanemask_t exhausted = ok & lane_mask(steps >= kMaxSteps);
            wf = lane_test(leave_hit) ? (wf | (uint32_t)WF_HIT) : wf;
            wf = lane_test(leave_oob | region_oob) ? (wf | (uint32_t)WF_OOB) : wf;
            const lanemask_t ending = leave_oob | leave_hit | region_oob | exhausted;
            st = lane_test(park) ? (uint32_t)ST_BOX : (lane_test(ending) ? (uint32_t)ST_END : st);
        }
    }

    // one group of VXRT_STEPS_PER_ROUND probes between two votes
    __device__ __forceinline__ void probe_group(const WorldView& W)
    {
#if VXRT_STEPS_PER_ROUND == 2 && !defined(VXRT_NO_STEP2)
        step2(W);
#else
        for (int s = 0; s < VXRT_STEPS_PER_ROUND; ++s)
            step(W);
#endif
    }

    // Raytrace's epilogue (:514-523)
    __device__ __forceinline__ void result(const WorldView& W, TraceResult& out) const
    {
        if (!LDS_COLD) {
            out.hit = ray_hit;
            out.steps = total;
            out.ncode = 0u;  // (register mode: the callers read entry_code / out_code themselves)
            out.normal = normal_decode(out_code);
            out.pos = hit_pos;
            out.vx = chx * W.f + min(cell_x, W.f - 1);
            out.vy = chy * W.f + min(cell_y, W.f - 1);
            out.vz = chz * W.f + min(cell_z, W.f - 1);
            if (ray_hit && total == 0) {
                out.pos = mk3(start.x * W.ff, start.y * W.ff, start.z * W.ff);
                out.normal = normal_decode(entry_code);
            }
            return;
        }
        const uint32_t ray_codes = cold[CF_RAY_CODES * 64];
        const bool hit = (ray_codes & 0x40u) != 0u;
        const int total_ = cget(CF_TOTAL, total);
        const int hx = cget(CF_CHX, chx), hy = cget(CF_CHY, chy), hz = cget(CF_CHZ, chz);
        out.hit = hit;
        out.steps = total_;
        // hitPosition of the walk that ended the ray, as phase_end computed it (the same operands, the same operations)
        const bool is_fine = fine != 0u;
        const float ox = (float)hx * W.ff, oy = (float)hy * W.ff, oz = (float)hz * W.ff;
        out.pos = mk3(is_fine ? point.x + ox : point.x * W.ff, is_fine ? point.y + oy : point.y * W.ff,
                      is_fine ? point.z + oz : point.z * W.ff);
        // brick HitCell = the clamped cell that was probed last (the walk does not advance past a hit)
        out.vx = hx * W.f + min(cell_x, W.f - 1);
        out.vy = hy * W.f + min(cell_y, W.f - 1);
        out.vz = hz * W.f + min(cell_z, W.f - 1);
        const bool at_entry = hit && total_ == 0;
        out.ncode = at_entry ? (ray_codes & 7u) : ((ray_codes >> 3) & 7u);
        out.normal = normal_decode(out.ncode);
        if (at_entry)
            out.pos = mk3(cget(CF_START_X, start.x) * W.ff, cget(CF_START_Y, start.y) * W.ff, cget(CF_START_Z, start.z) * W.ff);
    }
};

// Raytrace's prologue (:359-384) and the first walk's set-up (:199-213) of one ray as a 13-word record -- what
// WaveTracer::begin_ray + begin_walk compute, expression for expression (same operands, same operations, same order), so
// that WaveTracer::begin_prepared continues exactly where they would.  Runs in the shading kernels (vxrt_ts.hpp), all
// lanes busy, instead of in the traversal kernel's refill phase.
struct PreparedRay {
    uint4 a, b, c;
    uint32_t codes;
};
__device__ __forceinline__ PreparedRay prepare_ray(const WorldView& W, const f3 origin, const f3 ray, const int max_steps)
{
    const f3 d = unit3(ray);
    const float ivx = 1.0f / (d.x == 0 ? kFltEps : d.x);  // :127-129
    const float ivy = 1.0f / (d.y == 0 ? kFltEps : d.y);
    const float ivz = 1.0f / (d.z == 0 ? kFltEps : d.z);
    const int up_x = d.x > 0 ? 1 : 0, up_y = d.y > 0 ? 1 : 0, up_z = d.z > 0 ? 1 : 0;
    f3 s0 = mk3(origin.x * W.inv_f, origin.y * W.inv_f, origin.z * W.inv_f);
    uint32_t ec = 0u;
    if (!(s0.x >= 0 && s0.y >= 0 && s0.z >= 0 && s0.x < (float)W.cx && s0.y < (float)W.cy && s0.z < (float)W.cz)) {
        // the slab test of WaveTracer::slab against the world box [1e-6, dims - 1e-6]
        const float e = (float)1e-6;
        const float ax = (e - s0.x) * ivx, bx = (W.wmax_x - s0.x) * ivx;
        const float ay = (e - s0.y) * ivy, by = (W.wmax_y - s0.y) * ivy;
        const float az = (e - s0.z) * ivz, bz = (W.wmax_z - s0.z) * ivz;
        const float nx = lo(ax, bx), fx = hi(ax, bx);
        const float ny = lo(ay, by), fy = hi(ay, by);
        const float nz = lo(az, bz), fz = hi(az, bz);
        const float t_in = hi(hi(nx, ny), nz);
        const float t_out = lo(lo(fx, fy), fz);
        if (!(t_out < hi(t_in, 0.0f))) {
            s0 = mk3(s0.x + t_in * d.x, s0.y + t_in * d.y, s0.z + t_in * d.z);
            ec = (t_in == nx) ? (1u | (ivx < 0.0f ? 4u : 0u)) : (t_in == ny) ? (2u | (ivy < 0.0f ? 4u : 0u)) : (3u | (ivz < 0.0f ? 4u : 0u));
        }
    }
    const int cell_x = f2i(s0.x), cell_y = f2i(s0.y), cell_z = f2i(s0.z);
    const float tn_x = d.x != 0 ? ((float)(cell_x + up_x) - s0.x) / d.x : kInf;
    const float tn_y = d.y != 0 ? ((float)(cell_y + up_y) - s0.y) / d.y : kInf;
    const float tn_z = d.z != 0 ? ((float)(cell_z + up_z) - s0.z) / d.z : kInf;
    PreparedRay r;
    r.a = make_uint4(__float_as_uint(d.x), __float_as_uint(d.y), __float_as_uint(d.z), __float_as_uint(ivx));
    r.b = make_uint4(__float_as_uint(ivy), __float_as_uint(ivz), __float_as_uint(s0.x), __float_as_uint(s0.y));
    r.c = make_uint4(__float_as_uint(s0.z), __float_as_uint(tn_x), __float_as_uint(tn_y), __float_as_uint(tn_z));
    r.codes = ec | ((uint32_t)max_steps << 7);
    return r;
}

#if defined(VXRT_EXPERIMENTS) && !defined(VXRT_HOST_CHECK)
// Diagnostics: how many DISTINCT bricks do the lanes of this wave walk in right now?  One loop
// pass per distinct brick (readfirstlane of a lane's brick, ballot of the lanes in the same one); bin `n` of `hist` counts
// the iterations with n distinct bricks.  Probe-counting launches only.
template <class Tracer>
__device__ __forceinline__ void brick_histogram(const WorldView& W, const Tracer& T, unsigned long long* hist)
{
    const bool in_brick = T.st == ST_WALK && T.fine != 0u;
    const uint32_t key = (uint32_t)((T.bits - W.pool) / W.brick_words);
    unsigned long long m = __ballot(in_brick);
    uint32_t n = 0;
    while (m != 0ull) {
        const int first = __ffsll((long long)m) - 1;
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
        m &= ~__ballot(in_brick && key == k0);
        n += 1u;
    }
    const bool coarse_walkers = __ballot(T.st == ST_WALK && T.fine == 0u) != 0ull;
    if ((threadIdx.x & 63) == 0 && hist) {
        atomicAdd(&hist[n], 1ull);
        if (!coarse_walkers)
            atomicAdd(&hist[65 + n], 1ull);  // (kStatBrickHistFineOnly follows kStatBrickHist)
    }
}
#endif

// one ray per lane, entered by the whole wave at a converged point
template <bool STATS, bool MASKED_LOAD = false, bool LDS_COLD = false>
__device__ void trace_wave(const WorldView& W, const int max_steps, const bool active, const f3 origin, const f3 ray,
                           TraceResult& out, RayCounters& cnt, unsigned int* dbg = nullptr, uint32_t* cold_column = nullptr,
                           const bool prepared = false, unsigned long long* brick_hist = nullptr)
{
    WaveTracer<STATS, MASKED_LOAD, LDS_COLD> T;
    T.init(W, cold_column);
    if constexpr (LDS_COLD) {
        if (active && prepared) {  // the start the traversal kernel of vxrt_ts.hpp makes (host check, tests)
            const PreparedRay R = prepare_ray(W, origin, ray, max_steps);
            T.begin_prepared(W, R.a, R.b, R.c, R.codes);
        }
    }
    if (active && !(LDS_COLD && prepared))
        T.begin_ray(W, origin, ray, max_steps);
    for (;;) {
        const unsigned long long m_walk = __ballot(T.st == ST_WALK);
        const unsigned long long m_box = __ballot(T.st == ST_BOX);
        const unsigned long long m_end = __ballot(T.st == ST_END);
        if ((m_walk | m_box | m_end) == 0ull)
            break;  // every lane of the wave is done
        const int n_walk = __popcll(m_walk), n_box = __popcll(m_box), n_end = __popcll(m_end);
        if (STATS) {
            cnt.iters += 1;
            cnt.end_runs += vote_run(n_end, n_walk + n_box, VXRT_VOTE_END) ? 1u : 0u;
            cnt.box_runs += vote_run(n_box, n_walk, VXRT_VOTE_BOX) ? 1u : 0u;
        }
        if (vote_run(n_end, n_walk + n_box, VXRT_VOTE_END)) {
            if (T.st == ST_END)
                T.phase_end(W);
        }
        if (vote_run(n_box, n_walk, VXRT_VOTE_BOX)) {
            if (T.st == ST_BOX)
                T.phase_box(W);
        }
        if (STATS)
            cnt.walk_lanes += (uint32_t)__popcll(__ballot(T.st == ST_WALK));
#if defined(VXRT_EXPERIMENTS) && !defined(VXRT_HOST_CHECK)
        if (STATS && brick_hist)
            brick_histogram(W, T, brick_hist);
#endif
        if (STATS && dbg && cnt.iters <= 400u) {  // development trace of one lane (the caller passes dbg for lane 0 only)
            unsigned int* row = dbg + (cnt.iters - 1u) * 12u;
            row[0] = T.st; row[1] = T.fine; row[2] = (unsigned)T.cell_x; row[3] = (unsigned)T.cell_y; row[4] = (unsigned)T.cell_z;
            row[5] = __float_as_uint(T.tn_x); row[6] = __float_as_uint(T.tn_y); row[7] = __float_as_uint(T.tn_z);
            row[8] = (unsigned)T.steps; row[9] = (unsigned)T.total; row[10] = __float_as_uint(T.ws.x); row[11] = __float_as_uint(T.ws.y);
        }
        T.probe_group(W);
    }
    T.result(W, out);
    if (STATS) {
        cnt.coarse_probes += T.cnt.coarse_probes;
        cnt.brick_entries += T.cnt.brick_entries;
        cnt.fine_probes += T.cnt.fine_probes;
    }
}

}  // namespace vxrt
