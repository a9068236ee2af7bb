// vxrt_wave.hpp -- the two-level brickmap DDA as a flat, wave-level state machine for gfx950.
//
// Same results as the straightforward form in vxrt_device.hpp (and therefore as Raytrace/DDARayTraversal,
// VoxelRT/VolumeRaytracer.cu:176-525), but organised for 64-wide wavefronts:
//
//  * ONE loop.  Every iteration performs one cell probe + one DDA advance for every lane that is walking,
//    whether it is on the coarse grid or inside a brick, so lanes on different levels share the same VALU
//    instructions instead of serialising nested loops behind exec-mask branches.
//  * The rare, expensive events -- tight-box slab test on an occupied coarse cell (ST_BOX) and the
//    end-of-walk transitions: brick entry, brick exit + re-seed with the ulp nudge, ray end (ST_END) --
//    PARK the lane.  A __ballot vote runs a parked phase only when enough lanes wait for it (or nobody can
//    walk), so its cost is amortised over many lanes; __ballot == 0 is the wave's early-out.
//  * Per-ray invariants of the reference's per-walk / per-box arithmetic are hoisted: 1/(d or eps) of the slab
//    test (VolumeRaytracer.cu:127-129) and |1/d| of the DDA (:199-201) are the same IEEE quotients for every
//    walk of a ray, so they are divided once per ray.
//  * Occupancy bits are fetched as 64-bit z-slices of an 8x8x8 tile (one aligned global_load_dwordx2) and kept
//    in registers: x/y moves inside a tile re-use the slice without touching memory.  The next cell's slice is
//    requested right after the advance, one iteration ahead of its test.
//  * The 8-byte cell_meta record read for the slab test also carries the brick's pool slot, so brick entry
//    needs no second dependent load (the reference chases a 24-byte descriptor, then the brick pointer).
#pragma once

#include "vxrt_device.hpp"

namespace vxrt {

enum : uint32_t { ST_WALK = 0u, ST_BOX = 1u, ST_END = 2u, ST_DONE = 3u };

// unit normals as small codes: 0 = zero vector, (axis+1) | 4*negative
__device__ __forceinline__ f3 normal_decode(uint32_t c)
{
    float v = (c & 4u) ? -1.0f : 1.0f;
    uint32_t a = c & 3u;
    return mk3(a == 1u ? v : 0.0f, a == 2u ? v : 0.0f, a == 3u ? v : 0.0f);
}

// run a parked phase when its lanes are at least a quarter of the walking lanes (or nobody walks)
__device__ __forceinline__ bool vote_run(int parked, int walking) { return parked > 0 && parked * 4 >= walking; }

template <bool STATS>
__device__ void trace_wave(const WorldView& W, const int max_steps, const bool active, const f3 origin, const f3 ray,
                           TraceResult& out, RayCounters& cnt)
{
    // ---- per-ray constants ---------------------------------------------------------------------
    const f3 d = active ? unit3(ray) : mk3(1.0f, 0.0f, 0.0f);
    const int sgx = d.x > 0 ? 1 : -1, sgy = d.y > 0 ? 1 : -1, sgz = d.z > 0 ? 1 : -1;
    const float ivx = 1.0f / (d.x == 0 ? kFltEps : d.x);  // slab-test reciprocals, :127-129
    const float ivy = 1.0f / (d.y == 0 ? kFltEps : d.y);
    const float ivz = 1.0f / (d.z == 0 ? kFltEps : d.z);
    const float tdx = d.x != 0 ? fabsf(ivx) : kInf;       // |1/d|, :199-201 (same quotient as ivx when d != 0)
    const float tdy = d.y != 0 ? fabsf(ivy) : kInf;
    const float tdz = d.z != 0 ? fabsf(ivz) : kInf;

    // slab test against [bmin,bmax] from point s with the hoisted reciprocals (RayIntersectsAABB, :124-174)
    auto slab = [&](f3 s, f3 bmin, f3 bmax, f3& p, uint32_t& code) -> bool {
        float ax = (bmin.x - s.x) * ivx, bx = (bmax.x - s.x) * ivx;
        float ay = (bmin.y - s.y) * ivy, by = (bmax.y - s.y) * ivy;
        float az = (bmin.z - s.z) * ivz, bz = (bmax.z - s.z) * ivz;
        float nx = lo(ax, bx), fx = hi(ax, bx);
        float ny = lo(ay, by), fy = hi(ay, by);
        float nz = lo(az, bz), fz = hi(az, bz);
        float t_in = hi(hi(nx, ny), nz);
        float t_out = lo(lo(fx, fy), fz);
        if (t_out < hi(t_in, 0.0f))
            return false;
        p = mk3(s.x + t_in * d.x, s.y + t_in * d.y, s.z + t_in * d.z);
        code = (t_in == nx) ? (1u | (ivx < 0.0f ? 4u : 0u))
                            : (t_in == ny) ? (2u | (ivy < 0.0f ? 4u : 0u)) : (3u | (ivz < 0.0f ? 4u : 0u));
        return true;
    };

    // ---- Raytrace-level state (:359-384) -------------------------------------------------------
    f3 start = mk3(origin.x * W.inv_f, origin.y * W.inv_f, origin.z * W.inv_f);
    uint32_t entry_code = 0;
    if (active && !(start.x >= 0 && start.y >= 0 && start.z >= 0 && start.x < (float)W.cx && start.y < (float)W.cy &&
                    start.z < (float)W.cz)) {
        const float e = (float)1e-6;
        f3 p;
        uint32_t c;
        if (slab(start, mk3(e, e, e), mk3(W.wmax_x, W.wmax_y, W.wmax_z), p, c)) {
            start = p;
            entry_code = c;
        }
    }
    uint32_t last_ci = 0xFFFFFFFFu;  // previous_cell as its tiled index (unique per cell); none yet
    int total = 0;
    f3 hit_pos = mk3(0, 0, 0);
    bool ray_hit = false;
    uint32_t out_code = 0;
    int vx = 0, vy = 0, vz = 0;

    // ---- walk state (DDARayTraversal locals, :178-232) ------------------------------------------
    bool fine = false;                 // level of the current walk
    f3 ws = start;                     // Params.start of the current walk
    int cell_x, cell_y, cell_z;
    float tn_x, tn_y, tn_z;
    f3 point;
    int it = 0, steps = 0;
    bool w_hit = false, w_oob = false;
    uint32_t w_code = 0;               // HitNormal of the current walk
    int pad_x = 0, pad_y = 0, pad_z = 0;
    int hx = 0, hy = 0, hz = 0;        // HitCell of a brick hit
    // coarse results that outlive the coarse walk (:399-429,:438-488)
    int chx = 0, chy = 0, chz = 0;     // coarse HitCell
    uint32_t c_code = 0;               // coarse HitNormal (the tight box's)
    int nc_axis = 0;                   // NextCell = coarse HitCell + sgn on this axis
    uint32_t slot = kEmptySlot;
    // occupancy slice cache
    const uint32_t* bits = W.coarse_bits;
    uint32_t slice_key = 0xFFFFFFFFu;
    uint2 slice = make_uint2(0u, 0u);

    auto begin_walk = [&](f3 s, bool to_fine) {
        fine = to_fine;
        ws = s;
        cell_x = (int)s.x;
        cell_y = (int)s.y;
        cell_z = (int)s.z;
        tn_x = d.x != 0 ? ((float)(cell_x + (sgx > 0)) - s.x) / d.x : kInf;
        tn_y = d.y != 0 ? ((float)(cell_y + (sgy > 0)) - s.y) / d.y : kInf;
        tn_z = d.z != 0 ? ((float)(cell_z + (sgz > 0)) - s.z) / d.z : kInf;
        point = s;
        it = 0;
        steps = 0;
        w_hit = false;
        w_oob = false;
        w_code = 0;
        // an occupied coarse cell always owns a brick (checked at upload, guaranteed by the device builder),
        // so a brick walk's dimensions are always f
        const int dmx = to_fine ? W.f : W.cx, dmy = to_fine ? W.f : W.cy, dmz = to_fine ? W.f : W.cz;
        const bool edge = cell_x == dmx || cell_y == dmy || cell_z == dmz;  // :216-232
        pad_x = edge && d.x < 0;
        pad_y = edge && d.y < 0;
        pad_z = edge && d.z < 0;
        slice_key = 0xFFFFFFFFu;
    };

    uint32_t st = active ? ST_WALK : ST_DONE;
    if (active)
        begin_walk(start, false);

    for (;;) {
        const unsigned long long m_walk = __ballot(st == ST_WALK);
        const unsigned long long m_box = __ballot(st == ST_BOX);
        const unsigned long long m_end = __ballot(st == ST_END);
        if ((m_walk | m_box | m_end) == 0ull)
            break;  // every lane of the wave is done
        const int n_walk = __popcll(m_walk), n_box = __popcll(m_box), n_end = __popcll(m_end);

        // ---- parked phase: end of a walk (:395-511) ---------------------------------------------
        if (vote_run(n_end, n_walk + n_box)) {
            if (st == ST_END) {
                total += steps;
                if (!fine) {
                    f3 local = mk3(point.x * W.ff, point.y * W.ff, point.z * W.ff);
                    hit_pos = local;
                    const uint32_t ci = tiled_index(chx, chy, chz, W.ctw, W.ctwh);
                    if (!(w_hit && !w_oob) || ci == last_ci) {
                        st = ST_DONE;  // coarse miss / left the grid (:508-511), or the previous_cell guard (:402-407)
                    } else {
                        last_ci = ci;
                        const float fx = (float)chx, fy = (float)chy, fz = (float)chz;
                        local = mk3(local.x - fx * W.ff, local.y - fy * W.ff, local.z - fz * W.ff);
                        if (STATS)
                            cnt.brick_entries += 1;
                        bits = W.pool + (size_t)slot * W.brick_words;
                        begin_walk(local, true);
                        st = ST_WALK;
                    }
                } else {
                    const float fx = (float)chx, fy = (float)chy, fz = (float)chz;
                    hit_pos = mk3(point.x + fx * W.ff, point.y + fy * W.ff, point.z + fz * W.ff);
                    if (w_hit) {  // :493-506
                        out_code = (steps == 0) ? c_code : w_code;
                        vx = chx * W.f + hx;
                        vy = chy * W.f + hy;
                        vz = chz * W.f + hz;
                        ray_hit = true;
                        st = ST_DONE;
                    } else {  // brick missed: restart the coarse walk just past it (:431-491)
                        start = mk3(hit_pos.x * W.inv_f, hit_pos.y * W.inv_f, hit_pos.z * W.inv_f);
                        if (w_oob) {
                            bool same = fx == (float)(int)start.x && fy == (float)(int)start.y && fz == (float)(int)start.z;
                            if (same) {
                                start.x = ulp_step(start.x, d.x < 0);
                                start.y = ulp_step(start.y, d.y < 0);
                                start.z = ulp_step(start.z, d.z < 0);
                                same = fx == (float)(int)start.x && fy == (float)(int)start.y && fz == (float)(int)start.z;
                                if (same) {
                                    const int ncx = chx + (nc_axis == 0 ? sgx : 0), ncy = chy + (nc_axis == 1 ? sgy : 0),
                                              ncz = chz + (nc_axis == 2 ? sgz : 0);
                                    float gx = (float)ncx - start.x, gy = (float)ncy - start.y, gz = (float)ncz - start.z;
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
                        if (total < max_steps) {  // the while condition, checked only here (:386)
                            bits = W.coarse_bits;
                            begin_walk(start, false);
                            st = ST_WALK;
                        } else {
                            st = ST_DONE;
                        }
                    }
                }
            }
        }

        bool advance = false, leaving = false;
        const uint32_t st0 = st;  // lanes leaving the box test below must not probe again this iteration

        // ---- parked phase: tight-box test of an occupied coarse cell (:248-273) -------------------
        if (vote_run(n_box, n_walk) && st0 == ST_BOX) {
            {
                const int qx = min(cell_x, W.cx - 1), qy = min(cell_y, W.cy - 1), qz = min(cell_z, W.cz - 1);
                const uint32_t idx = tiled_index(qx, qy, qz, W.ctw, W.ctwh);
                const uint2 meta = W.cell_meta[idx];
                const uint32_t e = meta.y;
                f3 bmin = mk3(((float)(e & 31u) + 0) * W.inv_f + (float)qx, ((float)((e >> 5) & 31u) + 0) * W.inv_f + (float)qy,
                              ((float)((e >> 10) & 31u) + 0) * W.inv_f + (float)qz);
                f3 bmax = mk3(((float)((e >> 15) & 31u) + 1) * W.inv_f + (float)qx,
                              ((float)((e >> 20) & 31u) + 1) * W.inv_f + (float)qy,
                              ((float)((e >> 25) & 31u) + 1) * W.inv_f + (float)qz);
                f3 bp;
                uint32_t bc;
                if (bmin.x <= bmax.x && slab(ws, bmin, bmax, bp, bc)) {
                    w_hit = true;
                    w_code = bc;
                    if (it != 0)
                        point = bp;
                    chx = qx;
                    chy = qy;
                    chz = qz;
                    c_code = bc;
                    slot = meta.x;
                    leaving = true;
                }
                advance = true;
                st = ST_WALK;
            }
        } else if (st0 == ST_WALK) {
            // ---- probe the current cell ----------------------------------------------------------
            const int dmx = fine ? W.f : W.cx, dmy = fine ? W.f : W.cy, dmz = fine ? W.f : W.cz;
            const bool inside = (unsigned)cell_x < (unsigned)(dmx + pad_x) && (unsigned)cell_y < (unsigned)(dmy + pad_y) &&
                                (unsigned)cell_z < (unsigned)(dmz + pad_z);
            advance = true;
            if (!inside) {
                w_oob = true;
                leaving = true;
            } else {
                const int qx = min(cell_x, dmx - 1), qy = min(cell_y, dmy - 1), qz = min(cell_z, dmz - 1);
                if (STATS) {
                    if (fine)
                        cnt.fine_probes += 1;
                    else
                        cnt.coarse_probes += 1;
                }
                const uint32_t idx = tiled_index(qx, qy, qz, fine ? W.ftw : W.ctw, fine ? W.ftwh : W.ctwh);
                const uint32_t key = idx >> 6;
                if (key != slice_key) {
                    slice = reinterpret_cast<const uint2*>(bits)[key];
                    slice_key = key;
                }
                const unsigned long long s64 = ((unsigned long long)slice.y << 32) | slice.x;
                const bool solid = ((s64 >> (idx & 63u)) & 1ull) != 0ull;
                if (solid) {
                    if (fine) {
                        w_hit = true;
                        hx = qx;
                        hy = qy;
                        hz = qz;
                        leaving = true;
                    } else {
                        st = ST_BOX;  // park for the tight-box test; the cell is not advanced yet
                        advance = false;
                    }
                }
            }
        }

        // ---- advance one cell (also on the exit iteration, :290-349) --------------------------------
        if (advance) {
            const bool ax0 = tn_x < tn_y && tn_x < tn_z;
            const bool ax1 = !ax0 && (tn_y <= tn_x && tn_y < tn_z);
            const float t = ax0 ? tn_x : (ax1 ? tn_y : tn_z);
            const float crx = ax0 ? (float)(cell_x + (sgx > 0)) : ws.x + (t * d.x);
            const float cry = ax1 ? (float)(cell_y + (sgy > 0)) : ws.y + (t * d.y);
            const float crz = (!ax0 && !ax1) ? (float)(cell_z + (sgz > 0)) : ws.z + (t * d.z);
            if (ax0) {
                cell_x += sgx;
                tn_x += tdx;
            } else if (ax1) {
                cell_y += sgy;
                tn_y += tdy;
            } else {
                cell_z += sgz;
                tn_z += tdz;
            }
            if (leaving) {
                if (!fine)
                    nc_axis = ax0 ? 0 : (ax1 ? 1 : 2);  // coarse NextCell; must survive the brick walk that follows
                st = ST_END;
            } else {
                w_code = ax0 ? (1u | (sgx < 0 ? 4u : 0u)) : (ax1 ? (2u | (sgy < 0 ? 4u : 0u)) : (3u | (sgz < 0 ? 4u : 0u)));
                const float fmax = W.ff;
                if (fine && (crx < 0.0f || crx > fmax || cry < 0.0f || cry > fmax || crz < 0.0f || crz > fmax)) {
                    w_oob = true;  // region check on the crossing point (:325-341): step not counted
                    st = ST_END;
                } else {
                    steps += 1;
                    point = mk3(crx, cry, crz);
                    it += 1;
                    if (it >= kMaxSteps)
                        st = ST_END;  // walk exhausted without a verdict (:234)
                }
            }
        }
    }

    out.hit = ray_hit;
    out.steps = total;
    out.normal = normal_decode(out_code);
    out.pos = hit_pos;
    out.vx = vx;
    out.vy = vy;
    out.vz = vz;
    if (ray_hit && total == 0) {  // :518-522
        out.pos = mk3(start.x * W.ff, start.y * W.ff, start.z * W.ff);
        out.normal = normal_decode(entry_code);
    }
}

}  // namespace vxrt
