// vxrt_pool.hpp -- screenDispatch (VoxelRT/Renderer.cu:179-276) with the per-pixel work taken OFF the lanes that trace.
//
// k_render_persist (vxrt_persist.hpp) ties a pixel to a lane: when a ray ends, that lane runs the pixel's continuation
// (shading, the next ray's set-up with its seven divisions and square root, the store, the next pixel's camera ray) in a
// voted phase.  Measured, that phase is ~1000 vector instructions serving ~26 of 64 lanes -- the union of every
// continuation's branch -- and a lane waits ~4 rounds for its vote: a fifth of the kernel's time and 13 parked lanes.
//
// Here the 160 KB of LDS a CU has (unused by that kernel) hold, per wavefront, a POOL of pixel chains and a QUEUE of
// prepared rays:
//   * a lane whose ray ends RETIRES it -- a few LDS writes: hit, steps, normal code, position into the pixel's slot, the slot
//     onto the finished list -- and takes the next prepared ray from the queue at once (REFILL: 13 LDS words).  Both are
//     cheap, so they are voted eagerly and lanes hardly park;
//   * when the queue runs low, the whole wave runs a PASS: lane i takes finished slot i (whoever traced it) or a new pixel
//     of the tile queue, runs the continuation with all 64 lanes busy, and pushes the prepared ray -- Raytrace's prologue
//     already evaluated: direction, reciprocals, entry point, first DDA state -- onto the queue.  The walk state of the
//     64 rays in flight stays parked in registers meanwhile.
// Results are a pure function of each pixel, so they equal k_render_persist's bit for bit (tests run both).
#pragma once

#include "vxrt_persist.hpp"

namespace vxrt {

// branch weight for the voted phases: told that they are the rare path, the register allocator keeps its spills out of
// the probe code (experiment knob)
#ifdef VXRT_COLD_PHASES
#define VXRT_RARE(c) __builtin_expect(!!(c), 0)
#else
#define VXRT_RARE(c) (c)
#endif

#ifndef VXRT_POOL_SLOTS
#define VXRT_POOL_SLOTS 128
#endif
constexpr int kPoolSlots = VXRT_POOL_SLOTS;  // pixel chains a wavefront keeps in flight (lanes + queue + finished list)
constexpr int kPoolRays = 64;    // capacity of the queue of prepared rays
constexpr int kRayWords = 13;    // d, 1/d, start, tMax (3 each) + one packed word
constexpr int kPoolMaxSamples = 1023;  // the slot's state word holds the sample counter in 10 bits
enum : uint32_t { ST_FREE = 5u };  // a lane without a ray (WaveTracer's states are 0..4)

#ifndef VXRT_POOL_LOW
#define VXRT_POOL_LOW 8  // run a pass when at most this many prepared rays are left
#endif
#ifndef VXRT_POOL_MINPASS
#define VXRT_POOL_MINPASS 32  // ... and at least this many chains can be continued or started (or lanes are starving)
#endif
#ifndef VXRT_VOTE_RETIRE
#define VXRT_VOTE_RETIRE 6  // retire + refill when their lanes are a sixth of the busy ones
#endif

template <bool BOUNCE2>
struct PoolLds {
    uint32_t px[kPoolSlots];     // launch coordinates: tx | row << 16
    uint32_t state[kPoolSlots];  // stage (3) | view (4) | primary normal code (3) | sample (10) | occlusion in halves (11)
    uint32_t res[kPoolSlots];    // the slot's last finished ray: hit (1) | normal code (3) | steps (28)
    float pos[3][kPoolSlots];    // primary hit position
    float col[3][kPoolSlots];    // camera ray direction (the pixel's colour on a miss), later the shaded colour
    uint32_t vox[2][kPoolSlots]; // primary hit voxel (hit-index AOV)
    float rpos[BOUNCE2 ? 3 : 1][BOUNCE2 ? kPoolSlots : 1];  // a bounce sample's hit position (second-bounce extension)
    uint8_t finished[kPoolSlots], freelist[kPoolSlots];
    uint32_t ray[kPoolRays * kRayWords];
};

template <bool STATS, bool BOUNCE2, bool MULTI>
__global__ __launch_bounds__(64, VXRT_PERSIST_OCC) void k_render_pool(RenderArgs A)
{
    __shared__ PoolLds<BOUNCE2> L;
    const WorldView& W = A.W;
    const int lane = threadIdx.x & 63;
    const unsigned long long lane_below = (1ull << lane) - 1ull;
    auto lane_view = [&](uint32_t v) -> LaneView {
        if (MULTI) {
            const ViewArgs& S = A.views[v];
            return LaneView{S.origin, S.fwd, S.up, S.right, S.frame_number, S.fb, S.color_aov, S.hit_aov};
        }
        return LaneView{A.origin, A.fwd, A.up, A.right, A.frame_number, A.fb, A.color_aov, A.hit_aov};
    };

    for (int s = lane; s < kPoolSlots; s += 64)
        L.freelist[s] = (uint8_t)s;
    uint32_t nfree = kPoolSlots, nfin = 0, q_head = 0, q_count = 0;  // wave-uniform

    WaveTracer<STATS> T;
    T.init(W);
    T.st = ST_FREE;
    uint32_t my_tag = 0;  // the lane's ray: slot | primary << 8
    uint32_t n_primary = 0, n_shadow = 0, n_bounce = 0, n_hits = 0;

    const uint32_t ntx = (A.width + 7u) / 8u, nty = (A.launch_rows + 7u) / 8u, ntiles = ntx * nty;
    uint32_t tile = 0, tile_used = 64u, tile_view = 0;
    bool drained = false;
    unsigned long long dg_iters = 0, dg_walk = 0;
    unsigned int dg_runs[3] = {0, 0, 0}, dg_lanes[3] = {0, 0, 0};  // pass / end / box executions and the lanes they served
    const unsigned long long dg_t0 = STATS ? wall_clock64() : 0ull;
    unsigned long long dg_next_ticks = 0, dg_park_ticks = 0;

    const f3 Ld = A.light_dir;
    const f3 sray = unit3(Ld);

    auto store_pixel = [&](const PixelCoords& pc, const LaneView& V, f3 origin, bool hit, f3 normal, f3 pos, f3 shaded, int p_steps) __attribute__((always_inline)) {
        PixelSink sink{A, pc.out_row, V.fb, V.color_aov};
        const int Wd = (int)A.width, Hd = (int)A.height;
        if (hit) {
            if (A.mode == 1) {  // DEBUG_VIEW quadrants, Renderer.cu:215-243
                f3 dv = pos - origin;
                float dist = sqrtf(dot3(dv, dv));
                const float wrap = (float)(1.0 + 1e-6);
                f3 hp = mk3(fmodf(pos.x / 128.0f, wrap), fmodf(pos.y / 128.0f, wrap), fmodf(pos.z / 128.0f, wrap));
                if (pc.x < (Wd >> 1) && pc.y < (Hd >> 1))
                    sink.put(pc.x, pc.y, normal);
                else if (pc.x >= (Wd >> 1) && pc.y < (Hd >> 1))
                    sink.put(pc.x, pc.y, hp);
                else if (pc.x < (Wd >> 1)) {
                } else
                    sink.put(pc.x, pc.y, mk3(dist * 0.01f, 0, 0));
            } else {
                if (!MULTI && A.accum)  // temporal accumulation (extension, include/vxrt.h): the mean of the history is tonemapped
                    shaded = accumulate_color(A, pc.out_row, pc.x, shaded);
                f3 c = mk3(shaded.x / (shaded.x + 1.0f), shaded.y / (shaded.y + 1.0f), shaded.z / (shaded.z + 1.0f));  // Tonemap
                c = mk3(lo(hi(c.x, 0), 1), lo(hi(c.y, 0), 1), lo(hi(c.z, 0), 1));
                sink.put(pc.x, pc.y, c);
            }
        } else {
            sink.put(pc.x, pc.y, shaded);  // the ray direction, Renderer.cu:254-258
        }
        if (pc.tx == (A.width >> 1) && pc.ty == (A.height >> 1))  // crosshair on launch coordinates, :261-268
            sink.put(pc.x, pc.y, mk3(10, 10, 10));
        if (A.mode == 1 && pc.x < (Wd >> 1) && pc.y > (Hd >> 1))  // :270-275
            sink.put(pc.x, pc.y, mk3((float)p_steps / 256.0f, 0, 0));
    };

    // ---- a lane whose ray has ended hands the result to the ray's pixel slot ------------------------------------------
    // (always_inline: the closures must be gone before the first scalar-replacement pass sees the tracer, or its members
    // are still memory when the `?:` patterns over them are turned into selected addresses -- vxrt_wave.hpp's scratch trap)
    auto retire = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_wave_barrier();
        const unsigned long long m_done = __ballot(T.st == ST_DONE);
        if (T.st == ST_DONE) {
            TraceResult r;
            T.result(W, r);
            const uint32_t slot = my_tag & 0xFFu;
            const bool at_entry = r.hit && r.steps == 0;  // (selects against 0, OR-ed: never a select between two members)
            const uint32_t code = (at_entry ? T.entry_code : 0u) | (at_entry ? 0u : T.out_code);
            L.res[slot] = (r.hit ? 1u : 0u) | (code << 1) | ((uint32_t)r.steps << 4);
            if (my_tag >> 8) {  // a primary ray: the hit position is the pixel's position from here on
                L.pos[0][slot] = r.pos.x;
                L.pos[1][slot] = r.pos.y;
                L.pos[2][slot] = r.pos.z;
                if (A.want_hit_aov) {
                    const long long v = r.hit ? (long long)r.vx + (long long)W.X * ((long long)r.vy + (long long)W.Y * (long long)r.vz) : -1ll;
                    L.vox[0][slot] = (uint32_t)(unsigned long long)v;
                    L.vox[1][slot] = (uint32_t)((unsigned long long)v >> 32);
                }
            } else if (BOUNCE2) {
                L.rpos[0][BOUNCE2 ? slot : 0] = r.pos.x;
                L.rpos[BOUNCE2 ? 1 : 0][BOUNCE2 ? slot : 0] = r.pos.y;
                L.rpos[BOUNCE2 ? 2 : 0][BOUNCE2 ? slot : 0] = r.pos.z;
            }
            L.finished[nfin + (uint32_t)__popcll(m_done & lane_below)] = (uint8_t)slot;
            T.st = ST_FREE;
        }
        nfin += (uint32_t)__popcll(m_done);
        __builtin_amdgcn_wave_barrier();
    };

    // ---- a lane without a ray takes the next prepared ray of the queue --------------------------------------------------
    auto refill = [&]() __attribute__((always_inline)) {
        const unsigned long long m_free = __ballot(T.st == ST_FREE);
        const uint32_t want = (uint32_t)__popcll(m_free);
        const uint32_t n = want < q_count ? want : q_count;
        const uint32_t rank = (uint32_t)__popcll(m_free & lane_below);
        if (T.st == ST_FREE && rank < n) {
            const uint32_t* R = &L.ray[((q_head + rank) % (uint32_t)kPoolRays) * (uint32_t)kRayWords];
            T.d = mk3(__uint_as_float(R[0]), __uint_as_float(R[1]), __uint_as_float(R[2]));
            T.ivx = __uint_as_float(R[3]);
            T.ivy = __uint_as_float(R[4]);
            T.ivz = __uint_as_float(R[5]);
            T.start = mk3(__uint_as_float(R[6]), __uint_as_float(R[7]), __uint_as_float(R[8]));
            T.tn_x = __uint_as_float(R[9]);
            T.tn_y = __uint_as_float(R[10]);
            T.tn_z = __uint_as_float(R[11]);
            const uint32_t m = R[12];  // entry code (3) | 8-step budget (1) | edge padding x,y,z (3) | primary (1) | slot (8)
            // the rest of Raytrace's prologue and of the first walk's set-up follows from these (begin_ray / begin_walk)
            T.up_x = T.d.x > 0 ? 1 : 0;
            T.up_y = T.d.y > 0 ? 1 : 0;
            T.up_z = T.d.z > 0 ? 1 : 0;
            T.entry_code = m & 7u;
            T.max_steps = (m & 8u) ? 8 : kMaxSteps;
            T.last_ci = 0xFFFFFFFFu;
            T.total = 0;
            T.ray_hit = false;
            T.out_code = 0u;
            T.bits = W.coarse_bits;
            T.fine = 0u;
            T.ws = T.start;
            T.point = T.start;
            T.cell_x = f2i(T.start.x);
            T.cell_y = f2i(T.start.y);
            T.cell_z = f2i(T.start.z);
            T.steps = 0;
            T.wf = 0u;
            T.w_code = 0u;
            T.skip = 0u;
            T.lim_x = W.cx + (int)((m >> 4) & 1u);
            T.lim_y = W.cy + (int)((m >> 5) & 1u);
            T.lim_z = W.cz + (int)((m >> 6) & 1u);
            T.dm1_x = W.cx - 1;
            T.dm1_y = W.cy - 1;
            T.dm1_z = W.cz - 1;
            T.row = W.c_row;
            T.slice = W.c_slice;
            my_tag = m >> 7;
            T.st = ST_WALK;
        }
        q_head = (q_head + n) % (uint32_t)kPoolRays;
        q_count -= n;
        __builtin_amdgcn_wave_barrier();
    };

    // ---- the whole wave continues up to 64 pixel chains (finished slots first, then new pixels) ------------------------
    auto pass = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_wave_barrier();
        const uint32_t room = (uint32_t)kPoolRays - q_count;
        const uint32_t k_fin = nfin < room ? nfin : room;
        const bool has_slot = (uint32_t)lane < k_fin;
        uint32_t slot = has_slot ? (uint32_t)L.finished[nfin - 1u - (uint32_t)lane] : 0u;
        nfin -= k_fin;

        uint32_t stage = PX_NONE, px_tx = 0, px_row = 0, pcode = 0, sample = 0, occl_h = 0;
        f3 position = mk3(0, 0, 0), color = mk3(0, 0, 0);
        bool launch = false, c_hit = false, c_shadow = false, c_bounce = false;
        f3 l_origin = mk3(0, 0, 0), l_dir = mk3(1, 0, 0);
        int l_max = kMaxSteps;
        bool l_primary = false;
        if (has_slot) {
            const uint32_t sw = L.state[slot], pxw = L.px[slot], rw = L.res[slot];
            stage = sw & 7u;
            const uint32_t view = (sw >> 3) & 15u;
            pcode = (sw >> 7) & 7u;
            sample = (sw >> 10) & 1023u;
            occl_h = sw >> 20;
            px_tx = pxw & 0xFFFFu;
            px_row = (pxw >> 16) | (MULTI ? view << 16 : 0u);
            const bool r_hit = (rw & 1u) != 0u;
            const uint32_t r_code = (rw >> 1) & 7u;
            const int r_steps = (int)(rw >> 4);
            position = mk3(L.pos[0][slot], L.pos[1][slot], L.pos[2][slot]);
            color = mk3(L.col[0][slot], L.col[1][slot], L.col[2][slot]);

            const LaneView V = lane_view(MULTI ? view : 0u);
            const PixelCoords pc = pixel_coords(A, V.frame_number, px_tx, MULTI ? px_row & 0xFFFFu : px_row);
            const f3 origin = camera_origin(A, V, pc.x, pc.y);
            bool finalize = false, do_shade = false, shadowed = false, bounce = false, bounce2 = false;
            int p_steps = 0;
            if (stage == PX_PRIMARY) {
                pcode = r_code;
                p_steps = r_steps;
                if (V.hit_aov)
                    V.hit_aov[(size_t)pc.out_row * A.width + (size_t)pc.x] =
                        (long long)((unsigned long long)L.vox[0][slot] | ((unsigned long long)L.vox[1][slot] << 32));
                c_hit = r_hit;
                if (r_hit)
                    color = mk3(0, 0, 0);  // a miss keeps the camera ray's direction: it is the pixel's colour
                if (!(r_hit && A.mode == 0)) {
                    stage = r_hit ? PX_PRIMARY : PX_NONE;  // remember hit/miss for the store below
                    finalize = true;
                } else if (A.shadow) {
                    c_shadow = true;
                    launch = true;  // Renderer.cu:97-102
                    l_origin = position + sray * 0.01f;
                    l_dir = sray;
                    l_max = kMaxSteps;
                    stage = PX_SHADOW;
                } else {
                    do_shade = true;
                }
            } else if (stage == PX_SHADOW) {
                shadowed = r_hit;
                do_shade = true;
            }
            const f3 pn = normal_decode(pcode);
            const f3 normal = mk3(-pn.x, -pn.y, -pn.z);  // Renderer.cu:212
            if (do_shade) {  // calculateColor, Renderer.cu:104-118
                const float l_dot = hi(dot3(normal, Ld), 0) * (float)(shadowed ? 0 : 1);
                f3 diffuse = A.light_color * l_dot;
                float up_dot = normal.x * 0.0f + normal.y * 1.0f + normal.z * 0.0f;
                float t = (float)((double)up_dot * 0.5 + 0.5);
                color = diffuse + A.ambient * (0.25f + t * (1.0f - 0.25f));
                if (!shadowed) {
                    f3 view_dir = unit3(position - origin);
                    f3 refl = reflect3(Ld, normal);
                    float spec = pow32(hi(dot3(view_dir, refl), 0));
                    color.x += spec * A.light_color.x;
                    color.y += spec * A.light_color.y;
                    color.z += spec * A.light_color.z;
                }
                stage = PX_PRIMARY;
                if ((l_dot == 0 || A.bounce_all_hits) && A.bounce_samples > 0) {  // Renderer.cu:121
                    occl_h = 0u;
                    sample = 0u;
                    bounce = true;
                } else {
                    finalize = true;  // gate closed, or samples == 0: occlusion = 1 (Renderer.cu:159-164)
                }
            } else if (stage == PX_BOUNCE || (BOUNCE2 && stage == PX_BOUNCE2)) {
                if (!r_hit)
                    occl_h += stage == PX_BOUNCE ? 2u : 1u;  // a missing sample ray adds 1, a missing second-bounce ray 0.5
                bounce2 = BOUNCE2 && stage == PX_BOUNCE && r_hit;
                if (!bounce2) {
                    sample += 1u;
                    if ((int)sample < A.bounce_samples) {
                        bounce = true;
                    } else {
                        // the sum of 1.0 / 0.5 increments is exact in binary32 in any order: occl = halves * 0.5
                        float occl = (float)occl_h * 0.5f;
                        occl /= (float)A.bounce_samples;
                        color = color * occl;
                        stage = PX_PRIMARY;
                        finalize = true;
                    }
                }
            }
            if (bounce || bounce2) {  // one sample of Renderer.cu:128-142, around the primary hit or the sample ray's
                const uint32_t seed = pc.ty * A.width + pc.tx;
                const uint32_t si = seed + sample * 1000u + (V.frame_number + 1u) * 1000u + (bounce2 ? 500u : 0u);
                f3 bn = normal, bo = position;
                if (BOUNCE2 && bounce2) {
                    const f3 rn = normal_decode(r_code);
                    bn = mk3(-rn.x, -rn.y, -rn.z);
                    bo = mk3(L.rpos[0][BOUNCE2 ? slot : 0], L.rpos[BOUNCE2 ? 1 : 0][BOUNCE2 ? slot : 0], L.rpos[BOUNCE2 ? 2 : 0][BOUNCE2 ? slot : 0]);
                }
                f3 sd = mk3(random_float(si) * 2 - 1, random_float(si * 10u) * 2 - 1, random_float(si * 100u) * 2 - 1);
                sd = unit3(sd);
                if (dot3(sd, bn) < 0)
                    sd = reflect3(sd, bn);
                c_bounce = true;
                launch = true;
                l_origin = bo + bn * 0.01f;
                l_dir = sd;
                l_max = 8;
                stage = bounce2 ? PX_BOUNCE2 : PX_BOUNCE;
            }
            if (finalize) {
                store_pixel(pc, V, origin, stage != PX_NONE, normal, position, color, p_steps);
                stage = PX_NONE;
            }
        }

        // new pixels of the wave's tile(s) for the lanes whose chain has ended and for the lanes without a slot, as long as
        // there is room in the ray queue and (for a lane without a slot) a free slot
        const uint32_t n_cont = (uint32_t)__popcll(__ballot(launch));
        uint32_t cap = room - n_cont;                       // rays this pass may still add
        const unsigned long long m_reuse = __ballot(has_slot && stage == PX_NONE);
        const unsigned long long m_fresh = __ballot(!has_slot);
        {   // lanes that bring their own slot first, then as many slot-less lanes as there are free slots
            const uint32_t n_reuse = (uint32_t)__popcll(m_reuse);
            const uint32_t fresh_ok = cap > n_reuse ? (cap - n_reuse < nfree ? cap - n_reuse : nfree) : 0u;
            cap = cap < n_reuse ? cap : n_reuse + fresh_ok;
        }
        bool asks = (has_slot && stage == PX_NONE && (uint32_t)__popcll(m_reuse & lane_below) < cap) ||
                    (!has_slot && (uint32_t)__popcll(m_reuse) + (uint32_t)__popcll(m_fresh & lane_below) < cap);
        bool got = false;
        unsigned long long want = __ballot(asks);
        while (want != 0ull && !drained) {
            if (tile_used >= 64u) {
                uint32_t t = 0;
                if (lane == 0)
                    t = atomicAdd(A.tile_counter, 1u);
                tile = (uint32_t)__shfl((int)t, 0, 64);
                if (tile >= ntiles * (MULTI ? A.nviews : 1u)) {
                    drained = true;
                    break;
                }
                tile_used = 0u;
                if (MULTI) {
                    tile_view = tile / ntiles;
                    tile -= tile_view * ntiles;
                    const ViewArgs& S = A.views[tile_view];
                    if (S.row_order_n)
                        tile = (uint32_t)S.row_order[tile / ntx] * ntx + tile % ntx;
                } else if (A.tile_order) {
                    tile = A.tile_order[tile];
                } else if (A.row_order_n) {
                    tile = (uint32_t)A.row_order[tile / ntx] * ntx + tile % ntx;
                }
            }
            const uint32_t avail = 64u - tile_used;
            const bool wants = ((want >> lane) & 1ull) != 0ull;
            const uint32_t rank = (uint32_t)__popcll(want & lane_below);
            if (wants && rank < avail) {
                const uint32_t p = tile_used + rank;
                px_tx = (tile % ntx) * 8u + (p & 7u);
                px_row = (tile / ntx) * 8u + (p >> 3);
                const bool live = pixel_coords(A, MULTI ? A.views[tile_view].frame_number : A.frame_number, px_tx, px_row).live;
                if (MULTI)
                    px_row |= tile_view << 16;
                got = live;
                asks = live ? false : asks;  // a dead pixel of a ragged tile: ask again
                if (live)
                    asks = false;
            }
            const uint32_t asked = (uint32_t)__popcll(want);
            tile_used += asked < avail ? asked : avail;
            want = __ballot(asks && !got);
        }
        // slots: a fresh pixel on a slot-less lane takes one from the free list; a chain that ended without a new pixel gives
        // its slot back
        {
            const unsigned long long m_take = __ballot(got && !has_slot);
            if (got && !has_slot)
                slot = (uint32_t)L.freelist[nfree - 1u - (uint32_t)__popcll(m_take & lane_below)];
            nfree -= (uint32_t)__popcll(m_take);
            const unsigned long long m_give = __ballot(has_slot && stage == PX_NONE && !got);
            if (has_slot && stage == PX_NONE && !got)
                L.freelist[nfree + (uint32_t)__popcll(m_give & lane_below)] = (uint8_t)slot;
            nfree += (uint32_t)__popcll(m_give);
        }
        if (got) {
            const LaneView V = lane_view(MULTI ? px_row >> 16 : 0u);
            const PixelCoords pc = pixel_coords(A, V.frame_number, px_tx, MULTI ? px_row & 0xFFFFu : px_row);
            camera_ray(A, V, pc.x, pc.y, l_origin, l_dir);
            color = l_dir;  // the pixel's colour if the primary ray misses (Renderer.cu:254-258)
            l_max = kMaxSteps;
            launch = true;
            l_primary = true;
            stage = PX_PRIMARY;
            pcode = 0u;
            sample = 0u;
            occl_h = 0u;
        }
        // Raytrace's prologue for every ray of the pass, then onto the queue
        const unsigned long long m_launch = __ballot(launch);
        if (launch) {
            WaveTracer<false> P;
            P.begin_ray(W, l_origin, l_dir, l_max);
            uint32_t* R = &L.ray[((q_head + q_count + (uint32_t)__popcll(m_launch & lane_below)) % (uint32_t)kPoolRays) * (uint32_t)kRayWords];
            R[0] = __float_as_uint(P.d.x);
            R[1] = __float_as_uint(P.d.y);
            R[2] = __float_as_uint(P.d.z);
            R[3] = __float_as_uint(P.ivx);
            R[4] = __float_as_uint(P.ivy);
            R[5] = __float_as_uint(P.ivz);
            R[6] = __float_as_uint(P.start.x);
            R[7] = __float_as_uint(P.start.y);
            R[8] = __float_as_uint(P.start.z);
            R[9] = __float_as_uint(P.tn_x);
            R[10] = __float_as_uint(P.tn_y);
            R[11] = __float_as_uint(P.tn_z);
            R[12] = P.entry_code | (l_max == 8 ? 8u : 0u) | ((uint32_t)(P.lim_x - W.cx) << 4) | ((uint32_t)(P.lim_y - W.cy) << 5) |
                    ((uint32_t)(P.lim_z - W.cz) << 6) | ((slot | (l_primary ? 256u : 0u)) << 7);
            // the slot's state for the ray's return
            L.state[slot] = stage | ((MULTI ? px_row >> 16 : 0u) << 3) | (pcode << 7) | (sample << 10) | (occl_h << 20);
            L.px[slot] = px_tx | ((px_row & 0xFFFFu) << 16);
            L.col[0][slot] = color.x;
            L.col[1][slot] = color.y;
            L.col[2][slot] = color.z;
        }
        q_count += (uint32_t)__popcll(m_launch);
        n_primary += (uint32_t)__popcll(__ballot(got));
        n_shadow += (uint32_t)__popcll(__ballot(c_shadow));
        n_bounce += (uint32_t)__popcll(__ballot(c_bounce));
        n_hits += (uint32_t)__popcll(__ballot(c_hit));
        __builtin_amdgcn_wave_barrier();
    };

    for (uint32_t guard = 0; guard < (1u << 26); ++guard) {  // (the bound only keeps a logic error from hanging the GPU)
        const unsigned long long m_walk = __ballot(T.st == ST_WALK);
        const unsigned long long m_box = __ballot(T.st == ST_BOX);
        const unsigned long long m_end = __ballot(T.st == ST_END);
        const unsigned long long m_done = __ballot(T.st == ST_DONE);
        int c_walk = __popcll(m_walk), c_box = __popcll(m_box), c_end = __popcll(m_end), c_done = __popcll(m_done);
        if ((m_walk | m_box | m_end | m_done) == 0ull && q_count == 0u && nfin == 0u && (drained || nfree == 0u))
            break;
        if (STATS) {
            dg_iters += 1;
            dg_walk += (unsigned long long)c_walk;
        }
        // parked phases as a cascade on fresh votes (see k_render_persist)
        if (VXRT_RARE(vote_run(c_box, c_walk, VXRT_VOTE_BOX))) {
            if (STATS) {
                dg_runs[2] += 1u;
                dg_lanes[2] += (unsigned)c_box;
            }
            if (T.st == ST_BOX)
                T.phase_box(W);
            c_box = 0;
            c_walk = __popcll(__ballot(T.st == ST_WALK));
            c_end = __popcll(__ballot(T.st == ST_END));
        }
        if (VXRT_RARE(vote_run(c_end, c_walk + c_box, VXRT_VOTE_END))) {
            if (STATS) {
                dg_runs[1] += 1u;
                dg_lanes[1] += (unsigned)c_end;
            }
            if (T.st == ST_END)
                T.phase_end(W);
            c_end = 0;
            c_walk = __popcll(__ballot(T.st == ST_WALK));
            c_done = __popcll(__ballot(T.st == ST_DONE));
        }
        // retire + refill: cheap, so voted eagerly -- lanes hardly wait for a new ray
        {
            const int c_free = __popcll(__ballot(T.st == ST_FREE));
            const int feed = c_done + (c_free < (int)q_count ? c_free : (int)q_count);
            const int busy = c_walk + c_box + c_end;
            if (VXRT_RARE(vote_run(feed, busy, VXRT_VOTE_RETIRE))) {
                if (STATS)
                    dg_park_ticks -= wall_clock64();  // (in this kernel: the time in retire + refill)
                retire();
                refill();
                if (STATS)
                    dg_park_ticks += wall_clock64();
            }
            // the pass: when the queue runs low and there is something to continue or to start
            const uint32_t startable = drained ? 0u : nfree;
            const int hungry = __popcll(__ballot(T.st == ST_FREE));
            if (VXRT_RARE(q_count <= (uint32_t)VXRT_POOL_LOW && (nfin + startable) > 0u &&
                (nfin + startable >= (uint32_t)VXRT_POOL_MINPASS || hungry >= 8 || __popcll(__ballot(T.st == ST_WALK || T.st == ST_BOX || T.st == ST_END)) == 0))) {
                if (STATS) {
                    dg_runs[0] += 1u;
                    dg_next_ticks -= wall_clock64();
                }
                const uint32_t before = q_count;
                pass();
                if (STATS)
                    dg_lanes[0] += (unsigned)(q_count - before);
                refill();
                if (STATS)
                    dg_next_ticks += wall_clock64();
            }
        }
        for (int g = 0; g < VXRT_SUBROUNDS; ++g) {
            if (g > 0) {
                int m_w = __popcll(__ballot(T.st == ST_WALK)), m_b = __popcll(__ballot(T.st == ST_BOX)),
                    m_e = __popcll(__ballot(T.st == ST_END));
                if (VXRT_RARE(vote_run(m_b, m_w, VXRT_VOTE_BOX))) {
                    if (STATS) {
                        dg_runs[2] += 1u;
                        dg_lanes[2] += (unsigned)m_b;
                    }
                    if (T.st == ST_BOX)
                        T.phase_box(W);
                    m_b = 0;
                    m_w = __popcll(__ballot(T.st == ST_WALK));
                    m_e = __popcll(__ballot(T.st == ST_END));
                }
                if (VXRT_RARE(vote_run(m_e, m_w + m_b, VXRT_VOTE_END))) {
                    if (STATS) {
                        dg_runs[1] += 1u;
                        dg_lanes[1] += (unsigned)m_e;
                    }
                    if (T.st == ST_END)
                        T.phase_end(W);
                }
            }
            T.probe_group(W);
        }
    }

    const unsigned long long s0 = n_primary, s1 = n_shadow, s2 = n_bounce, s3 = n_hits;
    if (lane == 0 && A.stats) {
        atomicAdd(&A.stats[kStatPrimary], s0);
        atomicAdd(&A.stats[kStatShadow], s1);
        atomicAdd(&A.stats[kStatBounce], s2);
        atomicAdd(&A.stats[kStatPrimaryHits], s3);
    }
    if (STATS) {
        unsigned long long p0 = wave_sum(T.cnt.coarse_probes), p1 = wave_sum(T.cnt.brick_entries), p2 = wave_sum(T.cnt.fine_probes);
        if (lane == 0 && A.stats) {
            atomicAdd(&A.stats[kStatCoarseProbes], p0);
            atomicAdd(&A.stats[kStatBrickEntries], p1);
            atomicAdd(&A.stats[kStatFineProbes], p2);
            atomicAdd(&A.stats[kStatDbgIters], dg_iters);
            atomicAdd(&A.stats[kStatDbgWalkLanes], dg_walk);
            atomicAdd(&A.stats[kStatDbgNextRuns], (unsigned long long)dg_runs[0]);
            atomicAdd(&A.stats[kStatDbgEndRuns], (unsigned long long)dg_runs[1]);
            atomicAdd(&A.stats[kStatDbgBoxRuns], (unsigned long long)dg_runs[2]);
            atomicAdd(&A.stats[kStatDbgNextLanes], (unsigned long long)dg_lanes[0]);
            atomicAdd(&A.stats[kStatDbgEndLanes], (unsigned long long)dg_lanes[1]);
            atomicAdd(&A.stats[kStatDbgBoxLanes], (unsigned long long)dg_lanes[2]);
            atomicAdd(&A.stats[kStatDbgLifetime], wall_clock64() - dg_t0);
            atomicAdd(&A.stats[kStatDbgNextTicks], dg_next_ticks);
            atomicAdd(&A.stats[kStatDbgParkTicks], dg_park_ticks);
        }
    }
}

}  // namespace vxrt
