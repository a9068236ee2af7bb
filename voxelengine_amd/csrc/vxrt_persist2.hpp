// vxrt_persist2.hpp -- k_render_persist_lds (vxrt_persist_lds.hpp) on the tracer of vxrt_wave2.hpp (kernel variant 7).
//
// The same persistent kernel, statement for statement -- one pixel chain per lane, tile queue, parked phases voted, cold
// state in the wave's LDS block -- with the traversal replaced by WaveTracer2: speculative exec-masked DDA advance, packed
// step counters instead of cell coordinates, no crossing point in the probe (profiles/r03_instr_cost.md: the vector ALU
// pipe is what bounds these kernels, and a probe pair of the old tracer is 373 cycles of it, of the new one ~165).
// The STATS instantiation counts the probes of SURVEY 8(d) (WaveTracer2 derives them from its packed step counters at the
// end of each walk, so the probes themselves are the timed kernel's) and collects the loop diagnostics.
#pragma once

#include "vxrt_persist_lds.hpp"
#include "vxrt_wave2.hpp"

namespace vxrt {

#ifndef VXRT_PERSIST2_OCC
#define VXRT_PERSIST2_OCC 5
#endif
// Vote thresholds of this kernel (vote_run: a parked phase runs when parked * N >= the other live lanes).  The probes
// of this tracer cost less than half of WaveTracer's while the phases cost about the same, so waiting for more lanes pays:
// end-of-walk waits until its lanes are as many as the others (N = 1), ray-finished and the tight-box phase (which now
// also enters the brick) until they are half as many.  Sweep in profiles/r03_variant7.md.
// What runs between the probe pairs of an iteration: 0 = nothing (default: the pairs run back to back, the walking mask
// carried in scalar registers); A/B: 1 = tight box, then end of walk, on fresh votes (variant 5's schedule); 2 = the tight box only
#ifndef VXRT_INNER_CASCADE
#define VXRT_INNER_CASCADE 0
#endif
// probe pairs per loop iteration (between two rounds of votes)
#ifndef VXRT_SUBROUNDS2
#define VXRT_SUBROUNDS2 3
#endif
#ifndef VXRT_VOTE2_NEXT
#define VXRT_VOTE2_NEXT 2
#endif
#ifndef VXRT_VOTE2_END
#define VXRT_VOTE2_END 1
#endif
#ifndef VXRT_VOTE2_BOX
#define VXRT_VOTE2_BOX 2
#endif
// A/B knob: absolute thresholds instead (a phase runs when at least K lanes wait for it, or nobody else is left)
#ifdef VXRT_VOTE2_ABS_BOX
__device__ __forceinline__ bool vote2(int parked, int others, int num, int k)
{
    (void)num;
    return parked >= k || (parked > 0 && others == 0);
}
#else
#define VXRT_VOTE2_ABS_BOX 0
#define VXRT_VOTE2_ABS_END 0
#define VXRT_VOTE2_ABS_NEXT 0
__device__ __forceinline__ bool vote2(int parked, int others, int num, int k)
{
    (void)k;
    return vote_run(parked, others, num);
}
#endif
#define PX_LD_U(f, v) do { if (LDS) v = PX[(f) * 64]; } while (0)
#define PX_ST_U(f, v) do { if (LDS) PX[(f) * 64] = (uint32_t)(v); } while (0)
#define PX_LD_I(f, v) do { if (LDS) v = (int)PX[(f) * 64]; } while (0)
#define PX_LD_F(f, v) do { if (LDS) v = __uint_as_float(PX[(f) * 64]); } while (0)
#define PX_ST_F(f, v) do { if (LDS) PX[(f) * 64] = __float_as_uint(v); } while (0)
#define PX_LD_POS() do { PX_LD_F(PF_POS_X, position.x); PX_LD_F(PF_POS_Y, position.y); PX_LD_F(PF_POS_Z, position.z); } while (0)
#define PX_ST_POS() do { PX_ST_F(PF_POS_X, position.x); PX_ST_F(PF_POS_Y, position.y); PX_ST_F(PF_POS_Z, position.z); } while (0)
#define PX_LD_COL() do { PX_LD_F(PF_COL_X, color.x); PX_LD_F(PF_COL_Y, color.y); PX_LD_F(PF_COL_Z, color.z); } while (0)
#define PX_ST_COL() do { PX_ST_F(PF_COL_X, color.x); PX_ST_F(PF_COL_Y, color.y); PX_ST_F(PF_COL_Z, color.z); } while (0)

template <bool STATS, bool BOUNCE2, bool MULTI>
__global__ __launch_bounds__(64, VXRT_PERSIST2_OCC) void k_render_persist2(RenderArgs A_kern)
{
    constexpr bool LDS = true;  // (the PX_* macros and the tracer's LDS_COLD parameter)
    __shared__ uint32_t cold_block[(CF_TRACER_FIELDS + PF_PIXEL_FIELDS) * 64 + 4];  // + the wave's four ray counters
    const WorldView& W = A_kern.W;
    const int lane = threadIdx.x & 63;
    uint32_t* const PX = &cold_block[CF_TRACER_FIELDS * 64 + lane];  // this lane's column of the pixel fields
    // (staging the launch's per-view parameters in LDS instead of gathering them from L2 in the ray-finished phase
    // was measured: -0.3 %, the loads are not what that phase waits for)
    auto lane_view = [&](const RenderArgs& A, uint32_t v) -> LaneView {
        if (MULTI) {
            const ViewArgs& S = A.views[v];
            return LaneView{S.origin, S.fwd, S.up, S.right, S.frame_number, S.fb, S.color_aov, S.hit_aov};
        }
        return LaneView{A.origin, A.fwd, A.up, A.right, A.frame_number, A.fb, A.color_aov, A.hit_aov};
    };

    WaveTracer2 T;
    T.init(W, &cold_block[lane]);  // st = ST_DONE: every lane starts by asking for a pixel
    uint32_t stage = PX_NONE;
    uint32_t px_tx = 0, px_row = 0;
    f3 position = mk3(0, 0, 0), color = mk3(0, 0, 0);
    uint32_t pcode = 0;       // primary hit normal (step direction) code
    int p_steps = 0;
    float occl = 0.0f;
    int sample = 0;
    uint32_t n_primary = 0, n_shadow = 0, n_bounce = 0, n_hits = 0;  // wave-uniform (ballot counts): scalar registers
    PX_ST_U(PF_STAGE, stage);  // (the other fields are written before they are read: a pixel starts with its primary ray)
    if (lane < 4)
        cold_block[(CF_TRACER_FIELDS + PF_PIXEL_FIELDS) * 64 + lane] = 0u;

    // the wave's share of the tile queue (wave-uniform)
    // Whole tiles per ticket: one same-address atomic per 64 pixels.  A finer queue is limited by the atomic rate
    // (measured: 2x slower frames at 8 pixels per ticket), and handing out only the last tiles in smaller pieces
    // did not shorten the frame either.
    uint32_t tile = 0, tile_used = 64u, tile_view = 0;
    bool drained = false;
    unsigned long long dg_iters = 0, dg_walk = 0, dg_drain = 0;  // STATS only: loop diagnostics
    unsigned int dg_runs[3] = {0, 0, 0}, dg_lanes[3] = {0, 0, 0};  // next / end / box phase executions, lanes served
    const unsigned long long dg_t0 = STATS ? wall_clock64() : 0ull;
    unsigned long long dg_next_ticks = 0, dg_park_ticks = 0;
#ifdef VXRT_TAIL_DEBUG
    unsigned long long px_t0 = 0;
#endif


    // store one finished pixel (setPixelColor + the debug overlays of screenDispatch, Renderer.cu:213-275)
    // `shaded`: the shaded colour of a hit pixel; for a miss, the camera ray's direction (kept in `color` since launch)
    auto store_pixel = [&](const RenderArgs& A, const PixelCoords& pc, const LaneView& V, f3 origin, bool hit, f3 normal, f3 pos, f3 shaded) {
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

    for (;;) {
        const unsigned long long m_walk = __ballot(T.st == ST_WALK);
        const unsigned long long m_box = __ballot(T.st == ST_BOX);
        const unsigned long long m_end = __ballot(T.st == ST_END || T.st == ST_ENDHIT);
        const unsigned long long m_next = __ballot(T.st == ST_DONE);
        if ((m_walk | m_box | m_end | m_next) == 0ull)
            break;
        const int n_walk = __popcll(m_walk), n_box = __popcll(m_box), n_end = __popcll(m_end), n_next = __popcll(m_next);
        if (STATS) {
            dg_iters += 1;
            dg_walk += (unsigned long long)n_walk;
            dg_drain += drained ? 1ull : 0ull;
        }

        // The parked phases run box -> end -> next inside one round, each vote on fresh counts: a lane whose box test
        // hits can enter its brick, and a lane whose ray ends can start its next ray, in the same round instead of
        // waiting for the next round's vote (+4 % with several probes per round; with one probe per round it was +-0)
        int c_walk = n_walk, c_box = n_box, c_end = n_end, c_next = n_next;
        if (vote2(c_box, c_walk, VXRT_VOTE2_BOX, VXRT_VOTE2_ABS_BOX)) {
            if (STATS) {
                dg_runs[2] += 1u;
                dg_lanes[2] += (unsigned)c_box;
                dg_park_ticks -= wall_clock64();
            }
            T.phase_box<STATS>(W);
            if (STATS)
                dg_park_ticks += wall_clock64();
            c_box = 0;
            c_walk = __popcll(__ballot(T.st == ST_WALK));
            c_end = __popcll(__ballot(T.st == ST_END || T.st == ST_ENDHIT));
        }
        if (vote2(c_end, c_walk + c_box, VXRT_VOTE2_END, VXRT_VOTE2_ABS_END)) {
            if (STATS) {
                dg_runs[1] += 1u;
                dg_lanes[1] += (unsigned)c_end;
                dg_park_ticks -= wall_clock64();
            }
            T.phase_end<STATS>(W);
            if (STATS)
                dg_park_ticks += wall_clock64();
            c_end = 0;
            c_walk = __popcll(__ballot(T.st == ST_WALK));
            c_next = __popcll(__ballot(T.st == ST_DONE));
        }
        // ---- parked phase: a ray finished -> continue the pixel's chain, store, take the next pixel ------------
        // Every continuation (shadow ray, bounce sample, the next pixel's primary ray) only RECORDS the ray to
        // launch; one begin_ray at the end of the phase serves them all (its 7 divisions + square root are the
        // expensive part of this phase).
        if (vote2(c_next, c_walk + c_box + c_end, VXRT_VOTE2_NEXT, VXRT_VOTE2_ABS_NEXT)) {
            if (STATS) {
                dg_runs[0] += 1u;
                dg_lanes[0] += (unsigned)c_next;
                dg_next_ticks -= wall_clock64();
            }
            const RenderArgs& A = kernarg_reload(A_kern);
            const uint32_t ntx = (A.width + 7u) / 8u, nty = (A.launch_rows + 7u) / 8u, ntiles = ntx * nty;
            const f3 L = A.light_dir;
            const f3 sray = A.light_unit;  // unit3(L), evaluated once on the host
            PX_LD_U(PF_STAGE, stage);
            PX_LD_U(PF_TX, px_tx);
            PX_LD_U(PF_ROW, px_row);
            bool launch = false;
            bool c_hit = false, c_shadow = false, c_bounce = false;  // this lane's contribution to the ray counters
            f3 l_origin = mk3(0, 0, 0), l_dir = mk3(1, 0, 0);
            int l_max = kMaxSteps;
            if (T.st == ST_DONE && stage != PX_NONE) {
                const LaneView V = lane_view(A, MULTI ? px_row >> 16 : 0u);
                const PixelCoords pc = pixel_coords(A, V.frame_number, px_tx, MULTI ? px_row & 0xFFFFu : px_row);
                const f3 origin = camera_origin(A, V, pc.x, pc.y);
                TraceResult r;
                T.result(W, r);
                bool finalize = false, do_shade = false, shadowed = false, bounce = false, bounce2 = false;
                if (stage == PX_PRIMARY) {
                    pcode = r.ncode;
                    p_steps = r.steps;
                    position = r.pos;
                    PX_ST_U(PF_PCODE, pcode);
                    PX_ST_U(PF_PSTEPS, p_steps);
                    PX_ST_POS();
                    if (V.hit_aov)
                        V.hit_aov[(size_t)pc.out_row * A.width + (size_t)pc.x] =
                            r.hit ? (long long)r.vx + (long long)W.X * ((long long)r.vy + (long long)W.Y * (long long)r.vz) : -1ll;
                    c_hit = r.hit;
                    if (r.hit) {
                        color = mk3(0, 0, 0);  // a miss keeps the ray direction stored at launch: it is the pixel's colour
                        PX_ST_COL();
                    }
                    if (!(r.hit && A.mode == 0)) {
                        stage = r.hit ? PX_PRIMARY : PX_NONE;  // remember hit/miss for the store below
                        finalize = true;
                    } else if (A.shadow) {
                        c_shadow = true;
                        launch = true;  // Renderer.cu:97-102
                        l_origin = position + A.light_step;  // sray * 0.01f, the product evaluated on the host
                        l_dir = sray;
                        l_max = kMaxSteps;
                        stage = PX_SHADOW;
                    } else {
                        do_shade = true;
                    }
                } else if (stage == PX_SHADOW) {
                    shadowed = r.hit;
                    do_shade = true;
                }
                PX_LD_U(PF_PCODE, pcode);
                const f3 pn = normal_decode(pcode);
                const f3 normal = mk3(-pn.x, -pn.y, -pn.z);  // Renderer.cu:212
                if (do_shade) {  // calculateColor, Renderer.cu:104-118
                    const float l_dot = hi(dot3(normal, L), 0) * (float)(shadowed ? 0 : 1);
                    f3 diffuse = A.light_color * l_dot;
                    float up_dot = normal.x * 0.0f + normal.y * 1.0f + normal.z * 0.0f;
                    float t = (float)((double)up_dot * 0.5 + 0.5);
                    color = diffuse + A.ambient * (0.25f + t * (1.0f - 0.25f));
                    if (!shadowed) {
                        PX_LD_POS();
                        f3 view = unit3(position - origin);
                        f3 refl = reflect3(L, normal);
                        float spec = pow32(hi(dot3(view, refl), 0));
                        color.x += spec * A.light_color.x;
                        color.y += spec * A.light_color.y;
                        color.z += spec * A.light_color.z;
                    }
                    PX_ST_COL();
                    stage = PX_PRIMARY;
                    if ((l_dot == 0 || A.bounce_all_hits) && A.bounce_samples > 0) {  // Renderer.cu:121
                        occl = 0.0f;
                        sample = 0;
                        PX_ST_F(PF_OCCL, occl);
                        PX_ST_U(PF_SAMPLE, sample);
                        bounce = true;
                    } else {
                        finalize = true;  // gate closed, or samples == 0: occlusion = 1 (Renderer.cu:159-164)
                    }
                } else if (stage == PX_BOUNCE || (BOUNCE2 && stage == PX_BOUNCE2)) {
                    PX_LD_F(PF_OCCL, occl);
                    PX_LD_I(PF_SAMPLE, sample);
                    if (!r.hit)
                        occl += stage == PX_BOUNCE ? 1.0f : 0.5f;
                    // extension beyond the reference (bounce_depth 2): a sample ray that hits spawns one more ray
                    bounce2 = BOUNCE2 && stage == PX_BOUNCE && r.hit;
                    if (!bounce2) {
                        sample += 1;
                        if (sample < A.bounce_samples) {
                            bounce = true;
                        } else {
                            occl /= A.bounce_samples_f;
                            PX_LD_COL();
                            color = color * occl;
                            PX_ST_COL();
                            stage = PX_PRIMARY;
                            finalize = true;
                        }
                    }
                    PX_ST_F(PF_OCCL, occl);
                    PX_ST_U(PF_SAMPLE, sample);
                }
                if (bounce || bounce2) {  // one sample of Renderer.cu:128-142, around the primary hit or the sample ray's
                    PX_LD_POS();
                    const uint32_t seed = pc.ty * A.width + pc.tx;
                    const uint32_t si = seed + (uint32_t)sample * 1000u + (V.frame_number + 1u) * 1000u + (bounce2 ? 500u : 0u);
                    const f3 bn = mk3(bounce2 ? -r.normal.x : normal.x, bounce2 ? -r.normal.y : normal.y,
                                      bounce2 ? -r.normal.z : normal.z);
                    const f3 bo = mk3(bounce2 ? r.pos.x : position.x, bounce2 ? r.pos.y : position.y,
                                      bounce2 ? r.pos.z : position.z);
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
                    PX_LD_POS();
                    PX_LD_COL();
                    PX_LD_I(PF_PSTEPS, p_steps);
                    // (the view's buffer pointers are fetched here, where they are used, instead of being carried --
                    // spilled -- from the top of the phase)
                    const LaneView Vs = lane_view(A, MULTI ? px_row >> 16 : 0u);
                    store_pixel(A, pc, Vs, origin, stage != PX_NONE, normal, position, color);
#ifdef VXRT_TAIL_DEBUG  // development: when each pixel's chain started / ended (100 MHz ticks), and its primary steps
                    if (STATS && V.color_aov) {
                        float* o = V.color_aov + ((size_t)pc.out_row * A.width + (size_t)pc.x) * 3;
                        o[0] = (float)(px_t0 & 0xFFFFFFull);
                        o[1] = (float)(wall_clock64() & 0xFFFFFFull);
                        o[2] = (float)p_steps;
                    }
#endif
                    stage = PX_NONE;
                }
            }
            // hand out pixels of the wave's tile(s) to the lanes that are free
            bool got = false;
            unsigned long long want = __ballot(T.st == ST_DONE && stage == PX_NONE);
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
                    // hand-out order: expected-longest ray chains first, so that what is still in flight when the
                    // queue runs dry is cheap (the host ranks the tile rows by the elevation of their centre ray)
                    if (MULTI) {
                        tile_view = tile / ntiles;
                        tile -= tile_view * ntiles;
                        // (ranking only the last view's rows -- the only view with a tail of its own -- measured 0.8 %
                        // slower: horizon-first order inside every view also helps the overlap between views)
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
                // this lane's rank among the asking lanes: v_mbcnt counts the mask's bits below the lane (no 64-bit lane mask kept)
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                if (wants && rank < avail) {
                    const uint32_t p = tile_used + rank;
                    px_tx = (tile % ntx) * 8u + (p & 7u);
                    px_row = (tile / ntx) * 8u + (p >> 3);
                    got = pixel_coords(A, MULTI ? A.views[tile_view].frame_number : A.frame_number, px_tx, px_row).live;
                    if (MULTI)
                        px_row |= tile_view << 16;
                }
                const uint32_t asked = (uint32_t)__popcll(want);
                tile_used += asked < avail ? asked : avail;
                want = __ballot(T.st == ST_DONE && stage == PX_NONE && !got);
            }
            if (got) {
                const LaneView V = lane_view(A, MULTI ? px_row >> 16 : 0u);
                const PixelCoords pc = pixel_coords(A, V.frame_number, px_tx, MULTI ? px_row & 0xFFFFu : px_row);
                camera_ray(A, V, pc.x, pc.y, l_origin, l_dir);
                color = l_dir;  // the pixel's colour if the primary ray misses (Renderer.cu:254-258)
                PX_ST_COL();
                l_max = kMaxSteps;
                launch = true;
                stage = PX_PRIMARY;
#ifdef VXRT_TAIL_DEBUG
                px_t0 = wall_clock64();
#endif
            }
            if (launch)
                T.begin_ray(W, l_origin, l_dir, l_max);
            T.after_begin_ray(launch);
            if (drained && T.st == ST_DONE && stage == PX_NONE)
                T.st = ST_IDLE;
            PX_ST_U(PF_STAGE, stage);
            PX_ST_U(PF_TX, px_tx);
            PX_ST_U(PF_ROW, px_row);
            {   // ray counters: ballots here, where the whole wave is converged again; accumulated in LDS (as registers they
                // are vector registers spilled around this phase: the scalar file is full)
                const uint32_t d0 = (uint32_t)__popcll(__ballot(got)), d1 = (uint32_t)__popcll(__ballot(c_shadow)),
                               d2 = (uint32_t)__popcll(__ballot(c_bounce)), d3 = (uint32_t)__popcll(__ballot(c_hit));
                if (lane == 0) {
                    uint32_t* const C = &cold_block[(CF_TRACER_FIELDS + PF_PIXEL_FIELDS) * 64];
                    C[0] += d0;
                    C[1] += d1;
                    C[2] += d2;
                    C[3] += d3;
                }
            }
            if (STATS)
                dg_next_ticks += wall_clock64();
        }

        // A round = the cascade above (box, end, next), then VXRT_SUBROUNDS groups of VXRT_STEPS_PER_ROUND probes with
        // the cheap half of the cascade (box, end on fresh votes) between the groups: the ballots and branches of a
        // vote are paid once per group, box/end lanes wait at most one group, and the expensive ray-finished phase
        // is voted once per round.  Measured (groups x probes): 1x1 3.53, 1x2 3.96, 1x3 4.02 Grays/s without the
        // cascade; with it 1x3 4.18, 1x4 4.21, 2x2 4.31, 2x3 and 2x4 the same, 3x3 4.37, 3x2 4.12 (register allocation),
        // 4x2 falls into scratch.  The same schedule as a rolled loop (vote the ray-finished phase every 2nd or 3rd
        // round of 2 probes) pays the round's four ballots and the loop branch per group: 4.02.
#if VXRT_INNER_CASCADE == 0
        T.probe_pairs<VXRT_SUBROUNDS2>(W);
#else
        for (int g = 0; g < VXRT_SUBROUNDS2; ++g) {
            if (g > 0 && VXRT_INNER_CASCADE != 0) {
                int m_w = __popcll(__ballot(T.st == ST_WALK)), m_b = __popcll(__ballot(T.st == ST_BOX)),
                    m_e = VXRT_INNER_CASCADE == 2 ? 0 : __popcll(__ballot(T.st == ST_END || T.st == ST_ENDHIT));
                if (vote2(m_b, m_w, VXRT_VOTE2_BOX, VXRT_VOTE2_ABS_BOX)) {
                    if (STATS) {
                        dg_runs[2] += 1u;
                        dg_lanes[2] += (unsigned)m_b;
                        dg_park_ticks -= wall_clock64();
                    }
                    T.phase_box<STATS>(W);
                    if (STATS)
                        dg_park_ticks += wall_clock64();
                    m_b = 0;
                    if (VXRT_INNER_CASCADE == 1) {
                        m_w = __popcll(__ballot(T.st == ST_WALK));
                        m_e = __popcll(__ballot(T.st == ST_END || T.st == ST_ENDHIT));
                    }
                }
                if (VXRT_INNER_CASCADE == 1 && vote2(m_e, m_w + m_b, VXRT_VOTE2_END, VXRT_VOTE2_ABS_END)) {
                    if (STATS) {
                        dg_runs[1] += 1u;
                        dg_lanes[1] += (unsigned)m_e;
                        dg_park_ticks -= wall_clock64();
                    }
                    T.phase_end<STATS>(W);
                    if (STATS)
                        dg_park_ticks += wall_clock64();
                }
            }
            T.probe_group(W);
        }
#endif
    }

    if (lane == 0) {
        const uint32_t* const C = &cold_block[(CF_TRACER_FIELDS + PF_PIXEL_FIELDS) * 64];
        n_primary = C[0];
        n_shadow = C[1];
        n_bounce = C[2];
        n_hits = C[3];
    }
    const unsigned long long s0 = n_primary, s1 = n_shadow, s2 = n_bounce, s3 = n_hits;
    if (lane == 0 && A_kern.stats) {
        atomicAdd(&A_kern.stats[kStatPrimary], s0);
        atomicAdd(&A_kern.stats[kStatShadow], s1);
        atomicAdd(&A_kern.stats[kStatBounce], s2);
        atomicAdd(&A_kern.stats[kStatPrimaryHits], s3);
    }
    if (STATS) {
        const unsigned long long p0 = wave_sum(T.cnt.coarse_probes), p1 = wave_sum(T.cnt.brick_entries), p2 = wave_sum(T.cnt.fine_probes);
        if (lane == 0 && A_kern.stats) {
            atomicAdd(&A_kern.stats[kStatCoarseProbes], p0);
            atomicAdd(&A_kern.stats[kStatBrickEntries], p1);
            atomicAdd(&A_kern.stats[kStatFineProbes], p2);
            atomicAdd(&A_kern.stats[kStatDbgIters], dg_iters);
            atomicAdd(&A_kern.stats[kStatDbgWalkLanes], dg_walk);
            atomicAdd(&A_kern.stats[kStatDbgNextRuns], (unsigned long long)dg_runs[0]);
            atomicAdd(&A_kern.stats[kStatDbgEndRuns], (unsigned long long)dg_runs[1]);
            atomicAdd(&A_kern.stats[kStatDbgBoxRuns], (unsigned long long)dg_runs[2]);
            atomicAdd(&A_kern.stats[kStatDbgNextLanes], (unsigned long long)dg_lanes[0]);
            atomicAdd(&A_kern.stats[kStatDbgEndLanes], (unsigned long long)dg_lanes[1]);
            atomicAdd(&A_kern.stats[kStatDbgBoxLanes], (unsigned long long)dg_lanes[2]);
            atomicAdd(&A_kern.stats[kStatDbgLifetime], wall_clock64() - dg_t0);
            atomicAdd(&A_kern.stats[kStatDbgDrained], dg_drain);
            atomicAdd(&A_kern.stats[kStatDbgNextTicks], dg_next_ticks);
            atomicAdd(&A_kern.stats[kStatDbgParkTicks], dg_park_ticks);
        }
    }
}
#undef PX_LD_U
#undef PX_ST_U
#undef PX_LD_I
#undef PX_LD_F
#undef PX_ST_F
#undef PX_LD_POS
#undef PX_ST_POS
#undef PX_LD_COL
#undef PX_ST_COL

}  // namespace vxrt
