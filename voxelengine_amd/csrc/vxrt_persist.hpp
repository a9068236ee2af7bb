// vxrt_persist.hpp -- screenDispatch (VoxelRT/Renderer.cu:179-276) as a persistent wave-level kernel.
//
// The per-pixel ray chain of the reference -- primary ray, shadow ray for a hit (Renderer.cu:97-102), occlusion /
// bounce samples (:121-165), shading, tonemap, store -- is a dependent sequence per pixel but independent across
// pixels.  Launching "one lane = one pixel, three trace loops one after the other" leaves most lanes idle: measured
// useful-lane share of the traversal loop was ~46 % (rays of one 8x8 tile differ in length, sky pixels have no
// secondary rays, bounce rays go everywhere).  Here a wavefront is persistent:
//
//   * the launch grid's 8x8 pixel tiles are a queue (one global atomic per 64 pixels);
//   * each lane carries ONE pixel through its whole chain inside the single traversal loop of vxrt_wave.hpp; the
//     moment its chain ends it stores the pixel and takes the next pixel from the wave's current tile
//     (ranks from the ballot mask, no LDS, no per-lane atomics), so all 64 lanes stay in the walk phase;
//   * "ray finished" is one more parked state (ST_DONE), voted like the box/end phases: the per-pixel work (camera
//     ray, shading, next ray set-up with its divisions and square root) runs for many lanes at once.
//
// Results are identical to the one-lane-one-pixel kernels: every pixel is a pure function of its inputs.
#pragma once

#include "vxrt_kernels.hpp"
#include "vxrt_wave.hpp"

namespace vxrt {

enum : uint32_t { PX_NONE = 0u, PX_PRIMARY = 1u, PX_SHADOW = 2u, PX_BOUNCE = 3u, PX_BOUNCE2 = 4u };

struct PixelCoords {
    uint32_t tx, ty;  // launch coordinates of the reference's thread (crosshair, RNG seed)
    int x, y;         // frame pixel
    int out_row;      // row in the destination buffers
    bool live;
};

// launch (tx,row) -> pixel (Renderer.cu:183-196 + this build's strip sharding)
__device__ __forceinline__ PixelCoords pixel_coords(const RenderArgs& A, uint32_t frame_number, uint32_t tx, uint32_t row)
{
    PixelCoords c;
    c.tx = tx;
    c.ty = row;
    c.x = (int)tx;
    const bool sharded = A.strip_count > 1;
    if (sharded && !A.checkerboard) {
        // A shard's launch rows are its own frame rows in order: launch row = packed row, the frame row follows from
        // the strip arithmetic, and ownership holds by construction -- no division by the strip count, and none by the
        // strip height when it is a power of two (strip_shift >= 0; the default 16 is).
        const uint32_t sr = (uint32_t)A.strip_rows;
        const uint32_t q = A.strip_shift >= 0 ? row >> A.strip_shift : row / sr;
        c.ty = (q * (uint32_t)A.strip_count + (uint32_t)A.strip_index) * sr + (row - q * sr);
        c.y = (int)c.ty;
        c.live = row < A.launch_rows && (uint32_t)c.x < A.width && (uint32_t)c.y < A.height;
        c.out_row = A.compact ? (int)row : c.y;
        return c;
    }
    c.live = row < A.launch_rows;
    c.y = (int)c.ty;
    if (A.checkerboard) {
        c.y *= 2;
        if ((c.x % 2) == 0)
            c.y += 1;
        if (frame_number % 2 == 0)
            c.y += 1;
    }
    c.live = c.live && (uint32_t)c.x < A.width && (uint32_t)c.y < A.height;
    if (c.live && sharded && ((uint32_t)c.y / (uint32_t)A.strip_rows) % (uint32_t)A.strip_count != (uint32_t)A.strip_index)
        c.live = false;
    c.out_row = c.y;
    if (A.compact && sharded)
        c.out_row = (int)((((uint32_t)c.y / (uint32_t)A.strip_rows) / (uint32_t)A.strip_count) * (uint32_t)A.strip_rows +
                          (uint32_t)c.y % (uint32_t)A.strip_rows);
    return c;
}

// the per-view inputs of one lane's pixel: kernel arguments for a single-view launch, loaded from the launch's
// ViewArgs array for a multi-view one
struct LaneView {
    f3 origin, fwd, up, right;
    uint32_t frame_number;
    uint8_t* fb;
    float* color_aov;
    long long* hit_aov;
};

// getRayDirection / getRayDirectionOrtho (Renderer.cu:44-70)
__device__ __forceinline__ void camera_ray(const RenderArgs& A, const LaneView& V, int x, int y, f3& origin, f3& ray)
{
    const float u = (float)x / (float)(int)A.width, v = (float)y / (float)(int)A.height;
    origin = V.origin;
    if (A.ortho) {
        ray = V.fwd;
        origin = origin + ((V.right * (u * 2 - 1)) * A.ortho_x) * A.ratio;
        origin = origin + (V.up * (v * 2 - 1)) * A.ortho_y;
    } else {
        float su = u * 2 - 1, sv = v * 2 - 1;
        ray.x = V.fwd.x + su * A.kx * V.right.x + sv * A.ky * V.up.x;
        ray.y = V.fwd.y + su * A.kx * V.right.y + sv * A.ky * V.up.y;
        ray.z = V.fwd.z + su * A.kx * V.right.z + sv * A.ky * V.up.z;
        ray = unit3(ray);
    }
}

// the ray origin alone (perspective: the camera; ortho: per pixel) -- what shading and the debug view need of a
// pixel's camera ray once the primary ray has been traced
__device__ __forceinline__ f3 camera_origin(const RenderArgs& A, const LaneView& V, int x, int y)
{
    f3 origin = V.origin;
    if (A.ortho) {
        const float u = (float)x / (float)(int)A.width, v = (float)y / (float)(int)A.height;
        origin = origin + ((V.right * (u * 2 - 1)) * A.ortho_x) * A.ratio;
        origin = origin + (V.up * (v * 2 - 1)) * A.ortho_y;
    }
    return origin;
}

#ifndef VXRT_SUBROUNDS
#define VXRT_SUBROUNDS 2  // groups of VXRT_STEPS_PER_ROUND probes per round
#endif
#ifndef VXRT_PERSIST_OCC
#define VXRT_PERSIST_OCC 4  // waves per SIMD the register budget is sized for (128 VGPRs)
#endif
// BOUNCE2: the second-bounce extension (bounce_depth 2) is compiled into its own instantiation -- carried as a
// run-time branch it cost the reference ray set 29 more spilled VGPRs and 4 % of its speed.
// MULTI: several views of the world in one launch (vxrt_render_views).  The queue runs through view 0's tiles, then
// view 1's, ...: the next view's first tiles fill the lanes the previous view's last rays leave, so only the last
// view of a launch pays the low-occupancy tail.  A lane keeps its pixel's view in the upper half of px_row.
template <bool STATS, bool BOUNCE2, bool MULTI>
__global__ __launch_bounds__(64, VXRT_PERSIST_OCC) void k_render_persist(RenderArgs A)
{
    const WorldView& W = A.W;
    const int lane = threadIdx.x & 63;
    const unsigned long long lane_below = (1ull << lane) - 1ull;
    // (staging the launch's per-view parameters in LDS instead of gathering them from L2 in the ray-finished phase
    // was measured: -0.3 %, the loads are not what that phase waits for)
    auto lane_view = [&](uint32_t v) -> LaneView {
        if (MULTI) {
            const ViewArgs& S = A.views[v];
            return LaneView{S.origin, S.fwd, S.up, S.right, S.frame_number, S.fb, S.color_aov, S.hit_aov};
        }
        return LaneView{A.origin, A.fwd, A.up, A.right, A.frame_number, A.fb, A.color_aov, A.hit_aov};
    };

#ifdef VXRT_RENDER_MASKED_LOAD  // experiment knob: only walking lanes load (off: -3.6 % at 1080p, -3.4 % at 4K, -4 % on the 16k world)
    WaveTracer<STATS, true> T;
#else
    WaveTracer<STATS> T;
#endif
    T.init(W);  // st = ST_DONE: every lane starts by asking for a pixel
    uint32_t stage = PX_NONE;
    uint32_t px_tx = 0, px_row = 0;
    f3 position = mk3(0, 0, 0), color = mk3(0, 0, 0);
    uint32_t pcode = 0;       // primary hit normal (step direction) code
    int p_steps = 0;
    float occl = 0.0f;
    int sample = 0;
    uint32_t n_primary = 0, n_shadow = 0, n_bounce = 0, n_hits = 0;  // wave-uniform (ballot counts): scalar registers

    // the wave's share of the tile queue (wave-uniform)
    const uint32_t ntx = (A.width + 7u) / 8u, nty = (A.launch_rows + 7u) / 8u, ntiles = ntx * nty;
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

    const f3 L = A.light_dir;
    const f3 sray = unit3(L);

    // store one finished pixel (setPixelColor + the debug overlays of screenDispatch, Renderer.cu:213-275)
    // `shaded`: the shaded colour of a hit pixel; for a miss, the camera ray's direction (kept in `color` since launch)
    auto store_pixel = [&](const PixelCoords& pc, const LaneView& V, f3 origin, bool hit, f3 normal, f3 pos, f3 shaded) {
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
        const unsigned long long m_end = __ballot(T.st == ST_END);
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
        if (vote_run(c_box, c_walk, VXRT_VOTE_BOX)) {
            if (STATS) {
                dg_runs[2] += 1u;
                dg_lanes[2] += (unsigned)c_box;
                dg_park_ticks -= wall_clock64();
            }
            if (T.st == ST_BOX)
                T.phase_box(W);
            if (STATS)
                dg_park_ticks += wall_clock64();
            c_box = 0;
            c_walk = __popcll(__ballot(T.st == ST_WALK));
            c_end = __popcll(__ballot(T.st == ST_END));
        }
        if (vote_run(c_end, c_walk + c_box, VXRT_VOTE_END)) {
            if (STATS) {
                dg_runs[1] += 1u;
                dg_lanes[1] += (unsigned)c_end;
                dg_park_ticks -= wall_clock64();
            }
            if (T.st == ST_END)
                T.phase_end(W);
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
        if (vote_run(c_next, c_walk + c_box + c_end, VXRT_VOTE_NEXT)) {
            if (STATS) {
                dg_runs[0] += 1u;
                dg_lanes[0] += (unsigned)c_next;
                dg_next_ticks -= wall_clock64();
            }
            bool launch = false;
            bool c_hit = false, c_shadow = false, c_bounce = false;  // this lane's contribution to the ray counters
            f3 l_origin = mk3(0, 0, 0), l_dir = mk3(1, 0, 0);
            int l_max = kMaxSteps;
            if (T.st == ST_DONE && stage != PX_NONE) {
                const LaneView V = lane_view(MULTI ? px_row >> 16 : 0u);
                const PixelCoords pc = pixel_coords(A, V.frame_number, px_tx, MULTI ? px_row & 0xFFFFu : px_row);
                const f3 origin = camera_origin(A, V, pc.x, pc.y);
                TraceResult r;
                T.result(W, r);
                bool finalize = false, do_shade = false, shadowed = false, bounce = false, bounce2 = false;
                if (stage == PX_PRIMARY) {
                    pcode = (r.hit && r.steps == 0) ? T.entry_code : T.out_code;
                    p_steps = r.steps;
                    position = r.pos;
                    if (V.hit_aov)
                        V.hit_aov[(size_t)pc.out_row * A.width + (size_t)pc.x] =
                            r.hit ? (long long)r.vx + (long long)W.X * ((long long)r.vy + (long long)W.Y * (long long)r.vz) : -1ll;
                    c_hit = r.hit;
                    if (r.hit)
                        color = mk3(0, 0, 0);  // a miss keeps the ray direction stored at launch: it is the pixel's colour
                    if (!(r.hit && A.mode == 0)) {
                        stage = r.hit ? PX_PRIMARY : PX_NONE;  // remember hit/miss for the store below
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
                    shadowed = r.hit;
                    do_shade = true;
                }
                const f3 pn = normal_decode(pcode);
                const f3 normal = mk3(-pn.x, -pn.y, -pn.z);  // Renderer.cu:212
                if (do_shade) {  // calculateColor, Renderer.cu:104-118
                    const float l_dot = hi(dot3(normal, L), 0) * (float)(shadowed ? 0 : 1);
                    f3 diffuse = A.light_color * l_dot;
                    float up_dot = normal.x * 0.0f + normal.y * 1.0f + normal.z * 0.0f;
                    float t = (float)((double)up_dot * 0.5 + 0.5);
                    color = diffuse + A.ambient * (0.25f + t * (1.0f - 0.25f));
                    if (!shadowed) {
                        f3 view = unit3(position - origin);
                        f3 refl = reflect3(L, normal);
                        float spec = pow32(hi(dot3(view, refl), 0));
                        color.x += spec * A.light_color.x;
                        color.y += spec * A.light_color.y;
                        color.z += spec * A.light_color.z;
                    }
                    stage = PX_PRIMARY;
                    if ((l_dot == 0 || A.bounce_all_hits) && A.bounce_samples > 0) {  // Renderer.cu:121
                        occl = 0.0f;
                        sample = 0;
                        bounce = true;
                    } else {
                        finalize = true;  // gate closed, or samples == 0: occlusion = 1 (Renderer.cu:159-164)
                    }
                } else if (stage == PX_BOUNCE || (BOUNCE2 && stage == PX_BOUNCE2)) {
                    if (!r.hit)
                        occl += stage == PX_BOUNCE ? 1.0f : 0.5f;
                    // extension beyond the reference (bounce_depth 2): a sample ray that hits spawns one more ray
                    bounce2 = BOUNCE2 && stage == PX_BOUNCE && r.hit;
                    if (!bounce2) {
                        sample += 1;
                        if (sample < A.bounce_samples) {
                            bounce = true;
                        } else {
                            occl /= (float)A.bounce_samples;
                            color = color * occl;
                            stage = PX_PRIMARY;
                            finalize = true;
                        }
                    }
                }
                if (bounce || bounce2) {  // one sample of Renderer.cu:128-142, around the primary hit or the sample ray's
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
                    store_pixel(pc, V, origin, stage != PX_NONE, normal, position, color);
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
                const uint32_t rank = (uint32_t)__popcll(want & lane_below);
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
                const LaneView V = lane_view(MULTI ? px_row >> 16 : 0u);
                const PixelCoords pc = pixel_coords(A, V.frame_number, px_tx, MULTI ? px_row & 0xFFFFu : px_row);
                camera_ray(A, V, pc.x, pc.y, l_origin, l_dir);
                color = l_dir;  // the pixel's colour if the primary ray misses (Renderer.cu:254-258)
                l_max = kMaxSteps;
                launch = true;
                stage = PX_PRIMARY;
#ifdef VXRT_TAIL_DEBUG
                px_t0 = wall_clock64();
#endif
            }
            if (launch)
                T.begin_ray(W, l_origin, l_dir, l_max);
            if (drained && T.st == ST_DONE && stage == PX_NONE)
                T.st = ST_IDLE;
            // ray counters: ballots here, where the whole wave is converged again
            n_primary += (uint32_t)__popcll(__ballot(got));
            n_shadow += (uint32_t)__popcll(__ballot(c_shadow));
            n_bounce += (uint32_t)__popcll(__ballot(c_bounce));
            n_hits += (uint32_t)__popcll(__ballot(c_hit));
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
        for (int g = 0; g < VXRT_SUBROUNDS; ++g) {
            if (g > 0) {
                int m_w = __popcll(__ballot(T.st == ST_WALK)), m_b = __popcll(__ballot(T.st == ST_BOX)),
                    m_e = __popcll(__ballot(T.st == ST_END));
                if (vote_run(m_b, m_w, VXRT_VOTE_BOX)) {
                    if (STATS) {
                        dg_runs[2] += 1u;
                        dg_lanes[2] += (unsigned)m_b;
                        dg_park_ticks -= wall_clock64();
                    }
                    if (T.st == ST_BOX)
                        T.phase_box(W);
                    if (STATS)
                        dg_park_ticks += wall_clock64();
                    m_b = 0;
                    m_w = __popcll(__ballot(T.st == ST_WALK));
                    m_e = __popcll(__ballot(T.st == ST_END));
                }
                if (vote_run(m_e, m_w + m_b, VXRT_VOTE_END)) {
                    if (STATS) {
                        dg_runs[1] += 1u;
                        dg_lanes[1] += (unsigned)m_e;
                        dg_park_ticks -= wall_clock64();
                    }
                    if (T.st == ST_END)
                        T.phase_end(W);
                    if (STATS)
                        dg_park_ticks += wall_clock64();
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
            atomicAdd(&A.stats[kStatDbgDrained], dg_drain);
            atomicAdd(&A.stats[kStatDbgNextTicks], dg_next_ticks);
            atomicAdd(&A.stats[kStatDbgParkTicks], dg_park_ticks);
        }
    }
}

}  // namespace vxrt
