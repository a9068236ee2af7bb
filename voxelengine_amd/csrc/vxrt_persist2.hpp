// vxrt_persist2.hpp -- screenDispatch (VoxelRT/Renderer.cu:179-276) as a persistent wave-level kernel: k_render_persist2,
// the render kernel of the product (kernel variant 7; variant 1, k_render in vxrt_kernels.hip, is the straightforward
// cross-check).
//
// The per-pixel ray chain of the reference -- primary ray, shadow ray for a hit (Renderer.cu:97-102), occlusion /
// bounce samples (:121-165), shading, tonemap, store -- is a dependent sequence per pixel but independent across
// pixels.  Launching "one lane = one pixel, three trace loops one after the other" leaves most lanes idle: measured
// useful-lane share of the traversal loop was ~46 % (rays of one 8x8 tile differ in length, sky pixels have no
// secondary rays, bounce rays go everywhere).  Here a wavefront is persistent:
//
//   * the launch grid's 8x8 pixel tiles are a queue (one global atomic per 64 pixels);
//   * each lane carries ONE pixel through its whole chain inside the single traversal loop of vxrt_wave2.hpp; the
//     moment its chain ends it stores the pixel and takes the next pixel from the wave's current tile
//     (ranks from the ballot mask, no per-lane atomics), so all 64 lanes stay in the walk phase;
//   * "ray finished" is one more parked state (ST_DONE), voted like the box/end phases: the per-pixel work (camera
//     ray, shading, next ray set-up with its divisions and square root) runs for many lanes at once;
//   * the state only the parked phases touch -- the tracer's cold fields, the pixel's chain state, the wave's four ray
//     counters -- lives in the wave's LDS block (25 columns of 64 dwords, 6.4 KB per wave), and the view's buffer pointers
//     are fetched where they are used: the probe loop fits 96 VGPRs = 5 waves per SIMD (20 per CU, 128 KB of 160 KB LDS).
//
// Results are identical to the one-lane-one-pixel kernel: every pixel is a pure function of its inputs.
// The STATS instantiation counts the probes of SURVEY 8(d) (WaveTracer2 derives them from its packed step counters at the
// end of each walk, so the probes themselves are the timed kernel's) and collects the loop diagnostics.
#pragma once

#include "vxrt_kernels.hpp"
#include "vxrt_wave2.hpp"

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
    // (x / W and y / H: small integers over small integers, by the host's reciprocals and one correction step -- exact for
    // every such pair, tests/tools/exact_div_check.c)
    const float u = div_rn((float)x, (float)(int)A.width, A.inv_width), v = div_rn((float)y, (float)(int)A.height, A.inv_height);
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
        const float dd = dot3(ray, ray);
        const f3 plain = ray;
        ray = unit3_ordinary(plain, dd);
        if (__ballot(!ordinary(dd)) != 0ull)
            ray = unit3(plain);
    }
}

// the ray origin alone (perspective: the camera; ortho: per pixel) -- what shading and the debug view need of a
// pixel's camera ray once the primary ray has been traced
__device__ __forceinline__ f3 camera_origin(const RenderArgs& A, const LaneView& V, int x, int y)
{
    f3 origin = V.origin;
    if (A.ortho) {
        const float u = div_rn((float)x, (float)(int)A.width, A.inv_width), v = div_rn((float)y, (float)(int)A.height, A.inv_height);
        origin = origin + ((V.right * (u * 2 - 1)) * A.ortho_x) * A.ratio;
        origin = origin + (V.up * (v * 2 - 1)) * A.ortho_y;
    }
    return origin;
}

enum : int { PF_STAGE = 0, PF_TX, PF_ROW, PF_POS_X, PF_POS_Y, PF_POS_Z, PF_COL_X, PF_COL_Y, PF_COL_Z, PF_PCODE, PF_PSTEPS, PF_OCCL,
             PF_SAMPLE, PF_PIXEL_FIELDS };
// The launch's arguments as the ray-finished phase reads them.  The ~50 frame arguments (camera, light, mode words, strips,
// buffers) are used by that phase only, but as kernel arguments the compiler loads them once, before the loop, and keeps
// them in scalar registers for the kernel's lifetime -- beside the probe loop's wave masks they do not fit (74 spilled
// SGPRs: the phase fetched them back from vector-register lanes with ~150 v_readlane + s_nop pairs per execution).  With
// VXRT_KERNARG_RELOAD the phase reads them from the kernel-argument segment through a pointer the optimiser cannot see
// through (an empty asm "modifies" it at every execution), so the loads stay inside the phase -- a few s_load_dwordx8/x16
// from the scalar cache -- and the values occupy scalar registers only while the phase runs.
// Measured (profiles/r03_kernarg_reload.md): spilled SGPRs 74 -> 0, spilled VGPRs 12 -> 7 (one view: 7 -> 0, no scratch at
// all), v_readlane in the kernel 150 -> 8; 16 views per launch 5602 -> 5707 Mrays/s (+1.9 %), one view per launch +1.2 %.
#ifndef VXRT_NO_KERNARG_RELOAD
#define VXRT_KERNARG_RELOAD 1
#endif
__device__ __forceinline__ const RenderArgs& kernarg_reload(const RenderArgs& in_registers)
{
#ifdef VXRT_KERNARG_RELOAD
    auto* p = (const __attribute__((address_space(4))) RenderArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const RenderArgs*)p;
#else
    return in_registers;
#endif
}

// BOUNCE2: the second-bounce extension (bounce_depth 2) is compiled into its own instantiation -- carried as a
// run-time branch it cost the reference ray set 29 more spilled VGPRs and 4 % of its speed.
// MULTI: several views of the world in one launch (vxrt_render_views).  The queue runs through view 0's tiles, then
// view 1's, ...: the next view's first tiles fill the lanes the previous view's last rays leave, so only the last
// view of a launch pays the low-occupancy tail.  A lane keeps its pixel's view in the upper half of px_row.
#ifndef VXRT_PERSIST2_OCC
#define VXRT_PERSIST2_OCC 5
#endif
// Vote thresholds of this kernel (vote_run: a parked phase runs when parked * N >= the other live lanes).  The probes
// are cheap beside the phases, so waiting for more lanes pays:
// end-of-walk waits until its lanes are as many as the others (N = 1), ray-finished and the tight-box phase (which now
// also enters the brick) until they are half as many.  Sweep in profiles/r03_variant7.md.
// probe pairs per loop iteration (between two rounds of votes)
#ifndef VXRT_QUEUE_GRANULE
#define VXRT_QUEUE_GRANULE 4u  // consecutive tiles of the hand-out order that belong to the same shard of the queue
#endif
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

template <bool STATS, bool BOUNCE2, bool MULTI, bool WIDE>
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

    WaveTracerT<WIDE> T;
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
    uint32_t queue_shard = blockIdx.x % kQueueShards;  // (queue_take)
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
                // Tonemap c / (c + 1) (Renderer.cu:170-177): the short exact division for colours of ordinary size (or zero)
                const float tx = shaded.x + 1.0f, ty = shaded.y + 1.0f, tz = shaded.z + 1.0f;
                f3 c = mk3(div_rn(shaded.x, tx, rcp_rn(tx)), div_rn(shaded.y, ty, rcp_rn(ty)), div_rn(shaded.z, tz, rcp_rn(tz)));
                const bool plain = !(((shaded.x == 0.0f) | ordinary(shaded.x)) & ((shaded.y == 0.0f) | ordinary(shaded.y)) &
                                     ((shaded.z == 0.0f) | ordinary(shaded.z)) & ordinary(tx) & ordinary(ty) & ordinary(tz));
                if (__ballot(plain) != 0ull)
                    c = mk3(shaded.x / (shaded.x + 1.0f), shaded.y / (shaded.y + 1.0f), shaded.z / (shaded.z + 1.0f));
                c = mk3(lo(hi(c.x, 0), 1), lo(hi(c.y, 0), 1), lo(hi(c.z, 0), 1));
                sink.put(pc.x, pc.y, c);
            }
        } else {
            sink.put(pc.x, pc.y, shaded);  // the ray direction, Renderer.cu:254-258
        }
        if (pc.tx == (A.width >> 1) && pc.ty == (A.height >> 1)) {  // crosshair on launch coordinates, :261-268
            float ten = 10.0f;
            pin(ten);  // (materialised here: the compiler hoisted the constant vector to the top of the kernel and SPILLED it --
                       // the kernel's only scratch use)
            sink.put(pc.x, pc.y, mk3(ten, ten, ten));
        }
        if (A.mode == 1 && pc.x < (Wd >> 1) && pc.y > (Hd >> 1))  // :270-275
            sink.put(pc.x, pc.y, mk3((float)p_steps / 256.0f, 0, 0));
    };

    for (;;) {
        const unsigned long long m_walk = __ballot(T.st == ST_WALK);
        const unsigned long long m_box = __ballot(T.st == ST_BOX);
        const unsigned long long m_end = __ballot(waits_for_end(T.st));
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
            T.template phase_box<STATS>(W);
            if (STATS)
                dg_park_ticks += wall_clock64();
            c_box = 0;
            c_walk = __popcll(__ballot(T.st == ST_WALK));
            c_end = __popcll(__ballot(waits_for_end(T.st)));
        }
        if (vote2(c_end, c_walk + c_box, VXRT_VOTE2_END, VXRT_VOTE2_ABS_END)) {
            if (STATS) {
                dg_runs[1] += 1u;
                dg_lanes[1] += (unsigned)c_end;
                dg_park_ticks -= wall_clock64();
            }
            T.template phase_end<STATS>(W);
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
                        const f3 to_hit = position - origin;
                        const float vdd = dot3(to_hit, to_hit);
                        f3 view = unit3_ordinary(to_hit, vdd);
                        if (__ballot(!ordinary(vdd)) != 0ull)
                            view = unit3(to_hit);
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
                            occl = div_rn(occl, A.bounce_samples_f, A.inv_bounce_samples);  // (half-integers over a small integer: exact)
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
                    {
                        const float sdd = dot3(sd, sd);
                        const f3 raw = sd;
                        sd = unit3_ordinary(raw, sdd);
                        if (__ballot(!ordinary(sdd)) != 0ull)
                            sd = unit3(raw);
                    }
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
                    tile = queue_take(A.tile_counter, ntiles * (MULTI ? A.nviews : 1u), VXRT_QUEUE_GRANULE, queue_shard, lane);
                    if (tile == kQueueDry) {
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

        // A round = the cascade above (box, end, next), then VXRT_SUBROUNDS2 probe pairs back to back: the walking mask is
        // carried from probe to probe in scalar registers, the ballots and branches of a vote are paid once per round
        // (votes between the pairs only split the phases' lanes: -5 %, profiles/r03_variant7.md)
        // (Measured and not kept, profiles/r04_tail.md: one or two pairs per round once the tile queue has run dry -- on the idea
        // that the tail is a latency problem -- 16 views per launch -2.3 %, one view per launch -5 %; leaving a burst as soon as
        // no lane walks any more: -0.8 % / +-0.)
        T.template probe_pairs<VXRT_SUBROUNDS2, STATS>(W);
    }

    if (lane == 0) {
        const uint32_t* const C = &cold_block[(CF_TRACER_FIELDS + PF_PIXEL_FIELDS) * 64];
        n_primary = C[0];
        n_shadow = C[1];
        n_bounce = C[2];
        n_hits = C[3];
    }
    const unsigned long long s0 = n_primary, s1 = n_shadow, s2 = n_bounce, s3 = n_hits;
    unsigned long long* const stats = A_kern.stats ? stats_row_of<kStatRows, kStatRowStride>(A_kern.stats, blockIdx.x) : nullptr;
    if (lane == 0 && stats) {
        atomicAdd(&stats[kStatPrimary], s0);
        atomicAdd(&stats[kStatShadow], s1);
        atomicAdd(&stats[kStatBounce], s2);
        atomicAdd(&stats[kStatPrimaryHits], s3);
    }
    if (STATS) {
        const unsigned long long p0 = wave_sum(T.cnt.coarse_probes), p1 = wave_sum(T.cnt.brick_entries), p2 = wave_sum(T.cnt.fine_probes);
        const unsigned long long g0 = wave_sum(T.cnt.slack_loads), g1 = wave_sum(T.cnt.stray_loads);
        if (lane == 0 && stats) {
            atomicAdd(&stats[kStatCoarseProbes], p0);
            atomicAdd(&stats[kStatBrickEntries], p1);
            atomicAdd(&stats[kStatFineProbes], p2);
            atomicAdd(&stats[kStatGuardSlack], g0);
            atomicAdd(&stats[kStatGuardStray], g1);
            atomicAdd(&stats[kStatDbgIters], dg_iters);
            atomicAdd(&stats[kStatDbgWalkLanes], dg_walk);
            atomicAdd(&stats[kStatDbgNextRuns], (unsigned long long)dg_runs[0]);
            atomicAdd(&stats[kStatDbgEndRuns], (unsigned long long)dg_runs[1]);
            atomicAdd(&stats[kStatDbgBoxRuns], (unsigned long long)dg_runs[2]);
            atomicAdd(&stats[kStatDbgNextLanes], (unsigned long long)dg_lanes[0]);
            atomicAdd(&stats[kStatDbgEndLanes], (unsigned long long)dg_lanes[1]);
            atomicAdd(&stats[kStatDbgBoxLanes], (unsigned long long)dg_lanes[2]);
            atomicAdd(&stats[kStatDbgLifetime], wall_clock64() - dg_t0);
            atomicAdd(&stats[kStatDbgDrained], dg_drain);
            atomicAdd(&stats[kStatDbgNextTicks], dg_next_ticks);
            atomicAdd(&stats[kStatDbgParkTicks], dg_park_ticks);
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
