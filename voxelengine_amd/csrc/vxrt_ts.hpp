// vxrt_ts.hpp -- screenDispatch (VoxelRT/Renderer.cu:179-276) as a WAVEFRONT pipeline: a traversal kernel T and a
// shading kernel S that hand each other compact ray records through HBM, generation by generation.
//
// Why (round 3): in the fused persistent kernels (vxrt_persist*.hpp) the "ray finished" phase -- shading, the
// next ray's set-up with its seven divisions and square root, the pixel store, the next pixel's camera ray -- is ~1000
// vector instructions (the union of every continuation's branch) executed for the 17-28 lanes of a wave whose rays have
// ended, 23-27 % of the kernel's issue slots, and it is what keeps that kernel at 96 VGPRs / 74 spilled SGPRs.  The memory
// system meanwhile idles at 6 % of its bandwidth.  Here that work leaves the traversal loop:
//
//   k_ts_gen     one wave per 8x8 pixel tile of the launch grid, all lanes busy: camera ray (Renderer.cu:44-70) -> a PREPARED
//                ray record (Raytrace's prologue evaluated: direction, reciprocals, world entry, first tMax; vxrt_wave.hpp
//                prepare_ray) in the tile's GROUP of the generation's queue;
//   k_ts_trace   T: persistent wavefronts pull chunks of groups from the queue; a lane whose ray has ended writes a 16-byte
//                result and takes the next record -- four 16-byte loads, no division, no square root, no pixel state, no
//                frame arguments.  Only WaveTracer state is alive: 6 waves per SIMD;
//   k_ts_shade   S: one wave per group, one lane per FINISHED ray, all lanes busy: calculateColor's continuation for that ray
//                (Renderer.cu:90-168: shadow ray for a primary hit, shading, bounce samples), the pixel store (Tonemap,
//                setPixelColor, overlays) or the next generation's prepared ray, compacted inside the group by ballot ranks
//                (no global atomic: a same-address atomic per wave bounded the first version at ~100 M waves/s).
//
// Generations: primary -> shadow (primary hits) -> bounce sample(s) (gate lDot == 0) [-> second bounce, extension].  A
// pixel's chain state between generations lives in a 32-byte record in HBM (position, colour, occlusion sum, stage).
// Every value is computed by the same expressions as in the fused kernels, so frames, AOVs and counters are bit-identical
// (tests run this pipeline against the oracle and against the fused kernels).
#pragma once

#include "vxrt_persist.hpp"

namespace vxrt {

#ifndef VXRT_TS_OCC
#define VXRT_TS_OCC 6  // waves per SIMD of the traversal kernel (80 VGPRs)
#endif
// The ray-finished phase of T is ~100 instructions without a division, so it is voted sooner than the fused kernels'
// (~1000 instructions, a third of the other live lanes): a fifth
// The queue's counter is split into VXRT_TS_SHARDS counters on cache lines of their own (group g is handed out by counter
// g % SHARDS): same-address atomics serialise at ~10 ns each on this chip whoever issues them -- half a million tickets per
// generation through ONE counter bounded the first version of this kernel at ~5 ms per generation -- while counters on
// different lines run side by side.
#ifndef VXRT_TS_SHARDS
#define VXRT_TS_SHARDS 16
#endif
#ifndef VXRT_TS_VOTE_NEXT
#define VXRT_TS_VOTE_NEXT 4
#endif
struct TsTraceArgs {
    WorldView W;
    const uint4* rays;
    const uint32_t* gcount;
    uint4* res;
    long long* res_voxel;
    unsigned int* ticket;
    uint32_t groups;
    unsigned long long* stats;
};

// ---- T: traversal only -------------------------------------------------------------------------------------------------
// Loop shape of k_render_persist (phase cascade box -> end -> ray finished on fresh votes, then VXRT_SUBROUNDS groups of
// probe pairs); what differs is the ray-finished phase, which here is "write 16 bytes, read 64".
//
// The queue: groups of up to 64 rays (TsArgs), handed out one group per ticket -- the finest grain, so that the launch ends
// within one group's time of the last ticket (chunks of several groups per atomic were measured: they cut the atomics but
// left 29 % of the wave slots idle behind the slowest chunks).  A ticket costs a trip to its counter and then one to the
// group's ray count; both are taken off the critical path by keeping ONE TICKET IN FLIGHT: the atomic for the next group is
// issued when the current group is started, its count is requested at the following ray-finished phase, and by the time
// the current group is handed out both have long arrived.
template <bool STATS>
__global__ __launch_bounds__(64, VXRT_TS_OCC) void k_ts_trace(TsTraceArgs B)
{
    __shared__ uint32_t cold_block[CF_TRACER_FIELDS * 64];
    const WorldView& W = B.W;
    const int lane = threadIdx.x & 63;

    WaveTracer<STATS, false, true> T;
    T.init(W, &cold_block[lane]);  // st = ST_DONE: every lane starts by asking for a ray
    uint32_t my_ray = kTsNoRay;
    // wave-uniform queue state: the group being handed out, its ray count, its rays already taken; the ticket in flight
    constexpr uint32_t K = VXRT_TS_SHARDS;
    uint32_t cur_g = 0, cur_cnt = 0, used = 0;
    uint32_t shard = blockIdx.x % K, shards_left = K;  // this wave's counter; when it runs dry the wave moves on to the next one
    int pend = 0;           // 0 = no ticket in flight, 1 = atomic issued, 2 = group known and its count requested, 3 = every counter is dry
    uint32_t pend_g = 0;
    uint32_t pend_v = 0;    // pend == 1: the atomic's result (lane 0); pend == 2: the group's ray count (every lane)
    bool drained = false;
    auto issue_ticket = [&]() __attribute__((always_inline)) {
        uint32_t t = 0;
        if (lane == 0)
            t = atomicAdd(B.ticket + shard * 64u, 1u);
        pend_v = t;
        pend = 1;
    };
    auto resolve_ticket = [&]() __attribute__((always_inline)) {  // pend 1 -> 2; or on to the next counter; or 3
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)pend_v);
        const unsigned long long g = (unsigned long long)t * K + shard;
        if (g < B.groups) {
            pend_g = (uint32_t)g;
            pend_v = B.gcount[pend_g];
            pend = 2;
        } else {
            shards_left -= 1u;
            if (shards_left == 0u) {
                pend = 3;
            } else {
                shard = (shard + 1u) % K;
                issue_ticket();
            }
        }
    };
    unsigned long long dg_iters = 0, dg_walk = 0;  // STATS only: loop diagnostics
    unsigned int dg_runs[3] = {0, 0, 0}, dg_lanes[3] = {0, 0, 0};
    const unsigned long long dg_t0 = STATS ? wall_clock64() : 0ull;
    unsigned long long dg_next_ticks = 0, dg_park_ticks = 0;

    for (;;) {
        const unsigned long long m_walk = __ballot(T.st == ST_WALK);
        const unsigned long long m_box = __ballot(T.st == ST_BOX);
        const unsigned long long m_end = __ballot(T.st == ST_END);
        const unsigned long long m_next = __ballot(T.st == ST_DONE);
        if ((m_walk | m_box | m_end | m_next) == 0ull)
            break;
        int c_walk = __popcll(m_walk), c_box = __popcll(m_box), c_end = __popcll(m_end), c_next = __popcll(m_next);
        if (STATS) {
            dg_iters += 1;
            dg_walk += (unsigned long long)c_walk;
#ifdef VXRT_EXPERIMENTS
            if (B.stats)
                brick_histogram(W, T, B.stats + kStatBrickHist);
#endif
        }
        if (vote_run(c_box, c_walk, VXRT_VOTE_BOX)) {
            if (STATS) {
                dg_runs[2] += 1u;
                dg_lanes[2] += (unsigned)c_box;
            }
            if (STATS)
                dg_park_ticks -= wall_clock64();
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
            }
            if (STATS)
                dg_park_ticks -= wall_clock64();
            if (T.st == ST_END)
                T.phase_end(W);
            if (STATS)
                dg_park_ticks += wall_clock64();
            c_end = 0;
            c_walk = __popcll(__ballot(T.st == ST_WALK));
            c_next = __popcll(__ballot(T.st == ST_DONE));
        }
        // ---- parked phase: a ray finished -> write its result, take the next prepared ray of the queue ----------------
        if (vote_run(c_next, c_walk + c_box + c_end, VXRT_TS_VOTE_NEXT)) {
            if (STATS) {
                dg_runs[0] += 1u;
                dg_lanes[0] += (unsigned)c_next;
                dg_next_ticks -= wall_clock64();
            }
            if (T.st == ST_DONE && my_ray != kTsNoRay) {
                TraceResult t;
                T.result(W, t);
                B.res[my_ray] = make_uint4((t.hit ? 1u : 0u) | (t.ncode << 1) | ((uint32_t)t.steps << 4), __float_as_uint(t.pos.x),
                                           __float_as_uint(t.pos.y), __float_as_uint(t.pos.z));
                if (B.res_voxel)
                    B.res_voxel[my_ray] = t.hit ? (long long)t.vx + (long long)W.X * ((long long)t.vy + (long long)W.Y * (long long)t.vz) : -1ll;
                my_ray = kTsNoRay;
            }
            // one step of the ticket in flight per ray-finished phase (its trips to memory overlap the probes in between)
            if (pend == 1)
                resolve_ticket();
            else if (pend == 0)
                issue_ticket();
            bool got = false;
            unsigned long long want = __ballot(T.st == ST_DONE && my_ray == kTsNoRay);
            while (want != 0ull && !drained) {
                if (used >= cur_cnt) {  // this group is handed out: on to the ticket in flight (waiting for it if it is not there yet)
                    while (pend < 2) {
                        if (pend == 0)
                            issue_ticket();
                        resolve_ticket();
                    }
                    if (pend == 3) {
                        drained = true;
                        break;
                    }
                    cur_g = pend_g;
                    cur_cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)pend_v);
                    used = 0u;
                    issue_ticket();
                    continue;
                }
                const uint32_t avail = cur_cnt - used;
                const bool wants = ((want >> lane) & 1ull) != 0ull;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                if (wants && rank < avail) {
                    my_ray = cur_g * 64u + used + rank;
                    got = true;
                }
                const uint32_t asked = (uint32_t)__popcll(want);
                used += asked < avail ? asked : avail;
                want = __ballot(T.st == ST_DONE && my_ray == kTsNoRay);
            }
            if (got) {
#ifdef VXRT_TS_EXP_HOTREC  // timing experiment only (wrong frames): every lane reads its group's first record (cache-hot)
                const uint4* R = B.rays + 4ull * (my_ray & ~63u);
#else
                const uint4* R = B.rays + 4ull * my_ray;
#endif
                const uint4 a = R[0], b = R[1], c = R[2], e = R[3];
                T.begin_prepared(W, a, b, c, e.x);
            }
            if (drained && T.st == ST_DONE && my_ray == kTsNoRay)
                T.st = ST_IDLE;
            if (STATS) {
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): charge the records' latency to this phase, not to the next probe
                dg_next_ticks += wall_clock64();
            }
        }
        for (int g = 0; g < VXRT_SUBROUNDS; ++g) {
            if (g > 0) {
                int m_w = __popcll(__ballot(T.st == ST_WALK)), m_b = __popcll(__ballot(T.st == ST_BOX)),
                    m_e = __popcll(__ballot(T.st == ST_END));
                if (vote_run(m_b, m_w, VXRT_VOTE_BOX)) {
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
                if (vote_run(m_e, m_w + m_b, VXRT_VOTE_END)) {
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

    if (STATS && B.stats) {
        const unsigned long long p0 = wave_sum(T.cnt.coarse_probes), p1 = wave_sum(T.cnt.brick_entries), p2 = wave_sum(T.cnt.fine_probes);
        if (lane == 0) {
            atomicAdd(&B.stats[kStatCoarseProbes], p0);
            atomicAdd(&B.stats[kStatBrickEntries], p1);
            atomicAdd(&B.stats[kStatFineProbes], p2);
            atomicAdd(&B.stats[kStatDbgIters], dg_iters);
            atomicAdd(&B.stats[kStatDbgWalkLanes], dg_walk);
            atomicAdd(&B.stats[kStatDbgNextRuns], (unsigned long long)dg_runs[0]);
            atomicAdd(&B.stats[kStatDbgEndRuns], (unsigned long long)dg_runs[1]);
            atomicAdd(&B.stats[kStatDbgBoxRuns], (unsigned long long)dg_runs[2]);
            atomicAdd(&B.stats[kStatDbgNextLanes], (unsigned long long)dg_lanes[0]);
            atomicAdd(&B.stats[kStatDbgEndLanes], (unsigned long long)dg_lanes[1]);
            atomicAdd(&B.stats[kStatDbgBoxLanes], (unsigned long long)dg_lanes[2]);
            atomicAdd(&B.stats[kStatDbgLifetime], wall_clock64() - dg_t0);
            atomicAdd(&B.stats[kStatDbgNextTicks], dg_next_ticks);
            atomicAdd(&B.stats[kStatDbgParkTicks], dg_park_ticks);
        }
    }
}

// ---- S: per-ray work at full lane occupancy, one wavefront per group ---------------------------------------------------------

__device__ __forceinline__ LaneView ts_lane_view(const RenderArgs& A, uint32_t v)
{
    if (A.nviews) {
        const ViewArgs& S = A.views[v];
        return LaneView{S.origin, S.fwd, S.up, S.right, S.frame_number, S.fb, S.color_aov, S.hit_aov};
    }
    return LaneView{A.origin, A.fwd, A.up, A.right, A.frame_number, A.fb, A.color_aov, A.hit_aov};
}

// launch coordinates of pixel `in` (0..63) of group g: the group is one 8x8 tile of one view, tiles in the hand-out order of
// the persistent kernels (tile rows ranked longest-first per view by the host, or the caller's own permutation)
struct TsPixel {
    uint32_t view, tx, row;
};
__device__ __forceinline__ TsPixel ts_pixel_of(const RenderArgs& A, const TsArgs& S, uint32_t g, uint32_t in)
{
    const uint32_t ntx = (A.width + 7u) / 8u;
    const uint32_t gpv = S.slots_per_view >> 6;
    TsPixel P;
    P.view = g / gpv;
    uint32_t tile = g - P.view * gpv;
    if (A.nviews) {
        const ViewArgs& VA = A.views[P.view];
        if (VA.row_order_n)
            tile = (uint32_t)VA.row_order[tile / ntx] * ntx + tile % ntx;
    } else if (A.tile_order) {
        tile = A.tile_order[tile];
    } else if (A.row_order_n) {
        tile = (uint32_t)A.row_order[tile / ntx] * ntx + tile % ntx;
    }
    P.tx = (tile % ntx) * 8u + (in & 7u);
    P.row = (tile / ntx) * 8u + (in >> 3);
    return P;
}

// Append this wave's launching lanes to its group of the next generation: ranks from the ballot (no atomic), records back to
// back from the group's base (the 64-byte records of a wave land in whole cache lines), the group's ray count by one lane
__device__ __forceinline__ void ts_emit(const RenderArgs& A, uint4* rays, uint8_t* idx, uint32_t* gcount, uint32_t g, bool launch,
                                        uint32_t in, f3 origin, f3 dir, int max_steps)
{
    const unsigned long long m = __ballot(launch);
    const int lane = threadIdx.x & 63;
    if (lane == 0)
        gcount[g] = (uint32_t)__popcll(m);
    if (launch) {
        const uint32_t j = g * 64u + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const PreparedRay R = prepare_ray(A.W, origin, dir, max_steps);
        uint4* o = rays + 4ull * j;
        o[0] = R.a;
        o[1] = R.b;
        o[2] = R.c;
        o[3] = make_uint4(R.codes, 0u, 0u, 0u);
        idx[j] = (uint8_t)in;
    }
}

// store one finished pixel (setPixelColor + the debug overlays of screenDispatch, Renderer.cu:213-275): the store_pixel of
// the fused kernels.  `shaded`: the shaded colour of a hit pixel; for a miss, the camera ray's direction
__device__ __forceinline__ void ts_store_pixel(const RenderArgs& A, const PixelCoords& pc, const LaneView& V, f3 origin, bool hit, f3 normal,
                                               f3 pos, f3 shaded, int p_steps)
{
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
            if (A.nviews == 0 && A.accum)  // temporal accumulation (extension, include/vxrt.h)
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
}

// generation 0: the camera rays of the launch grid
__global__ __launch_bounds__(256) void k_ts_gen(RenderArgs A, TsArgs S)
{
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t n_primary = 0;
    for (uint32_t g = wave; g < S.groups; g += nwaves) {
        const TsPixel P = ts_pixel_of(A, S, g, lane);
        const LaneView V = ts_lane_view(A, P.view);
        const PixelCoords pc = pixel_coords(A, V.frame_number, P.tx, P.row);
        f3 l_origin = mk3(0, 0, 0), l_dir = mk3(1, 0, 0);
        if (pc.live)
            camera_ray(A, V, pc.x, pc.y, l_origin, l_dir);
        n_primary += pc.live ? 1u : 0u;
        ts_emit(A, S.rays[0], S.idx[0], S.gcount[0], g, pc.live, lane, l_origin, l_dir, kMaxSteps);
    }
    const unsigned long long s0 = wave_sum(n_primary);
    if (lane == 0 && A.stats && s0)
        atomicAdd(&A.stats[kStatPrimary], s0);
}

// generation g's finished rays -> pixel stores and generation g + 1 (the ray-finished phase of k_render_persist, per ray)
template <bool BOUNCE2>
__global__ __launch_bounds__(256) void k_ts_shade(RenderArgs A, TsArgs S, int gen)
{
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint8_t* __restrict__ idx_in = S.idx[gen & 1];
    const uint32_t* __restrict__ gcount_in = S.gcount[gen & 1];
    uint4* const out_rays = S.rays[(gen + 1) & 1];
    uint8_t* const out_idx = S.idx[(gen + 1) & 1];
    uint32_t* const out_gcount = S.gcount[(gen + 1) & 1];
    const f3 L = A.light_dir;
    const f3 sray = A.light_unit;
    uint32_t n_shadow = 0, n_bounce = 0, n_hits = 0;
    for (uint32_t g = wave; g < S.groups; g += nwaves) {
        const uint32_t cnt = gcount_in[g];
        const uint32_t i = g * 64u + lane;
        bool launch = false;
        uint32_t in = 0;
        f3 l_origin = mk3(0, 0, 0), l_dir = mk3(1, 0, 0);
        int l_max = kMaxSteps;
        if (lane < cnt) {
            in = idx_in[i];
            const uint4 rr = S.res[i];
            TraceResult r;
            r.hit = (rr.x & 1u) != 0u;
            r.ncode = (rr.x >> 1) & 7u;
            r.steps = (int)(rr.x >> 4);
            r.pos = mk3(__uint_as_float(rr.y), __uint_as_float(rr.z), __uint_as_float(rr.w));
            r.normal = normal_decode(r.ncode);
            const TsPixel P = ts_pixel_of(A, S, g, in);
            const LaneView V = ts_lane_view(A, P.view);
            const PixelCoords pc = pixel_coords(A, V.frame_number, P.tx, P.row);
            const f3 origin = camera_origin(A, V, pc.x, pc.y);
            uint4* const PS = S.pstate + 2ull * (g * 64u + in);
            // the pixel's chain state: generation 0 starts it, later generations read it back
            uint32_t stage = PX_PRIMARY, pcode = 0u;
            int sample = 0, p_steps = 0;
            f3 position = mk3(0, 0, 0), color = mk3(0, 0, 0);
            float occl = 0.0f;
            if (gen > 0) {
                const uint4 q0 = PS[0], q1 = PS[1];
                position = mk3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
                stage = q0.w & 7u;
                pcode = (q0.w >> 3) & 7u;
                sample = (int)(q0.w >> 6);
                color = mk3(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z));
                occl = __uint_as_float(q1.w);
            }
            bool finalize = false, do_shade = false, shadowed = false, bounce = false, bounce2 = false;
            if (stage == PX_PRIMARY) {
                pcode = r.ncode;
                p_steps = r.steps;
                position = r.pos;
                if (V.hit_aov)
                    V.hit_aov[(size_t)pc.out_row * A.width + (size_t)pc.x] = S.res_voxel[i];
                n_hits += r.hit ? 1u : 0u;
                if (!(r.hit && A.mode == 0)) {
                    stage = r.hit ? PX_PRIMARY : PX_NONE;  // remember hit/miss for the store below
                    finalize = true;
                    if (!r.hit) {  // a miss is coloured with the camera ray's direction (Renderer.cu:254-258)
                        f3 o_;
                        camera_ray(A, V, pc.x, pc.y, o_, color);
                    }
                } else if (A.shadow) {
                    n_shadow += 1u;
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
                        occl /= A.bounce_samples_f;
                        color = color * occl;
                        stage = PX_PRIMARY;
                        finalize = true;
                    }
                }
            }
            if (bounce || bounce2) {  // one sample of Renderer.cu:128-142, around the primary hit or the sample ray's
                const uint32_t seed = pc.ty * A.width + pc.tx;
                const uint32_t si = seed + (uint32_t)sample * 1000u + (V.frame_number + 1u) * 1000u + (bounce2 ? 500u : 0u);
                const f3 bn = mk3(bounce2 ? -r.normal.x : normal.x, bounce2 ? -r.normal.y : normal.y, bounce2 ? -r.normal.z : normal.z);
                const f3 bo = mk3(bounce2 ? r.pos.x : position.x, bounce2 ? r.pos.y : position.y, bounce2 ? r.pos.z : position.z);
                f3 sd = mk3(random_float(si) * 2 - 1, random_float(si * 10u) * 2 - 1, random_float(si * 100u) * 2 - 1);
                sd = unit3(sd);
                if (dot3(sd, bn) < 0)
                    sd = reflect3(sd, bn);
                n_bounce += 1u;
                launch = true;
                l_origin = bo + bn * 0.01f;
                l_dir = sd;
                l_max = 8;
                stage = bounce2 ? PX_BOUNCE2 : PX_BOUNCE;
            }
            if (finalize) {
                ts_store_pixel(A, pc, V, origin, stage != PX_NONE, normal, position, color, p_steps);
            } else {
                PS[0] = make_uint4(__float_as_uint(position.x), __float_as_uint(position.y), __float_as_uint(position.z),
                                   stage | (pcode << 3) | ((uint32_t)sample << 6));
                PS[1] = make_uint4(__float_as_uint(color.x), __float_as_uint(color.y), __float_as_uint(color.z), __float_as_uint(occl));
            }
        }
        ts_emit(A, out_rays, out_idx, out_gcount, g, launch, in, l_origin, l_dir, l_max);
    }
    const unsigned long long s1 = wave_sum(n_shadow), s2 = wave_sum(n_bounce), s3 = wave_sum(n_hits);
    if (lane == 0 && A.stats) {
        if (s1) atomicAdd(&A.stats[kStatShadow], s1);
        if (s2) atomicAdd(&A.stats[kStatBounce], s2);
        if (s3) atomicAdd(&A.stats[kStatPrimaryHits], s3);
    }
}

}  // namespace vxrt
