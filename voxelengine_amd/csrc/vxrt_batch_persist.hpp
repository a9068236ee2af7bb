// vxrt_batch_persist.hpp -- the batch query (VoxelRaytracer3D::Raytrace + kernel dispatch,
// VoxelRT/VolumeRaytracer.cu:95-117,574-618) as a persistent wave-level kernel on the tracer of vxrt_wave2.hpp.
//
// One ray per lane with a block-wide exit (k_trace_batch_wave2) leaves a lane idle from the end of its ray to the end
// of the slowest ray of its workgroup.  Here the rays are a queue, as the pixel tiles are in vxrt_persist2.hpp:
// persistent wavefronts take kBatchTicket consecutive rays per ticket (one same-address atomic per 64 rays), and a
// lane whose ray has ended writes its result and takes the next ray of the wave's ticket in the voted "ray finished"
// phase.  Incoherent rays through a large world gain 2.2x that way (tools/batch_probe.py); a million short coherent rays
// (BASELINE configs[0]'s fan, 12 probes per ray) lose, because there the per-ray trip through the voted phases costs
// more than the ray -- so the launcher takes this kernel only for batches of at least 8 rays per lane of the persistent
// grid and keeps one ray per lane below that.  Results are a pure function of each ray.
#pragma once

#include "vxrt_kernels.hpp"
#include "vxrt_wave2.hpp"

namespace vxrt {

// Vote thresholds, probe pairs per round and waves per SIMD of this kernel, from profiles/r04_batch_api.md (4 M incoherent
// rays through the 8192x512x8192 world): a batch has no coherence, memory latency is what its waves wait for -- one probe
// pair per round (a lane whose walk ended does not wait through two more pairs for its phase), four waves per SIMD (a fifth
// only adds misses), the render kernel's votes (more eager and more patient ones both lose)
#ifndef VXRT_BATCH_VOTE_NEXT
#define VXRT_BATCH_VOTE_NEXT 2
#endif
#ifndef VXRT_BATCH_VOTE_END
#define VXRT_BATCH_VOTE_END 1
#endif
#ifndef VXRT_BATCH_VOTE_BOX
#define VXRT_BATCH_VOTE_BOX 2
#endif
// only walking lanes load (probe_pairs' MASKED)
#ifndef VXRT_BATCH_MASKED
#define VXRT_BATCH_MASKED 1
#endif
#ifndef VXRT_BATCH_PAIRS
#define VXRT_BATCH_PAIRS 1
#endif
constexpr uint32_t kBatchTicket = 64u;  // rays per queue ticket (256 or 1024: 7 % slower on incoherent rays, no faster on short ones)

#ifndef VXRT_BATCH_OCC
#define VXRT_BATCH_OCC 4
#endif
template <bool STATS, bool WIDE>
__global__ __launch_bounds__(64, VXRT_BATCH_OCC) void k_trace_batch_persist(BatchArgs B)
{
    __shared__ uint32_t cold_block[CF_TRACER_FIELDS * 64];
    const WorldView& W = B.W;
    const int lane = threadIdx.x & 63;
    constexpr unsigned long long kNone = ~0ull;

    WaveTracerT<WIDE> T;
    T.init(W, &cold_block[lane]);  // st = ST_DONE: every lane starts by asking for a ray
    unsigned long long my_ray = kNone;
    unsigned long long chunk = 0;  // wave-uniform: first ray of the wave's current ticket
    uint32_t used = kBatchTicket;  // rays of the ticket already handed out
    const uint32_t tickets = (uint32_t)((B.n + kBatchTicket - 1u) / kBatchTicket);  // (the launcher refuses batches beyond 2^32 tickets)
    uint32_t queue_shard = blockIdx.x % kQueueShards;  // (queue_take)
    bool drained = false;
    uint32_t n_rays = 0, n_hits = 0;  // wave-uniform (ballot counts)

    for (;;) {
        const unsigned long long m_walk = __ballot(T.st == ST_WALK);
        const unsigned long long m_box = __ballot(T.st == ST_BOX);
        const unsigned long long m_end = __ballot(waits_for_end(T.st));
        const unsigned long long m_next = __ballot(T.st == ST_DONE);
        if ((m_walk | m_box | m_end | m_next) == 0ull)
            break;
        int c_walk = __popcll(m_walk), c_box = __popcll(m_box), c_end = __popcll(m_end), c_next = __popcll(m_next);
        // parked phases as a cascade on fresh votes (see k_render_persist2)
        if (vote_run(c_box, c_walk, VXRT_BATCH_VOTE_BOX)) {
            T.template phase_box<STATS>(W);
            c_box = 0;
            c_walk = __popcll(__ballot(T.st == ST_WALK));
            c_end = __popcll(__ballot(waits_for_end(T.st)));
        }
        if (vote_run(c_end, c_walk + c_box, VXRT_BATCH_VOTE_END)) {
            T.template phase_end<STATS>(W);
            c_end = 0;
            c_walk = __popcll(__ballot(T.st == ST_WALK));
            c_next = __popcll(__ballot(T.st == ST_DONE));
        }
        // ---- parked phase: a ray finished -> write its result, take the next ray of the ticket --------------------
        if (vote_run(c_next, c_walk + c_box + c_end, VXRT_BATCH_VOTE_NEXT)) {
            bool c_hit = false;
            if (T.st == ST_DONE && my_ray != kNone) {
                TraceResult t;
                T.result(W, t);
                const unsigned long long i = my_ray;
                const f3 p = t.hit ? t.pos : mk3(kInf, kInf, kInf);
                B.pos[3 * i] = p.x;
                B.pos[3 * i + 1] = p.y;
                B.pos[3 * i + 2] = p.z;
                B.normal[3 * i] = t.normal.x;  // (zero on a miss: out_normal is zeroed at :382 and set by a hit only)
                B.normal[3 * i + 1] = t.normal.y;
                B.normal[3 * i + 2] = t.normal.z;
                B.steps[i] = t.steps;
                if (B.hit)
                    B.hit[i] = t.hit ? 1 : 0;
                if (B.voxel)
                    B.voxel[i] = t.hit ? (long long)t.vx + (long long)W.X * ((long long)t.vy + (long long)W.Y * (long long)t.vz)
                                       : -1ll;
                c_hit = t.hit;
                my_ray = kNone;
            }
            bool got = false, launch = false;
            f3 o = mk3(0, 0, 0), d = mk3(1, 0, 0);
            unsigned long long want = __ballot(T.st == ST_DONE && my_ray == kNone);
            while (want != 0ull && !drained) {
                if (used >= kBatchTicket) {
                    const uint32_t t = queue_take(B.ticket, tickets, queue_shard, lane);
                    if (t == kQueueDry) {
                        drained = true;
                        break;
                    }
                    chunk = (unsigned long long)t * kBatchTicket;
                    used = 0u;
                }
                const uint32_t avail = kBatchTicket - used;
                const bool wants = ((want >> lane) & 1ull) != 0ull;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                if (wants && rank < avail) {
                    const unsigned long long i = chunk + used + rank;
                    if (i < B.n) {  // the last ticket may be partial: its surplus lanes ask again and find the queue dry
                        my_ray = i;
                        got = true;
                    }
                }
                const uint32_t asked = (uint32_t)__popcll(want);
                used += asked < avail ? asked : avail;
                want = __ballot(T.st == ST_DONE && my_ray == kNone);
            }
            if (got) {
                const unsigned long long j = my_ray;
                o = mk3(B.origins[3 * j], B.origins[3 * j + 1], B.origins[3 * j + 2]);
                d = mk3(B.dirs[3 * j], B.dirs[3 * j + 1], B.dirs[3 * j + 2]);
                if (ray_valid(o, d)) {
                    launch = true;
                } else {  // include/vxrt.h, ray validity: not traced, its result is a miss with 0 steps, written here
                    B.pos[3 * j] = kInf;
                    B.pos[3 * j + 1] = kInf;
                    B.pos[3 * j + 2] = kInf;
                    B.normal[3 * j] = 0.0f;
                    B.normal[3 * j + 1] = 0.0f;
                    B.normal[3 * j + 2] = 0.0f;
                    B.steps[j] = 0;
                    if (B.hit)
                        B.hit[j] = 0;
                    if (B.voxel)
                        B.voxel[j] = -1ll;
                    my_ray = kNone;  // the lane stays in ST_DONE and asks for its next ray in the next ray-finished phase
                }
            }
            if (launch)
                T.begin_ray(W, o, d, B.max_steps);
            T.after_begin_ray(launch);
            if (drained && T.st == ST_DONE && my_ray == kNone)
                T.st = ST_IDLE;
            n_rays += (uint32_t)__popcll(__ballot(got));
            n_hits += (uint32_t)__popcll(__ballot(c_hit));
        }
        T.template probe_pairs<VXRT_BATCH_PAIRS, STATS, VXRT_BATCH_MASKED != 0>(W);
    }

    if (STATS && B.stats) {
        const unsigned long long p0 = wave_sum(T.cnt.coarse_probes), p1 = wave_sum(T.cnt.brick_entries),
                                 p2 = wave_sum(T.cnt.fine_probes), g0 = wave_sum(T.cnt.slack_loads), g1 = wave_sum(T.cnt.stray_loads);
        if (lane == 0) {
            unsigned long long* const stats = stats_row_of<kStatRows, kStatRowStride>(B.stats, blockIdx.x);
            atomicAdd(&stats[kStatGuardSlack], g0);
            atomicAdd(&stats[kStatGuardStray], g1);
            atomicAdd(&stats[kStatPrimary], (unsigned long long)n_rays);
            atomicAdd(&stats[kStatPrimaryHits], (unsigned long long)n_hits);
            atomicAdd(&stats[kStatCoarseProbes], p0);
            atomicAdd(&stats[kStatBrickEntries], p1);
            atomicAdd(&stats[kStatFineProbes], p2);
        }
    }
}

}  // namespace vxrt
