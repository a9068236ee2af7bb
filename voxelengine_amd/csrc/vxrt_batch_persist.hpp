// vxrt_batch_persist.hpp -- the batch query (VoxelRaytracer3D::Raytrace + kernel dispatch,
// VoxelRT/VolumeRaytracer.cu:95-117,574-618) as a persistent wave-level kernel.
//
// One ray per lane with a block-wide exit (k_trace_batch_wave) leaves a lane idle from the end of its ray to the end
// of the slowest ray of its workgroup.  Here the rays are a queue, as the pixel tiles are in vxrt_persist.hpp:
// persistent wavefronts take kBatchTicket consecutive rays per ticket (one same-address atomic per 64 rays), and a
// lane whose ray has ended writes its result and takes the next ray of the wave's ticket in the voted "ray finished"
// phase.  Measured (tools/batch_probe.py): 4 M incoherent rays through the 8192x512x8192 world 0.83 -> 1.86 Grays/s;
// a million short coherent rays (BASELINE configs[0]'s fan, 12 probes per ray) 12.0 -> 3.8 Grays/s, because there
// the per-ray trip through the voted phases costs more than the ray -- so the launcher takes this kernel only for
// batches of at least 8 rays per lane of the persistent grid and keeps one ray per lane below that.
// Same loop shape as k_render_persist (phase cascade, two groups of probes per round); only walking lanes load
// (WaveTracer MASKED_LOAD): a batch has no coherence to rely on.  Results are a pure function of each ray.
#pragma once

#include "vxrt_kernels.hpp"
#include "vxrt_wave.hpp"

namespace vxrt {

// vote thresholds of this kernel: the render kernel's.  Voting 4x / 32x more eagerly -- on the idea that incoherent
// rays wait for memory, not for instructions -- measured 11 % / 20 % slower: lane efficiency still counts here.
#ifndef VXRT_BATCH_VOTE_NEXT
#define VXRT_BATCH_VOTE_NEXT VXRT_VOTE_NEXT
#endif
#ifndef VXRT_BATCH_VOTE_END
#define VXRT_BATCH_VOTE_END VXRT_VOTE_END
#endif
#ifndef VXRT_BATCH_VOTE_BOX
#define VXRT_BATCH_VOTE_BOX VXRT_VOTE_BOX
#endif
constexpr uint32_t kBatchTicket = 64u;  // rays per queue ticket (256 or 1024: 7 % slower on incoherent rays, no faster on short ones)

#ifndef VXRT_BATCH_OCC
#define VXRT_BATCH_OCC 4
#endif
template <bool STATS>
__global__ __launch_bounds__(64, VXRT_BATCH_OCC) void k_trace_batch_persist(BatchArgs B)
{
    const WorldView& W = B.W;
    const int lane = threadIdx.x & 63;
    const unsigned long long lane_below = (1ull << lane) - 1ull;
    constexpr unsigned long long kNone = ~0ull;

    WaveTracer<STATS, true> T;
    T.init(W);  // st = ST_DONE: every lane starts by asking for a ray
    unsigned long long my_ray = kNone;
    unsigned long long chunk = 0;  // wave-uniform: first ray of the wave's current ticket
    uint32_t used = kBatchTicket;  // rays of the ticket already handed out
    bool drained = false;
    uint32_t n_rays = 0, n_hits = 0;  // wave-uniform (ballot counts)

    for (;;) {
        const unsigned long long m_walk = __ballot(T.st == ST_WALK);
        const unsigned long long m_box = __ballot(T.st == ST_BOX);
        const unsigned long long m_end = __ballot(T.st == ST_END);
        const unsigned long long m_next = __ballot(T.st == ST_DONE);
        if ((m_walk | m_box | m_end | m_next) == 0ull)
            break;
        int c_walk = __popcll(m_walk), c_box = __popcll(m_box), c_end = __popcll(m_end), c_next = __popcll(m_next);
        // parked phases as a cascade on fresh votes (see k_render_persist)
        if (vote_run(c_box, c_walk, VXRT_BATCH_VOTE_BOX)) {
            if (T.st == ST_BOX)
                T.phase_box(W);
            c_box = 0;
            c_walk = __popcll(__ballot(T.st == ST_WALK));
            c_end = __popcll(__ballot(T.st == ST_END));
        }
        if (vote_run(c_end, c_walk + c_box, VXRT_BATCH_VOTE_END)) {
            if (T.st == ST_END)
                T.phase_end(W);
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
                B.normal[3 * i] = t.normal.x;
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
            bool got = false;
            unsigned long long want = __ballot(T.st == ST_DONE && my_ray == kNone);
            while (want != 0ull && !drained) {
                if (used >= kBatchTicket) {
                    uint32_t t = 0;
                    if (lane == 0)
                        t = atomicAdd(B.ticket, 1u);
                    t = (uint32_t)__shfl((int)t, 0, 64);
                    chunk = (unsigned long long)t * kBatchTicket;
                    if (chunk >= B.n) {
                        drained = true;
                        break;
                    }
                    used = 0u;
                }
                const uint32_t avail = kBatchTicket - used;
                const bool wants = ((want >> lane) & 1ull) != 0ull;
                const uint32_t rank = (uint32_t)__popcll(want & lane_below);
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
                const f3 o = mk3(B.origins[3 * j], B.origins[3 * j + 1], B.origins[3 * j + 2]);
                const f3 d = mk3(B.dirs[3 * j], B.dirs[3 * j + 1], B.dirs[3 * j + 2]);
                if (ray_valid(o, d)) {
                    T.begin_ray(W, o, d, B.max_steps);
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
            if (drained && T.st == ST_DONE && my_ray == kNone)
                T.st = ST_IDLE;
            n_rays += (uint32_t)__popcll(__ballot(got));
            n_hits += (uint32_t)__popcll(__ballot(c_hit));
        }
        for (int g = 0; g < 2; ++g) {
            if (g > 0) {
                int m_w = __popcll(__ballot(T.st == ST_WALK)), m_b = __popcll(__ballot(T.st == ST_BOX)),
                    m_e = __popcll(__ballot(T.st == ST_END));
                if (vote_run(m_b, m_w, VXRT_BATCH_VOTE_BOX)) {
                    if (T.st == ST_BOX)
                        T.phase_box(W);
                    m_b = 0;
                    m_w = __popcll(__ballot(T.st == ST_WALK));
                    m_e = __popcll(__ballot(T.st == ST_END));
                }
                if (vote_run(m_e, m_w + m_b, VXRT_BATCH_VOTE_END)) {
                    if (T.st == ST_END)
                        T.phase_end(W);
                }
            }
            T.probe_group(W);
        }
    }

    if (STATS && B.stats) {
        const unsigned long long p0 = wave_sum(T.cnt.coarse_probes), p1 = wave_sum(T.cnt.brick_entries),
                                 p2 = wave_sum(T.cnt.fine_probes);
        if (lane == 0) {
            atomicAdd(&B.stats[kStatPrimary], (unsigned long long)n_rays);
            atomicAdd(&B.stats[kStatPrimaryHits], (unsigned long long)n_hits);
            atomicAdd(&B.stats[kStatCoarseProbes], p0);
            atomicAdd(&B.stats[kStatBrickEntries], p1);
            atomicAdd(&B.stats[kStatFineProbes], p2);
        }
    }
}

}  // namespace vxrt
