// Debug harness: runs vxrt::trace_wave2 (one lane per wave) and vxrt::trace_direct on the host against the C oracle.
// build: g++ -O1 -g -std=c++17 -ffp-contract=off -Itests/tools/hoststub -Ioracle tests/tools/host_wave_check.cpp oracle/vxo_*.c -lm -lpthread
#include "../voxelengine_amd/csrc/vxrt_wave2.hpp"
extern "C" {
#include "vxo.h"
}
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace vxrt;
int main(int argc, char** argv)
{
    // usage: host_wave_check f S density n [Sy Sz [wide]]   (S = voxels along x; Sy, Sz default to S; wide = 1 forces the
    // wide-grid code of vxrt_wave2.hpp on a grid that would not need it)
    int f = argc > 1 ? atoi(argv[1]) : 8, S = argc > 2 ? atoi(argv[2]) : 64;
    double dens = argc > 3 ? atof(argv[3]) : 0.01;
    int n = argc > 4 ? atoi(argv[4]) : 30000;
    const int Sy = argc > 5 ? atoi(argv[5]) : S, Sz = argc > 6 ? atoi(argv[6]) : S;
    const int force_wide = argc > 7 ? atoi(argv[7]) : 0;
    std::vector<uint32_t> dense((size_t)S * Sy * Sz / 32, 0);
    srand(1);
    for (int z = 0; z < Sz; ++z) for (int y = 0; y < Sy; ++y) for (int x = 0; x < S; ++x)
        if (rand() / (double)RAND_MAX < dens) { uint64_t i = vxo_sample_index64(x, y, z, S, Sy); dense[i >> 5] |= 1u << (i & 31); }
    vxo_world* w = vxo_build_brickmap(dense.data(), S, Sy, Sz, f);
    // the oracle's tables are in the reference's tiled order; the tracer reads the HBM order (linear x, z, y on both
    // levels): re-order on the host what the library re-orders on the device (vxrt_worldgen.hip)
    const int cx = w->cdims[0], cy = w->cdims[1], cz = w->cdims[2];
    std::vector<uint2> meta(w->ncells);
    std::vector<uint32_t> coarse((w->ncells + 31) / 32, 0u);
    for (int z = 0; z < cz; ++z) for (int y = 0; y < cy; ++y) for (int x = 0; x < cx; ++x) {
        const uint64_t t = ref_tiled_index(x, y, z, cx / 8, cy / 8), i = hbm_index(x, y, z, cx, cz);
        uint32_t p = 0;
        if (w->brick_slot[t] != VXO_EMPTY_SLOT) for (int k = 0; k < 6; ++k) p |= (uint32_t)(int)w->bounds[t * 6 + k] << (5 * k);
        meta[i] = make_uint2(w->brick_slot[t], p);
        if ((w->coarse_bits[t >> 5] >> (t & 31)) & 1u) coarse[i >> 5] |= 1u << (i & 31);
    }
    const uint32_t bw = f * f * f / 32;
    std::vector<uint32_t> pool((size_t)w->nslots * bw, 0u);
    for (uint64_t s = 0; s < w->nslots; ++s)
        for (int z = 0; z < f; ++z) for (int y = 0; y < f; ++y) for (int x = 0; x < f; ++x) {
            const uint32_t t = ref_tiled_index(x, y, z, f / 8, f / 8), i = (uint32_t)hbm_index(x, y, z, f, f);
            if ((w->pool[s * bw + (t >> 5)] >> (t & 31)) & 1u) pool[s * bw + (i >> 5)] |= 1u << (i & 31);
        }
    // the second tracer (vxrt_wave2.hpp) needs addressable slack around its tables: one x-z slice around the coarse bits,
    // one brick around the pool (a lane that has just left the grid issues one more load)
    const size_t cslack = (size_t)cx * cz / 32 + 1;
    std::vector<uint32_t> coarse_pad(coarse.size() + 2 * cslack, 0xA5A5A5A5u), pool_pad(pool.size() + 2 * (size_t)bw, 0x5A5A5A5Au);
    memcpy(coarse_pad.data() + cslack, coarse.data(), coarse.size() * 4);
    memcpy(pool_pad.data() + bw, pool.data(), pool.size() * 4);
    WorldView W{};
    W.coarse_bits = coarse_pad.data() + cslack; W.cell_meta = meta.data(); W.pool = pool_pad.data() + bw;
    W.cx = cx; W.cy = cy; W.cz = cz; W.c_row = cx; W.c_slice = cx * cz;
    W.f = f; W.f_row = f; W.f_slice = f * f; W.brick_words = bw; W.ff = (float)f; W.inv_f = 1.0f / f;
    W.wmax_x = (float)((double)W.cx - 1e-6); W.wmax_y = (float)((double)W.cy - 1e-6); W.wmax_z = (float)((double)W.cz - 1e-6);
    W.X = S; W.Y = Sy;
    W.c_wide = (force_wide || grid_is_wide(cx, cy, cz)) ? 1 : 0;
    W.coarse_end = W.coarse_bits + coarse.size(); W.coarse_lo = coarse_pad.data(); W.coarse_hi = coarse_pad.data() + coarse_pad.size();
    W.pool_end = W.pool + pool.size(); W.pool_lo = pool_pad.data(); W.pool_hi = pool_pad.data() + pool_pad.size();
    unsigned long long slack_loads = 0, stray_loads = 0;  // load guard: loads beyond a table inside the slack / outside it
    const float ext[3] = {(float)S, (float)Sy, (float)Sz};
    // a tracer that lives across rays, as a lane of the persistent kernels does (k_render_persist2, k_trace_batch_persist):
    // whatever a ray leaves behind in the lane's state must not reach the next ray
    static uint32_t cold_persist[CF_TRACER_FIELDS * 64];
    WaveTracerT<false> TPn;
    WaveTracerT<true> TPw;
    TPn.init(W, cold_persist);
    TPw.init(W, cold_persist);
    auto trace_persistent_on = [&](auto& TP, const f3 o, const f3 d, TraceResult& out, RayCounters& c) {
        TP.cnt = RayCounters{0u, 0u, 0u};
        TP.begin_ray(W, o, d, 2048);
        TP.after_begin_ray(true);
        for (;;) {   // the cascade of the persistent kernels: tight box, end of walk, (ray finished), probes
            if (TP.st == ST_BOX) TP.template phase_box<true>(W);
            if (waits_for_end(TP.st)) TP.template phase_end<true>(W);
            if (TP.st == ST_DONE) break;
            TP.template probe_pairs<2, true>(W);
        }
        TP.result(W, out);
        c = TP.cnt;
    };
    auto trace_persistent = [&](const f3 o, const f3 d, TraceResult& out, RayCounters& c) {
        if (W.c_wide) trace_persistent_on(TPw, o, d, out, c); else trace_persistent_on(TPn, o, d, out, c);
    };
    int bad = 0, n_hits = 0, n_long = 0, n_exhausted = 0;  // coverage of the run: hits, walks beyond 1024 steps, rays that ran into MAX_STEPS
    for (int i = 0; i < n; ++i) {
        float o[3], d[3];
        for (int a = 0; a < 3; ++a) { o[a] = (rand() / (float)RAND_MAX) * (i % 3 ? ext[a] : 3 * ext[a]) - (i % 3 ? 0 : ext[a]); d[a] = rand() / (float)RAND_MAX * 2 - 1; }
        if (i % 7 == 0) d[i % 3] = 0;
        if (i % 11 == 0) { o[0] = floorf(o[0]); o[1] = floorf(o[1]); }
        // adversarial families: tiny / denormal direction components, starts exactly on the far faces (edge rule),
        // far-away origins aimed at the grid, axis-aligned rays along cell boundaries
        if (i % 13 == 0) d[(i / 13) % 3] *= 1e-30f;
        if (i % 17 == 0) d[(i / 17) % 3] = 1e-42f;
        if (i % 19 == 0) { o[(i / 19) % 3] = ext[(i / 19) % 3]; d[(i / 19) % 3] = -fabsf(d[(i / 19) % 3]) - 0.01f; }
        if (i % 23 == 0) { for (int a = 0; a < 3; ++a) { o[a] = o[a] * 1000.0f; d[a] = ext[a] * 0.5f - o[a]; } }
        if (i % 31 == 0) { o[0] = o[1] = Sy * (1.5f + (i % 7)); d[0] = d[1] = -fabsf(d[0]) - 0.1f; }  // exact x/y ties through a grid corner
        if (i % 37 == 0) { o[1] = o[2] = -Sy * 0.5f; d[1] = d[2] = fabsf(d[1]) + 0.1f; }
        // long walks along the long axis (wide grids: the packed counters are re-armed, a walk of MAX_STEPS steps ends the ray)
        if (i % 5 == 0 && S > 4 * Sy) { d[0] = (i & 8) ? 1.0f : -1.0f; d[1] *= 0.002f; d[2] *= 0.002f; if (i % 10 == 0) o[0] = d[0] > 0 ? -3.0f : S + 3.0f; }
        if (i % 29 == 0) { d[0] = (i & 1) ? 1.0f : -1.0f; d[1] = d[2] = 0; o[1] = floorf(o[1]); o[2] = floorf(o[2]); }
        // nearly axis-parallel rays from a face: the longest accumulation of one axis' tMax (the exit threshold's margin)
        if (i % 41 == 0) { int a = (i / 41) % 3; d[a] = (i & 2) ? 1.0f : -1.0f; d[(a + 1) % 3] *= 1e-4f; d[(a + 2) % 3] *= 1e-5f; o[a] = d[a] > 0 ? 0.0f : ext[a]; }
        int steps; float nn[3], pp[3] = {0, 0, 0}; int vox[3] = {0, 0, 0}; vxo_ray_stats st{};
        int h = vxo_raytrace(w, 2048, o, d, &steps, nn, pp, vox, &st);
        n_hits += h != 0; n_long += steps > 1024; n_exhausted += steps >= 2048 && !h;
        // the wave-level tracer with one lane per wave, one probe pair between two rounds of votes and three (as
        // k_render_persist2 runs it): results and the probe counters it derives from its packed step counters
        static uint32_t cold_column[CF_TRACER_FIELDS * 64];
        TraceResult t4{}, t5{};
        RayCounters c4{0, 0, 0}, c5{0, 0, 0};
        if (W.c_wide) {   // (the wide-grid instantiation of the tracer, as the launchers pick it)
            trace_wave2<1, true, true>(W, 2048, true, mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2]), t4, cold_column, &c4);
            trace_wave2<3, true, true>(W, 2048, true, mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2]), t5, cold_column, &c5);
        } else {
            trace_wave2<1, true, false>(W, 2048, true, mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2]), t4, cold_column, &c4);
            trace_wave2<3, true, false>(W, 2048, true, mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2]), t5, cold_column, &c5);
        }
        TraceResult t6{}; RayCounters c6{0, 0, 0};
        trace_persistent(mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2]), t6, c6);
        // ... and the straightforward loops the cross-check kernels run
        TraceResult t1{}; RayCounters c1{0, 0, 0};
        trace_direct(W, 2048, mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2]), t1, c1);
        auto same = [&](const TraceResult& t, const RayCounters& c) {
            bool ok = (t.hit == (h != 0)) && t.steps == steps && c.coarse_probes == st.coarse_probes && c.brick_entries == st.brick_entries &&
                      c.fine_probes == st.fine_probes;
            if (h) ok = ok && memcmp(&t.pos, pp, 12) == 0 && t.normal.x == nn[0] && t.normal.y == nn[1] && t.normal.z == nn[2] && t.vx == vox[0] && t.vy == vox[1] && t.vz == vox[2];
            return ok;
        };
        slack_loads += c4.slack_loads + c5.slack_loads + c6.slack_loads;
        stray_loads += c4.stray_loads + c5.stray_loads + c6.stray_loads;
        const bool ok = same(t4, c4) && same(t5, c5) && same(t1, c1) && same(t6, c6) && c4.stray_loads + c5.stray_loads + c6.stray_loads == 0;
        if (!ok && bad++ < 5)
            printf("ray %d o=(%.9g,%.9g,%.9g) d=(%.9g,%.9g,%.9g) oracle hit=%d steps=%d probes=%llu/%llu/%llu pos (%.9g,%.9g,%.9g) vox (%d,%d,%d) | wave2 hit=%d steps=%d probes=%u/%u/%u pos (%.9g,%.9g,%.9g) vox (%d,%d,%d) | x3 %d | direct %d | persistent lane %d: hit=%d steps=%d pos (%.9g,%.9g,%.9g)\n",
                   i, o[0], o[1], o[2], d[0], d[1], d[2], h, steps, (unsigned long long)st.coarse_probes, (unsigned long long)st.brick_entries,
                   (unsigned long long)st.fine_probes, pp[0], pp[1], pp[2], vox[0], vox[1], vox[2], t4.hit, t4.steps, c4.coarse_probes, c4.brick_entries,
                   c4.fine_probes, t4.pos.x, t4.pos.y, t4.pos.z, t4.vx, t4.vy, t4.vz, (int)same(t5, c5), (int)same(t1, c1), (int)same(t6, c6), t6.hit, t6.steps, t6.pos.x, t6.pos.y, t6.pos.z);
    }
    if (host_unsuspected_exits() != 0) {   // a lane left a grid without passing its walk's time threshold: the GPU probe would walk on
        printf("UNSUSPECTED EXITS: %llu\n", host_unsuspected_exits());
        bad += 1;
    }
    printf("mismatches %d of %d  (hits %d, rays of more than 1024 steps %d, of 2048 or more without a hit %d; loads in the tables' slack %llu, outside it %llu)\n", bad, n, n_hits, n_long, n_exhausted, slack_loads, stray_loads);
    return bad != 0;
}
