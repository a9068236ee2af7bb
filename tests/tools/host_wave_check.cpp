// Debug harness: runs vxrt::trace_wave (one lane) and vxrt::trace_direct on the host against the C oracle.
// build: g++ -O1 -g -std=c++17 -ffp-contract=off -Itests/tools/hoststub -Ioracle tests/tools/host_wave_check.cpp oracle/vxo_*.c -lm -lpthread
#include "../voxelengine_amd/csrc/vxrt_wave.hpp"
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
    int f = argc > 1 ? atoi(argv[1]) : 8, S = argc > 2 ? atoi(argv[2]) : 64;
    double dens = argc > 3 ? atof(argv[3]) : 0.01;
    int n = argc > 4 ? atoi(argv[4]) : 30000;
    std::vector<uint32_t> dense((size_t)S * S * S / 32, 0);
    srand(1);
    for (int z = 0; z < S; ++z) for (int y = 0; y < S; ++y) for (int x = 0; x < S; ++x)
        if (rand() / (double)RAND_MAX < dens) { uint64_t i = vxo_sample_index64(x, y, z, S, S); dense[i >> 5] |= 1u << (i & 31); }
    vxo_world* w = vxo_build_brickmap(dense.data(), S, S, S, f);
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
    W.X = S; W.Y = S;
    int bad = 0, bad2 = 0;
    for (int i = 0; i < n; ++i) {
        float o[3], d[3];
        for (int a = 0; a < 3; ++a) { o[a] = (rand() / (float)RAND_MAX) * (i % 3 ? S : 3 * S) - (i % 3 ? 0 : S); d[a] = rand() / (float)RAND_MAX * 2 - 1; }
        if (i % 7 == 0) d[i % 3] = 0;
        if (i % 11 == 0) { o[0] = floorf(o[0]); o[1] = floorf(o[1]); }
        // adversarial families: tiny / denormal direction components, starts exactly on the far faces (edge rule),
        // far-away origins aimed at the grid, axis-aligned rays along cell boundaries
        if (i % 13 == 0) d[(i / 13) % 3] *= 1e-30f;
        if (i % 17 == 0) d[(i / 17) % 3] = 1e-42f;
        if (i % 19 == 0) { o[(i / 19) % 3] = (float)S; d[(i / 19) % 3] = -fabsf(d[(i / 19) % 3]) - 0.01f; }
        if (i % 23 == 0) { for (int a = 0; a < 3; ++a) { o[a] = o[a] * 1000.0f; d[a] = S * 0.5f - o[a]; } }
        if (i % 31 == 0) { o[0] = o[1] = S * (1.5f + (i % 7)); d[0] = d[1] = -fabsf(d[0]) - 0.1f; }  // exact x/y ties through a grid corner
        if (i % 37 == 0) { o[1] = o[2] = -S * 0.5f; d[1] = d[2] = fabsf(d[1]) + 0.1f; }
        if (i % 29 == 0) { d[0] = (i & 1) ? 1.0f : -1.0f; d[1] = d[2] = 0; o[1] = floorf(o[1]); o[2] = floorf(o[2]); }
        int steps; float nn[3], pp[3] = {0, 0, 0}; int vox[3] = {0, 0, 0}; vxo_ray_stats st{};
        int h = vxo_raytrace(w, 2048, o, d, &steps, nn, pp, vox, &st);
        TraceResult t; RayCounters c{0, 0, 0};
        trace_wave<true>(W, 2048, true, mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2]), t, c);
        // the same tracer with its cold state outside the registers (on the GPU: an LDS column per lane) must agree field by field
        static uint32_t cold_column[CF_TRACER_FIELDS * 64];
        TraceResult t2; RayCounters c2{0, 0, 0};
        trace_wave<true, false, true>(W, 2048, true, mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2]), t2, c2, nullptr, cold_column);
        bool same2 = t2.hit == t.hit && t2.steps == t.steps && memcmp(&t2.normal, &t.normal, 12) == 0 && c2.coarse_probes == c.coarse_probes &&
                     c2.brick_entries == c.brick_entries && c2.fine_probes == c.fine_probes;
        if (t.hit) same2 = same2 && memcmp(&t2.pos, &t.pos, 12) == 0 && t2.vx == t.vx && t2.vy == t.vy && t2.vz == t.vz;
        // ... and so must a ray that starts from its prepared record (prepare_ray + begin_prepared: the traversal kernel's start)
        TraceResult t3; RayCounters c3{0, 0, 0};
        trace_wave<true, false, true>(W, 2048, true, mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2]), t3, c3, nullptr, cold_column, true);
        same2 = same2 && t3.hit == t.hit && t3.steps == t.steps && memcmp(&t3.normal, &t.normal, 12) == 0 && c3.coarse_probes == c.coarse_probes &&
                c3.brick_entries == c.brick_entries && c3.fine_probes == c.fine_probes;
        if (t.hit) same2 = same2 && memcmp(&t3.pos, &t.pos, 12) == 0 && t3.vx == t.vx && t3.vy == t.vy && t3.vz == t.vz;
        // the tracer built for the vector pipe's two instruction classes (vxrt_wave2.hpp): same results
        if (tracer2_fits(W)) {
            TraceResult t4{};
            RayCounters c4{0, 0, 0}, c5{0, 0, 0};
            trace_wave2<1, true>(W, 2048, true, mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2]), t4, cold_column, &c4);
            // ... and with three probe pairs between two rounds of votes, as k_render_persist2 runs it
            TraceResult t5{};
            trace_wave2<3, true>(W, 2048, true, mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2]), t5, cold_column, &c5);
            bool s4 = t4.hit == t.hit && t4.steps == t.steps && t5.hit == t.hit && t5.steps == t.steps;
            // the probe counters this tracer derives from its packed step counters at the end of each walk: the oracle's
            s4 = s4 && c4.coarse_probes == st.coarse_probes && c4.brick_entries == st.brick_entries && c4.fine_probes == st.fine_probes;
            s4 = s4 && c5.coarse_probes == st.coarse_probes && c5.brick_entries == st.brick_entries && c5.fine_probes == st.fine_probes;
            if (t.hit) s4 = s4 && memcmp(&t5.pos, &t.pos, 12) == 0 && memcmp(&t5.normal, &t.normal, 12) == 0 && t5.vx == t.vx && t5.vy == t.vy && t5.vz == t.vz;
            if (t.hit) s4 = s4 && memcmp(&t4.pos, &t.pos, 12) == 0 && memcmp(&t4.normal, &t.normal, 12) == 0 && t4.vx == t.vx && t4.vy == t.vy && t4.vz == t.vz;
            if (!s4 && bad2++ < 5)
                printf("tracer2: ray %d o=(%.9g,%.9g,%.9g) d=(%.9g,%.9g,%.9g): hit %d/%d steps %d/%d pos (%.9g,%.9g,%.9g)/(%.9g,%.9g,%.9g) vox (%d,%d,%d)/(%d,%d,%d) probes %u/%u/%u vs %llu/%llu/%llu\n", i, o[0], o[1], o[2], d[0], d[1], d[2],
                       t4.hit, t.hit, t4.steps, t.steps, t4.pos.x, t4.pos.y, t4.pos.z, t.pos.x, t.pos.y, t.pos.z, t4.vx, t4.vy, t4.vz, t.vx, t.vy, t.vz,
                       c4.coarse_probes, c4.brick_entries, c4.fine_probes, (unsigned long long)st.coarse_probes, (unsigned long long)st.brick_entries, (unsigned long long)st.fine_probes);
            same2 = same2 && s4;
        }
        bool ok = (t.hit == (h != 0)) && t.steps == steps && c.coarse_probes == st.coarse_probes && c.brick_entries == st.brick_entries && c.fine_probes == st.fine_probes;
        if (h) ok = ok && memcmp(&t.pos, pp, 12) == 0 && t.normal.x == nn[0] && t.normal.y == nn[1] && t.normal.z == nn[2] && t.vx == vox[0] && t.vy == vox[1] && t.vz == vox[2];
        ok = ok && same2;
        if (!ok && bad++ < 5)
            printf("ray %d o=(%.9g,%.9g,%.9g) d=(%.9g,%.9g,%.9g) cpu hit=%d steps=%d probes=%llu/%llu/%llu | wave hit=%d steps=%d probes=%u/%u/%u\n", i, o[0], o[1], o[2], d[0], d[1], d[2], h, steps,
                   (unsigned long long)st.coarse_probes, (unsigned long long)st.brick_entries, (unsigned long long)st.fine_probes, t.hit, t.steps, c.coarse_probes, c.brick_entries, c.fine_probes);
    }
    printf("mismatches %d of %d\n", bad, n);
    return bad != 0;
}
