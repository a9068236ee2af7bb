// vxrt_kernels.hip -- gfx950 kernels: per-pixel render (screenDispatch, VoxelRT/Renderer.cu:179-276),
// batch trace (dispatch, VoxelRT/VolumeRaytracer.cu:95-117) and the strip de-interleave used after the
// multi-GPU gather, with their launchers.  The product kernels are k_render_persist2 (vxrt_persist2.hpp),
// k_trace_batch_persist (vxrt_batch_persist.hpp) and k_trace_batch_wave2 below, all on the tracer of vxrt_wave2.hpp;
// k_render and k_trace_batch are the straightforward per-lane loops of vxrt_device.hpp, kept as the cross-check
// (kernel variant 1).
#include "vxrt_kernels.hpp"
#include "vxrt_wave2.hpp"

#include <algorithm>
#include <cstdlib>

namespace vxrt {

// x^32 by five binary64 squarings rounded once to binary32: this build's definition of
// powf(x, 32) at Renderer.cu:114 (libm powf differs in the last ulp between hosts and GPUs).
__device__ __forceinline__ float pow32(float x)
{
    double p = (double)x;
    p *= p;
    p *= p;
    p *= p;
    p *= p;
    p *= p;
    return (float)p;
}

// cudaNoise::hash / randomFloat (cuda_noise.cuh:44-54,66-71)
__device__ __forceinline__ uint32_t hash32(uint32_t s)
{
    s = (s + 0x7ed55d16u) + (s << 12);
    s = (s ^ 0xc761c23cu) ^ (s >> 19);
    s = (s + 0x165667b1u) + (s << 5);
    s = (s + 0xd3a2646cu) ^ (s << 9);
    s = (s + 0xfd7046c5u) + (s << 3);
    s = (s ^ 0xb55a4f09u) ^ (s >> 16);
    return s;
}
__device__ __forceinline__ float random_float(uint32_t s) { return (float)hash32(s) / 4294967296.0f; }

__device__ __forceinline__ unsigned long long wave_sum(uint32_t v)
{
    unsigned long long t = v;
    for (int off = 32; off > 0; off >>= 1)
        t += __shfl_xor(t, off, 64);
    return t;
}

struct PixelSink {
    const RenderArgs& A;
    int out_row;  // row inside the destination buffers (frame row, or packed shard row)
    uint8_t* fb;  // the view's framebuffer and optional colour AOV
    float* color_aov;
    __device__ void put(int x, int y, f3 c) const
    {
        if ((uint32_t)x >= A.width || (uint32_t)y >= A.height)
            return;
        size_t i = (size_t)out_row * A.width + (size_t)x;
        if (color_aov) {
            color_aov[i * 3 + 0] = c.x;
            color_aov[i * 3 + 1] = c.y;
            color_aov[i * 3 + 2] = c.z;
        }
        // setPixelColor (Renderer.cu:72-87): clamp, *255, truncate; bytes b,g,r,a
        float r = lo(hi(c.x, 0), 1), g = lo(hi(c.y, 0), 1), b = lo(hi(c.z, 0), 1);
        uint32_t px = (uint32_t)(b * 255) | ((uint32_t)(g * 255) << 8) | ((uint32_t)(r * 255) << 16) | 0xFF000000u;
        reinterpret_cast<uint32_t*>(fb)[i] = px;
    }
};

// Temporal accumulation (include/vxrt.h, vxrt_render_flags.d_accum): add the pre-tonemap colour of a shaded hit pixel to its
// history and return the mean; the first frame of a history returns the colour itself.  One float4 load + store per pixel.
__device__ __forceinline__ f3 accumulate_color(const RenderArgs& A, int out_row, int x, f3 c)
{
    float4* h = A.accum + ((size_t)out_row * A.width + (size_t)x);
    float4 v = *h;
    if (A.accum_reset || v.w == 0.0f) {
        *h = make_float4(c.x, c.y, c.z, 1.0f);
        return c;
    }
    v = make_float4(v.x + c.x, v.y + c.y, v.z + c.z, v.w + 1.0f);
    *h = v;
    return mk3(v.x / v.w, v.y / v.w, v.z / v.w);
}

// calculateColor (Renderer.cu:90-168) with the shadow ray (:102) and the sample count (:123) as run-time flags
__device__ f3 shade(const RenderArgs& A, uint32_t tx, uint32_t ty, f3 cam, f3 normal, f3 position,
                    RayCounters& cnt, uint32_t& n_shadow, uint32_t& n_bounce)
{
    const f3 L = A.light_dir;
    f3 sray = A.light_unit;
    f3 spos = position + sray * 0.01f;
    bool shadowed = false;
    if (A.shadow) {
        TraceResult t;
        n_shadow += 1;
        trace_direct(A.W, kMaxSteps, spos, sray, t, cnt);
        shadowed = t.hit;
    }
    float l_dot = hi(dot3(normal, L), 0) * (float)(shadowed ? 0 : 1);
    f3 diffuse = A.light_color * l_dot;
    float up_dot = normal.x * 0.0f + normal.y * 1.0f + normal.z * 0.0f;
    float t = (float)((double)up_dot * 0.5 + 0.5);
    f3 color = diffuse + A.ambient * (0.25f + t * (1.0f - 0.25f));
    if (!shadowed) {
        f3 view = unit3(position - cam);
        f3 refl = reflect3(L, normal);
        float spec = pow32(hi(dot3(view, refl), 0));
        color.x += spec * A.light_color.x;
        color.y += spec * A.light_color.y;
        color.z += spec * A.light_color.z;
    }
    if (l_dot == 0 || A.bounce_all_hits) {
        const int samples = A.bounce_samples;
        uint32_t seed = ty * A.width + tx;
        float occl = 0.0f;
        for (int i = 0; i < samples; ++i) {
            uint32_t si = seed + (uint32_t)i * 1000u + (A.frame_number + 1u) * 1000u;
            f3 sd = mk3(random_float(si) * 2 - 1, random_float(si * 10u) * 2 - 1, random_float(si * 100u) * 2 - 1);
            sd = unit3(sd);
            if (dot3(sd, normal) < 0)
                sd = reflect3(sd, normal);
            f3 sp = position + normal * 0.01f;
            TraceResult tr;
            n_bounce += 1;
            trace_direct(A.W, 8, sp, sd, tr, cnt);
            if (!tr.hit) {
                occl += 1.0f;
            } else if (A.bounce_depth >= 2) {  // extension: second bounce from the sample ray's hit point
                const f3 n2 = mk3(-tr.normal.x, -tr.normal.y, -tr.normal.z);
                const uint32_t s2 = si + 500u;
                f3 d2 = mk3(random_float(s2) * 2 - 1, random_float(s2 * 10u) * 2 - 1, random_float(s2 * 100u) * 2 - 1);
                d2 = unit3(d2);
                if (dot3(d2, n2) < 0)
                    d2 = reflect3(d2, n2);
                TraceResult t2;
                n_bounce += 1;
                trace_direct(A.W, 8, tr.pos + n2 * 0.01f, d2, t2, cnt);
                if (!t2.hit)
                    occl += 0.5f;
            }
        }
        if (samples > 0)
            occl /= (float)samples;
        else
            occl = 1.0f;
        color = color * occl;
    }
    return color;
}

template <bool STATS>
__global__ __launch_bounds__(256) void k_render(RenderArgs A)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t tile = blockDim.x == 64u ? 8u : 16u;  // one 8x8 wave tile per workgroup, or 2x2 of them
    const uint32_t tx = blockIdx.x * tile + (wave & 1) * 8 + (lane & 7);
    const uint32_t row = blockIdx.y * tile + (wave >> 1) * 8 + (lane >> 3);

    RayCounters cnt = {0, 0, 0};
    uint32_t n_primary = 0, n_shadow = 0, n_bounce = 0, n_hits = 0;

    // launch row -> the reference launch's thread row `ty`
    uint32_t ty = row;
    const bool sharded = A.strip_count > 1;
    if (sharded && !A.checkerboard) {
        uint32_t strip = (row / (uint32_t)A.strip_rows) * (uint32_t)A.strip_count + (uint32_t)A.strip_index;
        ty = strip * (uint32_t)A.strip_rows + row % (uint32_t)A.strip_rows;
    }
    bool live = row < A.launch_rows;
    int x = (int)tx, y = (int)ty;
    if (A.checkerboard) {  // Renderer.cu:186-194
        y *= 2;
        if ((x % 2) == 0)
            y += 1;
        if (A.frame_number % 2 == 0)
            y += 1;
    }
    live = live && (uint32_t)x < A.width && (uint32_t)y < A.height;
    if (live && sharded && ((uint32_t)y / (uint32_t)A.strip_rows) % (uint32_t)A.strip_count != (uint32_t)A.strip_index)
        live = false;

    if (live) {
        const int Wd = (int)A.width, Hd = (int)A.height;
        int out_row = y;
        if (A.compact && sharded)
            out_row = (int)((((uint32_t)y / (uint32_t)A.strip_rows) / (uint32_t)A.strip_count) * (uint32_t)A.strip_rows +
                            (uint32_t)y % (uint32_t)A.strip_rows);
        PixelSink sink{A, out_row, A.fb, A.color_aov};

        float u = (float)x / (float)Wd, v = (float)y / (float)Hd;
        f3 origin = A.origin;
        f3 ray;
        if (A.ortho) {  // getRayDirectionOrtho, Renderer.cu:61-70
            ray = A.fwd;
            origin = origin + ((A.right * (u * 2 - 1)) * A.ortho_x) * A.ratio;
            origin = origin + (A.up * (v * 2 - 1)) * A.ortho_y;
        } else {  // getRayDirection, Renderer.cu:44-59
            float su = u * 2 - 1, sv = v * 2 - 1;
            ray.x = A.fwd.x + su * A.kx * A.right.x + sv * A.ky * A.up.x;
            ray.y = A.fwd.y + su * A.kx * A.right.y + sv * A.ky * A.up.y;
            ray.z = A.fwd.z + su * A.kx * A.right.z + sv * A.ky * A.up.z;
            ray = unit3(ray);
        }
        TraceResult pr;
        n_primary = 1;
        trace_direct(A.W, kMaxSteps, origin, ray, pr, cnt);
        f3 normal = mk3(-pr.normal.x, -pr.normal.y, -pr.normal.z);
        int steps = pr.steps;
        if (A.hit_aov)
            A.hit_aov[(size_t)out_row * A.width + (size_t)x] =
                pr.hit ? (long long)pr.vx + (long long)A.W.X * ((long long)pr.vy + (long long)A.W.Y * (long long)pr.vz)
                       : -1ll;
        if (pr.hit) {
            n_hits = 1;
            if (A.mode == 1) {  // DEBUG_VIEW quadrants, Renderer.cu:215-243
                f3 dv = pr.pos - origin;
                float dist = sqrtf(dot3(dv, dv));
                const float wrap = (float)(1.0 + 1e-6);
                f3 hp = mk3(fmodf(pr.pos.x / 128.0f, wrap), fmodf(pr.pos.y / 128.0f, wrap),
                            fmodf(pr.pos.z / 128.0f, wrap));
                if (x < (Wd >> 1) && y < (Hd >> 1))
                    sink.put(x, y, normal);
                else if (x >= (Wd >> 1) && y < (Hd >> 1))
                    sink.put(x, y, hp);
                else if (x < (Wd >> 1)) {
                } else
                    sink.put(x, y, mk3(dist * 0.01f, 0, 0));
            } else {  // Renderer.cu:245-251
                f3 c = shade(A, tx, ty, origin, normal, pr.pos, cnt, n_shadow, n_bounce);
                if (A.accum)
                    c = accumulate_color(A, out_row, x, c);
                c = mk3(c.x / (c.x + 1.0f), c.y / (c.y + 1.0f), c.z / (c.z + 1.0f));  // Tonemap, :170-177
                c = mk3(lo(hi(c.x, 0), 1), lo(hi(c.y, 0), 1), lo(hi(c.z, 0), 1));
                sink.put(x, y, c);
            }
        } else {
            sink.put(x, y, ray);  // Renderer.cu:254-258
        }
        if (tx == (A.width >> 1) && ty == (A.height >> 1))  // crosshair on launch coordinates, :261-268
            sink.put(x, y, mk3(10, 10, 10));
        if (A.mode == 1 && x < (Wd >> 1) && y > (Hd >> 1))  // :270-275
            sink.put(x, y, mk3((float)steps / 256.0f, 0, 0));
    }

    // one set of atomics per wavefront
    unsigned long long s0 = wave_sum(n_primary), s1 = wave_sum(n_shadow), s2 = wave_sum(n_bounce),
                       s3 = wave_sum(n_hits);
    unsigned long long* const stats = A.stats ? stats_row_of<kStatRows, kStatRowStride>(A.stats, blockIdx.x + blockIdx.y * gridDim.x) : nullptr;
    if (lane == 0 && stats) {
        atomicAdd(&stats[kStatPrimary], s0);
        atomicAdd(&stats[kStatShadow], s1);
        atomicAdd(&stats[kStatBounce], s2);
        atomicAdd(&stats[kStatPrimaryHits], s3);
    }
    if (STATS) {
        unsigned long long p0 = wave_sum(cnt.coarse_probes), p1 = wave_sum(cnt.brick_entries),
                           p2 = wave_sum(cnt.fine_probes);
        if (lane == 0 && stats) {
            atomicAdd(&stats[kStatCoarseProbes], p0);
            atomicAdd(&stats[kStatBrickEntries], p1);
            atomicAdd(&stats[kStatFineProbes], p2);
        }
    }
}

template <bool STATS>
__global__ __launch_bounds__(256) void k_trace_batch(BatchArgs B)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    RayCounters cnt = {0, 0, 0};
    uint32_t rays = 0, hits = 0;
    if (i < B.n) {
        f3 o = mk3(B.origins[3 * i], B.origins[3 * i + 1], B.origins[3 * i + 2]);
        f3 d = mk3(B.dirs[3 * i], B.dirs[3 * i + 1], B.dirs[3 * i + 2]);
        TraceResult t;
        if (ray_valid(o, d)) {
            trace_direct(B.W, B.max_steps, o, d, t, cnt);
        } else {  // defined by this build (include/vxrt.h, ray validity): a miss with 0 steps
            t.hit = false;
            t.steps = 0;
            t.normal = mk3(0, 0, 0);
            t.pos = mk3(kInf, kInf, kInf);
            t.vx = t.vy = t.vz = 0;
        }
        rays = 1;
        hits = t.hit ? 1 : 0;
        f3 p = t.hit ? t.pos : mk3(kInf, kInf, kInf);  // dispatch, VolumeRaytracer.cu:105-113
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
            B.voxel[i] = t.hit ? (long long)t.vx + (long long)B.W.X * ((long long)t.vy + (long long)B.W.Y * (long long)t.vz)
                               : -1ll;
    }
    if (STATS && B.stats) {
        const int lane = threadIdx.x & 63;
        unsigned long long r = wave_sum(rays), h = wave_sum(hits), p0 = wave_sum(cnt.coarse_probes),
                           p1 = wave_sum(cnt.brick_entries), p2 = wave_sum(cnt.fine_probes);
        if (lane == 0) {
            unsigned long long* const stats = stats_row_of<kStatRows, kStatRowStride>(B.stats, blockIdx.x);
            atomicAdd(&stats[kStatPrimary], r);
            atomicAdd(&stats[kStatPrimaryHits], h);
            atomicAdd(&stats[kStatCoarseProbes], p0);
            atomicAdd(&stats[kStatBrickEntries], p1);
            atomicAdd(&stats[kStatFineProbes], p2);
        }
    }
}

// packed shard buffers -> full frame: 16 bytes (4 pixels) per lane, rows are whole strips
// blockIdx.y = view of a multi-view step: view j's packed rows sit `view_stride_vec` further into every shard's
// contribution, its frame `fb_stride_vec` further into the output
__global__ __launch_bounds__(256) void k_deinterleave(const uint4* __restrict__ shards, unsigned long long shard_stride_vec,
                                                      uint4* __restrict__ fb, uint32_t width_vec, uint32_t height,
                                                      uint32_t strip_rows, uint32_t strip_count,
                                                      unsigned long long view_stride_vec, unsigned long long fb_stride_vec)
{
    shards += (unsigned long long)blockIdx.y * view_stride_vec;
    fb += (unsigned long long)blockIdx.y * fb_stride_vec;
    const unsigned long long total = (unsigned long long)width_vec * height;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        uint32_t y = (uint32_t)(i / width_vec), xv = (uint32_t)(i % width_vec);
        uint32_t strip = y / strip_rows, shard = strip % strip_count;
        uint32_t local_row = (strip / strip_count) * strip_rows + y % strip_rows;
        fb[i] = shards[(unsigned long long)shard * shard_stride_vec + (unsigned long long)local_row * width_vec + xv];
    }
}

// ---- wave-level kernels (vxrt_wave.hpp): every trace is entered by the whole wavefront at a converged point
// with an `active` predicate, so the ballots inside see all 64 lanes ---------------------------------------

}  // namespace vxrt

#include "vxrt_persist2.hpp"

namespace vxrt {

// The batch query one ray per lane, the tracer's cold fields in LDS: what batches too small for the persistent queue take
// (BASELINE configs[0]'s million-ray fan: 17.7 Grays/s against 16.6 for the straightforward loops, tools/batch_probe.py).
template <bool STATS, bool WIDE>
__global__ __launch_bounds__(256) void k_trace_batch_wave2(BatchArgs B)
{
    __shared__ uint32_t cold_block[4][CF_TRACER_FIELDS * 64];
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < B.n;
    const unsigned long long j = live ? i : 0ull;
    f3 o = mk3(B.origins[3 * j], B.origins[3 * j + 1], B.origins[3 * j + 2]);
    f3 d = mk3(B.dirs[3 * j], B.dirs[3 * j + 1], B.dirs[3 * j + 2]);
    TraceResult t;
    t.hit = false;
    t.steps = 0;
    t.pos = t.normal = mk3(0, 0, 0);
    t.vx = t.vy = t.vz = 0;
    t.ncode = 0u;
    RayCounters cnt{0u, 0u, 0u, 0u, 0u};
    trace_wave2<1, STATS, WIDE, true>(B.W, B.max_steps, live && ray_valid(o, d), o, d, t, &cold_block[threadIdx.x >> 6][threadIdx.x & 63], &cnt);
    if (live) {
        f3 p = t.hit ? t.pos : mk3(kInf, kInf, kInf);
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
            B.voxel[i] = t.hit ? (long long)t.vx + (long long)B.W.X * ((long long)t.vy + (long long)B.W.Y * (long long)t.vz)
                               : -1ll;
    }
    if (STATS && B.stats) {
        const unsigned long long r = wave_sum(live ? 1u : 0u), h = wave_sum(live && t.hit ? 1u : 0u), p0 = wave_sum(cnt.coarse_probes),
                                 p1 = wave_sum(cnt.brick_entries), p2 = wave_sum(cnt.fine_probes), g0 = wave_sum(cnt.slack_loads),
                                 g1 = wave_sum(cnt.stray_loads);
        if ((threadIdx.x & 63) == 0) {
            unsigned long long* const stats = stats_row_of<kStatRows, kStatRowStride>(B.stats, blockIdx.x);
            atomicAdd(&stats[kStatGuardSlack], g0);
            atomicAdd(&stats[kStatGuardStray], g1);
            atomicAdd(&stats[kStatPrimary], r);
            atomicAdd(&stats[kStatPrimaryHits], h);
            atomicAdd(&stats[kStatCoarseProbes], p0);
            atomicAdd(&stats[kStatBrickEntries], p1);
            atomicAdd(&stats[kStatFineProbes], p2);
        }
    }
}

}  // namespace vxrt

#include "vxrt_batch_persist.hpp"

namespace vxrt {

// The kernel a render launch runs (vxrt_set_kernel_variant): 7 = the persistent kernel of vxrt_persist2.hpp, 1 = the
// straightforward per-lane loops (cross-check), 4 = the default, which is 7 for every launch shape and world.
int resolve_render_variant(const RenderArgs& A, int variant)
{
    (void)A;
    return variant == 1 ? 1 : 7;
}

// The persistent grid of a launch.  A full grid is VXRT_PERSIST2_OCC waves per SIMD: what a large launch wants (latency hiding
// in steady state).  A small launch ends in a tail in which every resident wave still carries a few pixel chains at low lane
// utilisation, and that tail's work grows with the number of waves while its length shrinks with the waves per SIMD; so a
// launch with few tiles per wave of the full grid starts fewer waves, kTilesPerWave tiles of 64 rays each (a tile counts once
// per ray kind enabled: primary, shadow, bounce), but never fewer than one wave per SIMD while there are four tiles for it.
// Measured on one MI355X (profiles/r04_grid_policy.md): 1080p primary rays only (32 400 tiles), one view per launch:
// 5120 / 3840 / 2560 / 1920 / 1280 / 960 waves = 2830 / 3111 / 3481 / 3654 / 3352 / 2948 Mrays/s; 1080p primary + shadow +
// bounce (97 200 weighted tiles): 5120 / 3840 / 2560 waves = 4626 / 4567 / 4098; 16 views per launch: the full grid.
// That sweep was taken while every wave's tickets and leaving atomics queued on single addresses, which penalised waves as
// such (17 tiles per wave then); with the sharded queue and the counter rows (profiles/r04_work_queue.md section 8) the 1080p
// primary-only frame runs 6276 / 6500 / 6284 / 5673-5686 / 4745 Mrays/s at 6 / 9 / 12 / 17 / 24 tiles per wave, the shaded
// one 5945-6029 at anything up to 17 (its grid is the full one from 19 down) and 5815 / 5215 at 24 / 34.
constexpr unsigned kTilesPerWave = 10u;
static unsigned render_grid_waves(const RenderArgs& A, unsigned long long ntiles)
{
    const unsigned resident = std::max(1u, A.persistent_waves / 4u * (unsigned)VXRT_PERSIST2_OCC);
    unsigned tiles_per_wave = kTilesPerWave;
#ifdef VXRT_EXPERIMENTS
    static const int env_tpw = getenv("VXRT_TILES_PER_WAVE") ? atoi(getenv("VXRT_TILES_PER_WAVE")) : 0;
    if (env_tpw > 0)
        tiles_per_wave = (unsigned)env_tpw;
#endif
    const unsigned long long kinds = 1ull + (A.mode == 0 && A.shadow ? 1ull : 0ull) + (A.mode == 0 && A.bounce_samples > 0 ? 1ull : 0ull);
    const unsigned long long weighted = ntiles * kinds;
    const unsigned long long by_work = (weighted + tiles_per_wave - 1u) / tiles_per_wave;
    const unsigned long long floor_waves = std::min<unsigned long long>(A.persistent_waves / 4u, (weighted + 3ull) / 4ull);  // one per SIMD
    const unsigned long long want = std::max(by_work, floor_waves);
    return (unsigned)std::max<unsigned long long>(1ull, std::min<unsigned long long>(std::min<unsigned long long>(want, resident), ntiles));
}

hipError_t launch_render(const RenderArgs& A, bool stats, int variant, hipStream_t stream)
{
    if (A.width == 0 || A.launch_rows == 0)
        return hipSuccess;
    variant = resolve_render_variant(A, variant);
    if (variant == 1) {
        // one wave per workgroup: no wave waits for a slower sibling before its slot is reused
        const dim3 block(64, 1, 1), grid((A.width + 7u) / 8u, (A.launch_rows + 7u) / 8u, 1);
        if (stats)
            hipLaunchKernelGGL(k_render<true>, grid, block, 0, stream, A);
        else
            hipLaunchKernelGGL(k_render<false>, grid, block, 0, stream, A);
        return hipSuccess;
    }
    const unsigned long long ntiles =
        (unsigned long long)((A.width + 7u) / 8u) * ((A.launch_rows + 7u) / 8u) * (A.nviews ? A.nviews : 1u);
    const unsigned waves = render_grid_waves(A, ntiles);
    const hipError_t e = hipMemsetAsync(A.tile_counter, 0, sizeof(unsigned int) * kQueueWords, stream);
    if (e != hipSuccess)  // a kernel started on a queue head that was not reset would skip or repeat tiles
        return e;
    const bool second_bounce = A.bounce_depth >= 2 && A.bounce_samples > 0;
    const dim3 g(waves), b(64);
    // (ordinary grids and wide ones -- beyond the tracer's packed step counters -- run their own instantiation of the kernel)
#define VXRT_LAUNCH_PERSIST(S, B2, M)                                                                    \
    do {                                                                                                 \
        if (A.W.c_wide)                                                                                  \
            hipLaunchKernelGGL((k_render_persist2<S, B2, M, true>), g, b, 0, stream, A);                 \
        else                                                                                             \
            hipLaunchKernelGGL((k_render_persist2<S, B2, M, false>), g, b, 0, stream, A);                \
    } while (0)
    if (A.nviews) {
        if (stats && second_bounce) VXRT_LAUNCH_PERSIST(true, true, true);
        else if (stats) VXRT_LAUNCH_PERSIST(true, false, true);
        else if (second_bounce) VXRT_LAUNCH_PERSIST(false, true, true);
        else VXRT_LAUNCH_PERSIST(false, false, true);
    } else {
        if (stats && second_bounce) VXRT_LAUNCH_PERSIST(true, true, false);
        else if (stats) VXRT_LAUNCH_PERSIST(true, false, false);
        else if (second_bounce) VXRT_LAUNCH_PERSIST(false, true, false);
        else VXRT_LAUNCH_PERSIST(false, false, false);
    }
#undef VXRT_LAUNCH_PERSIST
    return hipSuccess;
}

hipError_t launch_trace_batch(const BatchArgs& B, bool stats, int variant, hipStream_t stream)
{
    if (B.n == 0)
        return hipSuccess;
    dim3 block(256, 1, 1);
    dim3 grid((unsigned)((B.n + 255) / 256), 1, 1);
    if (variant == 1) {  // the straightforward loops (cross-check)
        if (stats)
            hipLaunchKernelGGL(k_trace_batch<true>, grid, block, 0, stream, B);
        else
            hipLaunchKernelGGL(k_trace_batch<false>, grid, block, 0, stream, B);
        return hipSuccess;
    }
    // persistent wavefronts pulling tickets of consecutive rays, for batches of at least 8 rays per lane of the persistent
    // grid (below that the queue cannot balance much, and short rays are cheaper one per lane)
    const unsigned resident = B.persistent_waves / 4u * (unsigned)VXRT_BATCH_OCC;
    const unsigned long long tickets = (B.n + kBatchTicket - 1) / kBatchTicket;
    const bool persistent = B.ticket && resident && B.n >= 8ull * 64ull * resident && tickets < (unsigned long long)kQueueDry;
    if (!persistent) {
        if (stats && B.W.c_wide)
            hipLaunchKernelGGL((k_trace_batch_wave2<true, true>), grid, block, 0, stream, B);
        else if (stats)
            hipLaunchKernelGGL((k_trace_batch_wave2<true, false>), grid, block, 0, stream, B);
        else if (B.W.c_wide)
            hipLaunchKernelGGL((k_trace_batch_wave2<false, true>), grid, block, 0, stream, B);
        else
            hipLaunchKernelGGL((k_trace_batch_wave2<false, false>), grid, block, 0, stream, B);
        return hipSuccess;
    }
    const hipError_t e = hipMemsetAsync(B.ticket, 0, sizeof(unsigned int) * kQueueWords, stream);
    if (e != hipSuccess)
        return e;
    const dim3 g((unsigned)(tickets < resident ? tickets : resident)), b(64);
    if (stats && B.W.c_wide)
        hipLaunchKernelGGL((k_trace_batch_persist<true, true>), g, b, 0, stream, B);
    else if (stats)
        hipLaunchKernelGGL((k_trace_batch_persist<true, false>), g, b, 0, stream, B);
    else if (B.W.c_wide)
        hipLaunchKernelGGL((k_trace_batch_persist<false, true>), g, b, 0, stream, B);
    else
        hipLaunchKernelGGL((k_trace_batch_persist<false, false>), g, b, 0, stream, B);
    return hipSuccess;
}

void launch_deinterleave(const void* shards, unsigned long long shard_stride_bytes, void* fb, uint32_t width,
                         uint32_t height, uint32_t strip_rows, uint32_t strip_count, hipStream_t stream, uint32_t n_views,
                         unsigned long long view_stride_bytes, unsigned long long fb_stride_bytes)
{
    uint32_t width_vec = width / 4;  // caller guarantees width % 4 == 0
    unsigned long long total = (unsigned long long)width_vec * height;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 2048)
        blocks = 2048;
    if (blocks == 0)
        return;
    if (n_views == 0)
        return;
    hipLaunchKernelGGL(k_deinterleave, dim3(blocks, n_views), dim3(256), 0, stream, (const uint4*)shards, shard_stride_bytes / 16,
                       (uint4*)fb, width_vec, height, strip_rows, strip_count, view_stride_bytes / 16, fb_stride_bytes / 16);
}

}  // namespace vxrt
