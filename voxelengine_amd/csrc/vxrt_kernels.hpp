// vxrt_kernels.hpp -- kernel argument blocks shared by the kernels and the host API.
#pragma once

#include "vxrt_device.hpp"

namespace vxrt {

// indices into the per-context device counter block
enum StatSlot {
    kStatPrimary = 0,
    kStatShadow,
    kStatBounce,
    kStatPrimaryHits,
    kStatCoarseProbes,
    kStatBrickEntries,
    kStatFineProbes,
    kStatDbgIters,      // wave-loop iterations, summed over waves (diagnostics)
    kStatDbgWalkLanes,  // walking lanes summed over those iterations
    kStatDbgEndRuns,    // executions of the end-of-walk phase, and (EndLanes) the lanes they served
    kStatDbgBoxRuns,
    kStatDbgNextRuns,   // persistent kernel: executions of the ray-finished phase
    kStatDbgEndLanes,
    kStatDbgBoxLanes,
    kStatDbgNextLanes,
    kStatDbgLifetime,   // persistent kernel: wave lifetime, 100 MHz ticks summed over waves
    kStatDbgDrained,    // persistent kernel: iterations after the tile queue ran dry
    kStatDbgNextTicks,  // persistent kernel: 100 MHz ticks inside the ray-finished phase, summed over waves
    kStatDbgParkTicks,  // ... inside the box and end-of-walk phases
    kStatGuardSlack,    // load guard (probe-counting launches): occupancy loads beyond a table, inside the allocator's slack
    kStatGuardStray,    // ... and outside everything addressable (must stay 0)
    kStatCount
};

// The counters are kStatRows rows of kStatRowStride words; a kernel adds to row (workgroup mod kStatRows) -- stats_row -- and
// vxrt_frame_stats_get sums the rows.  (One row: the four ray counters every wave of a 16-view launch adds when it leaves
// were 20 000 atomics on four addresses, ~12 ns each and one after the other: 1.7 % of the launch, profiles/r04_work_queue.md.)
#ifndef VXRT_STAT_ROWS
#define VXRT_STAT_ROWS 64
#endif
constexpr unsigned kStatRows = VXRT_STAT_ROWS, kStatRowStride = 32;
static_assert(kStatCount <= (int)kStatRowStride, "a row holds every counter");

constexpr unsigned kMaxScheduledTileRows = 512;  // frames up to 4096 launch rows get a tile schedule

// Everything screenDispatch read from dFrameInfo / g_env / its arguments (Renderer.cu:7-24,89,179-181),
// passed by value with the launch: no per-frame symbol copy.
struct RenderArgs {
    WorldView W;
    uint32_t width, height;
    uint32_t frame_number;
    uint32_t launch_rows;  // rows of the launch grid (height, height/2 with checkerboard, or the shard's rows)
    float kx, ky;          // tan(fov/2)*aspect, tan(fov/2), evaluated on the host (Renderer.cu:50-52)
    float ortho_x, ortho_y, ratio;
    f3 origin, fwd, up, right;
    f3 light_dir, light_color, ambient;
    int mode, checkerboard, shadow, bounce_samples, bounce_all_hits, ortho;
    int bounce_depth;  // 2: extension beyond the reference, a sample ray that hits spawns one more ray (include/vxrt.h)
    int strip_rows, strip_count, strip_index, compact;
    int strip_shift;  // log2(strip_rows) when that is a power of two, else -1
    uint8_t* fb;
    float* color_aov;
    long long* hit_aov;
    unsigned long long* stats;
    unsigned int* tile_counter;  // persistent kernel: the queue head of the launch's 8x8 tiles (kQueueWords words, zeroed per launch; queue_take)
    unsigned int persistent_waves;
    const unsigned int* tile_order;  // optional permutation of the launch grid's 8x8 tiles (hand-out order), or NULL
    // default hand-out order, tile-row granular: the k-th tile row handed out is row_order[k] (row_order_n = number
    // of tile rows, 0 = row-major).  Filled per frame by the host from the camera (vxrt_api.hip).
    unsigned int row_order_n;
    uint16_t row_order[kMaxScheduledTileRows];
    // multi-view launch (vxrt_render_views): the per-view members above are unused, every view's live in `views`
    // (device memory); the tile queue runs through view 0's tiles, then view 1's, ...
    const struct ViewArgs* views;
    unsigned int nviews;
    float4* accum;       // temporal accumulation history (vxrt_render_flags.d_accum), single-view launches only, or NULL
    int accum_reset;
    int want_hit_aov;  // some view of the launch has a hit-index AOV
    f3 light_unit;  // normalize(light_dir), the shadow ray (Renderer.cu:97): the same IEEE operations, evaluated once on the host
    f3 light_step;           // light_unit * 0.01f (the shadow ray's offset, Renderer.cu:97)
    float bounce_samples_f;  // (float)bounce_samples
    // correctly rounded reciprocals of (float)width, (float)height and bounce_samples_f, evaluated on the host: the render
    // kernel's x / W, y / H and occlusion mean take them with one correction step (div_rn, vxrt_device.hpp)
    float inv_width, inv_height, inv_bounce_samples;
};

// the per-view part of RenderArgs for a launch that renders several views of the same world
struct ViewArgs {
    f3 origin, fwd, up, right;
    uint32_t frame_number;
    uint32_t row_order_n;
    uint8_t* fb;
    float* color_aov;
    long long* hit_aov;
    uint16_t row_order[kMaxScheduledTileRows];
};
constexpr unsigned kMaxViews = 16;

struct BatchArgs {
    WorldView W;
    const float* origins;
    const float* dirs;
    unsigned long long n;
    float* pos;
    float* normal;
    int* steps;
    uint8_t* hit;
    long long* voxel;
    unsigned long long* stats;
    unsigned int* ticket;     // persistent batch kernel: the queue head of the 64-ray tickets (kQueueWords words, zeroed per launch), or NULL
    unsigned int persistent_waves;
    int max_steps;            // Raytrace's maxSteps (VolumeRaytracer.cu:354,386); kMaxSteps unless the caller lowered it
};

}  // namespace vxrt
