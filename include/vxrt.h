/*
 * vxrt.h -- C ABI of the MI355X-native voxel brickmap ray tracer (libvxrt.so).
 *
 * Drop-in boundary for the VoxelRT hot path of JoshuaLim007/VoxelEngine.  The
 * reference exposes this path as C++ (namespace GPUDDA); each entry point below
 * names the reference interface it replaces (paths relative to the reference
 * checkout).  A C++ facade with the reference's own names and signatures sits on
 * top of this ABI in include/GPUDDA/ (see INTEGRATION.md).
 *
 * Conventions: every call returns 0 on success or a negative vxrt_status; no
 * exceptions, no exit().  Pointers named d_* are device (HIP) pointers, all
 * others are host pointers.  `stream` is a hipStream_t passed as void* (NULL =
 * HIP's null stream, as in the HIP API).  A context belongs to one device; calls on one
 * context must be serialised by the caller.
 */
#ifndef VXRT_H
#define VXRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VXRT_ABI_VERSION 3
#define VXRT_EMPTY_SLOT 0xFFFFFFFFu
#define VXRT_MAX_STEPS 2048 /* MAX_STEPS, VoxelRT/VolumeRaytracer.cuh:235 */

typedef enum vxrt_status {
    VXRT_OK = 0,
    VXRT_ERR_INVALID = -1,   /* bad argument / unsupported shape */
    VXRT_ERR_HIP = -2,       /* a HIP runtime call failed (see vxrt_last_error) */
    VXRT_ERR_NO_WORLD = -3,  /* render/trace before a world was uploaded or built */
    VXRT_ERR_NOMEM = -4
} vxrt_status;

typedef struct vxrt_ctx vxrt_ctx;

/* ---- lifetime.  Replaces `new GPUDDA::VoxelRaytracer3D(count)` / `delete`
 * (VoxelRT/VolumeRaytracer.cuh:318-334, VoxelApp/main.cu:41,197). */
int vxrt_abi_version(void);
int vxrt_create(int device, vxrt_ctx **out);
int vxrt_destroy(vxrt_ctx *ctx);
/* message of the last failing call on this thread (never NULL) */
const char *vxrt_last_error(void);
int vxrt_synchronize(vxrt_ctx *ctx);
/* kernel implementation used by vxrt_render / vxrt_render_views and vxrt_trace_batch.  Both give identical results.
 *   7 = the product kernels, all on the wave-level tracer of csrc/vxrt_wave2.hpp: k_render_persist2 (persistent
 *       wavefronts, one pixel chain per lane, pixels from a tile queue, the state only the parked phases touch in LDS:
 *       96 VGPRs, 5 waves per SIMD) for every render launch, whatever its size and the world's; for batches
 *       k_trace_batch_persist (a persistent ray queue) from 8 rays per lane of the persistent grid upwards and
 *       k_trace_batch_wave2 (one ray per lane) below that;
 *   4 (default) = 7;
 *   1 = straightforward per-lane loops (k_render, k_trace_batch): the on-device cross-check.
 * Any other value is refused (rounds 1-3 carried further kernels under 0, 2, 3, 5 and 6; profiles/ keeps their
 * measurements). */
int vxrt_set_kernel_variant(vxrt_ctx *ctx, int variant);
/* Test hook of the load guard: on != 0 makes the guard treat the bit tables as if they had been allocated WITHOUT slack
 * (the real allocations are untouched), so that every slack load counts as stray -- the negative control of the test that
 * holds guard_stray_loads at zero. */
int vxrt_debug_guard_pretend_no_slack(vxrt_ctx *ctx, int on);
/* 1 when the library was built with -DVXRT_EXPERIMENTS (development knobs read from the environment; A/B builds) */
int vxrt_has_experiments(void);
/* Size of the persistent kernels' grid, in wavefronts per compute unit at 4 waves per SIMD (default 16 = 4 per SIMD; the
 * kernels, built for 5 waves per SIMD, scale it by 5/4).  For tests that need a small grid (a batch then takes the queue kernel at a few
 * thousand rays) and for occupancy measurements; 0 restores the default. */
int vxrt_set_persistent_waves_per_cu(vxrt_ctx *ctx, int waves_per_cu);

/* ---- world upload.  Replaces VoxelRaytracer3D::UploadVoxelBuffer,
 * ::UploadVoxelBufferDatas, ::UploadVoxelBufferDataBounds and ::SetFactor
 * (VoxelRT/VolumeRaytracer.cu:527-572, VolumeRaytracer.cuh:349).  The three
 * reference tables are handed over as flat host arrays in the reference's own order and copied into HBM (where the
 * library keeps them in an order of its own, x-fastest linear: vxrt_download_world and the brickmap file give the
 * reference's order back):
 *   coarse_bits : one bit per brick cell, tiled-linear order of GetSampleIndex
 *                 (VolumeRaytracer.cuh:107-131), (ncells+31)/32 words
 *   brick_slot  : per cell, index of the brick in `pool` or VXRT_EMPTY_SLOT
 *                 (reference: a VoxelBuffer3D descriptor with its own allocation)
 *   bounds      : per cell 6 floats {min xyz, max xyz}, brick-local inclusive voxel
 *                 extents, empty = {0,0,0,-1,-1,-1} (layout of Bounds3Df)
 *   pool        : nslots bricks of factor^3 bits each, tiled-linear inside the brick */
typedef struct vxrt_world_desc {
    uint32_t struct_size;
    int32_t factor;       /* brick edge: 8, 16 or 32 */
    int32_t cdims[3];     /* coarse cells per axis, each a multiple of 8 */
    uint64_t nslots;
    const uint32_t *coarse_bits;
    const uint32_t *brick_slot;
    const float *bounds;
    const uint32_t *pool;
} vxrt_world_desc;
int vxrt_upload_world(vxrt_ctx *ctx, const vxrt_world_desc *desc);

/* ---- on-device world construction.  Replaces CreateVoxels + PopulateVoxels
 * (VoxelRT/VoxelWorldBuilder.cuh:12-32, .cu:10-35) followed by
 * GenerateLowresVoxelBuffer (VoxelRT/VolumeRaytracer.cuh:379-516), without the
 * dense intermediate: one workgroup per brick evaluates the generator, packs the
 * brick, reduces its extents and sets the coarse bit. */
typedef enum vxrt_generator {
    VXRT_GEN_HASH_HEIGHTFIELD = 0, /* integer-only columns */
    VXRT_GEN_PERLIN_REF = 1,       /* PopulateVoxels' 32-octave Perlin fBm terrain */
    VXRT_GEN_INT_TERRAIN = 2       /* integer-only smooth terrain */
} vxrt_generator;
int vxrt_build_world_procedural(vxrt_ctx *ctx, int generator, int X, int Y, int Z, int factor);

typedef struct vxrt_world_info {
    int32_t factor;
    int32_t cdims[3];
    uint64_t ncells;
    uint64_t nslots;
    uint64_t hbm_bytes; /* bytes resident for the world tables */
} vxrt_world_info;
int vxrt_world_info_get(vxrt_ctx *ctx, vxrt_world_info *out);
/* copy the resident world back in vxrt_world_desc layout; the caller provides host
 * arrays sized from vxrt_world_info_get (pool may be NULL to skip it). */
int vxrt_download_world(vxrt_ctx *ctx, uint32_t *coarse_bits, uint32_t *brick_slot, float *bounds,
                        uint32_t *pool);

/* ---- brickmap file.  The reference rebuilds its world at every start (VoxelApp/main.cu:41-49:
 * CreateVoxels + GenerateLowresVoxelBuffer, minutes at 8k scale on its host threads); a built brickmap can be
 * kept instead.  Layout (little endian): 120-byte header {"VXBRKMAP", u32 version = 2, u32 header bytes,
 * i32 factor, i32 cdims[3], u64 ncells, u64 nslots, u64 bytes of the three streams, u64 word sums of the three
 * streams, u64 sums of the running word sums of the three streams (position-sensitive)}, then the streams as they lie in HBM: coarse_bits (as in vxrt_world_desc), one 8-byte
 * record per cell {u32 pool slot or VXRT_EMPTY_SLOT, u32 extents: min x,y,z then max x,y,z, 5 bits each from bit 0}
 * in the order of coarse_bits, and the pool.  Loading validates sizes, sums and the cell table against the coarse
 * bits, and streams through a 64 MiB staging buffer.  Version 1 files (104-byte header, plain word sums only; written
 * by the first round's builds) are refused: regenerate them with vxrt_save_world.  Shapes: every coarse dimension a
 * positive multiple of 8, at most 65535, with cy * cz < 2^24 and cx * cz * (cy + 2) < 2^32 (32-bit cell indices). */
int vxrt_save_world(vxrt_ctx *ctx, const char *path);
int vxrt_load_world(vxrt_ctx *ctx, const char *path);
/* header of a brickmap file (no GPU needed); hbm_bytes = bytes the three streams will occupy */
int vxrt_world_file_info(const char *path, vxrt_world_info *out);

/* ---- chunk streaming -- an EXTENSION: the reference lists "Chunking" and "Chunk data streaming" as to do
 * (README.md:15,20).  A world whose bricks need not all be resident: the coarse tables of the whole world stay in HBM
 * (128 KiB + 8 MiB for 8192x512x8192), the brick pool is a cache of `pool_capacity_bricks` bricks, and brick data is
 * read from a brickmap file (vxrt_save_world) for the CHUNKS near a focus point only.  A chunk is one 8x8x8 tile of
 * coarse cells -- 512 consecutive records of the file's tiled-linear tables and one contiguous run of bricks, so a
 * chunk arrives with one read, two copies and two small re-ordering launches.  A chunk that is not resident reads as EMPTY space: the kernels do not
 * change, and every frame equals the frame of the world with exactly the resident chunks' bricks (tests hold the HIP
 * frames equal to the oracle on that truncated world).
 *   vxrt_stream_open: replaces the context's world by the (so far empty) streamed world of `path`.
 *   vxrt_stream_focus: makes every chunk whose box lies within `radius` voxels of `focus` resident, nearest first;
 *     when the pool is full, resident chunks OUTSIDE the radius are evicted, farthest first; chunks inside the radius
 *     that still do not fit stay absent (stats.chunks_missing).  Synchronises the device before it touches the tables.
 *   vxrt_stream_resident: one byte per chunk (chunk = tile index of the coarse grid), 1 = resident.
 * The coarse tables are validated at open exactly as vxrt_load_world validates them (sizes, position-sensitive sums, slots
 * in cell order, brick extents); brick data read per chunk is NOT checksummed (the file's pool sum covers the whole
 * stream, not its chunks).  vxrt_download_world / vxrt_save_world on a streamed world give the CACHE as it stands: the
 * resident chunks' cells with the cache's own slot numbers, every other cell empty, the pool zero where nothing has been
 * loaded (it is cleared at open) and stale, unreferenced bricks where chunks have been evicted. */
typedef struct vxrt_stream_stats {
    uint64_t chunks_total, chunks_occupied;   /* tiles of the coarse grid; those holding at least one brick */
    uint64_t chunks_resident, bricks_resident;
    uint64_t chunks_loaded, chunks_evicted, chunks_missing;  /* by this call */
    uint64_t bytes_read;                                      /* from the file, by this call */
} vxrt_stream_stats;
int vxrt_stream_open(vxrt_ctx *ctx, const char *path, uint64_t pool_capacity_bricks);
int vxrt_stream_focus(vxrt_ctx *ctx, const float focus[3], float radius, vxrt_stream_stats *stats_or_null);
int vxrt_stream_resident(vxrt_ctx *ctx, uint8_t *flags, uint64_t n_chunks);
int vxrt_stream_close(vxrt_ctx *ctx);

/* ---- camera / lighting state.  Replaces Graphics::SetEnvironment, ::SetFOV,
 * ::SetOrthoWindowSize, ::GetDirections (VoxelRT/Renderer.cu:27-42,278-303). */
int vxrt_set_environment(vxrt_ctx *ctx, const float light_dir[3], const float light_color[3],
                         const float ambient[3]);
int vxrt_set_fov(vxrt_ctx *ctx, float fov_degrees);
int vxrt_set_ortho_window_size(vxrt_ctx *ctx, float size_x, float size_y);
void vxrt_get_directions(const float euler[3], float fwd[3], float up[3], float right[3]);

/* ---- per-frame render.  Replaces Graphics::RenderScreen + kernel screenDispatch
 * (VoxelRT/Renderer.cu:179-328).  The compile-time switches of the reference are
 * run-time flags here. */
typedef enum vxrt_mode { VXRT_MODE_SHADED = 0, VXRT_MODE_DEBUG = 1 } vxrt_mode;

typedef struct vxrt_frame_stats {
    uint64_t primary_rays, shadow_rays, bounce_rays, primary_hits;
    uint64_t coarse_probes; /* Nc: in-range coarse cell probes */
    uint64_t brick_entries; /* Nb */
    uint64_t fine_probes;   /* Nf: in-range brick cell probes */
    uint64_t dbg[12];       /* wave-loop diagnostics of collect_stats launches, summed over waves: [0] iterations,
                               [1] walking lanes over those iterations, [2] end-of-walk / [3] tight-box / [4] ray-finished
                               phase executions, [5..7] lanes those executions served (same order); persistent kernel
                               only: [8] wave lifetime in 100 MHz ticks, [9] iterations after the tile queue ran dry, [10] ticks inside
                               the ray-finished phase, [11] ticks inside the box and end-of-walk phases */
    /* Load guard of collect_stats launches (product kernels).  The tracer lets a lane that has just stepped out of a grid
     * issue one more occupancy load before it stops; the library allocates slack around both bit tables for it.  Counted
     * per load: beyond a table but inside its allocation (expected, > 0 on ordinary frames), and outside everything
     * addressable (must be 0: a world path that forgot the slack would show up here, not as a fault in a user's frame). */
    uint64_t guard_slack_loads, guard_stray_loads;
} vxrt_frame_stats;

typedef struct vxrt_render_flags {
    uint32_t struct_size;
    int32_t mode;            /* vxrt_mode; DEBUG = `#define DEBUG_VIEW` (Renderer.cu:4) */
    int32_t checkerboard;    /* ENABLE_CHECKERBOARD_RENDER (Renderer.cu:5) */
    int32_t shadow;          /* 1 = shadow ray of Renderer.cu:97-102 enabled */
    int32_t bounce_samples;  /* `samples` of Renderer.cu:123 */
    int32_t bounce_all_hits; /* 0 = reference gate `lDot == 0` (Renderer.cu:121); 1 = every hit pixel */
    int32_t bounce_depth;    /* <= 1 (default): the reference's one occlusion ray per sample.  2: EXTENSION beyond the
                                reference (BASELINE config 5): a sample ray that hits spawns one more 8-step ray from
                                its hit point, built like the first (outward normal there, seed + 500); a miss of
                                that ray adds 0.5 to the sample sum */
    int32_t ortho;           /* `#define ORTHO` (Renderer.cuh:13) */
    int64_t frame_number;    /* >= 0: value the kernel sees as FrameNumber; < 0: the context's own
                                counter with the reference's post-copy increment (Renderer.cu:310,322) */
    /* multi-GPU strip sharding: rows are cut into strips of `strip_rows`; strip s belongs to
     * shard s % strip_count.  strip_count <= 1 renders the whole frame. */
    int32_t strip_rows, strip_count, strip_index;
    int32_t compact;         /* 1: d_fb (and AOVs) hold only this shard's strips, packed in order */
    int32_t collect_stats;   /* 1: also count probes (the STATS instantiation of the same kernel); rays are always counted */
    int32_t tile_schedule;   /* persistent kernel: 1 (default) = hand out the rows of 8x8 pixel tiles expected-longest
                                first (ranked per frame on the host by the elevation of the row's centre ray in a
                                Y-up world); 0 = row-major.  Scheduling only: results do not depend on it */
    float *d_color_aov;      /* optional W*H*3 float colour handed to the pixel store, or NULL */
    int64_t *d_hit_aov;      /* optional W*H primary hit voxel index (x + X*(y + Y*z)) or -1, or NULL */
    const uint32_t *d_tile_order; /* optional caller-made hand-out order, overrides tile_schedule: a permutation of
                                0 .. ceil(W/8)*ceil(rows/8)-1 (tile = tx + ty*ceil(W/8), rows = the launch grid's), or NULL */
    void *stream;
    /* Temporal accumulation of the stochastic occlusion term -- an EXTENSION: the reference lists "denoise, temporal
     * accumulation" as to do (README.md:19).  d_accum: W*H*4 floats per pixel {sum r, g, b of the pre-tonemap colour, frames
     * in the history} (compact shards: their own rows), or NULL = off.  For every shaded hit pixel the colour of
     * calculateColor (Renderer.cu:90-168) is added to the history and the MEAN is tonemapped and stored; the first frame of
     * a history (accum_reset != 0, or frames == 0) stores the colour itself.  Miss pixels, the debug view and the overlays
     * are written as without it.  With a static camera and one frame number per call the bounce noise averages out as 1/n;
     * the caller resets the history when the camera moves.  vxrt_render only (a multi-view launch has no per-view
     * history). */
    float *d_accum;
    int32_t accum_reset;
    int32_t reserved_;
} vxrt_render_flags;

void vxrt_render_flags_default(vxrt_render_flags *flags);
/* d_fb: device BGRA8 framebuffer (bytes b,g,r,a = SDLRenderer.h:8-11 PixelData), W*H*4 bytes
 * (or the compact size, see vxrt_compact_rows).  Asynchronous on the stream. */
int vxrt_render(vxrt_ctx *ctx, uint32_t width, uint32_t height, void *d_fb, const float origin[3],
                const float fwd[3], const float up[3], const float right[3], const vxrt_render_flags *flags);
/* Several views of the resident world in ONE launch (this build's addition; the reference renders one view per
 * RenderScreen call).  Why: a frame ends with a stretch where only the longest ray chains are still running and most
 * of the GPU idles -- at 1080p about a quarter of the launch.  With n views in one launch the persistent kernel's
 * queue runs on into the next view's tiles, so only the last view pays that stretch (measured: 1.37x the rays/s of
 * one-view launches at 1080p).  Every view is exactly the frame vxrt_render would produce for the same camera,
 * frame number and flags.  `flags` applies to all views; its frame_number, d_color_aov, d_hit_aov and d_tile_order
 * are ignored (per-view members below).  1 <= n_views <= 16.  Launches issued on different streams may also be in
 * flight together: up to 64 launches (16 of them multi-view) per context share no state; the call that would exceed that
 * waits on the host for the oldest launch to finish before it reuses its queue head / view slot.  (Inside a stream capture
 * nothing can be waited for: the capturing caller keeps within those limits.)  Host calls on a context stay serialised. */
typedef struct vxrt_view {
    void *d_fb;             /* W*H*4 bytes BGRA8 (or the compact size) */
    float origin[3], fwd[3], up[3], right[3];
    int64_t frame_number;   /* as vxrt_render_flags.frame_number */
    float *d_color_aov;     /* optional, as in vxrt_render_flags */
    int64_t *d_hit_aov;
} vxrt_view;
int vxrt_render_views(vxrt_ctx *ctx, uint32_t width, uint32_t height, uint32_t n_views, const vxrt_view *views,
                      const vxrt_render_flags *flags);
/* the kernel (7 or 1) a vxrt_render (nviews = 0) or vxrt_render_views launch of this shape would run under the
 * context's current variant; -1 on bad arguments.  For tools that label measurements by kernel (bench.py). */
int vxrt_kernel_for_launch(const vxrt_ctx *ctx, uint32_t width, uint32_t height, const vxrt_render_flags *flags, uint32_t nviews);
/* number of frame rows owned by a shard, = rows of its compact buffer */
uint32_t vxrt_compact_rows(uint32_t height, int32_t strip_rows, int32_t strip_count, int32_t strip_index);
/* counters accumulated by the launches on this context since the previous read, whatever their streams; synchronises
 * the device (every stream).  The device-side counters only grow: "since the previous read" is a host-side snapshot, so a
 * read never clears memory that a running kernel adds to. */
int vxrt_frame_stats_get(vxrt_ctx *ctx, vxrt_frame_stats *out);
/* scatter `strip_count` compact shard buffers (laid out back to back, shard-major, each padded to
 * `shard_stride_bytes`) into a full W*H BGRA8 frame on the device; used by the root after the gather. */
int vxrt_deinterleave_strips(vxrt_ctx *ctx, uint32_t width, uint32_t height, int32_t strip_rows,
                             int32_t strip_count, const void *d_shards, uint64_t shard_stride_bytes,
                             void *d_fb, void *stream);
/* the same for the `n_views` views of one multi-view step in ONE launch: view j's packed rows start
 * `j * view_stride_bytes` into every shard's contribution, its frame `j * fb_stride_bytes` into d_fb */
int vxrt_deinterleave_views(vxrt_ctx *ctx, uint32_t width, uint32_t height, int32_t strip_rows, int32_t strip_count,
                            const void *d_shards, uint64_t shard_stride_bytes, uint64_t view_stride_bytes,
                            uint32_t n_views, void *d_fb, uint64_t fb_stride_bytes, void *stream);

/* ---- batch query.  Replaces VoxelRaytracer3D::Raytrace + kernel dispatch
 * (VoxelRT/VolumeRaytracer.cu:95-117,574-618).  Results follow the reference
 * convention: miss -> point = +inf; normal (step direction, zero on a miss) and
 * steps always written.  d_hit / d_voxel (optional) are this build's additions:
 * hit flag and global hit voxel index x + X*(y + Y*z), -1 on a miss. */
/* RAY VALIDITY (defined by this build; the reference leaves it undefined): a ray is valid when its origin's components
 * are finite (their absolute sum is a finite binary32) and the squared length of its direction, evaluated in binary32, is
 * positive and finite -- i.e. normalize(direction) is a vector of finite numbers.  NaN or infinite components, the zero
 * direction, and directions so short or long that the squared length leaves the binary32 range make Raytrace's prologue
 * (VolumeRaytracer.cu:359-367) produce NaN, which the reference then casts to int (undefined).  An INVALID ray of a batch
 * is not traced: its result is a miss with 0 steps (point = +inf, normal = 0, d_hit = 0, d_voxel = -1); it is counted as
 * a ray.  vxrt_render / vxrt_render_views return VXRT_ERR_INVALID for a camera with a non-finite component. */
int vxrt_trace_batch(vxrt_ctx *ctx, const float *d_origins, const float *d_dirs, uint64_t n, float *d_pos,
                     float *d_normal, int32_t *d_steps, uint8_t *d_hit, int64_t *d_voxel,
                     vxrt_frame_stats *stats_or_null, void *stream);
/* The device function's first argument, `Raytrace(int maxSteps, ...)` (VolumeRaytracer.cu:354): the budget that is tested
 * at the head of the two-level loop only (:386).  The batch kernel of the reference passes MAX_STEPS = 2048 (:105), the
 * secondary rays of calculateColor pass 8 (Renderer.cu:141).  Applies to the following vxrt_trace_batch* calls on this
 * context; default 2048.  1 <= max_steps <= 2048. */
int vxrt_set_batch_max_steps(vxrt_ctx *ctx, int32_t max_steps);
/* host-pointer convenience with the reference's copy-in / copy-out behaviour */
int vxrt_trace_batch_host(vxrt_ctx *ctx, const float *origins, const float *dirs, uint64_t n, float *pos,
                          float *normal, int32_t *steps, uint8_t *hit, int64_t *voxel,
                          vxrt_frame_stats *stats_or_null);

#ifdef __cplusplus
}
#endif
#endif /* VXRT_H */
