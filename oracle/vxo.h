/*
 * vxo.h -- CPU ORACLE for the voxel brickmap ray-tracing hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may load it; the shipped
 * library (voxelengine_amd/csrc) never links, imports or calls anything here.
 *
 * PARITY UNPINNED: the reference (JoshuaLim007/VoxelEngine) holds no tests,
 * golden vectors or fixtures for this path, and its sources are CUDA (.cu,
 * need nvcc + libcudart) so they cannot be built in this image without
 * writing stand-ins for the CUDA toolchain.  This file set is therefore a
 * hand restatement of the reference algorithm in plain C, executed with IEEE
 * float semantics (-ffp-contract=off, no fast-math), checked only by
 * hand-derived known-answer tests (tests/test_oracle_*.py).
 *
 * Every function cites the reference file:line it restates (paths relative
 * to the reference checkout).
 */
#ifndef VXO_H
#define VXO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VXO_MAX_STEPS 2048          /* VoxelRT/VolumeRaytracer.cuh:235 */
#define VXO_EMPTY_SLOT 0xFFFFFFFFu

/* ---- layout (VoxelRT/VolumeRaytracer.cuh:107-171, VolumeRaytracer.cu:61-68) */
uint32_t vxo_sample_index(uint32_t x, uint32_t y, uint32_t z, uint32_t width, uint32_t height);
void vxo_position_from_index(uint32_t index, uint32_t width, uint32_t height,
                             uint32_t *x, uint32_t *y, uint32_t *z);
/* 64-bit variant used by the builders for worlds past 2^32 bits (the reference
 * wraps there, VolumeRaytracer.cuh:107-131); identical below 2^31. */
uint64_t vxo_sample_index64(uint64_t x, uint64_t y, uint64_t z, uint64_t width, uint64_t height);
int vxo_bit_get(const uint32_t *words, uint64_t nbits, uint64_t index);
void vxo_bit_set(uint32_t *words, uint64_t index, int value);

/* ---- world: the reference's three tables (VolumeRaytracer.cuh:227-233,303-315)
 * coarse grid  = one bit per brick cell, tiled-linear order
 * bricks       = per coarse cell a descriptor {bits ptr, dims}; here a slot
 *                number into one pool (VXO_EMPTY_SLOT <=> dims 0, no bits)
 * bounds       = per coarse cell {min xyz, max xyz} floats, brick-local,
 *                inclusive, empty = min 0 / max -1 (VolumeRaytracer.cuh:454-467) */
typedef struct vxo_world {
    int factor;               /* brick edge f */
    int cdims[3];             /* coarse cells per axis (cols, rows, slices) */
    uint64_t ncells;
    uint32_t *coarse_bits;    /* (ncells+31)/32 words */
    uint32_t *brick_slot;     /* ncells */
    float *bounds;            /* ncells * 6 */
    uint64_t nslots;
    uint32_t *pool;           /* nslots * f^3/32 words */
    int owns;                 /* 1: arrays malloc'ed by the oracle */
} vxo_world;

void vxo_world_free(vxo_world *w);
/* wrap caller-owned arrays (no copy) */
vxo_world *vxo_world_wrap(int factor, const int cdims[3], uint32_t *coarse_bits, uint32_t *brick_slot,
                          float *bounds, uint64_t nslots, uint32_t *pool);

/* GenerateLowresVoxelBuffer (VolumeRaytracer.cuh:379-516): dense tiled-linear
 * bits of an X*Y*Z world -> brickmap.  Non-empty bricks get pool slots in
 * coarse tiled-index order. */
vxo_world *vxo_build_brickmap(const uint32_t *dense_bits, int X, int Y, int Z, int factor);

/* ---- procedural dense worlds (tiled-linear bits, caller frees with free()) */
enum { VXO_GEN_HASH_HEIGHTFIELD = 0, VXO_GEN_PERLIN_REF = 1, VXO_GEN_INT_TERRAIN = 2 };
/* solid(x,y,z) for generator g in a world of X*Y*Z */
int vxo_gen_solid(int g, int x, int y, int z, int X, int Y, int Z);
uint32_t *vxo_gen_dense(int g, int X, int Y, int Z, int nthreads);
/* brickmap built brick by brick without the dense intermediate (same result
 * as vxo_build_brickmap(vxo_gen_dense(...))) */
vxo_world *vxo_gen_brickmap(int g, int X, int Y, int Z, int factor, int nthreads);
/* cuda_noise.cuh:44-54 / :66-71 */
uint32_t vxo_hash32(uint32_t seed);
float vxo_random_float(uint32_t seed);
/* cuda_noise.cuh:565-628 with the a12 parameters (VoxelWorldBuilder.cu:4-8) */
float vxo_fbm_perlin(float x, float y, float z);

/* ---- traversal */
/* RayIntersectsAABB (VolumeRaytracer.cu:124-174); out_p/out_n may be NULL */
int vxo_ray_aabb(const float start[3], const float dir[3], const float bmin[3], const float bmax[3],
                 float out_p[3], float out_n[3]);

typedef struct vxo_dda_params {     /* DDARayParams, VolumeRaytracer.cuh:237-264 */
    const uint32_t *bits;           /* VoxelBuffer.grid */
    uint64_t nbits;
    int dims[3];                    /* VoxelBuffer.dimensions */
    float start[3];
    float dir[3];
    int has_bounds;                 /* bounds != nullptr */
    float bounds_min[3], bounds_max[3];
    int max_steps;
    const float *cell_bounds;       /* per_voxel_bounds (6 floats per cell) or NULL */
    int cell_bounds_scale;          /* per_voxel_bounds_scale */
    int take_initial_step;
} vxo_dda_params;

typedef struct vxo_dda_result {     /* DDARayResults, VolumeRaytracer.cuh:266-275 */
    int hit;
    int out_of_bounds;
    float hit_cell[3];
    float point[3];                 /* HitIntersectedPoint */
    float next_cell[3];
    float normal[3];                /* HitNormal */
    int steps;                      /* stepsTaken */
    int probes;                     /* in-range cell probes (oracle bookkeeping) */
} vxo_dda_result;

/* DDARayTraversal (VolumeRaytracer.cu:176-352) */
void vxo_dda(const vxo_dda_params *p, vxo_dda_result *r);

typedef struct vxo_ray_stats {      /* bookkeeping for SURVEY 8(d) byte accounting */
    uint64_t coarse_probes;         /* Nc */
    uint64_t brick_entries;         /* Nb */
    uint64_t fine_probes;           /* Nf */
} vxo_ray_stats;

/* Raytrace (VolumeRaytracer.cu:354-525). out_pos is written only on a hit.
 * hit_voxel (may be NULL): global voxel coords of the solid voxel that ended
 * the ray (coarse HitCell*f + brick HitCell), defined by this build. */
int vxo_raytrace(const vxo_world *w, int max_steps, const float origin[3], const float ray[3],
                 int *out_steps, float out_normal[3], float out_pos[3], int hit_voxel[3],
                 vxo_ray_stats *stats);

/* dispatch + VoxelRaytracer3D::Raytrace result convention (VolumeRaytracer.cu:95-117):
 * miss -> point = +inf; normal and steps always written.  hit_voxel_index =
 * gx + X*(gy + Y*gz) as int64, -1 on miss. */
void vxo_trace_batch(const vxo_world *w, const float *origins, const float *dirs, size_t n,
                     float *out_pos, float *out_normal, int32_t *out_steps, uint8_t *out_hit,
                     int64_t *out_voxel, vxo_ray_stats *stats_sum, int nthreads);

/* ---- renderer */
typedef struct vxo_env {            /* Graphics::Environment, Renderer.cuh:33-37 */
    float light_dir[3];
    float light_color[3];
    float ambient[3];
} vxo_env;

enum { VXO_MODE_SHADED = 0, VXO_MODE_DEBUG = 1 };

typedef struct vxo_render_params {
    uint32_t width, height;         /* full frame resolution */
    uint32_t frame_number;          /* value the kernel sees in dFrameInfo.FrameNumber */
    float fov_deg;                  /* SetFOV */
    float ortho_size[2];            /* SetOrthoWindowSize */
    int ortho;                      /* #define ORTHO (Renderer.cuh:13) */
    int mode;                       /* VXO_MODE_* (#define DEBUG_VIEW, Renderer.cu:4) */
    int checkerboard;               /* ENABLE_CHECKERBOARD_RENDER (Renderer.cu:5) */
    int shadow;                     /* 1: shadow ray enabled (Renderer.cu:102 un-commented) */
    int bounce_samples;             /* `samples` at Renderer.cu:123 */
    int bounce_all_hits;            /* 0: reference gate lDot==0 (Renderer.cu:121); 1: every hit pixel */
    int bounce_depth;               /* <= 1: the reference's single occlusion ray per sample.  2: EXTENSION beyond the
                                       reference (BASELINE config 5): a sample ray that hits spawns one more 8-step ray
                                       from its hit point (outward normal, seed + 500); a miss of that ray adds 0.5 */
    float origin[3], fwd[3], up[3], right[3];
    vxo_env env;
    /* rows [row_begin,row_end) of the frame are rendered (multi-GPU strips); 0,height = all */
    uint32_t row_begin, row_end;
} vxo_render_params;

typedef struct vxo_frame_stats {
    uint64_t primary_rays, shadow_rays, bounce_rays;
    uint64_t primary_hits;
    vxo_ray_stats probes;           /* summed over every ray */
    uint64_t pixels_written;
} vxo_frame_stats;

/* GetDirections (Renderer.cu:27-42) */
void vxo_get_directions(const float euler[3], float fwd[3], float up[3], float right[3]);

/* screenDispatch over the launch grid RenderScreen would use (Renderer.cu:179-328).
 * fb = width*height BGRA8 (bytes b,g,r,a); pixels the launch does not cover keep
 * their contents.  color_aov (optional, width*height*3 floats) receives the float
 * colour handed to setPixelColor before clamping, for pixels written.
 * hit_aov (optional, width*height int64) receives the primary hit voxel index or -1. */
void vxo_render(const vxo_world *w, const vxo_render_params *p, uint8_t *fb, float *color_aov,
                int64_t *hit_aov, vxo_frame_stats *stats, int nthreads);
/* the same with temporal accumulation (include/vxrt.h, vxrt_render_flags.d_accum): `accum` = W*H*4 floats {sum of the
 * pre-tonemap colour, frames in the history}; shaded hit pixels only */
void vxo_render_accum(const vxo_world *w, const vxo_render_params *p, uint8_t *fb, float *color_aov,
                      int64_t *hit_aov, float *accum, int accum_reset, vxo_frame_stats *stats, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
