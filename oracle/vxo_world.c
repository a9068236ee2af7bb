/*
 * vxo_world.c -- ORACLE (test infrastructure; parity unpinned, see vxo.h).
 * Brickmap construction (VoxelRT/VolumeRaytracer.cuh:379-516) and the
 * procedural worlds used by tests and bench (VoxelRT/VoxelWorldBuilder.cu:4-35,
 * VoxelRT/cuda_noise.cuh:44-71,162-202,580-628).
 */
#include "vxo.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------ world object */

void vxo_world_free(vxo_world *w)
{
    if (!w)
        return;
    if (w->owns) {
        free(w->coarse_bits);
        free(w->brick_slot);
        free(w->bounds);
        free(w->pool);
    }
    free(w);
}

vxo_world *vxo_world_wrap(int factor, const int cdims[3], uint32_t *coarse_bits, uint32_t *brick_slot,
                          float *bounds, uint64_t nslots, uint32_t *pool)
{
    vxo_world *w = (vxo_world *)calloc(1, sizeof(*w));
    w->factor = factor;
    memcpy(w->cdims, cdims, 3 * sizeof(int));
    w->ncells = (uint64_t)cdims[0] * cdims[1] * cdims[2];
    w->coarse_bits = coarse_bits;
    w->brick_slot = brick_slot;
    w->bounds = bounds;
    w->nslots = nslots;
    w->pool = pool;
    w->owns = 0;
    return w;
}

/* ------------------------------------------------------------- noise pieces */

/* cudaNoise::hash (cuda_noise.cuh:44-54) */
uint32_t vxo_hash32(uint32_t seed)
{
    seed = (seed + 0x7ed55d16u) + (seed << 12);
    seed = (seed ^ 0xc761c23cu) ^ (seed >> 19);
    seed = (seed + 0x165667b1u) + (seed << 5);
    seed = (seed + 0xd3a2646cu) ^ (seed << 9);
    seed = (seed + 0xfd7046c5u) + (seed << 3);
    seed = (seed ^ 0xb55a4f09u) ^ (seed >> 16);
    return seed;
}

/* cudaNoise::randomFloat (cuda_noise.cuh:66-71): hash / (float)0xffffffff */
float vxo_random_float(uint32_t seed)
{
    return (float)vxo_hash32(seed) / 4294967296.0f; /* (float)0xffffffff rounds to 2^32 */
}

/* float -> unsigned conversion as AMD/NVIDIA GPUs do it (saturating, NaN -> 0).
 * The reference's casts at cuda_noise.cuh:114,120 overflow for high octaves,
 * which is undefined on the host; this build fixes the GPU behaviour as the
 * definition (world generation parity is unpinned). */
static inline uint32_t sat_u32(float v)
{
    if (!(v > 0.0f))
        return 0u;
    if (v >= 4294967296.0f)
        return 0xFFFFFFFFu;
    return (uint32_t)v;
}

/* randomIntGrid (cuda_noise.cuh:118-121) */
static inline uint32_t lattice_hash(float x, float y, float z, float seed)
{
    return vxo_hash32(sat_u32(x * 1723.0f + y * 93241.0f + z * 149812.0f + 3824.0f + seed));
}

/* grad (cuda_noise.cuh:174-196) */
static inline float corner_grad(uint32_t h, float x, float y, float z)
{
    switch (h & 0xFu) {
    case 0x0: return x + y;
    case 0x1: return -x + y;
    case 0x2: return x - y;
    case 0x3: return -x - y;
    case 0x4: return x + z;
    case 0x5: return -x + z;
    case 0x6: return x - z;
    case 0x7: return -x - z;
    case 0x8: return y + z;
    case 0x9: return -y + z;
    case 0xA: return y - z;
    case 0xB: return -y - z;
    case 0xC: return y + x;
    case 0xD: return -y + z;
    case 0xE: return y - x;
    default: return -y - z;
    }
}

static inline float quintic(float t) { return t * t * t * (t * (t * 6.0f - 15.0f) + 10.0f); } /* fade :199-202 */
static inline float mix(float a, float b, float r) { return a * (1.0f - r) + b * r; }         /* lerp :162-165 */

/* perlinNoise (cuda_noise.cuh:565-607) with scale 1 */
static float perlin3(float px, float py, float pz, int seed)
{
    float fseed = (float)seed;
    float ix = floorf(px), iy = floorf(py), iz = floorf(pz);
    px -= ix;
    py -= iy;
    pz -= iz;
    float u = quintic(px), v = quintic(py), w = quintic(pz);
    float g000 = corner_grad(lattice_hash(ix, iy, iz, fseed), px, py, pz);
    float g100 = corner_grad(lattice_hash(ix + 1.0f, iy, iz, fseed), px - 1.0f, py, pz);
    float g010 = corner_grad(lattice_hash(ix, iy + 1.0f, iz, fseed), px, py - 1.0f, pz);
    float g110 = corner_grad(lattice_hash(ix + 1.0f, iy + 1.0f, iz, fseed), px - 1.0f, py - 1.0f, pz);
    float g001 = corner_grad(lattice_hash(ix, iy, iz + 1.0f, fseed), px, py, pz - 1.0f);
    float g101 = corner_grad(lattice_hash(ix + 1.0f, iy, iz + 1.0f, fseed), px - 1.0f, py, pz - 1.0f);
    float g011 = corner_grad(lattice_hash(ix, iy + 1.0f, iz + 1.0f, fseed), px, py - 1.0f, pz - 1.0f);
    float g111 = corner_grad(lattice_hash(ix + 1.0f, iy + 1.0f, iz + 1.0f, fseed), px - 1.0f, py - 1.0f,
                             pz - 1.0f);
    float x00 = mix(g000, g100, u), x10 = mix(g010, g110, u);
    float x01 = mix(g001, g101, u), x11 = mix(g011, g111, u);
    return mix(mix(x00, x10, v), mix(x01, x11, v), w);
}

/* repeaterPerlin(pos, 1.0f, <ignored seed>, 32, 2.0f, 0.5f)
 * (cuda_noise.cuh:612-628 called from VoxelWorldBuilder.cu:4-8); octave seeds
 * (i+38)*27389482 in wrapping 32-bit arithmetic. */
float vxo_fbm_perlin(float x, float y, float z)
{
    float acc = 0.0f, amp = 1.0f, scale = 1.0f;
    for (int i = 0; i < 32; ++i) {
        int seed = (int)((uint32_t)(i + 38) * 27389482u);
        acc += perlin3(x * scale, y * scale, z * scale, seed) * amp;
        scale *= 2.0f;
        amp *= 0.5f;
    }
    return acc;
}

/* ------------------------------------------------------------- generators */

static inline uint32_t hash2(uint32_t a, uint32_t b, uint32_t seed)
{
    return vxo_hash32(a * 73856093u ^ b * 19349663u ^ seed);
}

/* VXO_GEN_HASH_HEIGHTFIELD: integer-only columns on an 8x8
 * footprint, h = 3Y/16 + hash(x>>3, z>>3, 1) % (3Y/8); solid <=> y < h.
 * VXO_GEN_PERLIN_REF: PopulateVoxels (VoxelWorldBuilder.cu:10-35): solid <=>
 * !(y > max(0, 1000*fBm(0.005x, 0.005y, 0.005z))).
 * VXO_GEN_INT_TERRAIN (this build's own; integer-only):
 * four octaves of bilinearly interpolated hashed lattice heights, so terrain has
 * large smooth features and long grazing rays without any floating point. */
int vxo_gen_solid(int g, int x, int y, int z, int X, int Y, int Z)
{
    (void)X;
    (void)Z;
    if (g == VXO_GEN_HASH_HEIGHTFIELD) {
        uint32_t base = (uint32_t)(3 * Y / 16), range = (uint32_t)(3 * Y / 8);
        if (range == 0)
            range = 1;
        uint32_t h = base + hash2((uint32_t)x >> 3, (uint32_t)z >> 3, 1u) % range;
        return (uint32_t)y < h;
    }
    if (g == VXO_GEN_PERLIN_REF) {
        float scale = 0.005f;
        float t = vxo_fbm_perlin((float)x * scale, (float)y * scale, (float)z * scale) * 1000.0f;
        t = t > 0.0f ? t : 0.0f;
        return !((float)y > t);
    }
    /* integer terrain */
    uint64_t h = (uint64_t)Y / 8u;
    for (int o = 0; o < 4; ++o) {
        int k = 8 - o;                         /* lattice pitch 256,128,64,32 voxels */
        uint64_t S = 1ull << k;
        uint64_t amp = ((uint64_t)Y / 2u) >> o;
        if (amp == 0)
            break;
        uint32_t cx = (uint32_t)x >> k, cz = (uint32_t)z >> k;
        uint64_t fx = (uint64_t)x & (S - 1), fz = (uint64_t)z & (S - 1);
        uint64_t v00 = hash2(cx, cz, 7u + (uint32_t)o) % amp;
        uint64_t v10 = hash2(cx + 1u, cz, 7u + (uint32_t)o) % amp;
        uint64_t v01 = hash2(cx, cz + 1u, 7u + (uint32_t)o) % amp;
        uint64_t v11 = hash2(cx + 1u, cz + 1u, 7u + (uint32_t)o) % amp;
        uint64_t top = v00 * (S - fx) + v10 * fx;
        uint64_t bot = v01 * (S - fx) + v11 * fx;
        h += (top * (S - fz) + bot * fz) >> (2 * k);
    }
    return (uint64_t)y < h;
}

typedef struct gen_job {
    int g, X, Y, Z, factor;
    uint32_t *dense;
    /* brick mode */
    vxo_world *w;
    uint32_t *scratch_pool; /* ncells * words_per_brick when building bricks directly */
    uint64_t begin, end;    /* z-slab range (dense) or coarse index range (bricks) */
} gen_job;

static void *gen_dense_worker(void *arg)
{
    gen_job *j = (gen_job *)arg;
    /* z range is a multiple of 8 so no two workers share a word */
    for (uint64_t z = j->begin; z < j->end; ++z)
        for (int y = 0; y < j->Y; ++y)
            for (int x = 0; x < j->X; ++x) {
                uint64_t idx = vxo_sample_index64((uint64_t)x, (uint64_t)y, z, (uint64_t)j->X, (uint64_t)j->Y);
                if (vxo_gen_solid(j->g, x, y, (int)z, j->X, j->Y, j->Z))
                    j->dense[idx >> 5] |= 1u << (idx & 31u);
            }
    return NULL;
}

static void run_jobs(void *(*fn)(void *), gen_job *jobs, int n)
{
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n);
    for (int i = 0; i < n; ++i)
        pthread_create(&th[i], NULL, fn, &jobs[i]);
    for (int i = 0; i < n; ++i)
        pthread_join(th[i], NULL);
    free(th);
}

uint32_t *vxo_gen_dense(int g, int X, int Y, int Z, int nthreads)
{
    uint64_t nbits = (uint64_t)X * Y * Z;
    uint32_t *dense = (uint32_t *)calloc((nbits + 31) / 32, sizeof(uint32_t));
    if (nthreads < 1)
        nthreads = 1;
    int slabs = Z / 8;
    if (nthreads > slabs)
        nthreads = slabs > 0 ? slabs : 1;
    gen_job *jobs = (gen_job *)calloc((size_t)nthreads, sizeof(gen_job));
    for (int i = 0; i < nthreads; ++i) {
        jobs[i].g = g;
        jobs[i].X = X;
        jobs[i].Y = Y;
        jobs[i].Z = Z;
        jobs[i].dense = dense;
        jobs[i].begin = (uint64_t)(slabs * (int64_t)i / nthreads) * 8u;
        jobs[i].end = (uint64_t)(slabs * (int64_t)(i + 1) / nthreads) * 8u;
    }
    run_jobs(gen_dense_worker, jobs, nthreads);
    free(jobs);
    return dense;
}

/* ------------------------------------------------------------ brickmap build */

/* One coarse cell of GenerateLowresVoxelBuffer's HandleThread body
 * (VolumeRaytracer.cuh:417-468): copy f^3 bits (brick-local tiled index <-
 * world tiled index), track inclusive extents, report emptiness.
 * `solid` supplies the source bit: from a dense array or from a generator. */
static int fill_brick(int f, uint32_t bx, uint32_t by, uint32_t bz, const uint32_t *dense, int X, int Y, int Z,
                      int g, uint32_t *brick_words, float out_bounds[6])
{
    int mn[3] = {2147483647, 2147483647, 2147483647};
    int mx[3] = {(-2147483647 - 1), (-2147483647 - 1), (-2147483647 - 1)};
    int any = 0;
    memset(brick_words, 0, (size_t)f * f * f / 8);
    for (int dz = 0; dz < f; ++dz)
        for (int dy = 0; dy < f; ++dy)
            for (int dx = 0; dx < f; ++dx) {
                uint64_t gx = (uint64_t)dx + (uint64_t)f * bx, gy = (uint64_t)dy + (uint64_t)f * by,
                         gz = (uint64_t)dz + (uint64_t)f * bz;
                int bit;
                if (dense) {
                    uint64_t hi_idx = vxo_sample_index64(gx, gy, gz, (uint64_t)X, (uint64_t)Y);
                    bit = (int)((dense[hi_idx >> 5] >> (hi_idx & 31u)) & 1u);
                } else {
                    bit = vxo_gen_solid(g, (int)gx, (int)gy, (int)gz, X, Y, Z);
                }
                if (bit) {
                    uint32_t lo_idx = vxo_sample_index((uint32_t)dx, (uint32_t)dy, (uint32_t)dz, (uint32_t)f,
                                                       (uint32_t)f);
                    brick_words[lo_idx >> 5] |= 1u << (lo_idx & 31u);
                    any = 1;
                    if (dx < mn[0]) mn[0] = dx;
                    if (dy < mn[1]) mn[1] = dy;
                    if (dz < mn[2]) mn[2] = dz;
                    if (dx > mx[0]) mx[0] = dx;
                    if (dy > mx[1]) mx[1] = dy;
                    if (dz > mx[2]) mx[2] = dz;
                }
            }
    if (!any) {
        mn[0] = mn[1] = mn[2] = 0;
        mx[0] = mx[1] = mx[2] = -1;
    }
    for (int a = 0; a < 3; ++a) {
        out_bounds[a] = (float)mn[a];
        out_bounds[3 + a] = (float)mx[a];
    }
    return any;
}

typedef struct brick_job {
    gen_job base;
    const uint32_t *dense;
    uint8_t *any_flags;
} brick_job;

static void *brick_worker(void *arg)
{
    brick_job *bj = (brick_job *)arg;
    gen_job *j = &bj->base;
    vxo_world *w = j->w;
    const int f = j->factor;
    const uint64_t bw = (uint64_t)f * f * f / 32u;
    for (uint64_t t = j->begin; t < j->end; ++t) {
        uint32_t x, y, z;
        vxo_position_from_index((uint32_t)t, (uint32_t)w->cdims[0], (uint32_t)w->cdims[1], &x, &y, &z);
        uint32_t idx = vxo_sample_index(x, y, z, (uint32_t)w->cdims[0], (uint32_t)w->cdims[1]);
        bj->any_flags[idx] = (uint8_t)fill_brick(f, x, y, z, bj->dense, j->X, j->Y, j->Z, j->g,
                                                 j->scratch_pool + (uint64_t)idx * bw, w->bounds + (uint64_t)idx * 6);
    }
    return NULL;
}

static vxo_world *build_common(const uint32_t *dense, int g, int X, int Y, int Z, int factor, int nthreads)
{
    if (factor <= 0 || factor % 8 || X % factor || Y % factor || Z % factor)
        return NULL;
    int cd[3] = {X / factor, Y / factor, Z / factor};
    if (cd[0] % 8 || cd[1] % 8 || cd[2] % 8)
        return NULL; /* tiled-linear needs whole 8^3 tiles of coarse cells (SURVEY 8 preamble) */
    vxo_world *w = (vxo_world *)calloc(1, sizeof(*w));
    w->factor = factor;
    memcpy(w->cdims, cd, sizeof(cd));
    w->ncells = (uint64_t)cd[0] * cd[1] * cd[2];
    w->owns = 1;
    const uint64_t bw = (uint64_t)factor * factor * factor / 32u;
    w->coarse_bits = (uint32_t *)calloc((w->ncells + 31) / 32, sizeof(uint32_t));
    w->brick_slot = (uint32_t *)malloc(w->ncells * sizeof(uint32_t));
    w->bounds = (float *)calloc(w->ncells * 6, sizeof(float));
    uint32_t *scratch = (uint32_t *)malloc(w->ncells * bw * sizeof(uint32_t));
    uint8_t *any_flags = (uint8_t *)calloc(w->ncells, 1);

    if (nthreads < 1)
        nthreads = 1;
    if ((uint64_t)nthreads > w->ncells)
        nthreads = (int)w->ncells;
    brick_job *jobs = (brick_job *)calloc((size_t)nthreads, sizeof(brick_job));
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int i = 0; i < nthreads; ++i) {
        jobs[i].base.g = g;
        jobs[i].base.X = X;
        jobs[i].base.Y = Y;
        jobs[i].base.Z = Z;
        jobs[i].base.factor = factor;
        jobs[i].base.w = w;
        jobs[i].base.scratch_pool = scratch;
        jobs[i].base.begin = w->ncells * (uint64_t)i / (uint64_t)nthreads;
        jobs[i].base.end = w->ncells * (uint64_t)(i + 1) / (uint64_t)nthreads;
        jobs[i].dense = dense;
        jobs[i].any_flags = any_flags;
        pthread_create(&th[i], NULL, brick_worker, &jobs[i]);
    }
    for (int i = 0; i < nthreads; ++i)
        pthread_join(th[i], NULL);
    free(th);
    free(jobs);

    /* coarse bit = any (VolumeRaytracer.cuh:504-507); pool slots in index order */
    uint64_t nslots = 0;
    for (uint64_t i = 0; i < w->ncells; ++i)
        nslots += any_flags[i];
    w->nslots = nslots;
    w->pool = (uint32_t *)malloc((nslots ? nslots : 1) * bw * sizeof(uint32_t));
    uint64_t next = 0;
    for (uint64_t i = 0; i < w->ncells; ++i) {
        if (any_flags[i]) {
            w->coarse_bits[i >> 5] |= 1u << (i & 31u);
            memcpy(w->pool + next * bw, scratch + i * bw, bw * sizeof(uint32_t));
            w->brick_slot[i] = (uint32_t)next++;
        } else {
            w->brick_slot[i] = VXO_EMPTY_SLOT;
        }
    }
    free(scratch);
    free(any_flags);
    return w;
}

vxo_world *vxo_build_brickmap(const uint32_t *dense_bits, int X, int Y, int Z, int factor)
{
    return build_common(dense_bits, 0, X, Y, Z, factor, 8);
}

vxo_world *vxo_gen_brickmap(int g, int X, int Y, int Z, int factor, int nthreads)
{
    return build_common(NULL, g, X, Y, Z, factor, nthreads);
}
