/*
 * vxo_render.c -- ORACLE (test infrastructure; parity unpinned, see vxo.h).
 * Camera, per-pixel primary/secondary shading and the BGRA8 store, restated
 * from the reference's VoxelRT/Renderer.cu, plus the batch entry point
 * (VoxelRT/VolumeRaytracer.cu:95-117,574-618).  Threads split the launch grid's
 * rows; every pixel is a pure function of its inputs so the thread count does
 * not change results.
 */
#include "vxo.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct v3 { float x, y, z; } v3;

static inline v3 mk(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 from(const float *p) { return mk(p[0], p[1], p[2]); }
static inline v3 add(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 scl(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
static inline v3 mul(v3 a, v3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* helper_math.h:1264-1267 */
static inline float lo(float a, float b) { return a < b ? a : b; }
static inline float hi(float a, float b) { return a > b ? a : b; }
/* normalize = v * rsqrtf(dot), host rsqrtf = 1.0f/sqrtf (helper_math.h:78-81,1325-1329) */
static inline v3 unit(v3 v) { return scl(v, 1.0f / sqrtf(dot3(v, v))); }
/* reflect(i,n) = i - 2.0f*n*dot(n,i) (helper_math.h:1427-1430) */
static inline v3 bounce_dir(v3 i, v3 n) { return sub(i, scl(scl(n, 2.0f), dot3(n, i))); }

/* GetDirections (Renderer.cu:27-42): cos/sin on float arguments resolve to the
 * float overloads (cosf/sinf); returned forward and up are negated. */
void vxo_get_directions(const float euler[3], float fwd_out[3], float up_out[3], float right_out[3])
{
    v3 f, r;
    f.x = cosf(euler[0]) * sinf(euler[1]);
    f.y = -sinf(euler[0]);
    f.z = cosf(euler[0]) * cosf(euler[1]);
    r.x = cosf(euler[1]);
    r.y = 0;
    r.z = -sinf(euler[1]);
    /* cross(fwd, rgt), helper_math.h:1436-1439 */
    v3 u = mk(f.y * r.z - f.z * r.y, f.z * r.x - f.x * r.z, f.x * r.y - f.y * r.x);
    fwd_out[0] = f.x * -1; fwd_out[1] = f.y * -1; fwd_out[2] = f.z * -1;
    up_out[0] = u.x * -1;  up_out[1] = u.y * -1;  up_out[2] = u.z * -1;
    right_out[0] = r.x;    right_out[1] = r.y;    right_out[2] = r.z;
}

/* powf(x, 32) at Renderer.cu:114.  libm powf implementations differ in the last
 * ulp between hosts and GPUs; this build defines the value as x^32 by five
 * squarings in binary64 rounded once to binary32, which is the correctly rounded
 * power except on astronomically rare near-ties (tests compare it with libm). */
static inline float pow32(float x)
{
    double p = (double)x;
    p *= p; p *= p; p *= p; p *= p; p *= p;
    return (float)p;
}

typedef struct frame_ctx {
    const vxo_world *w;
    const vxo_render_params *p;
    uint8_t *fb;
    float *color_aov;
    int64_t *hit_aov;
    float *accum;       /* temporal accumulation (this build's extension): W*H*4 floats {sum r,g,b, frames}, or NULL */
    int accum_reset;
    uint32_t first_chunk, chunk_stride, rows; /* launch-grid rows for this worker, in chunks of ROW_CHUNK */
    uint32_t grid_w;
    vxo_frame_stats stats;
} frame_ctx;

/* setPixelColor (Renderer.cu:72-87): clamp to [0,1], scale by 255, truncate;
 * memory order b,g,r,a (Renderer.cuh:29-31). */
static void put_pixel(frame_ctx *c, int x, int y, v3 col)
{
    const vxo_render_params *p = c->p;
    if ((uint32_t)x >= p->width || (uint32_t)y >= p->height)
        return;
    size_t i = (size_t)y * p->width + (size_t)x;
    if (c->color_aov) {
        c->color_aov[i * 3 + 0] = col.x;
        c->color_aov[i * 3 + 1] = col.y;
        c->color_aov[i * 3 + 2] = col.z;
    }
    col.x = lo(hi(col.x, 0), 1);
    col.y = lo(hi(col.y, 0), 1);
    col.z = lo(hi(col.z, 0), 1);
    uint8_t *px = c->fb + i * 4;
    px[2] = (uint8_t)(col.x * 255);
    px[1] = (uint8_t)(col.y * 255);
    px[0] = (uint8_t)(col.z * 255);
    px[3] = 255;
    c->stats.pixels_written += 1;
}

static int trace(frame_ctx *c, int max_steps, v3 o, v3 d, int *steps, v3 *n, v3 *pos, int vox[3])
{
    float of[3] = {o.x, o.y, o.z}, df[3] = {d.x, d.y, d.z}, nf[3], pf[3] = {pos->x, pos->y, pos->z};
    int h = vxo_raytrace(c->w, max_steps, of, df, steps, nf, pf, vox, &c->stats.probes);
    *n = from(nf);
    *pos = from(pf);
    return h;
}

/* calculateColor (Renderer.cu:90-168) with the two switches the checked-in file
 * hard-codes exposed: `shadow` re-enables the call commented out at :102,
 * `bounce_samples` is `samples` at :123.  The inner-scope `occlusion` at :146-147
 * shadows the accumulator, so a hit sample adds nothing and a miss adds one.
 * `normal` is the outward (already negated) normal. */
static v3 shade(frame_ctx *c, uint32_t tx, uint32_t ty, v3 cam, v3 normal, v3 position)
{
    const vxo_render_params *p = c->p;
    v3 L = from(p->env.light_dir), Lc = from(p->env.light_color), Amb = from(p->env.ambient);

    v3 sray = unit(L);
    v3 spos = add(position, scl(sray, 0.01f));
    int shadowed = 0;
    if (p->shadow) {
        int st;
        v3 sn;
        c->stats.shadow_rays += 1;
        shadowed = trace(c, VXO_MAX_STEPS, spos, sray, &st, &sn, &spos, NULL);
    }
    float l_dot = hi(dot3(normal, L), 0) * (float)(shadowed ? 0 : 1);
    v3 diffuse = scl(Lc, l_dot);
    /* lerp(0.25, 1.0, dot(n,(0,1,0))*0.5+0.5): the blend factor is computed in double and
     * narrowed at the call (helper_math.h:1146-1149 lerp = a + t*(b-a)) */
    float up_dot = normal.x * 0.0f + normal.y * 1.0f + normal.z * 0.0f;
    float t = (float)((double)up_dot * 0.5 + 0.5);
    v3 color = add(diffuse, scl(Amb, 0.25f + t * (1.0f - 0.25f)));

    if (!shadowed) {
        v3 view = unit(sub(position, cam));
        v3 refl = bounce_dir(L, normal);
        float spec = pow32(hi(dot3(view, refl), 0));
        color.x += spec * Lc.x;
        color.y += spec * Lc.y;
        color.z += spec * Lc.z;
    }

    if (l_dot == 0 || p->bounce_all_hits) {
        const int samples = p->bounce_samples;
        /* seed from the un-remapped launch coordinates (Renderer.cu:124-126) */
        uint32_t seed = ty * p->width + tx;
        float occl = 0.0f;
        for (int i = 0; i < samples; ++i) {
            uint32_t si = seed + (uint32_t)i * 1000u + (p->frame_number + 1u) * 1000u;
            v3 sd = mk(vxo_random_float(si) * 2 - 1, vxo_random_float(si * 10u) * 2 - 1,
                       vxo_random_float(si * 100u) * 2 - 1);
            sd = unit(sd);
            if (dot3(sd, normal) < 0)
                sd = bounce_dir(sd, normal);
            v3 sp = add(position, scl(normal, 0.01f)), sn;
            int st;
            c->stats.bounce_rays += 1;
            if (!trace(c, 8, sp, sd, &st, &sn, &sp, NULL)) {
                occl += 1.0f;
            } else if (p->bounce_depth >= 2) {
                /* extension (not in the reference): second diffuse bounce from the sample ray's hit point `sp`,
                 * built exactly like the first one around the outward normal there */
                v3 n2 = mk(-sn.x, -sn.y, -sn.z);
                uint32_t s2 = si + 500u;
                v3 d2 = mk(vxo_random_float(s2) * 2 - 1, vxo_random_float(s2 * 10u) * 2 - 1,
                           vxo_random_float(s2 * 100u) * 2 - 1);
                d2 = unit(d2);
                if (dot3(d2, n2) < 0)
                    d2 = bounce_dir(d2, n2);
                v3 o2 = add(sp, scl(n2, 0.01f)), sn2;
                c->stats.bounce_rays += 1;
                if (!trace(c, 8, o2, d2, &st, &sn2, &o2, NULL))
                    occl += 0.5f;
            }
        }
        if (samples > 0)
            occl /= (float)samples;
        else
            occl = 1.0f;
        color = scl(color, occl);
    }
    return color;
}

/* Tonemap (Renderer.cu:170-177) */
static v3 tonemap(v3 c)
{
    v3 t = mk(c.x / (c.x + 1.0f), c.y / (c.y + 1.0f), c.z / (c.z + 1.0f));
    return mk(lo(hi(t.x, 0), 1), lo(hi(t.y, 0), 1), lo(hi(t.z, 0), 1));
}

/* Temporal accumulation of the stochastic occlusion term (the reference's README lists "denoise, temporal accumulation"
 * as to do, README.md:19; defined by this build, include/vxrt.h): the pre-tonemap colour of a shaded hit pixel is added
 * to the pixel's history and the MEAN is what gets tonemapped.  First frame of a history (reset, or no frames yet): the
 * colour itself. */
static v3 accumulate(frame_ctx *c, int x, int y, v3 col)
{
    float *h = c->accum + ((size_t)y * c->p->width + (size_t)x) * 4;
    if (c->accum_reset || h[3] == 0.0f) {
        h[0] = col.x; h[1] = col.y; h[2] = col.z; h[3] = 1.0f;
        return col;
    }
    h[0] = h[0] + col.x; h[1] = h[1] + col.y; h[2] = h[2] + col.z; h[3] = h[3] + 1.0f;
    return mk(h[0] / h[3], h[1] / h[3], h[2] / h[3]);
}

/* one launch thread of screenDispatch (Renderer.cu:179-276) */
static void pixel_thread(frame_ctx *c, uint32_t tx, uint32_t ty)
{
    const vxo_render_params *p = c->p;
    int x = (int)tx, y = (int)ty;
    if (p->checkerboard) {                       /* :186-194 */
        y *= 2;
        if ((x % 2) == 0)
            y += 1;
        if (p->frame_number % 2 == 0)
            y += 1;
    }
    if ((uint32_t)x >= p->width || (uint32_t)y >= p->height)
        return;
    if ((uint32_t)y < p->row_begin || (uint32_t)y >= p->row_end)
        return;                                  /* strip sharding (this build) */
    const int W = (int)p->width, H = (int)p->height;
    float u = (float)x / (float)W, v = (float)y / (float)H;
    v3 origin = from(p->origin), fwd = from(p->fwd), up = from(p->up), right = from(p->right);
    v3 ray;
    if (p->ortho) {                              /* getRayDirectionOrtho, :61-70 */
        float ratio = (float)p->width / (float)p->height;
        ray = fwd;
        origin = add(origin, scl(scl(scl(right, u * 2 - 1), p->ortho_size[0]), ratio));
        origin = add(origin, scl(scl(up, v * 2 - 1), p->ortho_size[1]));
    } else {                                     /* getRayDirection, :44-59 */
        float aspect = (float)p->width / (float)p->height;
        float su = u * 2 - 1, sv = v * 2 - 1;
        float fov = (float)((double)p->fov_deg * 3.1415 / 180.0);
        float kx = tanf(fov / 2.0f) * aspect, ky = tanf(fov / 2.0f);
        ray.x = fwd.x + su * kx * right.x + sv * ky * up.x;
        ray.y = fwd.y + su * kx * right.y + sv * ky * up.y;
        ray.z = fwd.z + su * kx * right.z + sv * ky * up.z;
        ray = unit(ray);
    }

    int steps = 0, vox[3] = {0, 0, 0};
    v3 normal, pos = mk(0, 0, 0);
    c->stats.primary_rays += 1;
    int hit = trace(c, VXO_MAX_STEPS, origin, ray, &steps, &normal, &pos, vox);
    normal = mk(-normal.x, -normal.y, -normal.z);
    if (c->hit_aov) {
        int X = c->w->cdims[0] * c->w->factor, Y = c->w->cdims[1] * c->w->factor;
        c->hit_aov[(size_t)y * p->width + (size_t)x] =
            hit ? (int64_t)vox[0] + (int64_t)X * ((int64_t)vox[1] + (int64_t)Y * (int64_t)vox[2]) : -1;
    }
    if (hit) {
        c->stats.primary_hits += 1;
        if (p->mode == VXO_MODE_DEBUG) {         /* :215-243 */
            v3 dvec = sub(pos, origin);
            float dist = sqrtf(dot3(dvec, dvec));
            const float wrap = (float)(1.0 + 1e-6); /* fmodf(x, 1.0f + FLT_EPS_DDA): double sum narrowed */
            v3 hp = mk(fmodf(pos.x / 128.0f, wrap), fmodf(pos.y / 128.0f, wrap), fmodf(pos.z / 128.0f, wrap));
            if (x < (W >> 1) && y < (H >> 1))
                put_pixel(c, x, y, normal);
            else if (x >= (W >> 1) && y < (H >> 1))
                put_pixel(c, x, y, hp);
            else if (x < (W >> 1)) {
            } else
                put_pixel(c, x, y, mk(dist * 0.01f, 0, 0));
        } else {                                 /* :245-251 */
            v3 col = shade(c, tx, ty, origin, normal, pos);
            if (c->accum)
                col = accumulate(c, x, y, col);
            put_pixel(c, x, y, tonemap(col));
        }
    } else {
        put_pixel(c, x, y, ray);                 /* :254-258 */
    }
    /* crosshair keyed on the un-remapped launch coordinates (:261-268) */
    if (tx == (p->width >> 1) && ty == (p->height >> 1))
        put_pixel(c, x, y, mk(10, 10, 10));
    if (p->mode == VXO_MODE_DEBUG && x < (W >> 1) && y > (H >> 1))   /* :270-275 */
        put_pixel(c, x, y, mk((float)steps / 256.0f, 0, 0));
}

#define ROW_CHUNK 4u

/* rows are dealt to workers in interleaved chunks so sky and terrain rows balance */
static void *frame_worker(void *arg)
{
    frame_ctx *c = (frame_ctx *)arg;
    for (uint32_t ch = c->first_chunk; ch * ROW_CHUNK < c->rows; ch += c->chunk_stride) {
        uint32_t end = ch * ROW_CHUNK + ROW_CHUNK < c->rows ? ch * ROW_CHUNK + ROW_CHUNK : c->rows;
        for (uint32_t ty = ch * ROW_CHUNK; ty < end; ++ty)
            for (uint32_t tx = 0; tx < c->grid_w; ++tx)
                pixel_thread(c, tx, ty);
    }
    return NULL;
}

static void stats_add(vxo_frame_stats *a, const vxo_frame_stats *b)
{
    a->primary_rays += b->primary_rays;
    a->shadow_rays += b->shadow_rays;
    a->bounce_rays += b->bounce_rays;
    a->primary_hits += b->primary_hits;
    a->pixels_written += b->pixels_written;
    a->probes.coarse_probes += b->probes.coarse_probes;
    a->probes.brick_entries += b->probes.brick_entries;
    a->probes.fine_probes += b->probes.fine_probes;
}

/* RenderScreen's launch shape (Renderer.cu:311-316): blocks of 32x1 threads over
 * width x (height>>1 if checkerboard). */
void vxo_render(const vxo_world *w, const vxo_render_params *p, uint8_t *fb, float *color_aov,
                int64_t *hit_aov, vxo_frame_stats *stats, int nthreads)
{
    vxo_render_accum(w, p, fb, color_aov, hit_aov, NULL, 0, stats, nthreads);
}

void vxo_render_accum(const vxo_world *w, const vxo_render_params *p, uint8_t *fb, float *color_aov,
                      int64_t *hit_aov, float *accum, int accum_reset, vxo_frame_stats *stats, int nthreads)
{
    uint32_t rows = p->checkerboard ? (p->height >> 1) : p->height;
    uint32_t grid_w = ((p->width + 31u) / 32u) * 32u;
    if (nthreads < 1)
        nthreads = 1;
    if ((uint32_t)nthreads > rows)
        nthreads = rows ? (int)rows : 1;
    frame_ctx *ctx = (frame_ctx *)calloc((size_t)nthreads, sizeof(frame_ctx));
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    vxo_frame_stats total;
    memset(&total, 0, sizeof(total));
    for (int i = 0; i < nthreads; ++i) {
        frame_ctx *c = &ctx[i];
        c->w = w;
        c->p = p;
        c->fb = fb;
        c->color_aov = color_aov;
        c->hit_aov = hit_aov;
        c->accum = accum;
        c->accum_reset = accum_reset;
        c->grid_w = grid_w;
        c->rows = rows;
        c->first_chunk = (uint32_t)i;
        c->chunk_stride = (uint32_t)nthreads;
        pthread_create(&th[i], NULL, frame_worker, c);
    }
    for (int i = 0; i < nthreads; ++i) {
        pthread_join(th[i], NULL);
        stats_add(&total, &ctx[i].stats);
    }
    free(th);
    free(ctx);
    if (stats)
        *stats = total;
}

/* ------------------------------------------------------------------- batch */

typedef struct batch_ctx {
    const vxo_world *w;
    const float *origins, *dirs;
    size_t begin, end;
    float *out_pos, *out_normal;
    int32_t *out_steps;
    uint8_t *out_hit;
    int64_t *out_voxel;
    vxo_ray_stats stats;
} batch_ctx;

static void *batch_worker(void *arg)
{
    batch_ctx *b = (batch_ctx *)arg;
    const int64_t X = (int64_t)b->w->cdims[0] * b->w->factor, Y = (int64_t)b->w->cdims[1] * b->w->factor;
    for (size_t i = b->begin; i < b->end; ++i) {
        int steps = 0, vox[3] = {0, 0, 0};
        float n[3], pos[3] = {0, 0, 0};
        int h = vxo_raytrace(b->w, VXO_MAX_STEPS, b->origins + 3 * i, b->dirs + 3 * i, &steps, n, pos, vox,
                             &b->stats);
        for (int a = 0; a < 3; ++a) {
            b->out_pos[3 * i + a] = h ? pos[a] : (float)INFINITY; /* dispatch, VolumeRaytracer.cu:105-113 */
            b->out_normal[3 * i + a] = n[a];
        }
        b->out_steps[i] = steps;
        if (b->out_hit)
            b->out_hit[i] = (uint8_t)h;
        if (b->out_voxel)
            b->out_voxel[i] = h ? (int64_t)vox[0] + X * ((int64_t)vox[1] + Y * (int64_t)vox[2]) : -1;
    }
    return NULL;
}

void vxo_trace_batch(const vxo_world *w, const float *origins, const float *dirs, size_t n, float *out_pos,
                     float *out_normal, int32_t *out_steps, uint8_t *out_hit, int64_t *out_voxel,
                     vxo_ray_stats *stats_sum, int nthreads)
{
    if (nthreads < 1)
        nthreads = 1;
    if ((size_t)nthreads > n)
        nthreads = n ? (int)n : 1;
    batch_ctx *ctx = (batch_ctx *)calloc((size_t)nthreads, sizeof(batch_ctx));
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int i = 0; i < nthreads; ++i) {
        ctx[i].w = w;
        ctx[i].origins = origins;
        ctx[i].dirs = dirs;
        ctx[i].begin = n * (size_t)i / (size_t)nthreads;
        ctx[i].end = n * (size_t)(i + 1) / (size_t)nthreads;
        ctx[i].out_pos = out_pos;
        ctx[i].out_normal = out_normal;
        ctx[i].out_steps = out_steps;
        ctx[i].out_hit = out_hit;
        ctx[i].out_voxel = out_voxel;
        pthread_create(&th[i], NULL, batch_worker, &ctx[i]);
    }
    vxo_ray_stats sum = {0, 0, 0};
    for (int i = 0; i < nthreads; ++i) {
        pthread_join(th[i], NULL);
        sum.coarse_probes += ctx[i].stats.coarse_probes;
        sum.brick_entries += ctx[i].stats.brick_entries;
        sum.fine_probes += ctx[i].stats.fine_probes;
    }
    if (stats_sum)
        *stats_sum = sum;
    free(th);
    free(ctx);
}
