/*
 * vxo_trace.c -- ORACLE (test infrastructure; parity unpinned, see vxo.h).
 * Bit layout, slab test, single-level DDA and the two-level brickmap trace,
 * restated from the reference's VoxelRT/VolumeRaytracer.cu / .cuh.
 * Build with -ffp-contract=off and without fast-math: results are meant to be
 * the reference source evaluated with IEEE-754 binary32 semantics.
 */
#include "vxo.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define VXO_INF ((float)INFINITY)
#define VXO_FLT_EPS 1.1920928955078125e-7f /* std::numeric_limits<float>::epsilon, VolumeRaytracer.cuh:22 */
#define VXO_EPS_DDA 1e-6                   /* a DOUBLE in the reference, VolumeRaytracer.cuh:20 */

/* ------------------------------------------------------------------ layout */

/* GetSampleIndex, SAMPLE_MODE_TILED_LINEAR (VolumeRaytracer.cuh:107-131):
 * 8x8x8 tiles stored linearly, tiles ordered x-fastest. Computed in int and
 * returned as uint32_t like the reference. */
uint32_t vxo_sample_index(uint32_t x, uint32_t y, uint32_t z, uint32_t width, uint32_t height)
{
    int tiles_w = (int)(width / 8u);
    int tiles_h = (int)(height / 8u);
    int tile = (int)(x / 8u) + (int)(y / 8u) * tiles_w + (int)(z / 8u) * tiles_w * tiles_h;
    int inside = (int)(x % 8u) + (int)(y % 8u) * 8 + (int)(z % 8u) * 64;
    return (uint32_t)(tile * 512 + inside);
}

uint64_t vxo_sample_index64(uint64_t x, uint64_t y, uint64_t z, uint64_t width, uint64_t height)
{
    uint64_t tiles_w = width / 8u, tiles_h = height / 8u;
    uint64_t tile = (x / 8u) + (y / 8u) * tiles_w + (z / 8u) * tiles_w * tiles_h;
    return tile * 512u + (x % 8u) + (y % 8u) * 8u + (z % 8u) * 64u;
}

/* GetPositionFromSampleIndex (VolumeRaytracer.cuh:138-171) */
void vxo_position_from_index(uint32_t index, uint32_t width, uint32_t height, uint32_t *x, uint32_t *y,
                             uint32_t *z)
{
    uint32_t tiles_w = width / 8u, tiles_h = height / 8u;
    uint32_t tile = index / 512u, inside = index % 512u;
    *x = (tile % tiles_w) * 8u + inside % 8u;
    *y = ((tile / tiles_w) % tiles_h) * 8u + (inside / 8u) % 8u;
    *z = (tile / (tiles_w * tiles_h)) * 8u + inside / 64u;
}

/* BitArray::operator[] const (VolumeRaytracer.cu:61-68): LSB-first in u32 words,
 * reads past the end give 0. */
int vxo_bit_get(const uint32_t *words, uint64_t nbits, uint64_t index)
{
    if (index >= nbits)
        return 0;
    return (int)((words[index / 32u] >> (index % 32u)) & 1u);
}

/* BitRef::operator= (VolumeRaytracer.cu:19-36); the oracle's builders are
 * single-writer per word or use it under their own partitioning, so no atomics. */
void vxo_bit_set(uint32_t *words, uint64_t index, int value)
{
    uint32_t mask = 1u << (index & 31u);
    if (value)
        words[index / 32u] |= mask;
    else
        words[index / 32u] &= ~mask;
}

/* ---------------------------------------------------------------- slab test */

/* helper_math.h:56-64 host fminf/fmaxf are plain comparisons */
static inline float lo(float a, float b) { return a < b ? a : b; }
static inline float hi(float a, float b) { return a > b ? a : b; }

/* RayIntersectsAABB (VolumeRaytracer.cu:124-174).  Zero direction components are
 * replaced by FLT_EPSILON before the reciprocal (:127-129); rejection is
 * t_exit < max(t_enter, 0) (:148) so a start inside the box is a hit whose point
 * lies behind the start; the normal is the sign of travel on the first axis (x,y,z
 * order) whose near-plane time equals t_enter (:157-171). */
int vxo_ray_aabb(const float start[3], const float dir[3], const float bmin[3], const float bmax[3],
                 float out_p[3], float out_n[3])
{
    float inv[3], near_t[3], far_t[3];
    for (int a = 0; a < 3; ++a) {
        inv[a] = 1.0f / (dir[a] == 0 ? VXO_FLT_EPS : dir[a]);
        float ta = (bmin[a] - start[a]) * inv[a];
        float tb = (bmax[a] - start[a]) * inv[a];
        near_t[a] = lo(ta, tb);
        far_t[a] = hi(ta, tb);
    }
    float t_enter = hi(hi(near_t[0], near_t[1]), near_t[2]);
    float t_exit = lo(lo(far_t[0], far_t[1]), far_t[2]);
    if (t_exit < hi(t_enter, 0.0f))
        return 0;
    if (out_p) {
        for (int a = 0; a < 3; ++a)
            out_p[a] = start[a] + t_enter * dir[a];
    }
    if (out_n) {
        int axis = (t_enter == near_t[0]) ? 0 : (t_enter == near_t[1]) ? 1 : 2;
        out_n[0] = out_n[1] = out_n[2] = 0.0f;
        out_n[axis] = (inv[axis] < 0.0f) ? -1.0f : 1.0f;
    }
    return 1;
}

/* ------------------------------------------------------------ single-level DDA */

/* float -> int as the GPUs the reference targets do it (cvt.rzi / v_cvt_i32_f32): truncation that SATURATES, NaN -> 0.
 * C leaves out-of-range conversions undefined (x86 yields INT_MIN); rays with infinite or huge intermediate
 * positions (denormal direction components make 1/d overflow) reach such conversions, so the definition matters. */
static inline int sat_i32(float v)
{
    if (!(v == v))
        return 0;
    if (v >= 2147483648.0f)
        return 2147483647;
    if (v <= -2147483648.0f)
        return (-2147483647 - 1);
    return (int)v;
}

/* min(max(v, lo), hi) exactly as written at VolumeRaytracer.cu:242-244 */
static inline int clampi(int v, int lo_, int hi_)
{
    int m = v > lo_ ? v : lo_;
    return m < hi_ ? m : hi_;
}

/* DDARayTraversal (VolumeRaytracer.cu:176-352), Amanatides-Woo with the
 * reference's quirks kept:
 *  - edge rule (:216-232,240): a start cell equal to the dimension on any axis
 *    widens the accepted range by one on axes travelled in the negative sense;
 *    lookups use the clamped cell (:242-244);
 *  - with per-cell bounds the tight box is tested from the DDA *start* (:260) and
 *    its entry point replaces the running point only when step != 0 (:266-269);
 *  - the ray always advances once more on the exit iteration to produce NextCell
 *    (:290-322, :345-349);
 *  - axis choice x if tx<ty && tx<tz, else y if ty<=tx && ty<tz, else z (:293-313);
 *  - the optional region check truncates min/max to int, is inclusive, applies to
 *    the boundary point, and on failure neither counts the step nor moves the
 *    point (:325-341).
 * Fields the reference leaves uninitialised (HitCell/NextCell/HitNormal) start
 * at zero here. */
void vxo_dda(const vxo_dda_params *p, vxo_dda_result *r)
{
    const float *s = p->start, *d = p->dir;
    int cell[3], sgn[3], pad[3] = {0, 0, 0};
    float t_delta[3], t_next[3];
    for (int a = 0; a < 3; ++a) {
        cell[a] = sat_i32(s[a]);
        sgn[a] = (d[a] > 0) ? 1 : -1;
        t_delta[a] = (d[a] != 0) ? fabsf(1.0f / d[a]) : VXO_INF;
        t_next[a] = (d[a] != 0) ? (((float)(int)((unsigned)cell[a] + (unsigned)(sgn[a] > 0)) - s[a]) / d[a]) : VXO_INF;
    }
    memset(r, 0, sizeof(*r));
    for (int a = 0; a < 3; ++a)
        r->point[a] = s[a];

    const int *dim = p->dims;
    if (cell[0] == dim[0] || cell[1] == dim[1] || cell[2] == dim[2]) {
        for (int a = 0; a < 3; ++a)
            pad[a] = (d[a] < 0) ? 1 : 0;
    }

    int leaving = 0;
    for (int it = 0; it < p->max_steps; ++it) {
        int skip = p->take_initial_step && it == 0;
        if (!skip) {
            int inside = 1;
            for (int a = 0; a < 3; ++a)
                inside = inside && 0 <= cell[a] && cell[a] < dim[a] + pad[a];
            if (inside) {
                int c[3];
                for (int a = 0; a < 3; ++a) {
                    c[a] = clampi(cell[a], 0, dim[a] - 1);
                    r->hit_cell[a] = (float)c[a];
                }
                r->probes += 1;
                uint32_t idx = vxo_sample_index((uint32_t)c[0], (uint32_t)c[1], (uint32_t)c[2],
                                                (uint32_t)dim[0], (uint32_t)dim[1]);
                /* non-const BitArray::operator[] -> BitRef: no size check (VolumeRaytracer.cu:70-73,15-18).
                 * A NULL grid (empty brick descriptor; never entered when the tables are consistent)
                 * reads as empty instead of faulting. */
                int solid = p->bits ? (int)((p->bits[idx / 32u] >> (idx % 32u)) & 1u) : 0;
                if (p->cell_bounds) {
                    const float *cb = p->cell_bounds + (size_t)idx * 6;
                    float scale = (float)p->cell_bounds_scale;
                    float bmin[3], bmax[3];
                    for (int a = 0; a < 3; ++a) {
                        bmin[a] = (cb[a] + 0) / scale + (float)c[a];
                        bmax[a] = (cb[3 + a] + 1) / scale + (float)c[a];
                    }
                    if (solid && bmin[0] <= bmax[0]) {
                        float bp[3] = {0, 0, 0}, bn[3] = {0, 0, 0};
                        if (vxo_ray_aabb(s, d, bmin, bmax, bp, bn)) {
                            r->hit = 1;
                            memcpy(r->normal, bn, sizeof(bn));
                            if (it != 0)
                                memcpy(r->point, bp, sizeof(bp));
                            leaving = 1;
                        }
                    }
                } else if (solid) {
                    r->hit = 1;
                    leaving = 1;
                }
            } else {
                r->out_of_bounds = 1;
                leaving = 1;
            }
        }

        int axis;
        if (t_next[0] < t_next[1] && t_next[0] < t_next[2])
            axis = 0;
        else if (t_next[1] <= t_next[0] && t_next[1] < t_next[2])
            axis = 1;
        else
            axis = 2;
        float t = t_next[axis];
        float crossing[3];
        for (int a = 0; a < 3; ++a)
            crossing[a] = (a == axis) ? (float)(int)((unsigned)cell[a] + (unsigned)(sgn[a] > 0)) : s[a] + (t * d[a]);
        cell[axis] = (int)((unsigned)cell[axis] + (unsigned)sgn[axis]);  /* wraps like the hardware add */
        t_next[axis] += t_delta[axis];

        if (leaving) {
            for (int a = 0; a < 3; ++a)
                r->next_cell[a] = (float)cell[a];
            break;
        }
        r->normal[0] = r->normal[1] = r->normal[2] = 0.0f;
        r->normal[axis] = (float)sgn[axis];
        if (p->has_bounds) {
            int outside = 0;
            for (int a = 0; a < 3; ++a) {
                int mn = (int)p->bounds_min[a], mx = (int)p->bounds_max[a];
                outside = outside || crossing[a] < (float)mn || crossing[a] > (float)mx;
            }
            if (outside) {
                r->out_of_bounds = 1;
                break;
            }
        }
        r->steps += 1;
        memcpy(r->point, crossing, sizeof(crossing));
    }
}

/* ----------------------------------------------------------- two-level trace */

static inline float step_ulp(float v, float toward_sign_dir)
{
    /* nextafterf(v, dir<0 ? -inf : +inf) (VolumeRaytracer.cu:452-460) */
    return nextafterf(v, toward_sign_dir < 0 ? -VXO_INF : VXO_INF);
}

/* Raytrace (VolumeRaytracer.cu:354-525). */
int vxo_raytrace(const vxo_world *w, int max_steps, const float origin[3], const float ray[3],
                 int *out_steps, float out_normal[3], float out_pos[3], int hit_voxel[3],
                 vxo_ray_stats *stats)
{
    const int f = w->factor;
    const float ff = (float)f;
    const uint64_t brick_words = ((uint64_t)f * f * f) / 32u;
    float last_cell[3] = {-1, -1, -1};
    int total = 0;

    float start[3], dir[3], entry_normal[3] = {0, 0, 0};
    for (int a = 0; a < 3; ++a)
        start[a] = origin[a] / ff;                                   /* :362-365 */
    {   /* normalize = v * rsqrtf(dot(v,v)), host rsqrtf = 1/sqrtf (helper_math.h:78-81,1325-1329) */
        float inv_len = 1.0f / sqrtf(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
        for (int a = 0; a < 3; ++a)
            dir[a] = ray[a] * inv_len;
    }
    int inside_grid = 1;
    for (int a = 0; a < 3; ++a)
        inside_grid = inside_grid && start[a] >= 0 && start[a] < (float)w->cdims[a];
    if (!inside_grid) {                                              /* :369-381 */
        float bmin[3], bmax[3], entry[3];
        for (int a = 0; a < 3; ++a) {
            bmin[a] = (float)VXO_EPS_DDA;
            bmax[a] = (float)((double)w->cdims[a] - VXO_EPS_DDA);    /* double subtraction, then to float */
        }
        if (vxo_ray_aabb(start, dir, bmin, bmax, entry, entry_normal))
            memcpy(start, entry, sizeof(entry));
    }
    out_normal[0] = out_normal[1] = out_normal[2] = 0.0f;           /* :382 */
    float hit_pos[3] = {0, 0, 0};
    int hit = 0;

    while (total < max_steps) {                                      /* :386, checked at the head only */
        vxo_dda_params cp;
        memset(&cp, 0, sizeof(cp));
        cp.bits = w->coarse_bits;
        cp.nbits = w->ncells;
        memcpy(cp.dims, w->cdims, sizeof(cp.dims));
        memcpy(cp.start, start, sizeof(start));
        memcpy(cp.dir, dir, sizeof(dir));
        cp.max_steps = VXO_MAX_STEPS;
        cp.cell_bounds = w->bounds;
        cp.cell_bounds_scale = f;
        vxo_dda_result cr;
        vxo_dda(&cp, &cr);
        if (stats)
            stats->coarse_probes += (uint64_t)cr.probes;

        total += cr.steps;
        float local[3];
        for (int a = 0; a < 3; ++a) {
            local[a] = cr.point[a] * ff;                             /* :396-398 */
            hit_pos[a] = local[a];
        }
        if (!(cr.hit && !cr.out_of_bounds))
            break;                                                   /* :508-511 */
        if (last_cell[0] == cr.hit_cell[0] && last_cell[1] == cr.hit_cell[1] &&
            last_cell[2] == cr.hit_cell[2])
            break;                                                   /* :402-407 */
        memcpy(last_cell, cr.hit_cell, sizeof(last_cell));
        for (int a = 0; a < 3; ++a)
            local[a] -= cr.hit_cell[a] * ff;                         /* :415-417 */

        uint32_t ci = vxo_sample_index((uint32_t)cr.hit_cell[0], (uint32_t)cr.hit_cell[1],
                                       (uint32_t)cr.hit_cell[2], (uint32_t)w->cdims[0],
                                       (uint32_t)w->cdims[1]);       /* :419 */
        uint32_t slot = w->brick_slot[ci];
        vxo_dda_params bp;
        memset(&bp, 0, sizeof(bp));
        if (slot != VXO_EMPTY_SLOT) {
            bp.bits = w->pool + (uint64_t)slot * brick_words;
            bp.nbits = (uint64_t)f * f * f;
            bp.dims[0] = bp.dims[1] = bp.dims[2] = f;
        }
        memcpy(bp.start, local, sizeof(local));
        memcpy(bp.dir, dir, sizeof(dir));
        bp.max_steps = VXO_MAX_STEPS;
        bp.has_bounds = 1;                                           /* [0,f]^3, :409-414,:422 */
        bp.bounds_max[0] = bp.bounds_max[1] = bp.bounds_max[2] = ff;
        vxo_dda_result br;
        vxo_dda(&bp, &br);
        if (stats) {
            stats->brick_entries += 1;
            stats->fine_probes += (uint64_t)br.probes;
        }

        total += br.steps;
        for (int a = 0; a < 3; ++a)
            hit_pos[a] = br.point[a] + cr.hit_cell[a] * ff;          /* :427-429 */

        if (br.hit) {                                                /* :493-506 */
            const float *n = (br.steps == 0) ? cr.normal : br.normal;
            memcpy(out_normal, n, 3 * sizeof(float));
            if (hit_voxel) {
                for (int a = 0; a < 3; ++a)
                    hit_voxel[a] = (int)cr.hit_cell[a] * f + (int)br.hit_cell[a];
            }
            hit = 1;
            break;
        }
        /* brick missed: restart the coarse walk just past it (:431-491) */
        for (int a = 0; a < 3; ++a)
            start[a] = hit_pos[a] / ff;
        if (br.out_of_bounds) {
            int same = 1;
            for (int a = 0; a < 3; ++a)
                same = same && cr.hit_cell[a] == (float)sat_i32(start[a]);
            if (same) {
                for (int a = 0; a < 3; ++a) {
                    if (cr.hit_cell[a] == (float)sat_i32(start[a]))
                        start[a] = step_ulp(start[a], dir[a]);
                }
                same = 1;
                for (int a = 0; a < 3; ++a)
                    same = same && cr.hit_cell[a] == (float)sat_i32(start[a]);
                if (same) {
                    float gap[3], mag[3];
                    for (int a = 0; a < 3; ++a) {
                        gap[a] = cr.next_cell[a] - start[a];
                        mag[a] = fabsf(gap[a]);
                    }
                    if (mag[0] < mag[1] && mag[0] < mag[2])
                        start[0] += gap[0];
                    else if (mag[1] < mag[0] && mag[1] < mag[2])
                        start[1] += gap[1];
                    else
                        start[2] += gap[2];
                }
            }
        }
    }

    *out_steps = total;                                              /* :514-523 */
    if (hit) {
        for (int a = 0; a < 3; ++a)
            out_pos[a] = hit_pos[a];
        if (total == 0) {
            for (int a = 0; a < 3; ++a) {
                out_pos[a] = start[a] * ff;
                out_normal[a] = entry_normal[a];
            }
        }
    }
    return hit;
}
