"""A second, independent restatement of the traversal in pure Python + numpy binary32 scalars, with EVENT TRACING.

TEST INFRASTRUCTURE ONLY (like the rest of oracle/): imported by tests/ alone.  It exists for two reasons:
  * two separately written restatements (this one, and the C one in vxo_trace.c) agreeing ray for ray is a stronger pin
    than either alone -- the reference holds no vectors and cannot be built here (parity with it: UNPINNED);
  * it records WHICH of the reference's quirks a ray exercised (edge padding, exact ties, region check, previous_cell
    break, ulp nudge, NextCell snap and its branch), so that the known-answer tests can prove that their cases do
    reach the lines they are named for.
Written from the reference's text, expression by expression; every function cites the lines it follows.  Slow
(a few thousand rays per second): small cases only.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
FLT_EPS = f32(1.1920928955078125e-7)   # VolumeRaytracer.cuh:22
FLT_EPS_DDA = 1e-6                     # a double in the reference (VolumeRaytracer.cuh:20)
INF = f32(np.inf)
MAX_STEPS = 2048                       # VolumeRaytracer.cuh:235


def f2i(v) -> int:
    """static_cast<int>(float) as the GPU does it (v_cvt_i32_f32): truncation, saturating, NaN -> 0.  (C leaves the
    out-of-range cases undefined; this build's definition is in DESIGN.md section 2.)"""
    v = float(v)
    if v != v:
        return 0
    if v >= 2147483648.0:
        return 2147483647
    if v <= -2147483648.0:
        return -2147483648
    return int(v)


def lo(a, b):   # fminf as helper_math.h:56-59 defines it on the host: a < b ? a : b
    return a if a < b else b


def hi(a, b):   # fmaxf, helper_math.h:61-64
    return a if a > b else b


def sample_index(x, y, z, w, h) -> int:
    """GetSampleIndex, VolumeRaytracer.cuh:107-131 (tiled-linear)."""
    tw, th = w // 8, h // 8
    return ((x // 8) + (y // 8) * tw + (z // 8) * tw * th) * 512 + (x % 8) + (y % 8) * 8 + (z % 8) * 64


def bit(words, idx) -> int:
    """BitArray::operator[] const, VolumeRaytracer.cu:61-68: false beyond the end."""
    if (idx >> 5) >= len(words):
        return 0
    return (int(words[idx >> 5]) >> (idx & 31)) & 1


def ray_aabb(start, d, bmin, bmax):
    """RayIntersectsAABB, VolumeRaytracer.cu:124-174.  Returns (hit, point, normal)."""
    with np.errstate(all="ignore"):
        inv = [f32(1.0) / (FLT_EPS if d[a] == 0 else d[a]) for a in range(3)]
        t1, t2 = [], []
        for a in range(3):
            ta = f32(f32(bmin[a] - start[a]) * inv[a])
            tb = f32(f32(bmax[a] - start[a]) * inv[a])
            t1.append(lo(ta, tb))
            t2.append(hi(ta, tb))
        t_min = hi(hi(t1[0], t1[1]), t1[2])
        t_max = lo(lo(t2[0], t2[1]), t2[2])
        if t_max < hi(t_min, f32(0.0)):
            return False, None, None
        p = [f32(start[a] + f32(t_min * d[a])) for a in range(3)]
    if t_min == t1[0]:
        n = [f32(-1.0 if inv[0] < 0 else 1.0), f32(0), f32(0)]
    elif t_min == t1[1]:
        n = [f32(0), f32(-1.0 if inv[1] < 0 else 1.0), f32(0)]
    else:
        n = [f32(0), f32(0), f32(-1.0 if inv[2] < 0 else 1.0)]
    return True, p, n


class Walk:
    """DDARayResults (VolumeRaytracer.cuh:262-270) + probe count."""
    def __init__(self):
        self.hit = False
        self.oob = False
        self.hit_cell = [f32(0)] * 3
        self.point = None
        self.next_cell = [f32(0)] * 3
        self.normal = [f32(0)] * 3
        self.steps = 0
        self.probes = 0


def dda(words, dims, start, d, events, per_cell_bounds=None, scale=0, bounds=None, max_steps=MAX_STEPS, level=""):
    """DDARayTraversal, VolumeRaytracer.cu:176-352.  `per_cell_bounds`: (ncells, 6) float array or None; `bounds`:
    (min3, max3) floats or None.  Appends the quirks met to `events`."""
    cols, rows, depth = dims
    x, y, z = start
    dx, dy, dz = d
    cell = [f2i(x), f2i(y), f2i(z)]
    step = [1 if dx > 0 else -1, 1 if dy > 0 else -1, 1 if dz > 0 else -1]
    with np.errstate(all="ignore"):
        t_delta = [f32(abs(f32(1.0) / c)) if c != 0 else INF for c in (dx, dy, dz)]                      # :199-201
        t_max = [f32(f32(f32(cell[a] + (1 if step[a] > 0 else 0)) - start[a]) / d[a]) if d[a] != 0 else INF
                 for a in range(3)]                                                                        # :203-205
    R = Walk()
    R.point = [x, y, z]
    pad = [0, 0, 0]
    if cell[0] == cols or cell[1] == rows or cell[2] == depth:                                             # :216-232
        pad = [1 if c < 0 else 0 for c in (dx, dy, dz)]
        events.append(level + "edge")
        if any(pad):
            events.append(level + "edge_pad")
    leaving = False
    for it in range(max_steps):
        if (0 <= cell[0] < cols + pad[0]) and (0 <= cell[1] < rows + pad[1]) and (0 <= cell[2] < depth + pad[2]):
            q = [min(max(cell[0], 0), cols - 1), min(max(cell[1], 0), rows - 1), min(max(cell[2], 0), depth - 1)]
            if q != cell:
                events.append(level + "clamped_lookup")
            R.hit_cell = [f32(v) for v in q]
            idx = sample_index(q[0], q[1], q[2], cols, rows)
            R.probes += 1
            if per_cell_bounds is not None:                                                                # :248-273
                b = per_cell_bounds[idx]
                sc = f32(scale)
                bmin = [f32(f32(f32(b[a]) + f32(0)) / sc + f32(q[a])) for a in range(3)]
                bmax = [f32(f32(f32(b[3 + a]) + f32(1)) / sc + f32(q[a])) for a in range(3)]
                if bit(words, idx) == 1 and bmin[0] <= bmax[0]:
                    h, p, n = ray_aabb(start, d, bmin, bmax)
                    if h:
                        R.hit = True
                        R.normal = n
                        if it != 0:
                            R.point = p
                        else:
                            events.append(level + "box_hit_at_step0")
                        leaving = True
            elif bit(words, idx) == 1:                                                                     # :276-280
                R.hit = True
                leaving = True
        else:
            R.oob = True
            leaving = True
        with np.errstate(all="ignore"):
            finite = [v for v in t_max if v != INF]
            if len(set(float(v) for v in finite)) < len(finite):
                events.append(level + "tie")
            if t_max[0] < t_max[1] and t_max[0] < t_max[2]:                                                # :293-322
                a = 0
            elif t_max[1] <= t_max[0] and t_max[1] < t_max[2]:
                a = 1
            else:
                a = 2
            t = t_max[a]
            cross = [f32(start[c] + f32(t * d[c])) for c in range(3)]
            cross[a] = f32(cell[a] + (1 if step[a] > 0 else 0))
            cell[a] += step[a]
            t_max[a] = f32(t_max[a] + t_delta[a])
        if leaving:
            R.next_cell = [f32(c) for c in cell]
            break
        R.normal = [f32(step[c]) if c == a else f32(0) for c in range(3)]
        if bounds is not None:                                                                             # :325-341
            mn = [f2i(v) for v in bounds[0]]
            mx = [f2i(v) for v in bounds[1]]
            if any(cross[c] < mn[c] or cross[c] > mx[c] for c in range(3)):
                R.oob = True
                events.append(level + "region_oob")
                break
        R.steps += 1
        R.point = cross
    return R


class PyWorld:
    """The reference-shaped tables of a brickmap (coarse bits, per-cell brick slot, per-cell bounds, pool of bricks)."""
    def __init__(self, factor, cdims, coarse_bits, brick_slot, bounds, pool):
        self.f = int(factor)
        self.cdims = tuple(int(c) for c in cdims)
        self.coarse_bits = np.asarray(coarse_bits, np.uint32)
        self.brick_slot = np.asarray(brick_slot, np.uint32)
        self.bounds = np.asarray(bounds, np.float32).reshape(-1, 6)
        self.pool = np.asarray(pool, np.uint32)
        self.bw = self.f ** 3 // 32


def raytrace(W: PyWorld, origin, ray, max_steps=MAX_STEPS):
    """Raytrace, VolumeRaytracer.cu:354-525.  Returns dict(hit, steps, pos, normal, voxel, stats, events)."""
    events = []
    f = f32(W.f)
    origin = [f32(v) for v in origin]
    ray = [f32(v) for v in ray]
    start = [f32(origin[a] / f) for a in range(3)]                                                         # :362-365
    with np.errstate(all="ignore"):
        dot = f32(f32(f32(ray[0] * ray[0]) + f32(ray[1] * ray[1])) + f32(ray[2] * ray[2]))
        inv_len = f32(f32(1.0) / np.sqrt(dot, dtype=f32))                                                  # normalize, helper_math.h:1325
        d = [f32(ray[a] * inv_len) for a in range(3)]
    start_normal = [f32(0)] * 3
    cd = W.cdims
    if not (start[0] >= 0 and start[1] >= 0 and start[2] >= 0 and start[0] < cd[0] and start[1] < cd[1] and start[2] < cd[2]):
        e = f32(FLT_EPS_DDA)                                                                               # :373-376
        far = [f32(np.float64(c) - FLT_EPS_DDA) for c in cd]   # int - double, rounded to float ONCE
        h, p, n = ray_aabb(start, d, [e, e, e], far)
        events.append("outside_start")
        if h:
            start = p
            start_normal = n
            events.append("world_entry")
    out_normal = [f32(0)] * 3
    hit_pos = [f32(0)] * 3
    hit = False
    total = 0
    previous = [f32(-1)] * 3
    voxel = None
    probes = [0, 0, 0]
    while total < max_steps:                                                                               # :386
        c = dda(W.coarse_bits, cd, start, d, events, per_cell_bounds=W.bounds, scale=W.f, level="c:")
        probes[0] += c.probes
        total += c.steps
        hit_pos = [f32(c.point[a] * f) for a in range(3)]
        if not (c.hit and not c.oob):
            break
        if previous == c.hit_cell:                                                                         # :402-407
            events.append("previous_cell_break")
            break
        previous = list(c.hit_cell)
        local = [f32(hit_pos[a] - f32(c.hit_cell[a] * f)) for a in range(3)]
        hc = [int(v) for v in c.hit_cell]
        ci = sample_index(hc[0], hc[1], hc[2], cd[0], cd[1])
        slot = int(W.brick_slot[ci])
        probes[1] += 1
        if slot == 0xFFFFFFFF:
            words, bdims = np.zeros(0, np.uint32), (0, 0, 0)
        else:
            words, bdims = W.pool[slot * W.bw:(slot + 1) * W.bw], (W.f, W.f, W.f)
        b = dda(words, bdims, local, d, events, bounds=([f32(0)] * 3, [f] * 3), level="b:")               # :421-424
        probes[2] += b.probes
        total += b.steps
        hit_pos = [f32(b.point[a] + f32(c.hit_cell[a] * f)) for a in range(3)]
        if b.hit:
            out_normal = c.normal if b.steps == 0 else b.normal                                            # :496-503
            if b.steps == 0:
                events.append("normal_from_coarse")
            voxel = tuple(hc[a] * W.f + int(b.hit_cell[a]) for a in range(3))
            hit = True
            break
        start = [f32(hit_pos[a] / f) for a in range(3)]
        if b.oob:
            same = all(c.hit_cell[a] == f32(f2i(start[a])) for a in range(3))
            if same:                                                                                       # :449-461
                events.append("ulp_nudge")
                start = [np.nextafter(start[a], -INF if d[a] < 0 else INF) for a in range(3)]
                same = all(c.hit_cell[a] == f32(f2i(start[a])) for a in range(3))
                if same:                                                                                   # :470-487
                    diff = [f32(c.next_cell[a] - start[a]) for a in range(3)]
                    ad = [f32(abs(v)) for v in diff]
                    if ad[0] < ad[1] and ad[0] < ad[2]:
                        start[0] = f32(start[0] + diff[0])
                        events.append("snap_x")
                    elif ad[1] < ad[0] and ad[1] < ad[2]:
                        start[1] = f32(start[1] + diff[1])
                        events.append("snap_y")
                    else:
                        start[2] = f32(start[2] + diff[2])
                        events.append("snap_z")
    pos = None
    if hit:
        pos = hit_pos
        if total == 0:                                                                                     # :518-522
            pos = [f32(start[a] * f) for a in range(3)]
            out_normal = start_normal
            events.append("zero_steps")
    return dict(hit=hit, steps=total, pos=pos, normal=out_normal, voxel=voxel, stats=tuple(probes), events=events)
