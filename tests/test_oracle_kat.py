"""Hand-derived known-answer tests for the CPU oracle.

The reference holds no tests or golden vectors for this path and
cannot be built here, so the oracle's parity with it is UNPINNED; these cases were
worked out by hand from the reference source (file:line in each docstring) and pin
the restatement against transcription slips.
"""
import numpy as np

f32 = np.float32


def _world_with(vxo, voxels, size=64, factor=8):
    v = np.zeros((size, size, size), bool)
    for (x, y, z) in voxels:
        v[x, y, z] = True
    return vxo.World.from_voxels(v, factor)


def test_sample_index_tiled_linear(vxo):
    """VolumeRaytracer.cuh:107-131: tile = x/8 + y/8*tW + z/8*tW*tH; bit = tile*512 + x%8 + y%8*8 + z%8*64."""
    assert vxo.sample_index(0, 0, 0, 16, 16) == 0
    assert vxo.sample_index(7, 7, 7, 16, 16) == 511
    assert vxo.sample_index(8, 0, 0, 16, 16) == 512
    assert vxo.sample_index(0, 8, 0, 16, 16) == 1024
    assert vxo.sample_index(0, 0, 8, 16, 16) == 2048
    assert vxo.sample_index(9, 10, 11, 16, 16) == 7 * 512 + 1 + 2 * 8 + 3 * 64
    # inverse (VolumeRaytracer.cuh:138-171) round-trips on a non-cubic grid
    for idx in (0, 511, 512, 3793, 32 * 8 * 32 - 1):
        x, y, z = vxo.position_from_index(idx, 32, 8)
        assert vxo.sample_index(x, y, z, 32, 8) == idx


def test_dense_from_voxels_matches_sample_index(vxo):
    rng = np.random.default_rng(0)
    v = rng.random((16, 8, 24)) < 0.3
    words = vxo.dense_from_voxels(v)
    for _ in range(200):
        x, y, z = rng.integers(0, 16), rng.integers(0, 8), rng.integers(0, 24)
        i = vxo.sample_index(int(x), int(y), int(z), 16, 8)
        assert ((words[i >> 5] >> (i & 31)) & 1) == int(v[x, y, z])


def test_hash_known_values(vxo):
    """cuda_noise.cuh:44-54 evaluated by hand in Python integers."""
    def h(s):
        M = 0xFFFFFFFF
        s = ((s + 0x7ed55d16) + (s << 12)) & M
        s = ((s ^ 0xc761c23c) ^ (s >> 19)) & M
        s = ((s + 0x165667b1) + (s << 5)) & M
        s = ((s + 0xd3a2646c) ^ (s << 9)) & M
        s = ((s + 0xfd7046c5) + (s << 3)) & M
        s = ((s ^ 0xb55a4f09) ^ (s >> 16)) & M
        return s
    for s in (0, 1, 12345, 0xFFFFFFFF, 2073600 * 100):
        assert vxo.hash32(s) == h(s)
        assert vxo.random_float(s) == f32(f32(h(s)) / f32(4294967296.0))


def test_ray_aabb_cases(vxo):
    """VolumeRaytracer.cu:124-174."""
    hit, p, n = vxo.ray_aabb((0.5, 0.5, -1), (0, 0, 1), (0, 0, 0), (1, 1, 1))
    assert hit and p.tolist() == [0.5, 0.5, 0.0] and n.tolist() == [0, 0, 1]
    # start inside: accepted, point lies behind the start (t_enter < 0), normal = travel sign
    hit, p, n = vxo.ray_aabb((0.5, 0.5, 0.5), (1, 0, 0), (0, 0, 0), (1, 1, 1))
    assert hit and p.tolist() == [0.0, 0.5, 0.5] and n.tolist() == [1, 0, 0]
    # box behind the ray: t_exit < max(t_enter, 0)
    hit, _, _ = vxo.ray_aabb((2, 0.5, 0.5), (1, 0, 0), (0, 0, 0), (1, 1, 1))
    assert not hit
    # negative travel: normal carries the sign of the direction
    hit, p, n = vxo.ray_aabb((3, 0.5, 0.5), (-1, 0, 0), (0, 0, 0), (1, 1, 1))
    assert hit and p.tolist() == [1.0, 0.5, 0.5] and n.tolist() == [-1, 0, 0]


def test_dda_single_level_dense(vxo):
    """VolumeRaytracer.cu:176-352 without per-cell bounds (dense mode): 16^3 grid, one voxel at (5,3,3)."""
    import ctypes as C
    v = np.zeros((16, 16, 16), bool)
    v[5, 3, 3] = True
    words = vxo.dense_from_voxels(v)
    P, R = vxo.DDAParams(), vxo.DDAResult()
    P.bits = words.ctypes.data_as(C.POINTER(C.c_uint32))
    P.nbits = 4096
    P.dims = (C.c_int * 3)(16, 16, 16)
    P.start = (C.c_float * 3)(0.5, 3.5, 3.5)
    P.dir = (C.c_float * 3)(1, 0, 0)
    P.max_steps = 2048
    vxo.lib().vxo_dda(C.byref(P), C.byref(R))
    assert R.hit == 1 and R.out_of_bounds == 0
    assert R.steps == 5 and R.probes == 6
    assert list(R.point) == [5.0, 3.5, 3.5]
    assert list(R.hit_cell) == [5, 3, 3]
    assert list(R.normal) == [1, 0, 0]
    assert list(R.next_cell) == [6, 3, 3]          # advanced once more on exit (:290-322,:345-349)
    # leaving the grid: out of bounds after 16 crossings, the last one counted
    P.start = (C.c_float * 3)(0.5, 8.5, 8.5)
    vxo.lib().vxo_dda(C.byref(P), C.byref(R))
    assert R.hit == 0 and R.out_of_bounds == 1 and R.steps == 16 and R.probes == 16
    assert list(R.point) == [16.0, 8.5, 8.5]


def test_raytrace_direct_hit(vxo):
    """Two-level walk, VolumeRaytracer.cu:354-525.  64^3, f=8, voxel (21,11,11) = brick (2,1,1) local (5,3,3).
    Coarse: 2 crossings, tight box [2.625,2.75]x[1.375,1.5]^2 entered at t=2.125 -> point 2.625.
    Brick: start (5,3.5,3.5) is solid at step 0 -> normal taken from the coarse result (:496-499)."""
    w = _world_with(vxo, [(21, 11, 11)])
    assert w.nslots == 1
    ci = vxo.sample_index(2, 1, 1, 8, 8)
    assert w.bounds[ci].tolist() == [5, 3, 3, 5, 3, 3]
    assert w.brick_slot[ci] == 0 and (w.coarse_bits[ci >> 5] >> (ci & 31)) & 1
    r = w.raytrace((4.0, 11.5, 11.5), (1, 0, 0))
    assert r["hit"] and r["steps"] == 2
    assert r["pos"].tolist() == [21.0, 11.5, 11.5]
    assert r["normal"].tolist() == [1, 0, 0]
    assert r["voxel"] == (21, 11, 11)
    assert r["stats"] == (3, 1, 1)


def test_raytrace_brick_miss_then_hit(vxo):
    """Brick (2,1,1) holds (21,12,11),(21,10,11): box spans the ray but no voxel on it -> 3 brick
    crossings, exit at local x=8 (inclusive bound, :325-341), restart at coarse x=3 (:433-436).
    Second coarse DDA hits at step 0 so its point stays the start (:266-269); brick (3,1,1) voxel
    (29,11,11) reached after 5 crossings."""
    w = _world_with(vxo, [(21, 12, 11), (21, 10, 11), (29, 11, 11)])
    ci = vxo.sample_index(2, 1, 1, 8, 8)
    assert w.bounds[ci].tolist() == [5, 2, 3, 5, 4, 3]
    r = w.raytrace((4.0, 11.5, 11.5), (1, 0, 0))
    assert r["hit"] and r["steps"] == 10
    assert r["pos"].tolist() == [29.0, 11.5, 11.5]
    assert r["normal"].tolist() == [1, 0, 0]
    assert r["voxel"] == (29, 11, 11)
    assert r["stats"] == (4, 2, 9)


def test_raytrace_negative_direction_nudge(vxo):
    """Travelling -x the brick is left at local x=0, so the restart point truncates to the same
    coarse cell and all three components are moved one ulp along the ray (:438-461; a zero
    direction component moves toward +inf)."""
    w = _world_with(vxo, [(26, 12, 11), (26, 10, 11), (18, 11, 11)])
    r = w.raytrace((60.0, 11.5, 11.5), (-1, 0, 0))
    up = np.nextafter(f32(1.4375), f32(np.inf))
    y_local = f32(f32(up * f32(8)) - f32(8))
    y_world = f32(y_local + f32(8))
    assert r["hit"] and r["steps"] == 13
    assert r["pos"].tolist() == [19.0, float(y_world), float(y_world)]
    assert r["normal"].tolist() == [-1, 0, 0]
    assert r["voxel"] == (18, 11, 11)
    assert r["stats"] == (6, 2, 10)


def test_raytrace_from_outside_zero_steps(vxo):
    """Start outside the grid: moved to the slab entry of [1e-6, C-1e-6]^3 (:369-381).  A voxel
    right at the entry gives total_steps == 0 -> position = start*f, normal = entry normal (:518-522)."""
    w = _world_with(vxo, [(0, 11, 11)])
    r = w.raytrace((-10.0, 11.5, 11.5), (1, 0, 0))
    t = f32(f32(1e-6) - f32(-1.25))
    sx = f32(f32(-1.25) + f32(t * f32(1)))
    assert r["hit"] and r["steps"] == 0
    assert r["pos"].tolist() == [float(f32(sx * f32(8))), 11.5, 11.5]
    assert r["normal"].tolist() == [1, 0, 0]
    assert r["voxel"] == (0, 11, 11)


def test_raytrace_miss_conventions(vxo):
    w = _world_with(vxo, [(21, 11, 11)])
    r = w.raytrace((4.0, 40.5, 40.5), (1, 0, 0))
    assert not r["hit"] and r["normal"].tolist() == [0, 0, 0]
    assert r["steps"] == 8      # 8 coarse crossings incl. the one onto x=8 (counted, then out of range)
    b = w.trace_batch([(4.0, 40.5, 40.5), (4.0, 11.5, 11.5)], [(1, 0, 0), (1, 0, 0)])
    assert np.isinf(b["pos"][0]).all() and b["hit"].tolist() == [0, 1]    # dispatch, VolumeRaytracer.cu:105-113
    assert b["voxel"].tolist() == [-1, 21 + 64 * (11 + 64 * 11)]
    # a ray that never touches the grid box
    r = w.raytrace((-10.0, 100.0, 11.5), (1, 0, 0))
    assert not r["hit"] and r["steps"] == 0


def test_brickmap_builder_tables(vxo):
    """GenerateLowresVoxelBuffer, VolumeRaytracer.cuh:379-516: brick bits, inclusive extents, empty = 0/-1."""
    rng = np.random.default_rng(3)
    v = rng.random((64, 64, 128)) < 0.002
    v[:, 40:, :] = False
    w = vxo.World.from_voxels(v, 8)
    assert w.cdims == (8, 8, 16) and w.ncells == 1024
    slots_seen = 0
    for cz in range(16):
        for cy in range(8):
            for cx in range(8):
                ci = vxo.sample_index(cx, cy, cz, 8, 8)
                blk = v[cx * 8:cx * 8 + 8, cy * 8:cy * 8 + 8, cz * 8:cz * 8 + 8]
                bit = (w.coarse_bits[ci >> 5] >> (ci & 31)) & 1
                assert bit == int(blk.any())
                if blk.any():
                    xs, ys, zs = np.nonzero(blk)
                    assert w.bounds[ci].tolist() == [xs.min(), ys.min(), zs.min(), xs.max(), ys.max(), zs.max()]
                    s = int(w.brick_slot[ci])
                    words = w.pool[s * 16:(s + 1) * 16]
                    assert np.array_equal(words, vxo.dense_from_voxels(blk))
                    slots_seen += 1
                else:
                    assert w.bounds[ci].tolist() == [0, 0, 0, -1, -1, -1]
                    assert w.brick_slot[ci] == vxo.EMPTY_SLOT
    assert slots_seen == w.nslots
    # slots are handed out in coarse tiled-index order
    used = w.brick_slot[w.brick_slot != vxo.EMPTY_SLOT]
    assert np.array_equal(used, np.arange(w.nslots, dtype=np.uint32))


def test_generator_direct_equals_dense_path(vxo):
    for g in (vxo.GEN_HASH_HEIGHTFIELD, vxo.GEN_INT_TERRAIN):
        a = vxo.World.generate(g, 128, 128, 128, 16)
        ptr = vxo.lib().vxo_gen_dense(g, 128, 128, 128, 4)
        import ctypes as C
        dense = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), (128 * 128 * 128 // 32,)).copy()
        vxo.lib().free(ptr)
        b = vxo.World.from_dense(dense, 128, 128, 128, 16)
        for name in ("coarse_bits", "brick_slot", "bounds", "pool"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), name
        assert 0 < a.nslots < a.ncells


def test_perlin_ref_generator_is_deterministic_and_sane(vxo):
    # y=0 layer is always solid (t clamped to >= 0, VoxelWorldBuilder.cu:25-33)
    L = vxo.lib()
    assert all(L.vxo_gen_solid(vxo.GEN_PERLIN_REF, x, 0, z, 64, 64, 64) == 1 for x in (0, 5, 63) for z in (0, 9))
    a = [L.vxo_fbm_perlin(f32(0.005 * x), f32(0.01), f32(0.02)) for x in range(0, 2000, 37)]
    b = [L.vxo_fbm_perlin(f32(0.005 * x), f32(0.01), f32(0.02)) for x in range(0, 2000, 37)]
    assert a == b and max(a) < 3 and min(a) > -3 and len(set(a)) > 10
