"""BASELINE configs[0] / SURVEY 8d config 1: the reference's CPU plumbing case -- ONE batch of 1 000 000 rays
(RAYS, DDATestCpp/DDATestCpp.cpp:21) fanned out from a single origin (the 2-D tester's angular fan, :443-448, taken
to 3-D as a Fibonacci sphere) through a 128^3 grid: the build's integer HASH_HEIGHTFIELD world, brick edge 8.

CPU half: the oracle reproduces the committed fixture, is a pure function of its inputs (thread count), and its
two-level brickmap trace finds the same voxel as the single-level DDA through the whole grid as ONE dense buffer
(per_voxel_bounds == nullptr, the 2-D tester's dense mode, DDATestCpp.cpp:83-90 / VolumeRaytracer.cu:276-280).
GPU half: the batch API (VoxelRaytracer3D::Raytrace, VolumeRaytracer.cu:574-618) on the same million rays."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
G = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(G)


@pytest.fixture(scope="module")
def case(vxo):
    w, o, d = G.config0_case()
    g = np.load(os.path.join(HERE, "golden", "config0_fan.npz"))
    assert str(g["inputs"]) == G.digest(o, d, w.coarse_bits, w.pool), "the seeded inputs changed"
    return w, o, d, g


def _check_against_fixture(r, g, pos_key):
    sub = G.CONFIG0["subset"]
    assert int(r["hit"].sum()) == int(g["hits_total"]) and int(r["steps"].astype(np.int64).sum()) == int(g["steps_total"])
    assert np.array_equal(r["hit"][sub], g["hit"]) and np.array_equal(r["steps"][sub], g["steps"])
    assert np.array_equal(r["voxel"][sub], g["voxel"])
    assert np.array_equal(r[pos_key][sub].view(np.uint32), g["pos_bits"])
    assert np.array_equal(r["normal"][sub].astype(np.int8), g["normal"])
    assert G.batch_digest(r, pos_key) == str(g["outputs"])


def test_oracle_million_ray_batch(vxo, case):
    w, o, d, g = case
    r = w.trace_batch(o, d)
    assert len(r["hit"]) == 1_000_000
    assert 0.1 < r["hit"].mean() < 0.9            # the fan sees terrain below and sky above
    _check_against_fixture(r, g, "pos")
    r1 = w.trace_batch(o, d, nthreads=1)          # Raytrace is a pure function of the ray: no thread-count dependence
    assert G.batch_digest(r1) == str(g["outputs"])


def test_two_level_equals_single_level_dense(vxo, case):
    """The same rays through the same voxels as ONE 128^3 buffer, no brickmap: same hit voxel, same hit point."""
    w, o, d, g = case
    c = G.CONFIG0
    X, Y, Z = c["dims"]
    vox = np.zeros((X, Y, Z), bool)               # the generator restated: h = 3Y/16 + hash(x>>3, z>>3, 1) % (3Y/8)
    for cx in range(X // 8):
        for cz in range(Z // 8):
            h = 3 * Y // 16 + vxo.hash32(((cx * 73856093) & 0xFFFFFFFF) ^ ((cz * 19349663) & 0xFFFFFFFF) ^ 1) % (3 * Y // 8)
            vox[cx * 8:cx * 8 + 8, :h, cz * 8:cz * 8 + 8] = True
    words = vxo.dense_from_voxels(vox)
    w2 = vxo.World.from_dense(words, X, Y, Z, c["factor"])   # GenerateLowresVoxelBuffer on the dense image
    assert np.array_equal(w.coarse_bits, w2.coarse_bits) and np.array_equal(w.pool, w2.pool)
    assert np.array_equal(w.bounds, w2.bounds) and np.array_equal(w.brick_slot, w2.brick_slot)
    sub = np.arange(1_000_000)[c["subset"]]
    P, R = vxo.DDAParams(), vxo.DDAResult()
    P.bits = words.ctypes.data_as(C.POINTER(C.c_uint32))
    P.nbits = X * Y * Z
    P.dims = (C.c_int * 3)(X, Y, Z)
    P.max_steps = 2048
    hits = 0
    for k, i in enumerate(sub):
        v = d[i]
        n = (v * (np.float32(1.0) / np.sqrt(np.float32(v @ v)))).astype(np.float32)   # Raytrace normalises (:367)
        P.start = (C.c_float * 3)(*c["origin"])
        P.dir = (C.c_float * 3)(*n)
        vxo.lib().vxo_dda(C.byref(P), C.byref(R))
        assert bool(R.hit) == bool(g["hit"][k]), i
        if R.hit:
            hits += 1
            vx_ = int(g["voxel"][k])
            assert tuple(R.hit_cell) == (vx_ % X, (vx_ // X) % Y, vx_ // (X * Y)), i
            assert np.allclose(np.array(list(R.point), np.float32), g["pos_bits"][k].view(np.float32), atol=1e-3), i
    assert hits == int(g["hit"].sum()) > 500


@pytest.mark.gpu
def test_hip_million_ray_batch(vxo, case):
    import torch
    import voxelengine_amd as vx
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    w, o, d, g = case
    ctx = vx.Context(0)
    try:
        ctx.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
        for variant in (4, 1):
            ctx.set_kernel_variant(variant)
            _check_against_fixture(ctx.Raytrace(o, d), g, "hitPoint")
        # the same world built on the device
        ctx.set_kernel_variant(4)
        c = G.CONFIG0
        ctx.build_world(c["gen"], c["dims"][0], c["dims"][1], c["dims"][2], c["factor"])
        _check_against_fixture(ctx.Raytrace(o, d), g, "hitPoint")
    finally:
        ctx.close()
