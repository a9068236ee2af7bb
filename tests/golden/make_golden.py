"""Generates tests/golden/*.npz -- small input/output vectors of the path, produced by THIS repository's CPU
oracle (oracle/, a restatement of the reference algorithm).

What they pin: the oracle's behaviour (and with it the HIP path's) against drift between rounds -- any change to
either must reproduce these vectors bit for bit.  What they do NOT pin: parity with the reference itself (the
reference's CUDA sources cannot be built here and hold no fixtures of their own).

    python tests/golden/make_golden.py        # rewrites the fixtures (only when a definition deliberately changes)

Fixtures (inputs are regenerated from seeds by the tests; the files hold the expected outputs plus a digest of the
inputs so that a change of the input builders cannot pass unnoticed):
  trace_terrain.npz   6000 mixed rays (tests/helpers.mixed_rays) through INT_TERRAIN 128^3, f = 16
  trace_sparse.npz    6000 mixed rays through a random 2 % voxel world 64^3, f = 8
  frame_shaded.npz    160x96 frame: camera A, shadow ray, 2 bounce samples, reference gate, FrameNumber 5
  frame_debug.npz     160x96 frame: camera D, DEBUG_VIEW + checkerboard on stale contents, FrameNumber 2
  frame_second_bounce.npz  144x80 frame: camera B, bounce_depth 2 (this build's extension), all-hits gate
  worlds.npz          SHA-256 of the four brickmap tables of the three generators on small grids
  config0_fan.npz     BASELINE configs[0]: ONE batch of 1 000 000 rays, Fibonacci-sphere fan from
                      (64, 100, 64) through the 128^3 HASH_HEIGHTFIELD world, f = 8: digest of all outputs + the
                      outputs of a 4096-ray subset
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import vxo  # noqa: E402
from tests import helpers  # noqa: E402


def digest(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def world_terrain():
    return vxo.World.generate(vxo.GEN_INT_TERRAIN, 128, 128, 128, 16)


def world_sparse():
    return helpers.random_voxel_world(vxo, (64, 64, 64), 8, 0.02, 77)


TRACES = {"trace_terrain": (world_terrain, 6000, 41), "trace_sparse": (world_sparse, 6000, 42)}

FRAMES = {
    "frame_shaded": (world_terrain, 160, 96, "A", dict(frame_number=5, shadow=1, bounce_samples=2)),
    "frame_debug": (world_terrain, 160, 96, "D", dict(frame_number=2, mode=1, checkerboard=1)),
    "frame_second_bounce": (world_terrain, 144, 80, "B", dict(frame_number=9, shadow=1, bounce_samples=1, bounce_all_hits=1,
                                                              bounce_depth=2)),
}

WORLDS = {
    "hash_heightfield_128_f16": (vxo.GEN_HASH_HEIGHTFIELD, (128, 128, 128), 16),
    "int_terrain_128x64x128_f8": (vxo.GEN_INT_TERRAIN, (128, 64, 128), 8),
    "perlin_ref_64_f8": (vxo.GEN_PERLIN_REF, (64, 64, 64), 8),
}


CONFIG0 = dict(gen=vxo.GEN_HASH_HEIGHTFIELD, dims=(128, 128, 128), factor=8, rays=1_000_000, origin=(64.0, 100.0, 64.0),
               subset=slice(0, 1_000_000, 244))


def config0_case():
    c = CONFIG0
    w = vxo.World.generate(c["gen"], c["dims"][0], c["dims"][1], c["dims"][2], c["factor"])
    o, d = helpers.fibonacci_fan(c["rays"], c["origin"])
    return w, o, d


def batch_digest(r, pos_key="pos") -> str:
    return digest(r["hit"].astype(np.uint8), r["steps"].astype(np.int32), r["voxel"].astype(np.int64),
                  r[pos_key].view(np.uint32), r["normal"].astype(np.int8))


def trace_case(name):
    make, n, seed = TRACES[name]
    w = make()
    o, d = helpers.mixed_rays(w.dims, n, seed)
    return w, o, d


def frame_case(name):
    make, W, H, cam, kw = FRAMES[name]
    w = make()
    pos, f, u, r = helpers.camera(cam, w.dims, vxo)
    stale = np.random.default_rng(11).integers(0, 255, size=(H, W, 4), dtype=np.uint8)
    return w, W, H, (pos, f, u, r), kw, stale


def world_tables(name):
    gen, dims, f = WORLDS[name]
    w = vxo.World.generate(gen, dims[0], dims[1], dims[2], f)
    return w


def main():
    for name in TRACES:
        w, o, d = trace_case(name)
        r = w.trace_batch(o, d)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), inputs=digest(o, d, w.coarse_bits, w.pool), hit=r["hit"],
                            steps=r["steps"], voxel=r["voxel"], pos_bits=r["pos"].view(np.uint32),
                            normal=r["normal"].astype(np.int8))
    for name in FRAMES:
        w, W, H, (pos, f, u, r), kw, stale = frame_case(name)
        out = w.render(vxo.make_params(W, H, pos, f, u, r, **kw), fb=stale.copy(), want_hit=True)
        st = out["stats"]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), inputs=digest(stale, w.coarse_bits, w.pool), fb=out["fb"],
                            hit=out["hit"], rays=np.array([st.primary_rays, st.shadow_rays, st.bounce_rays, st.primary_hits],
                                                          np.int64))
    w, o, d = config0_case()
    r = w.trace_batch(o, d)
    sub = CONFIG0["subset"]
    np.savez_compressed(os.path.join(HERE, "config0_fan.npz"), inputs=digest(o, d, w.coarse_bits, w.pool), outputs=batch_digest(r),
                        hit=r["hit"][sub], steps=r["steps"][sub], voxel=r["voxel"][sub], pos_bits=r["pos"][sub].view(np.uint32),
                        normal=r["normal"][sub].astype(np.int8), hits_total=np.int64(r["hit"].sum()),
                        steps_total=np.int64(r["steps"].astype(np.int64).sum()))
    sums = {}
    for name in WORLDS:
        w = world_tables(name)
        sums[name] = digest(w.coarse_bits, w.brick_slot, w.bounds, w.pool)
    np.savez_compressed(os.path.join(HERE, "worlds.npz"), **sums)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-28s %7d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
