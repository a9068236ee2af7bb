"""Randomised parity run (development; not part of the test suite): random worlds, random ray batches and random
render configurations through the HIP path (the product kernels -- queue and one-ray-per-lane batch kernels, the persistent
render kernel, timed and probe-counting instantiations -- and the straightforward cross-check) against the CPU oracle, for a
time budget.  One round in six uses a WIDE grid (1024 or 2048 coarse cells along x: the re-armed step counters).
Prints one summary line per round and a final tally; exits non-zero at the first mismatch after dumping the seed.

usage: fuzz_parity.py [seconds=120] [first_seed=1000]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import voxelengine_amd as vx  # noqa: E402
from oracle import vxo  # noqa: E402
from tests import helpers  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctx = vx.Context(0)
small = vx.Context(0)
small.set_persistent_waves_per_cu(1)   # a small persistent grid: batches of this size take the queue kernel
t_end = time.time() + budget
rays_checked = frames_checked = rounds = 0


def fail(what, seed, detail):
    print("MISMATCH seed=%d: %s %s" % (seed, what, detail), flush=True)
    sys.exit(1)


while time.time() < t_end:
    rng = np.random.default_rng(seed)
    factor = int(rng.choice([8, 16, 32]))
    cells = [int(rng.choice([8, 16])) for _ in range(3)]
    wide = seed % 6 == 0
    if wide:
        factor, cells = 8, [int(rng.choice([1024, 2048])), 8, 8]
    dims = tuple(c * factor for c in cells)
    kind = int(rng.integers(0, 4))
    if wide:
        kind = int(rng.choice([0, 2]))
    if kind == 0:
        w = helpers.random_voxel_world(vxo, dims, factor, float(rng.choice([0.00002, 0.0002]) if wide else rng.choice([0.0005, 0.005, 0.05, 0.4])), seed)
    else:
        gen = [vxo.GEN_HASH_HEIGHTFIELD, vxo.GEN_INT_TERRAIN, vxo.GEN_PERLIN_REF][kind - 1]
        if gen == vxo.GEN_PERLIN_REF and dims[0] * dims[1] * dims[2] > 128 ** 3:
            gen = vxo.GEN_INT_TERRAIN
        w = vxo.World.generate(gen, dims[0], dims[1], dims[2], factor)
    ctx.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
    n = int(rng.integers(20000, 120000))
    o, d = helpers.mixed_rays(w.dims, n, seed)
    if wide:   # a third of the rays nearly along the long axis: walks of hundreds to thousands of coarse cells
        d[::3, 1:] *= np.float32(0.003)
    cpu = w.trace_batch(o, d)
    small.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
    want_stats = seed % 2 == 0   # alternately the probe-counting and the timed instantiations
    for what, c, variant in (("lane", ctx, 4), ("queue", small, 4), ("direct", ctx, 1)):
        c.set_kernel_variant(variant)
        g = c.Raytrace(o, d, want_stats=want_stats)
        for k_gpu, k_cpu in (("hit", "hit"), ("steps", "steps"), ("voxel", "voxel")):
            if not np.array_equal(g[k_gpu], cpu[k_cpu]):
                bad = int(np.flatnonzero(g[k_gpu] != cpu[k_cpu])[0])
                fail("batch %s field %s" % (what, k_gpu), seed, "ray %d o=%r d=%r" % (bad, o[bad], d[bad]))
        if not np.array_equal(helpers.float_bits(g["hitPoint"]), helpers.float_bits(cpu["pos"])):
            fail("batch %s" % what, seed, "positions differ")
        if not np.array_equal(g["normal"], cpu["normal"]):
            fail("batch %s" % what, seed, "normals differ")
        if want_stats:
            st, cs = g["stats"], cpu["stats"]
            if (st.coarse_probes, st.brick_entries, st.fine_probes) != (cs.coarse_probes, cs.brick_entries, cs.fine_probes):
                fail("batch %s" % what, seed, "probe counters differ")
        c.set_kernel_variant(4)
    rays_checked += 3 * n
    # one random frame configuration, all three render kernels
    W, H = int(rng.integers(40, 400)), int(rng.integers(30, 260))
    cam = str(rng.choice(["A", "B", "C", "D"]))
    pos, f, u, r = helpers.camera(cam, w.dims, vxo)
    kw = dict(frame_number=int(rng.integers(0, 9)), mode=int(rng.integers(0, 2)), checkerboard=int(rng.integers(0, 2)),
              shadow=int(rng.integers(0, 2)), bounce_samples=int(rng.integers(0, 3)), bounce_all_hits=int(rng.integers(0, 2)),
              bounce_depth=int(rng.integers(1, 3)), ortho=int(rng.integers(0, 4) == 0))
    size = float(rng.choice([10.0, 40.0, 120.0]))
    p = vxo.make_params(W, H, pos, f, u, r, ortho_size=(size, size), **kw)
    fb0 = rng.integers(0, 255, size=(H, W, 4), dtype=np.uint8)
    want = w.render(p, fb=fb0.copy(), want_hit=True)
    ctx.SetEnvironment(list(p.env.light_dir), list(p.env.light_color), list(p.env.ambient))
    ctx.SetFOV(p.fov_deg)
    ctx.SetOrthoWindowSize(size, size)
    opts = vx.RenderOptions(mode=kw["mode"], checkerboard=bool(kw["checkerboard"]), shadow=bool(kw["shadow"]),
                            bounce_samples=kw["bounce_samples"], bounce_all_hits=bool(kw["bounce_all_hits"]),
                            bounce_depth=kw["bounce_depth"], ortho=bool(kw["ortho"]), frame_number=kw["frame_number"])
    for variant in (4, 1):
        ctx.set_kernel_variant(variant)
        for stats in (False, True):   # the timed and the probe-counting instantiation
            opts.collect_stats = stats
            ctx.frame_stats()
            d_fb = torch.from_numpy(fb0.copy()).cuda()
            d_hit = torch.full((H, W), -1, dtype=torch.int64, device="cuda")
            ctx.RenderScreen(W, H, d_fb, pos, f, u, r, opts, hit_aov=d_hit)
            if not np.array_equal(d_fb.cpu().numpy(), want["fb"]):
                fail("frame variant %d stats %d" % (variant, stats), seed, "cam %s %dx%d %r" % (cam, W, H, kw))
            if not np.array_equal(d_hit.cpu().numpy(), want["hit"]):
                fail("hit indices variant %d" % variant, seed, "cam %s %dx%d %r" % (cam, W, H, kw))
            st, cs = ctx.frame_stats(), want["stats"]
            if (st.primary_rays, st.shadow_rays, st.bounce_rays, st.primary_hits) != (cs.primary_rays, cs.shadow_rays, cs.bounce_rays, cs.primary_hits):
                fail("ray counters variant %d" % variant, seed, "cam %s %dx%d %r" % (cam, W, H, kw))
            if stats and (st.coarse_probes, st.brick_entries, st.fine_probes) != (cs.probes.coarse_probes, cs.probes.brick_entries, cs.probes.fine_probes):
                fail("probe counters variant %d" % variant, seed, "cam %s %dx%d %r" % (cam, W, H, kw))
    opts.collect_stats = False
    # the same configuration as a multi-view launch: this view plus two others, each against its own oracle frame
    ctx.set_kernel_variant(4)
    views, wants = [], []
    for j in range(3):
        cj = cam if j == 0 else str(rng.choice(["A", "B", "C", "D"]))
        pj, fj, uj, rj = helpers.camera(cj, w.dims, vxo)
        kwj = dict(kw, frame_number=kw["frame_number"] + j)
        wants.append(want["fb"] if j == 0 else w.render(vxo.make_params(W, H, pj, fj, uj, rj, ortho_size=(size, size), **kwj),
                                                         fb=fb0.copy())["fb"])
        views.append(dict(fb=torch.from_numpy(fb0.copy()).cuda(), origin=pj, fwd=fj, up=uj, right=rj,
                          frame_number=kwj["frame_number"]))
    ctx.RenderViews(W, H, views, opts)
    for j in range(3):
        if not np.array_equal(views[j]["fb"].cpu().numpy(), wants[j]):
            fail("multi-view launch, view %d" % j, seed, "cam %s %dx%d %r" % (cam, W, H, kw))
    ctx.set_kernel_variant(4)
    ctx.frame_stats()
    frames_checked += 4 + 3
    rounds += 1
    if rounds % 10 == 0:
        print("round %d (seed %d): %d rays, %d frames checked, all equal" % (rounds, seed, rays_checked, frames_checked), flush=True)
    seed += 1

print("FUZZ OK: %d rounds, %d batch rays (queue kernel, one ray per lane, straightforward loops), %d frames (persistent kernel and straightforward loops, timed and counting, + multi-view launches), 0 mismatches, seeds %s..%d" % (
    rounds, rays_checked, frames_checked, sys.argv[2] if len(sys.argv) > 2 else "1000", seed - 1), flush=True)
