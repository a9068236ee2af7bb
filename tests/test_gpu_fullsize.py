"""Parity at BASELINE.json's full sizes (8192x512x8192 brickmap, 1920x1080, primary + shadow + 1 bounce), where
the oracle cannot render whole frames in test time: size-independent properties plus oracle spot checks.
  * a random sample of the frame's primary rays: HIP batch trace == oracle (bit-exact), on the downloaded bricks
  * the two entry points agree: hit voxel AOV of vxrt_render == vxrt_trace_batch of the same camera rays
  * idempotence: the same frame rendered twice is byte-identical
  * oracle-rendered bands of the shaded 1080p frame, every render kernel of the product and a 16-view launch
  * the separately written kernels (variants 1, 2, 5, 6) produce the same frame
  * 8 interleaved strip shards reassemble into the single-GPU frame
  * ray accounting: shadow rays == primary hits; rays <= 3 * pixels
"""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

X, Y, Z, F = 8192, 512, 8192, 32
W, H = 1920, 1080
f32 = np.float32


@pytest.fixture(scope="module")
def big():
    import torch
    import voxelengine_amd as vx
    ctx = vx.Context(0)
    info = ctx.build_world(vx.GEN_PERLIN_REF, X, Y, Z, F)
    inv = float(f32(1.0) / np.sqrt(f32(3.0), dtype=f32))
    ctx.SetEnvironment((inv, inv, inv), (2, 2, 2), (0.5, 0.5, 0.5))
    ctx.SetFOV(90.0)
    yield vx, ctx, torch, info
    ctx.close()


def _camera_rays(vx, name):
    """getRayDirection (Renderer.cu:44-59) in numpy float32, op for op (the kernel normalises again inside the
    trace, so un-normalised directions are equivalent inputs only if built identically: we pass the normalised one)."""
    (fx, fy, fz), euler = helpers.CAMERAS[name]
    pos = np.array((fx * X, fy * Y, fz * Z), f32)
    fwd, up, right = vx.GetDirections(euler)
    aspect = f32(W) / f32(H)
    fov = f32(np.float64(f32(90.0)) * 3.1415 / 180.0)
    import math
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.tanf.restype = ctypes.c_float
    libm.tanf.argtypes = [ctypes.c_float]
    t = f32(libm.tanf(ctypes.c_float(float(fov / f32(2.0)))))
    kx, ky = f32(t * aspect), t
    xs = (np.arange(W, dtype=f32) / f32(W)) * f32(2) - f32(1)
    ys = (np.arange(H, dtype=f32) / f32(H)) * f32(2) - f32(1)
    d = np.empty((H, W, 3), f32)
    for a in range(3):
        d[..., a] = (fwd[a] + (xs[None, :] * kx) * right[a]) + (ys[:, None] * ky) * up[a]
    n = f32(1.0) / np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2], dtype=f32)
    d = d * n[..., None]
    return pos, fwd, up, right, d


@pytest.mark.parametrize("cam", ["A", "B", "D"])
def test_render_and_batch_entry_points_agree(big, cam):
    vx, ctx, torch, _ = big
    pos, fwd, up, right, d = _camera_rays(vx, cam)
    fb = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    hit = torch.full((H, W), -1, dtype=torch.int64, device="cuda")
    ctx.frame_stats()
    ctx.RenderScreen(W, H, fb, pos, fwd, up, right, vx.RenderOptions(shadow=True, bounce_samples=1, frame_number=7),
                     hit_aov=hit)
    st = ctx.frame_stats()
    assert st.primary_rays == W * H and st.shadow_rays == st.primary_hits and st.total_rays() <= 3 * W * H
    o = np.broadcast_to(pos, d.shape).reshape(-1, 3)
    b = ctx.Raytrace(o, d.reshape(-1, 3))
    assert np.array_equal(b["voxel"].reshape(H, W), hit.cpu().numpy())
    assert int(b["hit"].sum()) == st.primary_hits


def test_shaded_1080p_frame_bands_against_the_oracle(big, vxo):
    """BASELINE configs[2] itself -- the bench workload: 1080p, primary + shadow + 1 bounce on the 8192x512x8192 world --
    against oracle-rendered BANDS of the shaded frame (sky / horizon rows, terrain rows, the last rows), for two cameras,
    through every render kernel of the product and through a 16-view launch (bench.py's step)."""
    vx, ctx, torch, info = big
    w = ctx.download_world()
    world = vxo.World.wrap(w["factor"], w["cdims"], w["coarse_bits"], w["brick_slot"], w["bounds"], w["pool"])
    default = ctx.kernel_variant
    bands = ((0, 16), (520, 552), (H - 16, H))
    try:
        for cam, fn in (("A", 3), ("D", 8)):
            pos, fwd, up, right, _ = _camera_rays(vx, cam)
            refs = []
            for r0, r1 in bands:
                refs.append(world.render(vxo.make_params(W, H, pos, fwd, up, right, frame_number=fn, shadow=1, bounce_samples=1,
                                                         row_begin=r0, row_end=r1), fb=np.zeros((H, W, 4), np.uint8),
                                         want_hit=True, nthreads=16))
            for variant in (4, 1):   # the product kernel, the straightforward loops
                ctx.set_kernel_variant(variant)
                fb = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
                hit = torch.full((H, W), -1, dtype=torch.int64, device="cuda")
                ctx.RenderScreen(W, H, fb, pos, fwd, up, right, vx.RenderOptions(shadow=True, bounce_samples=1, frame_number=fn),
                                 hit_aov=hit)
                got, ghit = fb.cpu().numpy(), hit.cpu().numpy()
                for (r0, r1), ref in zip(bands, refs):
                    assert np.array_equal(got[r0:r1], ref["fb"][r0:r1]), (cam, variant, r0)
                    assert np.array_equal(ghit[r0:r1], ref["hit"][r0:r1]), (cam, variant, r0)
            # the same view as one of 16 in a multi-view launch
            for variant in (4,):
                ctx.set_kernel_variant(variant)
                fbs = torch.zeros((16, H, W, 4), dtype=torch.uint8, device="cuda")
                views = [dict(fb=fbs[j], origin=pos, fwd=fwd, up=up, right=right, frame_number=fn + (j != 5)) for j in range(16)]
                ctx.RenderViews(W, H, views, vx.RenderOptions(shadow=True, bounce_samples=1))
                got = fbs[5].cpu().numpy()
                for (r0, r1), ref in zip(bands, refs):
                    assert np.array_equal(got[r0:r1], ref["fb"][r0:r1]), (cam, variant, "multi-view", r0)
    finally:
        ctx.set_kernel_variant(default)
        ctx.frame_stats()


def test_sampled_rays_against_oracle_on_the_full_world(big, vxo):
    vx, ctx, torch, info = big
    w = ctx.download_world()
    world = vxo.World.wrap(w["factor"], w["cdims"], w["coarse_bits"], w["brick_slot"], w["bounds"], w["pool"])
    rng = np.random.default_rng(5)
    for cam in ("A", "C"):
        pos, fwd, up, right, d = _camera_rays(vx, cam)
        pick = rng.choice(W * H, size=60000, replace=False)
        dd = d.reshape(-1, 3)[pick]
        oo = np.broadcast_to(pos, dd.shape).copy()
        g = ctx.Raytrace(oo, dd, want_stats=True)
        c = world.trace_batch(oo, dd, nthreads=16)
        assert np.array_equal(g["hit"], c["hit"]) and np.array_equal(g["steps"], c["steps"])
        assert np.array_equal(g["voxel"], c["voxel"])
        assert np.array_equal(g["hitPoint"].view(np.uint32), c["pos"].view(np.uint32))
        assert np.array_equal(g["normal"], c["normal"])
        assert (g["stats"].coarse_probes, g["stats"].brick_entries, g["stats"].fine_probes) == (
            c["stats"].coarse_probes, c["stats"].brick_entries, c["stats"].fine_probes)
    # secondary-ray style inputs: short random rays from just above hit points (max_steps is 2048 in the batch API)
    hp = g["hitPoint"][g["hit"] == 1][:20000] + f32(0.01)
    dirs = rng.normal(size=hp.shape).astype(f32)
    g2, c2 = ctx.Raytrace(hp, dirs), world.trace_batch(hp, dirs, nthreads=16)
    assert np.array_equal(g2["steps"], c2["steps"]) and np.array_equal(g2["voxel"], c2["voxel"])


def test_idempotence_variants_and_strip_shards(big):
    vx, ctx, torch, _ = big
    (fx, fy, fz), euler = helpers.CAMERAS["A"]
    pos = (fx * X, fy * Y, fz * Z)
    fwd, up, right = vx.GetDirections(euler)
    base = dict(shadow=True, bounce_samples=1, frame_number=3)
    a = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    b = torch.zeros_like(a)
    ctx.RenderScreen(W, H, a, pos, fwd, up, right, vx.RenderOptions(**base))
    ctx.RenderScreen(W, H, b, pos, fwd, up, right, vx.RenderOptions(**base))
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    default = ctx.kernel_variant
    try:
        for variant in (1, 7):  # two separately written kernels, one frame
            ctx.set_kernel_variant(variant)
            b.zero_()
            ctx.RenderScreen(W, H, b, pos, fwd, up, right, vx.RenderOptions(**base))
            torch.cuda.synchronize()
            assert torch.equal(a, b), variant
    finally:
        ctx.set_kernel_variant(default)
    count, rows = 8, 16
    max_rows = max(vx.compact_rows(H, rows, count, i) for i in range(count))
    stride = max_rows * W * 4
    shards = torch.zeros((count, stride), dtype=torch.uint8, device="cuda")
    for i in range(count):
        ctx.RenderScreen(W, H, shards[i], pos, fwd, up, right,
                         vx.RenderOptions(strip_rows=rows, strip_count=count, strip_index=i, compact=True, **base))
    out = torch.zeros_like(a)
    ctx.deinterleave_strips(W, H, rows, count, shards, stride, out)
    torch.cuda.synchronize()
    assert torch.equal(out, a)
    ctx.frame_stats()
