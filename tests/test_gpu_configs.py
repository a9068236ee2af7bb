"""Every BASELINE.json configuration on the HIP path (configs[0] is tests/test_config0_plumbing.py, configs[2] is
tests/test_gpu_fullsize.py).  Reference lines restated by what is compared: Renderer.cu:179-276 (screenDispatch),
VolumeRaytracer.cu:354-525 (Raytrace).

  configs[1]  1080p primary-ray-only, 1024x256x1024 world: the oracle renders WHOLE frames here, so the four benchmark
              cameras are compared byte for byte (framebuffer) and index for index (hit voxel AOV).
  configs[3]  3840x2160 on the 8192x512x8192 world (the frame 8 GPUs split by strips): sampled primary rays against the
              oracle, an oracle-rendered band of rows of the shaded frame (primary + shadow + bounce) against the same rows
              of the HIP frame, render-vs-batch hit indices over the whole 8.3 M pixel frame, 8-strip reassembly.
  configs[4]  16384x1024x16384 world (9 GiB resident, 64-bit addressing), 4K, bounce_depth 2: sampled rays and an
              oracle band on the downloaded world, idempotence, the three separately written kernels agree, 8-strip
              reassembly, ray accounting.
"""
import ctypes

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu
f32 = np.float32


def _make_ctx(X, Y, Z, F):
    import torch
    import voxelengine_amd as vx
    ctx = vx.Context(0)
    info = ctx.build_world(vx.GEN_PERLIN_REF, X, Y, Z, F)
    inv = float(f32(1.0) / np.sqrt(f32(3.0), dtype=f32))
    ctx.SetEnvironment((inv, inv, inv), (2, 2, 2), (0.5, 0.5, 0.5))
    ctx.SetFOV(90.0)
    return vx, ctx, torch, info


def _oracle_world(ctx, vxo):
    w = ctx.download_world()
    return vxo.World.wrap(w["factor"], w["cdims"], w["coarse_bits"], w["brick_slot"], w["bounds"], w["pool"])


def _camera(vx, name, dims):
    (fx, fy, fz), euler = helpers.CAMERAS[name]
    pos = np.array((fx * dims[0], fy * dims[1], fz * dims[2]), f32)
    fwd, up, right = vx.GetDirections(euler)
    return pos, fwd, up, right


def _camera_rays(vx, name, dims, W, H):
    """getRayDirection (Renderer.cu:44-59) in numpy binary32, op for op; returns the normalised directions."""
    pos, fwd, up, right = _camera(vx, name, dims)
    libm = ctypes.CDLL("libm.so.6")
    libm.tanf.restype = ctypes.c_float
    libm.tanf.argtypes = [ctypes.c_float]
    fov = f32(np.float64(f32(90.0)) * 3.1415 / 180.0)
    t = f32(libm.tanf(ctypes.c_float(float(fov / f32(2.0)))))
    kx, ky = f32(t * (f32(W) / f32(H))), t
    xs = (np.arange(W, dtype=f32) / f32(W)) * f32(2) - f32(1)
    ys = (np.arange(H, dtype=f32) / f32(H)) * f32(2) - f32(1)
    d = np.empty((H, W, 3), f32)
    for a in range(3):
        d[..., a] = (fwd[a] + (xs[None, :] * kx) * right[a]) + (ys[:, None] * ky) * up[a]
    n = f32(1.0) / np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2], dtype=f32)
    return pos, fwd, up, right, d * n[..., None]


def _assert_rays_equal(g, c):
    assert np.array_equal(g["hit"], c["hit"]) and np.array_equal(g["steps"], c["steps"])
    assert np.array_equal(g["voxel"], c["voxel"])
    assert np.array_equal(g["hitPoint"].view(np.uint32), c["pos"].view(np.uint32))
    assert np.array_equal(g["normal"], c["normal"])


def _strip_reassembly(vx, ctx, torch, W, H, pos, fwd, up, right, base, full, count=8, rows=16):
    max_rows = max(vx.compact_rows(H, rows, count, i) for i in range(count))
    stride = max_rows * W * 4
    shards = torch.zeros((count, stride), dtype=torch.uint8, device="cuda")
    for i in range(count):
        ctx.RenderScreen(W, H, shards[i], pos, fwd, up, right,
                         vx.RenderOptions(strip_rows=rows, strip_count=count, strip_index=i, compact=True, **base))
    out = torch.zeros_like(full)
    ctx.deinterleave_strips(W, H, rows, count, shards, stride, out)
    torch.cuda.synchronize()
    assert torch.equal(out, full)


# ---------------------------------------------------------------------------------------------------------------
# configs[1]: 1080p primary-ray-only brickmap trace, 1024x256x1024 world
@pytest.fixture(scope="module")
def cfg1(vxo):
    vx, ctx, torch, info = _make_ctx(1024, 256, 1024, 32)
    world = _oracle_world(ctx, vxo)
    yield vx, ctx, torch, world
    ctx.close()


@pytest.mark.parametrize("cam", ["A", "B", "C", "D"])
def test_config1_whole_1080p_frames_equal_the_oracle(cfg1, vxo, cam):
    vx, ctx, torch, world = cfg1
    W, H, dims = 1920, 1080, (1024, 256, 1024)
    pos, fwd, up, right = _camera(vx, cam, dims)
    fb = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    hit = torch.full((H, W), -1, dtype=torch.int64, device="cuda")
    ctx.frame_stats()
    ctx.RenderScreen(W, H, fb, pos, fwd, up, right, vx.RenderOptions(frame_number=1), hit_aov=hit)
    st = ctx.frame_stats()
    ref = world.render(vxo.make_params(W, H, pos, fwd, up, right, frame_number=1), fb=np.zeros((H, W, 4), np.uint8),
                       want_hit=True, nthreads=16)
    assert np.array_equal(fb.cpu().numpy(), ref["fb"])
    assert np.array_equal(hit.cpu().numpy(), ref["hit"])
    assert st.primary_rays == W * H and st.shadow_rays == 0 and st.bounce_rays == 0
    assert st.primary_hits == ref["stats"].primary_hits
    assert st.primary_hits > 0
    if cam in ("A", "B"):  # these two see sky as well as terrain (the top-down and the grazing camera see terrain only)
        assert st.primary_hits < W * H
    # the probe-counting instantiation counts what the oracle counts (the roofline's algorithmic bytes)
    ctx.RenderScreen(W, H, fb, pos, fwd, up, right, vx.RenderOptions(frame_number=1, collect_stats=True))
    sp = ctx.frame_stats()
    assert (sp.coarse_probes, sp.brick_entries, sp.fine_probes) == (
        ref["stats"].probes.coarse_probes, ref["stats"].probes.brick_entries, ref["stats"].probes.fine_probes)


def test_config1_multi_view_launch_of_the_four_cameras(cfg1, vxo):
    """bench.py's step shape on this configuration: the four cameras in ONE launch, every view equal to the oracle."""
    vx, ctx, torch, world = cfg1
    W, H, dims = 1920, 1080, (1024, 256, 1024)
    frames = torch.zeros((4, H, W, 4), dtype=torch.uint8, device="cuda")
    cams = [_camera(vx, c, dims) for c in "ABCD"]
    ctx.RenderViews(W, H, [dict(fb=frames[j], origin=c[0], fwd=c[1], up=c[2], right=c[3], frame_number=j + 1)
                           for j, c in enumerate(cams)], vx.RenderOptions())
    got = frames.cpu().numpy()
    for j, c in enumerate(cams):
        ref = world.render(vxo.make_params(W, H, *c, frame_number=j + 1), fb=np.zeros((H, W, 4), np.uint8), nthreads=16)
        assert np.array_equal(got[j], ref["fb"]), "ABCD"[j]
    ctx.frame_stats()


# ---------------------------------------------------------------------------------------------------------------
# configs[3]: 4K frame on the 8192x512x8192 world (what 8 GPUs split by strips)
@pytest.fixture(scope="module")
def cfg3(vxo):
    vx, ctx, torch, info = _make_ctx(8192, 512, 8192, 32)
    world = _oracle_world(ctx, vxo)
    yield vx, ctx, torch, world
    ctx.close()


DIMS8K = (8192, 512, 8192)
W4K, H4K = 3840, 2160


@pytest.mark.parametrize("cam", ["A", "D"])
def test_config3_sampled_4k_primary_rays_against_the_oracle(cfg3, cam):
    vx, ctx, torch, world = cfg3
    pos, fwd, up, right, d = _camera_rays(vx, cam, DIMS8K, W4K, H4K)
    pick = np.random.default_rng(11).choice(W4K * H4K, size=60000, replace=False)
    dd = d.reshape(-1, 3)[pick]
    oo = np.broadcast_to(pos, dd.shape).copy()
    g, c = ctx.Raytrace(oo, dd, want_stats=True), world.trace_batch(oo, dd, nthreads=16)
    _assert_rays_equal(g, c)
    assert (g["stats"].coarse_probes, g["stats"].brick_entries, g["stats"].fine_probes) == (
        c["stats"].coarse_probes, c["stats"].brick_entries, c["stats"].fine_probes)


def test_config3_4k_frame_bands_entry_points_and_strips(cfg3, vxo):
    vx, ctx, torch, world = cfg3
    pos, fwd, up, right, d = _camera_rays(vx, "A", DIMS8K, W4K, H4K)
    base = dict(shadow=True, bounce_samples=1, frame_number=9)
    fb = torch.zeros((H4K, W4K, 4), dtype=torch.uint8, device="cuda")
    hit = torch.full((H4K, W4K), -1, dtype=torch.int64, device="cuda")
    ctx.frame_stats()
    ctx.RenderScreen(W4K, H4K, fb, pos, fwd, up, right, vx.RenderOptions(**base), hit_aov=hit)
    st = ctx.frame_stats()
    assert st.primary_rays == W4K * H4K and st.shadow_rays == st.primary_hits and st.total_rays() <= 3 * W4K * H4K
    got = fb.cpu().numpy()
    # oracle-rendered bands of the SHADED frame (primary + shadow + bounce): sky/horizon rows, terrain rows, last rows
    for r0, r1 in ((0, 24), (1040, 1088), (H4K - 24, H4K)):
        ref = world.render(vxo.make_params(W4K, H4K, pos, fwd, up, right, frame_number=9, shadow=1, bounce_samples=1,
                                           row_begin=r0, row_end=r1), fb=np.zeros((H4K, W4K, 4), np.uint8), want_hit=True,
                           nthreads=16)
        assert np.array_equal(got[r0:r1], ref["fb"][r0:r1]), (r0, r1)
        assert np.array_equal(hit[r0:r1].cpu().numpy(), ref["hit"][r0:r1]), (r0, r1)
    # the two entry points agree over the whole frame: hit voxel AOV of vxrt_render == vxrt_trace_batch of the camera rays
    b = ctx.Raytrace(np.broadcast_to(pos, d.shape).reshape(-1, 3), d.reshape(-1, 3))
    assert np.array_equal(b["voxel"].reshape(H4K, W4K), hit.cpu().numpy())
    assert int(b["hit"].sum()) == st.primary_hits
    _strip_reassembly(vx, ctx, torch, W4K, H4K, pos, fwd, up, right, base, fb)
    ctx.frame_stats()


# ---------------------------------------------------------------------------------------------------------------
# configs[4]: 16384x1024x16384 world, 4K, 2 bounces
@pytest.fixture(scope="module")
def cfg4(vxo):
    vx, ctx, torch, info = _make_ctx(16384, 1024, 16384, 32)
    assert info.hbm_bytes > 8 * 2**30  # the pool alone needs 64-bit byte offsets
    yield vx, ctx, torch, info
    ctx.close()


DIMS16K = (16384, 1024, 16384)


def test_config4_properties_at_4k_with_two_bounces(cfg4):
    vx, ctx, torch, info = cfg4
    pos, fwd, up, right = _camera(vx, "A", DIMS16K)
    base = dict(shadow=True, bounce_samples=1, bounce_depth=2, frame_number=5)
    a = torch.zeros((H4K, W4K, 4), dtype=torch.uint8, device="cuda")
    b = torch.zeros_like(a)
    ctx.frame_stats()
    ctx.RenderScreen(W4K, H4K, a, pos, fwd, up, right, vx.RenderOptions(**base))
    st = ctx.frame_stats()
    # ray accounting: one shadow ray per primary hit; a bounce sample may spawn one more ray (the extension)
    assert st.primary_rays == W4K * H4K and st.shadow_rays == st.primary_hits
    assert st.bounce_rays <= 2 * st.primary_hits and st.total_rays() <= 4 * W4K * H4K
    ctx.RenderScreen(W4K, H4K, b, pos, fwd, up, right, vx.RenderOptions(**base))
    torch.cuda.synchronize()
    assert torch.equal(a, b)  # idempotence
    default = ctx.kernel_variant
    try:
        for variant in (1, 7):  # two separately written kernels, one frame
            ctx.set_kernel_variant(variant)
            b.zero_()
            ctx.RenderScreen(W4K, H4K, b, pos, fwd, up, right, vx.RenderOptions(**base))
            torch.cuda.synchronize()
            assert torch.equal(a, b), variant
    finally:
        ctx.set_kernel_variant(default)
    _strip_reassembly(vx, ctx, torch, W4K, H4K, pos, fwd, up, right, base, a)
    # the same frame inside a multi-view launch
    b.zero_()
    ctx.RenderViews(W4K, H4K, [dict(fb=b, origin=pos, fwd=fwd, up=up, right=right, frame_number=5)],
                    vx.RenderOptions(shadow=True, bounce_samples=1, bounce_depth=2))
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    ctx.frame_stats()


def test_config4_oracle_on_the_downloaded_16k_world(cfg4, vxo):
    """The whole 9 GiB brick pool is copied to the host and handed to the oracle: sampled primary and secondary rays, and
    bands of rows of the 4K frame with the second bounce."""
    vx, ctx, torch, info = cfg4
    world = _oracle_world(ctx, vxo)
    rng = np.random.default_rng(4)
    for cam in ("A", "B"):
        pos, fwd, up, right, d = _camera_rays(vx, cam, DIMS16K, W4K, H4K)
        pick = rng.choice(W4K * H4K, size=40000, replace=False)
        dd = d.reshape(-1, 3)[pick]
        oo = np.broadcast_to(pos, dd.shape).copy()
        g, c = ctx.Raytrace(oo, dd), world.trace_batch(oo, dd, nthreads=16)
        _assert_rays_equal(g, c)
    hp = g["hitPoint"][g["hit"] == 1][:20000] + f32(0.01)  # secondary-ray style inputs from hit points deep in the world
    dirs = rng.normal(size=hp.shape).astype(f32)
    _assert_rays_equal(ctx.Raytrace(hp, dirs), world.trace_batch(hp, dirs, nthreads=16))
    pos, fwd, up, right = _camera(vx, "A", DIMS16K)
    fb = torch.zeros((H4K, W4K, 4), dtype=torch.uint8, device="cuda")
    ctx.RenderScreen(W4K, H4K, fb, pos, fwd, up, right,
                     vx.RenderOptions(shadow=True, bounce_samples=1, bounce_depth=2, frame_number=5))
    got = fb.cpu().numpy()
    for r0, r1 in ((1064, 1096), (H4K - 16, H4K)):
        ref = world.render(vxo.make_params(W4K, H4K, pos, fwd, up, right, frame_number=5, shadow=1, bounce_samples=1,
                                           bounce_depth=2, row_begin=r0, row_end=r1), fb=np.zeros((H4K, W4K, 4), np.uint8),
                           nthreads=16)
        assert np.array_equal(got[r0:r1], ref["fb"][r0:r1]), (r0, r1)
    ctx.frame_stats()
