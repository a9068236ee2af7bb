"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs.
Integer/index outputs (hit, steps, voxel index, 8-bit pixels, probe counters) must be bit-exact; positions
and normals are compared bit-exact too (same IEEE expressions on both sides); the float colour AOV within
1e-4 per channel (north_star tolerance)."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

COLOR_TOL = 1e-4


@pytest.fixture(scope="module")
def eng():
    import torch
    import voxelengine_amd as vx
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    ctx = vx.Context(0)
    yield vx, ctx, torch
    ctx.close()


def _upload(ctx, w):
    ctx.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)


def _assert_batch_equal(gpu, cpu):
    assert np.array_equal(gpu["hit"], cpu["hit"])
    assert np.array_equal(gpu["steps"], cpu["steps"])
    assert np.array_equal(gpu["voxel"], cpu["voxel"])
    # positions bit for bit; a NaN equals a NaN (a ray with a denormal direction component can make the reference's own
    # arithmetic produce inf * 0: x86 and gfx950 then differ in the sign of the default NaN, which carries no information)
    assert np.array_equal(helpers.float_bits(gpu["hitPoint"]), helpers.float_bits(cpu["pos"]))
    assert np.array_equal(gpu["normal"].view(np.uint32) & 0x7FFFFFFF, cpu["normal"].view(np.uint32) & 0x7FFFFFFF)
    assert np.array_equal(gpu["normal"], cpu["normal"])


@pytest.mark.parametrize("factor,size,density,seed", [
    (8, (64, 64, 64), 0.01, 1), (8, (128, 64, 64), 0.08, 2), (16, (128, 128, 128), 0.002, 3),
    (32, (256, 256, 256), 0.0005, 4), (8, (64, 64, 64), 0.6, 5),
])
def test_trace_batch_random_worlds(eng, vxo, factor, size, density, seed):
    vx, ctx, _ = eng
    w = helpers.random_voxel_world(vxo, size, factor, density, seed)
    _upload(ctx, w)
    o, d = helpers.mixed_rays(w.dims, 30000, seed)
    cpu = w.trace_batch(o, d)
    gpu = ctx.Raytrace(o, d, want_stats=True)
    _assert_batch_equal(gpu, cpu)
    _assert_batch_equal(ctx.Raytrace(o, d), cpu)      # the timed instantiation
    st = gpu["stats"]
    assert st.primary_rays == len(o) and st.primary_hits == int(cpu["hit"].sum())
    assert (st.coarse_probes, st.brick_entries, st.fine_probes) == (
        cpu["stats"].coarse_probes, cpu["stats"].brick_entries, cpu["stats"].fine_probes)
    assert 0 < int(cpu["hit"].sum()) < len(o)


@pytest.mark.parametrize("n", [262145, 300000, 524288 + 63])
def test_trace_batch_persistent_queue(eng, vxo, n):
    """Batches of at least 8 rays per lane of the persistent grid take k_trace_batch_persist (a queue of rays instead
    of one ray per lane): same results, same probe counters, ragged last ticket included.  The context is created
    with one persistent wave per CU so that these batch sizes are over the threshold."""
    import os
    vx, _, _ = eng
    ctx = vx.Context(0)
    ctx.set_persistent_waves_per_cu(1)   # a small persistent grid: batches of this size take the queue kernel
    try:
        w = helpers.random_voxel_world(vxo, (128, 128, 128), 16, 0.004, 21)
        _upload(ctx, w)
        o, d = helpers.mixed_rays(w.dims, n, 5)
        cpu = w.trace_batch(o, d)
        gpu = ctx.Raytrace(o, d, want_stats=True)
        _assert_batch_equal(gpu, cpu)
        st = gpu["stats"]
        assert st.primary_rays == n and st.primary_hits == int(cpu["hit"].sum())
        assert (st.coarse_probes, st.brick_entries, st.fine_probes) == (
            cpu["stats"].coarse_probes, cpu["stats"].brick_entries, cpu["stats"].fine_probes)
        _assert_batch_equal(ctx.Raytrace(o, d), cpu)      # the timed instantiation
        assert 0 < int(cpu["hit"].sum()) < n
    finally:
        ctx.close()


def test_invalid_rays_are_defined_misses_and_a_bad_camera_is_rejected(eng, vxo):
    """Ray validity at the C ABI (include/vxrt.h): a ray with a NaN / infinite component, the zero direction, or a
    direction whose squared length leaves the binary32 range is not traced; its result is a miss with 0 steps.  The valid
    rays around it are unaffected (equal to the oracle), in every batch kernel -- the persistent queue, one ray per lane,
    the straightforward loops.  A camera with a non-finite component is an error, not a frame."""
    import os
    vx, ctx0, torch = eng
    ctx = vx.Context(0)
    ctx.set_persistent_waves_per_cu(1)   # a small persistent grid: batches of this size take the queue kernel
    try:
        w = helpers.random_voxel_world(vxo, (128, 128, 128), 16, 0.004, 33)
        _upload(ctx, w)
        n = 300000
        o, d = helpers.mixed_rays(w.dims, n, 11)
        bad = np.zeros(n, bool)
        nan, inf = np.float32(np.nan), np.float32(np.inf)
        poison = [("o", 0, nan), ("o", 2, inf), ("o", 1, -inf), ("d", 0, nan), ("d", 1, inf), ("d", None, 0.0),
                  ("d", None, 1e-30), ("d", None, 1e30)]
        for k, (what, comp, val) in enumerate(poison):
            idx = np.arange(7 + k * 13, n, 997)
            arr = o if what == "o" else d
            if comp is None:
                arr[idx] = (np.sign(arr[idx]) + (arr[idx] == 0)) * np.float32(val)   # all three components that small / large
            else:
                arr[idx, comp] = val
            bad[idx] = True
        good = ~bad
        cpu = w.trace_batch(o[good], d[good])
        # the queue kernel (small persistent grid) and one ray per lane, each counting probes and as timed; the
        # straightforward loops
        for variant, waves_per_cu, want_stats in ((4, 1, True), (4, 1, False), (4, 0, True), (4, 0, False), (1, 0, True)):
            ctx.set_kernel_variant(variant)
            ctx.set_persistent_waves_per_cu(waves_per_cu)
            g = ctx.Raytrace(o, d, want_stats=want_stats)
            if want_stats:
                assert g["stats"].primary_rays == n
            sub = {k: g[k][good] for k in ("hit", "steps", "voxel", "hitPoint", "normal")}
            _assert_batch_equal(sub, cpu)
            assert not g["hit"][bad].any() and not g["steps"][bad].any() and (g["voxel"][bad] == -1).all(), variant
            assert np.isposinf(g["hitPoint"][bad]).all() and not g["normal"][bad].any(), variant
        W, H = 64, 48
        fb = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        pos, f, u, r = helpers.camera("A", w.dims, vxo)
        for which in range(4):
            args = [list(pos), list(f), list(u), list(r)]
            args[which][which % 3] = float("nan") if which % 2 == 0 else float("inf")
            with pytest.raises(vx.VxrtError):
                ctx.RenderScreen(W, H, fb, *args)
        ctx.RenderScreen(W, H, fb, pos, f, u, r)   # and the context still renders afterwards
        torch.cuda.synchronize()
    finally:
        ctx.close()


def test_trace_batch_terrain_and_empty_inputs(eng, vxo):
    vx, ctx, _ = eng
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, 256, 256, 256, 32)
    _upload(ctx, w)
    o, d = helpers.mixed_rays(w.dims, 60000, 9)
    _assert_batch_equal(ctx.Raytrace(o, d), w.trace_batch(o, d))
    out = ctx.Raytrace(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert out["steps"].shape == (0,)
    # ragged tail (not a multiple of the workgroup size) and a single ray
    _assert_batch_equal(ctx.Raytrace(o[:257], d[:257]), w.trace_batch(o[:257], d[:257]))
    _assert_batch_equal(ctx.Raytrace(o[:1], d[:1]), w.trace_batch(o[:1], d[:1]))


@pytest.mark.parametrize("variant", [1, 7])
def test_every_kernel_variant_agrees_with_the_oracle(eng, vxo, variant):
    """Variants: 7 = the product kernels (persistent wavefronts on the tracer of vxrt_wave2.hpp: speculative exec-masked
    advance, packed step counters; what the default runs), 1 = the straightforward per-lane loops.  Both give the oracle's
    bits, in every render mode -- frame, colour AOV and hit indices from the timed instantiation, probe counters from the
    counting one (_render_both)."""
    vx, ctx, torch = eng
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, 256, 256, 256, 32)
    _upload(ctx, w)
    o, d = helpers.mixed_rays(w.dims, 20000, 21)
    cpu = w.trace_batch(o, d)
    default = ctx.kernel_variant
    try:
        ctx.set_kernel_variant(variant)
        _assert_batch_equal(ctx.Raytrace(o, d), cpu)
        _assert_frame_equal(*_render_both(eng, vxo, w, 160, 96, "A", frame_number=5, shadow=1, bounce_samples=2))
        _assert_frame_equal(*_render_both(eng, vxo, w, 150, 90, "B", frame_number=4, shadow=1, bounce_samples=1,
                                          bounce_all_hits=1))
        _assert_frame_equal(*_render_both(eng, vxo, w, 160, 90, "D", frame_number=2, mode=1, checkerboard=1))
        _assert_frame_equal(*_render_both(eng, vxo, w, 128, 96, "A", ortho=1, ortho_size=(60.0, 60.0), shadow=1))
        _assert_frame_equal(*_render_both(eng, vxo, w, 131, 77, "C"))
        _assert_frame_equal(*_render_both(eng, vxo, w, 144, 80, "A", frame_number=6, shadow=1, bounce_samples=2,
                                          bounce_all_hits=1, bounce_depth=2))
    finally:
        ctx.set_kernel_variant(default)


def test_known_answer_rays_on_gpu(eng, vxo):
    """The hand-derived cases of tests/test_oracle_kat.py, through the HIP path."""
    vx, ctx, _ = eng
    v = np.zeros((64, 64, 64), bool)
    for p in [(26, 12, 11), (26, 10, 11), (18, 11, 11)]:
        v[p] = True
    w = vxo.World.from_voxels(v, 8)
    _upload(ctx, w)
    up = np.nextafter(np.float32(1.4375), np.float32(np.inf))
    yw = np.float32(np.float32(np.float32(up * np.float32(8)) - np.float32(8)) + np.float32(8))
    default = ctx.kernel_variant
    try:
        for variant in (4, 7, 1):   # (the tracer of the headline kernel, with and without its probe counters; the loops)
            ctx.set_kernel_variant(variant)
            for want_stats in (True, False):
                r = ctx.Raytrace([(60.0, 11.5, 11.5)], [(-1, 0, 0)], want_stats=want_stats)
                assert r["steps"][0] == 13 and r["hit"][0] == 1, variant
                assert r["hitPoint"][0].tolist() == [19.0, float(yw), float(yw)], variant
                assert r["normal"][0].tolist() == [-1, 0, 0], variant
                assert r["voxel"][0] == 18 + 64 * (11 + 64 * 11), variant
                if want_stats:
                    assert (r["stats"].coarse_probes, r["stats"].brick_entries, r["stats"].fine_probes) == (6, 2, 10), variant
    finally:
        ctx.set_kernel_variant(default)


def test_quirk_cases_on_gpu(eng, vxo):
    """The hand-derived quirk cases of tests/quirk_cases.py (edge padding, exact ties, head-only maxSteps with the bounce
    rays' budget of 8, double-rounded world-entry box, previous_cell hole, NextCell snap order, region check) through the HIP
    path: the hand-derived expectations themselves, not just equality with the oracle; all three kernel variants."""
    from tests import quirk_cases
    vx, ctx, _ = eng
    default = ctx.kernel_variant
    try:
        for name, case in quirk_cases.all_cases().items():
            w, o, d = quirk_cases.build_case(vxo, case)
            _upload(ctx, w)
            e = case["expect"]
            ctx.set_batch_max_steps(case["max_steps"])
            # (a one-ray batch: 7 = the tracer of the headline kernel, with its probe counters and -- `False` -- as timed;
            # 1 = the straightforward loops)
            for variant, want_stats in ((7, True), (7, False), (4, False), (1, True)):
                ctx.set_kernel_variant(variant)
                r = ctx.Raytrace([o], [d], want_stats=want_stats)
                assert bool(r["hit"][0]) == e["hit"] and int(r["steps"][0]) == e["steps"], (name, variant)
                assert r["normal"][0].tolist() == [float(x) for x in e["normal"]], (name, variant)
                if want_stats:
                    assert (r["stats"].coarse_probes, r["stats"].brick_entries, r["stats"].fine_probes) == tuple(e["stats"]), (name, variant)
                if e["hit"]:
                    assert r["hitPoint"][0].tolist() == [float(np.float32(x)) for x in e["pos"]], (name, variant)
                    assert int(r["voxel"][0]) == quirk_cases.voxel_index(e["voxel"], case["size"]), (name, variant)
                else:
                    assert np.isinf(r["hitPoint"][0]).all() and int(r["voxel"][0]) == -1, (name, variant)
    finally:
        ctx.set_batch_max_steps(2048)
        ctx.set_kernel_variant(default)


def test_batch_max_steps_matches_the_oracle(eng, vxo):
    """Raytrace's maxSteps on whole batches (8 = the secondary rays' budget, Renderer.cu:141), all batch kernels."""
    import os
    vx, ctx, _ = eng
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, 256, 256, 256, 32)
    o, d = helpers.mixed_rays(w.dims, 40000, 33)
    small = vx.Context(0)
    small.set_persistent_waves_per_cu(1)   # a small persistent grid: batches of this size take the queue kernel
    try:
        for c in (ctx, small):
            _upload(c, w)
            for ms in (8, 1, 100):
                c.set_batch_max_steps(ms)
                got = c.Raytrace(o, d)
                hit = np.empty(len(o), np.uint8)
                steps = np.empty(len(o), np.int32)
                for i in range(0, len(o), 997):   # the oracle's single-ray entry point takes max_steps
                    r = w.raytrace(o[i], d[i], ms)
                    assert bool(got["hit"][i]) == r["hit"] and int(got["steps"][i]) == r["steps"], (ms, i)
            c.set_batch_max_steps(2048)
            _assert_batch_equal(c.Raytrace(o, d), w.trace_batch(o, d))
        with pytest.raises(vx.VxrtError):
            ctx.set_batch_max_steps(0)
    finally:
        ctx.set_batch_max_steps(2048)
        small.close()


@pytest.mark.parametrize("depth", [1, 2])
def test_eighty_launches_in_flight_over_four_streams(eng, vxo, depth):
    """More launches in flight than the context has queue heads (64 single-view, 16 multi-view): launch 65 waits for launch 1
    instead of sharing its tile counter (vxrt_api.hip ring_acquire).  80 single-view launches and 20 multi-view launches
    over 4 streams, no synchronisation in between; every frame must be the frame a lone launch renders, and the ray
    counters must add up.  depth 2 = the second-bounce instantiation of the kernels (the one with the most private state)."""
    vx, ctx, torch = eng
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, 256, 256, 256, 32)
    _upload(ctx, w)
    W, H = 200, 120
    cams = [helpers.camera(c, w.dims, vxo) for c in "ABCD"]
    inv = float(np.float32(1.0) / np.sqrt(np.float32(3.0)))
    ctx.SetEnvironment((inv, inv, inv), (2, 2, 2), (0.5, 0.5, 0.5))
    ctx.SetFOV(90.0)
    base = dict(shadow=True, bounce_samples=1, bounce_depth=depth)
    ref, rays_ref = [], []
    for j, (pos, f, u, r) in enumerate(cams):     # one at a time
        fb = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        ctx.frame_stats()
        ctx.RenderScreen(W, H, fb, pos, f, u, r, vx.RenderOptions(frame_number=j + 1, **base))
        rays_ref.append(ctx.frame_stats().total_rays())
        ref.append(fb.clone())
    streams = [torch.cuda.Stream() for _ in range(4)]
    fbs = torch.zeros((80, H, W, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.frame_stats()
    for i in range(80):
        pos, f, u, r = cams[i % 4]
        with torch.cuda.stream(streams[i % 4]):
            ctx.RenderScreen(W, H, fbs[i], pos, f, u, r, vx.RenderOptions(frame_number=i % 4 + 1, **base))
    torch.cuda.synchronize()
    st = ctx.frame_stats()
    assert st.total_rays() == 20 * sum(rays_ref)
    for i in range(80):
        assert torch.equal(fbs[i], ref[i % 4]), i
    mv = torch.zeros((20, 4, H, W, 4), dtype=torch.uint8, device="cuda")
    for k in range(20):
        with torch.cuda.stream(streams[k % 4]):
            ctx.RenderViews(W, H, [dict(fb=mv[k, j], origin=c[0], fwd=c[1], up=c[2], right=c[3], frame_number=j + 1)
                                   for j, c in enumerate(cams)], vx.RenderOptions(**base))
    torch.cuda.synchronize()
    assert ctx.frame_stats().total_rays() == 20 * sum(rays_ref)
    for k in range(20):
        for j in range(4):
            assert torch.equal(mv[k, j], ref[j]), (k, j)


def _render_both(eng, vxo, w, W, H, cam, frame_number=1, **kw):
    vx, ctx, torch = eng
    pos, f, u, r = helpers.camera(cam, w.dims, vxo)
    p = vxo.make_params(W, H, pos, f, u, r, frame_number=frame_number, **kw)
    fb0 = np.random.default_rng(7).integers(0, 255, size=(H, W, 4), dtype=np.uint8)  # stale contents survive
    cpu = w.render(p, fb=fb0.copy(), want_color=True, want_hit=True)
    d_fb = torch.from_numpy(fb0.copy()).cuda()
    d_col = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    d_hit = torch.full((H, W), -1, dtype=torch.int64, device="cuda")
    ctx.SetEnvironment(list(p.env.light_dir), list(p.env.light_color), list(p.env.ambient))
    ctx.SetFOV(p.fov_deg)
    ctx.SetOrthoWindowSize(p.ortho_size[0], p.ortho_size[1])
    okw = dict(mode=kw.get("mode", 0), checkerboard=bool(kw.get("checkerboard", 0)),
               shadow=bool(kw.get("shadow", 0)), bounce_samples=kw.get("bounce_samples", 0),
               bounce_all_hits=bool(kw.get("bounce_all_hits", 0)), ortho=bool(kw.get("ortho", 0)),
               bounce_depth=kw.get("bounce_depth", 1), frame_number=frame_number)
    # The frame, the colour AOV and the hit indices that are compared with the oracle come from the TIMED instantiation of
    # the kernel (no probe counting), and the kernel is the one the test names: a forced variant must not be swapped.
    opts = vx.RenderOptions(**okw)
    if ctx.kernel_variant != 4:
        assert ctx.kernel_for_launch(W, H, opts) == ctx.kernel_variant
    ctx.frame_stats()   # counters accumulate until read: start this frame from zero
    ctx.RenderScreen(W, H, d_fb, pos, f, u, r, opts, color_aov=d_col, hit_aov=d_hit)
    plain = ctx.frame_stats()
    # ... and a second launch of the probe-counting instantiation of the same kernel gives the counters (and the same frame)
    d_fb2 = torch.from_numpy(fb0.copy()).cuda()
    ctx.RenderScreen(W, H, d_fb2, pos, f, u, r, vx.RenderOptions(collect_stats=True, **okw))
    st = ctx.frame_stats()
    assert torch.equal(d_fb2, d_fb)
    assert (plain.primary_rays, plain.shadow_rays, plain.bounce_rays, plain.primary_hits) == (
        st.primary_rays, st.shadow_rays, st.bounce_rays, st.primary_hits)
    return cpu, d_fb.cpu().numpy(), d_col.cpu().numpy(), d_hit.cpu().numpy(), st


def _assert_frame_equal(cpu, fb, col, hit, st):
    cst = cpu["stats"]
    assert (st.primary_rays, st.shadow_rays, st.bounce_rays, st.primary_hits) == (
        cst.primary_rays, cst.shadow_rays, cst.bounce_rays, cst.primary_hits)
    assert (st.coarse_probes, st.brick_entries, st.fine_probes) == (
        cst.probes.coarse_probes, cst.probes.brick_entries, cst.probes.fine_probes)
    assert np.array_equal(hit, cpu["hit"])
    assert np.nanmax(np.abs(col - cpu["color"])) <= COLOR_TOL
    assert np.array_equal(fb, cpu["fb"])


@pytest.mark.parametrize("cam", ["A", "B", "C", "D"])
def test_render_shaded_primary_only(eng, vxo, cam):
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, 256, 256, 256, 32)
    _upload(eng[1], w)
    cpu, fb, col, hit, st = _render_both(eng, vxo, w, 200, 120, cam)
    _assert_frame_equal(cpu, fb, col, hit, st)
    assert st.primary_rays == 200 * 120


@pytest.mark.parametrize("cam,gate", [("A", 0), ("D", 0), ("A", 1)])
def test_render_shadow_and_bounce(eng, vxo, cam, gate):
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, 256, 256, 256, 32)
    _upload(eng[1], w)
    cpu, fb, col, hit, st = _render_both(eng, vxo, w, 192, 108, cam, frame_number=3, shadow=1, bounce_samples=1,
                                         bounce_all_hits=gate)
    _assert_frame_equal(cpu, fb, col, hit, st)
    assert st.shadow_rays == st.primary_hits and st.bounce_rays > 0


@pytest.mark.parametrize("cam,gate", [("A", 0), ("B", 1), ("D", 1)])
def test_render_second_bounce_extension(eng, vxo, cam, gate):
    """bounce_depth = 2 (BASELINE config 5; an extension beyond the reference, defined in include/vxrt.h and restated
    in oracle/vxo_render.c): sample rays that hit spawn one more ray; more bounce rays than with depth 1, same bits
    as the oracle."""
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, 256, 256, 256, 32)
    _upload(eng[1], w)
    kw = dict(frame_number=4, shadow=1, bounce_samples=2, bounce_all_hits=gate)
    one = _render_both(eng, vxo, w, 192, 108, cam, **kw)
    two = _render_both(eng, vxo, w, 192, 108, cam, bounce_depth=2, **kw)
    _assert_frame_equal(*one)
    _assert_frame_equal(*two)
    assert two[4].bounce_rays > one[4].bounce_rays and two[4].primary_rays == one[4].primary_rays
    assert not np.array_equal(one[1], two[1])


@pytest.mark.parametrize("frame", [0, 1])
def test_render_checkerboard_and_debug_view(eng, vxo, frame):
    """The shipped configuration: DEBUG_VIEW quadrants + checkerboard (Renderer.cu:4-5); only half the pixels
    are written per frame, the rest keep their previous contents."""
    w = vxo.World.generate(vxo.GEN_HASH_HEIGHTFIELD, 128, 128, 128, 16)
    _upload(eng[1], w)
    cpu, fb, col, hit, st = _render_both(eng, vxo, w, 160, 90, "A", frame_number=frame, mode=1, checkerboard=1)
    _assert_frame_equal(cpu, fb, col, hit, st)
    assert st.primary_rays < 160 * 90
    cpu, fb, col, hit, st = _render_both(eng, vxo, w, 160, 90, "A", frame_number=frame, mode=0, checkerboard=1,
                                         shadow=1, bounce_samples=2)
    _assert_frame_equal(cpu, fb, col, hit, st)


def test_render_ortho_and_f8(eng, vxo):
    w = helpers.random_voxel_world(vxo, (128, 64, 128), 8, 0.03, 11)
    _upload(eng[1], w)
    cpu, fb, col, hit, st = _render_both(eng, vxo, w, 128, 96, "A", ortho=1, ortho_size=(30.0, 30.0), shadow=1)
    _assert_frame_equal(cpu, fb, col, hit, st)
    cpu, fb, col, hit, st = _render_both(eng, vxo, w, 133, 77, "B", shadow=1, bounce_samples=1)   # ragged frame
    _assert_frame_equal(cpu, fb, col, hit, st)


def test_context_frame_counter_matches_reference_increment(eng, vxo):
    """RenderScreen copies hFrameInfo then increments FrameNumber (Renderer.cu:310,322): frames see 0,1,2..."""
    vx, ctx, torch = eng
    import ctypes as C
    w = vxo.World.generate(vxo.GEN_HASH_HEIGHTFIELD, 128, 128, 128, 16)
    c2 = vx.Context(0)
    c2.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
    pos, f, u, r = helpers.camera("A", w.dims, vxo)
    p0 = vxo.make_params(160, 90, pos, f, u, r)
    c2.SetEnvironment(list(p0.env.light_dir), list(p0.env.light_color), list(p0.env.ambient))
    for n in range(3):
        d_fb = torch.full((90, 160, 4), 255, dtype=torch.uint8, device="cuda")
        c2.RenderScreen(160, 90, d_fb, pos, f, u, r, vx.RenderOptions(checkerboard=True, bounce_samples=1))
        p = vxo.make_params(160, 90, pos, f, u, r, frame_number=n, checkerboard=1, bounce_samples=1)
        assert np.array_equal(d_fb.cpu().numpy(), w.render(p)["fb"])
    c2.close()


@pytest.mark.parametrize("count,rows", [(2, 16), (8, 16), (3, 8), (4, 12), (5, 1)])
def test_strip_sharding_reassembles_the_frame(eng, vxo, count, rows):
    """Each shard renders its interleaved strips into a packed buffer; de-interleaving the shard buffers gives
    the single-GPU frame byte for byte."""
    vx, ctx, torch = eng
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, 256, 256, 256, 32)
    _upload(ctx, w)
    W, H = 192, 108
    pos, f, u, r = helpers.camera("A", w.dims, vxo)
    full = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    base = dict(shadow=True, bounce_samples=1, frame_number=2)
    ctx.RenderScreen(W, H, full, pos, f, u, r, vx.RenderOptions(**base))
    assert ctx.frame_stats().primary_rays == W * H
    max_rows = max(vx.compact_rows(H, rows, count, i) for i in range(count))
    stride = max_rows * W * 4
    shards = torch.zeros((count, stride), dtype=torch.uint8, device="cuda")
    total_primary = 0
    for i in range(count):
        ctx.RenderScreen(W, H, shards[i], pos, f, u, r,
                         vx.RenderOptions(strip_rows=rows, strip_count=count, strip_index=i, compact=True, **base))
        total_primary += ctx.frame_stats().primary_rays
    out = torch.zeros_like(full)
    ctx.deinterleave_strips(W, H, rows, count, shards, stride, out)
    torch.cuda.synchronize()
    assert total_primary == W * H
    assert torch.equal(out, full)
    p = vxo.make_params(W, H, pos, f, u, r, frame_number=2, shadow=1, bounce_samples=1)
    assert np.array_equal(full.cpu().numpy(), w.render(p, fb=np.zeros((H, W, 4), np.uint8))["fb"])


def test_tile_hand_out_order_never_changes_the_frame(eng, vxo):
    """The persistent kernel hands out 8x8 tiles in an order (device-made horizon-first schedule by default, or the
    caller's permutation).  Frames large enough for the queue to matter (> waves tiles) must come out identical
    to the oracle whatever the order, also for a sharded launch grid."""
    vx, ctx, torch = eng
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, 256, 256, 256, 32)
    _upload(ctx, w)
    W, H = 1004, 516   # 126 x 65 tiles, ragged on both edges
    ntiles = ((W + 7) // 8) * ((H + 7) // 8)
    pos, f, u, r = helpers.camera("A", w.dims, vxo)
    p = vxo.make_params(W, H, pos, f, u, r, frame_number=3, shadow=1, bounce_samples=1)
    want = w.render(p, fb=np.zeros((H, W, 4), np.uint8))["fb"]
    base = dict(shadow=True, bounce_samples=1, frame_number=3)
    rng = np.random.default_rng(5)
    orders = {
        "row-major": (dict(tile_schedule=False), None),
        "device schedule": (dict(tile_schedule=True), None),
        "host schedule": (dict(), torch.from_numpy(vx.tile_schedule(W, range(H), f, u, r, 90.0, H).astype(np.int32)).cuda()),
        "random": (dict(), torch.from_numpy(rng.permutation(ntiles).astype(np.int32)).cuda()),
    }
    for name, (kw, order) in orders.items():
        fb = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        ctx.RenderScreen(W, H, fb, pos, f, u, r, vx.RenderOptions(**base, **kw), tile_order=order)
        st = ctx.frame_stats()
        assert st.primary_rays == W * H, name
        assert np.array_equal(fb.cpu().numpy(), want), name
    # sharded launch grid (rows of the shard only) with the device schedule
    count, rows = 2, 16
    max_rows = max(vx.compact_rows(H, rows, count, i) for i in range(count))
    stride = max_rows * W * 4
    shards = torch.zeros((count, stride), dtype=torch.uint8, device="cuda")
    for i in range(count):
        ctx.RenderScreen(W, H, shards[i], pos, f, u, r,
                         vx.RenderOptions(strip_rows=rows, strip_count=count, strip_index=i, compact=True, **base))
    out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    ctx.deinterleave_strips(W, H, rows, count, shards, stride, out)
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("W,H", [(8, 8), (20, 8), (72, 40), (264, 136), (1000, 24)])
def test_tile_queue_hands_out_every_tile_once(eng, vxo, W, H):
    """The tile queue is cut into interleaved shards with a counter each (queue_take): frames with fewer tiles than shards,
    with tile counts that are no multiple of a shard round, and launches of several such views must trace every pixel
    exactly once -- the ray counter equals the pixel count -- and equal the oracle."""
    vx, ctx, torch = eng
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, 128, 128, 128, 16)
    _upload(ctx, w)
    pos, f, u, r = helpers.camera("A", w.dims, vxo)
    p = vxo.make_params(W, H, pos, f, u, r, frame_number=1)
    want = w.render(p, fb=np.zeros((H, W, 4), np.uint8))["fb"]
    ctx.frame_stats()
    for nviews in (1, 3, 16):
        views = [dict(fb=torch.full((H, W, 4), 9, dtype=torch.uint8, device="cuda"), origin=pos, fwd=f, up=u, right=r, frame_number=1)
                 for _ in range(nviews)]
        ctx.RenderViews(W, H, views, vx.RenderOptions())
        assert ctx.kernel_for_launch(W, H, vx.RenderOptions(), nviews=nviews) == 7
        assert ctx.frame_stats().primary_rays == W * H * nviews
        for v in views:
            assert np.array_equal(v["fb"].cpu().numpy(), want)


@pytest.mark.parametrize("variant", [4, 1])
def test_multi_view_launch_equals_single_view_launches(eng, vxo, variant):
    """vxrt_render_views: several views in one launch (the queue runs on from one view's tiles into the next's).
    Every view must be byte for byte the frame RenderScreen produces for it -- frame, hit-index AOV and colour AOV --
    for plain, checkerboard (per-view frame numbers) and strip-sharded launches, and equal to the oracle."""
    vx, ctx, torch = eng
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, 256, 256, 256, 32)
    _upload(ctx, w)
    W, H = 724, 412   # 91 x 52 tiles per view: more tiles than waves, ragged edges
    cams = ["A", "B", "D", "C", "A"]
    frames = [3, 4, 7, 8, 11]
    default = ctx.kernel_variant
    try:
        ctx.set_kernel_variant(variant)
        for kw in (dict(shadow=True, bounce_samples=1), dict(shadow=True, bounce_samples=2, bounce_depth=2, checkerboard=True),
                   dict(mode=1), dict(shadow=True, bounce_samples=1, strip_rows=16, strip_count=3, strip_index=1, compact=True)):
            rows = vx.compact_rows(H, 16, 3, 1) if kw.get("compact") else H
            singles, views = [], []
            ctx.frame_stats()   # counters accumulate until read: start from zero
            stale =torch.from_numpy(np.random.default_rng(3).integers(0, 255, size=(rows, W, 4), dtype=np.uint8)).cuda()
            for cam, fn in zip(cams, frames):
                pos, f, u, r = helpers.camera(cam, w.dims, vxo)
                fb, hit = stale.clone(), torch.full((rows, W), -7, dtype=torch.int64, device="cuda")
                col = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda")
                ctx.RenderScreen(W, H, fb, pos, f, u, r, vx.RenderOptions(frame_number=fn, **kw), color_aov=col, hit_aov=hit)
                singles.append((fb, hit, col))
                views.append(dict(fb=stale.clone(), origin=pos, fwd=f, up=u, right=r, frame_number=fn,
                                  hit_aov=torch.full((rows, W), -7, dtype=torch.int64, device="cuda"),
                                  color_aov=torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda")))
            rays_single = ctx.frame_stats().total_rays()
            ctx.RenderViews(W, H, views, vx.RenderOptions(**kw))
            assert ctx.frame_stats().total_rays() == rays_single
            for (fb, hit, col), v in zip(singles, views):
                assert torch.equal(v["fb"], fb) and torch.equal(v["hit_aov"], hit) and torch.equal(v["color_aov"], col)
        # and against the oracle for one of them
        pos, f, u, r = helpers.camera("D", w.dims, vxo)
        p = vxo.make_params(W, H, pos, f, u, r, frame_number=7, shadow=1, bounce_samples=1)
        views = [dict(fb=torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda"), origin=pos, fwd=f, up=u, right=r,
                      frame_number=7) for _ in range(2)]
        ctx.RenderViews(W, H, views, vx.RenderOptions(shadow=True, bounce_samples=1))
        want = w.render(p, fb=np.zeros((H, W, 4), np.uint8))["fb"]
        assert np.array_equal(views[0]["fb"].cpu().numpy(), want) and np.array_equal(views[1]["fb"].cpu().numpy(), want)
        with pytest.raises(vx.VxrtError):
            ctx.RenderViews(W, H, views * 9, vx.RenderOptions())   # 18 views: more than one launch takes
    finally:
        ctx.set_kernel_variant(default)
        ctx.frame_stats()


def test_multi_view_launch_sixteen_views_back_to_back(eng, vxo):
    """The largest launch (16 views, bench.py's default step) issued 20 times without a host sync in between: more
    launches than the context has argument slots (16) or tile counters, each with its own frame numbers and its own
    output buffers.  Every frame of every launch equals the single-view render of that frame."""
    vx, ctx, torch = eng
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, 256, 256, 256, 32)
    _upload(ctx, w)
    W, H = 200, 120
    opts = dict(shadow=True, bounce_samples=1)
    names = ["A", "B", "C", "D"]
    cams = [helpers.camera(n, w.dims, vxo) for n in names]
    launches = []
    for k in range(20):
        views = []
        for j in range(16):
            pos, f, u, r = cams[(k + j) % 4]
            views.append(dict(fb=torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda"), origin=pos, fwd=f, up=u, right=r,
                              frame_number=1 + 16 * k + j))
        ctx.RenderViews(W, H, views, vx.RenderOptions(**opts))
        launches.append(views)
    torch.cuda.synchronize()
    ref = {}
    for k in (0, 7, 15, 16, 19):            # spot-check launches on both sides of the slot ring's wrap
        for j in (0, 5, 15):
            fn = 1 + 16 * k + j
            pos, f, u, r = cams[(k + j) % 4]
            fb = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            ctx.RenderScreen(W, H, fb, pos, f, u, r, vx.RenderOptions(frame_number=fn, **opts))
            assert torch.equal(launches[k][j]["fb"], fb), (k, j)
            ref[fn] = fb
    # one of them against the oracle
    pos, f, u, r = cams[(19 + 15) % 4]
    p = vxo.make_params(W, H, pos, f, u, r, frame_number=1 + 16 * 19 + 15, shadow=1, bounce_samples=1)
    assert np.array_equal(launches[19][15]["fb"].cpu().numpy(), w.render(p, fb=np.zeros((H, W, 4), np.uint8))["fb"])
    ctx.frame_stats()


@pytest.mark.parametrize("count,rows,views", [(2, 16, 3), (8, 16, 16), (3, 8, 5)])
def test_deinterleave_views_equals_per_view_deinterleave(eng, count, rows, views):
    """vxrt_deinterleave_views: the strips of all views of a multi-view step in one launch = one
    vxrt_deinterleave_strips per view (what the gather's root runs), on a ragged frame height."""
    vx, ctx, torch = eng
    W, H = 256, 150
    shard_bytes = max(vx.compact_rows(H, rows, count, i) for i in range(count)) * W * 4
    step_bytes = views * shard_bytes
    g = torch.Generator(device="cuda").manual_seed(7)
    shards = torch.randint(0, 256, (count, step_bytes), dtype=torch.uint8, device="cuda", generator=g)
    want = torch.zeros((views, H, W, 4), dtype=torch.uint8, device="cuda")
    for j in range(views):
        ctx.deinterleave_strips(W, H, rows, count, shards.data_ptr() + j * shard_bytes, step_bytes, want[j])
    got = torch.full((views, H, W, 4), 9, dtype=torch.uint8, device="cuda")
    ctx.deinterleave_views(W, H, rows, count, shards, step_bytes, shard_bytes, views, got, W * H * 4)
    assert torch.equal(got, want)
    with pytest.raises(vx.VxrtError):
        ctx.deinterleave_views(W, H, rows, count, shards, step_bytes, shard_bytes + 4, views, got, W * H * 4)


@pytest.mark.parametrize("gen,shape,factor", [(0, (128, 128, 128), 16), (2, (256, 256, 256), 32),
                                              (1, (128, 128, 128), 16), (2, (64, 64, 128), 8),
                                              (1, (512, 64, 64), 8)])
def test_device_world_builder_matches_oracle(eng, vxo, gen, shape, factor):
    """vxrt_build_world_procedural produces the oracle's tables bit for bit (PERLIN_REF included: every float
    op in the generator is exactly specified)."""
    vx, ctx, _ = eng
    info = ctx.build_world(gen, *shape, factor)
    w = vxo.World.generate(gen, *shape, factor)
    got = ctx.download_world()
    assert info.nslots == w.nslots and tuple(info.cdims) == w.cdims
    assert np.array_equal(got["coarse_bits"], w.coarse_bits)
    assert np.array_equal(got["brick_slot"], w.brick_slot)
    assert np.array_equal(got["bounds"], w.bounds)
    assert np.array_equal(got["pool"], w.pool)


@pytest.mark.parametrize("shape,factor", [((64, 128, 256), 8), ((256, 128, 384), 16), ((512, 256, 768), 32)])
def test_uploaded_tables_come_back_in_the_reference_order(eng, vxo, shape, factor):
    """The tables cross the C ABI in the reference's tiled bit order and live in HBM in the library's own linear order:
    what vxrt_upload_world takes, vxrt_download_world gives back bit for bit (coarse bits, slots, extents, every brick),
    on grids whose three dimensions differ, and the re-ordered world traces like the oracle's."""
    vx, ctx, _ = eng
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, *shape, factor)
    _upload(ctx, w)
    got = ctx.download_world()
    assert np.array_equal(got["coarse_bits"], w.coarse_bits)
    assert np.array_equal(got["brick_slot"], w.brick_slot)
    assert np.array_equal(got["bounds"], w.bounds)
    assert np.array_equal(got["pool"], w.pool)
    rng = np.random.default_rng(shape[0] + factor)
    o = (rng.random((20000, 3)) * np.array(shape)).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    _assert_batch_equal(ctx.Raytrace(o, d), w.trace_batch(o, d))


def test_errors_are_reported_not_swallowed(eng, vxo):
    vx, ctx, torch = eng
    c2 = vx.Context(0)
    with pytest.raises(vx.VxrtError):
        c2.Raytrace([(0, 0, 0)], [(1, 0, 0)])            # no world resident
    with pytest.raises(vx.VxrtError):
        c2.build_world(0, 100, 100, 100, 32)               # not a multiple of the brick edge
    with pytest.raises(vx.VxrtError):
        c2.build_world(0, 128, 128, 128, 32)               # coarse dims not multiples of 8
    c2.close()


@pytest.mark.parametrize("cells_x,density", [(8192, 0.000004), (2048, 0.00005), (1024, 0.00003)])
def test_wide_grids_run_the_product_kernels(eng, vxo, cells_x, density):
    """Coarse grids beyond the packed step counters of the tracer (11-bit fields: 1022 cells) and long enough for a single
    walk to reach DDARayTraversal's MAX_STEPS (VolumeRaytracer.cuh:235): 8192, 2048 and 1024 cells along x at f = 8.  The
    product kernels run them (no fallback): rays along the long axis -- walks of thousands of cells, the counters re-armed
    on the way, rays that end after 2048 steps of one walk without a hit -- through the queue kernel, one ray per lane and
    the straightforward loops, probe counters included; and frames through the persistent render kernel."""
    vx, ctx, torch = eng
    X = cells_x * 8
    rng = np.random.default_rng(cells_x)
    v = np.zeros((X, 64, 64), bool)
    n_vox = int(X * 64 * 64 * density)
    v[rng.integers(0, X, n_vox), rng.integers(0, 64, n_vox), rng.integers(0, 64, n_vox)] = True
    v[:, 0, :] = True                  # a floor, so that frames have something to shade
    w = vxo.World.from_voxels(v, 8)
    assert w.cdims[0] == cells_x
    _upload(ctx, w)
    n = 60000
    o, d = helpers.mixed_rays(w.dims, n, cells_x)
    d[::2, 1:] *= np.float32(0.002)    # half of the rays nearly along the long axis
    d[::2, 1] = np.abs(d[::2, 1])      # ... and not into the floor
    o[::2, 1] = np.float32(8.0) + np.abs(o[::2, 1]) % np.float32(40.0)
    cpu = w.trace_batch(o, d)
    assert int((cpu["steps"] > 900).sum()) > 100       # walks about as long as a field's range, or beyond it
    if cells_x >= 2048:
        assert int(((cpu["steps"] >= 2048) & (cpu["hit"] == 0)).sum()) > 100   # walks that end by MAX_STEPS
    small = vx.Context(0)
    small.set_persistent_waves_per_cu(1)
    try:
        _upload(small, w)
        for c, variant in ((ctx, 4), (small, 4), (ctx, 1)):
            c.set_kernel_variant(variant)
            g = c.Raytrace(o, d, want_stats=True)
            _assert_batch_equal(g, cpu)
            assert (g["stats"].coarse_probes, g["stats"].brick_entries, g["stats"].fine_probes) == (
                cpu["stats"].coarse_probes, cpu["stats"].brick_entries, cpu["stats"].fine_probes)
            _assert_batch_equal(c.Raytrace(o, d), cpu)
            c.set_kernel_variant(4)
    finally:
        small.close()
    shaded = vx.RenderOptions(shadow=True, bounce_samples=1)
    assert ctx.kernel_for_launch(1920, 1080, shaded, nviews=16) == 7 and ctx.kernel_for_launch(160, 96, shaded) == 7
    _assert_frame_equal(*_render_both(eng, vxo, w, 160, 96, "A", frame_number=3, shadow=1, bounce_samples=1))
    _assert_frame_equal(*_render_both(eng, vxo, w, 200, 64, "D", frame_number=2, shadow=1, bounce_samples=2, bounce_all_hits=1))


def test_kernel_for_launch_reports_the_kernel(eng, vxo):
    """vxrt_kernel_for_launch: the default runs the persistent kernel (7) for every launch shape -- one view or sixteen, a
    1080p frame of primary rays or a 1/8 shard -- and the cross-check variant is reported as itself; the kernels of
    earlier rounds are refused."""
    vx, ctx, torch = eng
    _upload(ctx, vxo.World.generate(vxo.GEN_INT_TERRAIN, 256, 256, 256, 32))
    default = ctx.kernel_variant
    try:
        ctx.set_kernel_variant(4)
        shaded = vx.RenderOptions(shadow=True, bounce_samples=1)
        for W, H, o, nv in ((1920, 1080, shaded, 16), (1920, 1080, shaded, 0), (1920, 1080, vx.RenderOptions(), 0),
                            (1280, 720, vx.RenderOptions(mode=1, checkerboard=True), 0), (3840, 2160, vx.RenderOptions(), 0),
                            (3840, 2160, vx.RenderOptions(shadow=True, strip_rows=16, strip_count=8, strip_index=3, compact=True), 0)):
            assert ctx.kernel_for_launch(W, H, o, nviews=nv) == 7
        for v in (1, 7):
            ctx.set_kernel_variant(v)
            assert ctx.kernel_for_launch(640, 480, shaded) == v
        assert ctx.KERNEL_NAMES[7] == "k_render_persist2" and ctx.KERNEL_NAMES[1] == "k_render"
        for v in (0, 2, 3, 5, 6, 8, -1):
            with pytest.raises(vx.VxrtError):
                ctx.set_kernel_variant(v)
    finally:
        ctx.set_kernel_variant(default)


def test_speculative_loads_stay_inside_the_allocators_slack(eng, vxo, tmp_path):
    """The slack contract of the tracer (csrc/vxrt_wave2.hpp): a lane that has just stepped out of the coarse grid or of a
    brick issues one more occupancy load before it stops, and every path that makes a world resident (upload, device
    builder, brickmap file, chunk streaming) goes through the allocator that leaves addressable slack around both bit
    tables.  The probe-counting kernels classify every load address: frames of the four cameras (sky rays leave through
    the top face: camera D grazing, B from outside) and a batch with far-face starts and rays leaving through every face
    must produce loads in the slack (the test does exercise it) and NONE outside what the allocator made addressable.
    Negative control: with the guard told to pretend the tables have no slack, exactly those loads count as stray -- so a
    future world path that forgets the slack fails here, not with a fault in a user's frame."""
    vx, _, torch = eng
    ctx = vx.Context(0)
    small = vx.Context(0)
    small.set_persistent_waves_per_cu(1)   # batches of this size take the queue kernel
    try:
        w = vxo.World.generate(vxo.GEN_INT_TERRAIN, 256, 256, 256, 32)
        o, d = helpers.mixed_rays(w.dims, 200000, 3)
        inv = float(np.float32(1.0) / np.sqrt(np.float32(3.0)))
        path = str(tmp_path / "guard.vxb")

        def exercise(c):
            slack = stray = 0
            c.SetEnvironment((inv, inv, inv), (2, 2, 2), (0.5, 0.5, 0.5))
            c.frame_stats()
            for cam in "ABCD":
                pos, f, u, r = helpers.camera(cam, w.dims, vxo)
                fb = torch.zeros((120, 200, 4), dtype=torch.uint8, device="cuda")
                c.RenderScreen(200, 120, fb, pos, f, u, r, vx.RenderOptions(shadow=True, bounce_samples=1, frame_number=2,
                                                                           collect_stats=True))
                st = c.frame_stats()
                slack, stray = slack + st.guard_slack_loads, stray + st.guard_stray_loads
            st = c.Raytrace(o, d, want_stats=True)["stats"]
            return slack + st.guard_slack_loads, stray + st.guard_stray_loads

        def load_from_file(c):
            c.load_world(path)

        def stream_from_file(c):
            c.stream_open(path, 4096)
            c.stream_focus((128.0, 128.0, 128.0), 1000.0)

        paths = [("upload", lambda c: _upload(c, w)), ("device builder", lambda c: c.build_world(vxo.GEN_INT_TERRAIN, 256, 256, 256, 32)),
                 ("brickmap file", load_from_file), ("chunk streaming", stream_from_file)]
        _upload(ctx, w)
        ctx.save_world(path)
        for c in (ctx, small):
            for name, make in paths:
                make(c)
                c.guard_pretend_no_slack(False)
                slack, stray = exercise(c)
                assert stray == 0, (name, stray)
                assert slack > 0, name            # the frames and the batch do send lanes one load beyond the tables
                c.guard_pretend_no_slack(True)
                slack2, stray2 = exercise(c)
                # (the count is not a constant of the workload: parked lanes re-issue their last load with every probe of
                # their wave, so it depends on how the rays met in the waves)
                assert slack2 == 0 and 0.7 * slack <= stray2 <= 1.4 * slack, (name, slack, stray2)
                c.guard_pretend_no_slack(False)
                if name == "chunk streaming":
                    c.stream_close()
    finally:
        ctx.close()
        small.close()
