"""Temporal accumulation of the stochastic occlusion term (the reference lists it as to do, README.md:19;
defined by this build in include/vxrt.h, vxrt_render_flags.d_accum, and restated in oracle/vxo_render.c).

CPU: the oracle's accumulating render against the definition evaluated in numpy from its own per-frame colour AOV.
GPU: every render kernel against the oracle, bit for bit (framebuffer and history), over a sequence of frames with a reset."""
import numpy as np
import pytest

from tests import helpers

f32 = np.float32
W, H = 160, 96


def _scene(vxo):
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, 256, 256, 256, 32)
    pos, f, u, r = helpers.camera("A", w.dims, vxo)
    return w, pos, f, u, r


def _tonemap_bgra(c):
    """Tonemap + setPixelColor (Renderer.cu:170-177,72-87) in numpy binary32."""
    t = (c / (c + f32(1.0))).astype(f32)
    t = np.minimum(np.maximum(t, f32(0)), f32(1))
    px = (t * f32(255)).astype(np.uint8)       # truncation
    return px[..., ::-1]                        # b, g, r


def test_oracle_accumulation_is_the_running_mean_of_the_frames(vxo):
    w, pos, f, u, r = _scene(vxo)
    acc = np.zeros((H, W, 4), f32)
    total = np.zeros((H, W, 3), f32)
    n = np.zeros((H, W), f32)
    fb = np.zeros((H, W, 4), np.uint8)
    for frame in range(1, 6):
        reset = frame == 4        # a camera cut: the history starts again
        p = vxo.make_params(W, H, pos, f, u, r, frame_number=frame, shadow=1, bounce_samples=1, bounce_all_hits=1)
        plain = w.render(p, fb=np.zeros((H, W, 4), np.uint8), want_color=True, want_hit=True)
        p = vxo.make_params(W, H, pos, f, u, r, frame_number=frame, shadow=1, bounce_samples=1, bounce_all_hits=1)
        out = w.render(p, fb=fb, accum=acc, accum_reset=reset)
        hit = plain["hit"] >= 0
        # the pre-tonemap colour of the frame, recovered from the plain frame's tonemapped AOV is not exact; evaluate the
        # definition on the HISTORY instead: it must hold sum and count, and the stored pixel must be tonemap(sum / n)
        if reset:
            n[:] = 0
        n[hit] += 1
        assert np.array_equal(acc[..., 3], n)
        mean = (acc[..., :3] / np.maximum(acc[..., 3:4], f32(1))).astype(f32)
        want = _tonemap_bgra(mean)
        got = out["fb"]
        cross = (H // 2, W // 2)
        mask = hit.copy()
        mask[cross] = False   # the crosshair is drawn over the centre pixel
        assert np.array_equal(got[..., :3][mask], want[mask])
        assert np.array_equal(got[..., :3][~hit], plain["fb"][..., :3][~hit])   # miss pixels: written as without history
        if frame == 1 or reset:   # first frame of a history: the frame itself
            assert np.array_equal(got, plain["fb"])
    # the noise of the single bounce sample averages out: consecutive accumulated frames differ less and less
    assert acc[..., 3].max() == 2


def test_accumulation_reduces_frame_to_frame_noise(vxo):
    w, pos, f, u, r = _scene(vxo)
    acc = np.zeros((H, W, 4), f32)
    prev, deltas = None, []
    for frame in range(1, 9):
        p = vxo.make_params(W, H, pos, f, u, r, frame_number=frame, shadow=1, bounce_samples=1, bounce_all_hits=1)
        fb = w.render(p, fb=np.zeros((H, W, 4), np.uint8), accum=acc)["fb"].astype(np.int32)
        if prev is not None:
            deltas.append(np.abs(fb - prev).mean())
        prev = fb
    assert deltas[-1] < 0.5 * deltas[0]


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [4, 1])
def test_gpu_accumulation_equals_the_oracle(vxo, variant):
    import torch
    import voxelengine_amd as vx
    w, pos, f, u, r = _scene(vxo)
    ctx = vx.Context(0)
    try:
        ctx.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
        inv = float(f32(1.0) / np.sqrt(f32(3.0)))
        ctx.SetEnvironment((inv, inv, inv), (2, 2, 2), (0.5, 0.5, 0.5))
        ctx.SetFOV(90.0)
        ctx.set_kernel_variant(variant)
        acc_c = np.zeros((H, W, 4), f32)
        fb_c = np.zeros((H, W, 4), np.uint8)
        acc_g = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        fb_g = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        for frame in range(1, 6):
            reset = frame == 4
            p = vxo.make_params(W, H, pos, f, u, r, frame_number=frame, shadow=1, bounce_samples=2)
            w.render(p, fb=fb_c, accum=acc_c, accum_reset=reset)
            ctx.RenderScreen(W, H, fb_g, pos, f, u, r, vx.RenderOptions(shadow=True, bounce_samples=2, frame_number=frame),
                             accum=acc_g, accum_reset=reset)
            assert np.array_equal(fb_g.cpu().numpy(), fb_c), frame
            assert np.array_equal(acc_g.cpu().numpy().view(np.uint32), acc_c.view(np.uint32)), frame
        # a multi-view launch has no per-view history: clean error
        with pytest.raises(vx.VxrtError):
            fl_views = [dict(fb=fb_g, origin=pos, fwd=f, up=u, right=r, frame_number=1)]
            o = vx.RenderOptions(shadow=True)
            o.extra["accum"] = acc_g
            ctx.RenderViews(W, H, fl_views, o)
    finally:
        ctx.close()
