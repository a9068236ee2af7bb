"""Chunk streaming (the reference lists "Chunking" / "Chunk data streaming" as to do, README.md:15,20;
defined by this build in include/vxrt.h, vxrt_stream_*): the coarse tables of the whole world resident, brick data read
from a brickmap file only for the chunks (8x8x8 tiles of coarse cells) near a focus point, a non-resident chunk reads as
empty space.  Every frame must equal the ORACLE's frame of the world truncated to exactly the resident chunks; with
everything resident it must equal the frame of the fully loaded world; eviction under a small pool keeps both true."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu
f32 = np.float32
W, H = 200, 120
X, Y, Z, F = 512, 256, 512, 16      # coarse 32 x 16 x 32 = 4 x 2 x 4 chunks


def _truncated(vxo, w, flags):
    """The oracle world with the bricks of non-resident chunks removed (cells unoccupied)."""
    coarse = w.coarse_bits.copy()
    slot = w.brick_slot.copy()
    for ch in np.flatnonzero(flags == 0):
        coarse[ch * 16:(ch + 1) * 16] = 0
        slot[ch * 512:(ch + 1) * 512] = 0xFFFFFFFF
    return vxo.World.wrap(w.factor, w.cdims, coarse, slot, w.bounds, w.pool)


def _render(vx, ctx, torch, cam, dims, vxo):
    pos, f, u, r = helpers.camera(cam, dims, vxo)
    fb = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    hit = torch.full((H, W), -1, dtype=torch.int64, device="cuda")
    ctx.RenderScreen(W, H, fb, pos, f, u, r, vx.RenderOptions(shadow=True, bounce_samples=1, frame_number=3), hit_aov=hit)
    return fb.cpu().numpy(), hit.cpu().numpy()


def _oracle(vxo, world, cam, dims):
    pos, f, u, r = helpers.camera(cam, dims, vxo)
    out = world.render(vxo.make_params(W, H, pos, f, u, r, frame_number=3, shadow=1, bounce_samples=1),
                       fb=np.zeros((H, W, 4), np.uint8), want_hit=True, nthreads=16)
    return out["fb"], out["hit"]


def test_streamed_frames_equal_the_oracle_on_the_resident_world(vxo, tmp_path):
    import torch
    import voxelengine_amd as vx
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, X, Y, Z, F, nthreads=16)
    dims = w.dims
    path = str(tmp_path / "world.vxb")
    full = vx.Context(0)
    ctx = vx.Context(0)
    try:
        inv = float(f32(1.0) / np.sqrt(f32(3.0)))
        for c in (full, ctx):
            c.SetEnvironment((inv, inv, inv), (2, 2, 2), (0.5, 0.5, 0.5))
            c.SetFOV(90.0)
        full.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
        full.save_world(path)
        nbricks = w.pool.size // (F ** 3 // 32)
        info = ctx.stream_open(path, nbricks)            # room for everything
        assert (info.factor, tuple(info.cdims)) == (F, tuple(w.cdims))
        nchunks = int(np.prod(w.cdims)) // 512
        assert ctx.stream_resident().sum() == 0
        fb, hit = _render(vx, ctx, torch, "A", dims, vxo)
        want_fb, want_hit = _oracle(vxo, _truncated(vxo, w, np.zeros(nchunks, np.uint8)), "A", dims)
        assert np.array_equal(fb, want_fb) and (hit == -1).all()      # nothing resident: empty space

        # a focus near camera A with a small radius: part of the world
        pos = helpers.camera("A", dims, vxo)[0]
        st = ctx.stream_focus(pos, 120.0)
        flags = ctx.stream_resident()
        assert 0 < st.chunks_resident == flags.sum() < st.chunks_occupied and st.chunks_loaded == st.chunks_resident
        assert st.bytes_read == st.bricks_resident * (F ** 3 // 8) and st.chunks_missing == 0
        for cam in ("A", "B"):
            fb, hit = _render(vx, ctx, torch, cam, dims, vxo)
            want_fb, want_hit = _oracle(vxo, _truncated(vxo, w, flags), cam, dims)
            assert np.array_equal(fb, want_fb), cam
            assert np.array_equal(hit, want_hit), cam

        # everything within reach: the streamed world IS the world
        st = ctx.stream_focus(pos, 1.0e6)
        assert st.chunks_resident == st.chunks_occupied and st.bricks_resident == nbricks and st.chunks_evicted == 0
        for cam in ("A", "D"):
            fb, hit = _render(vx, ctx, torch, cam, dims, vxo)
            ref_fb, ref_hit = _render(vx, full, torch, cam, dims, vxo)
            assert np.array_equal(fb, ref_fb) and np.array_equal(hit, ref_hit), cam
        # its tables are the world's tables up to the numbering of the slots
        d = ctx.download_world()
        assert np.array_equal(d["coarse_bits"], w.coarse_bits) and np.array_equal(d["bounds"], w.bounds.reshape(-1, 6))
        bw = F ** 3 // 32
        for cell in np.flatnonzero(w.brick_slot != 0xFFFFFFFF)[::97]:
            a, b = int(d["brick_slot"][cell]), int(w.brick_slot[cell])
            assert np.array_equal(d["pool"][a * bw:(a + 1) * bw], w.pool[b * bw:(b + 1) * bw])
        ctx.stream_close()
        with pytest.raises(vx.VxrtError):
            ctx.stream_focus(pos, 10.0)
    finally:
        ctx.close()
        full.close()


def test_a_small_pool_evicts_the_farthest_chunks_as_the_focus_moves(vxo, tmp_path):
    import torch
    import voxelengine_amd as vx
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, X, Y, Z, F, nthreads=16)
    dims = w.dims
    path = str(tmp_path / "world.vxb")
    ctx = vx.Context(0)
    try:
        inv = float(f32(1.0) / np.sqrt(f32(3.0)))
        ctx.SetEnvironment((inv, inv, inv), (2, 2, 2), (0.5, 0.5, 0.5))
        ctx.SetFOV(90.0)
        ctx.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
        ctx.save_world(path)
        nbricks = w.pool.size // (F ** 3 // 32)
        ctx.stream_open(path, nbricks // 3)               # a third of the world fits
        evicted = 0
        for step, fx in enumerate((0.1, 0.3, 0.5, 0.7, 0.9, 0.5)):    # fly across the world and back
            focus = (fx * X, 0.5 * Y, 0.5 * Z)
            st = ctx.stream_focus(focus, 150.0)
            evicted += int(st.chunks_evicted)
            flags = ctx.stream_resident()
            assert st.bricks_resident <= nbricks // 3 and st.chunks_resident == flags.sum()
            # every chunk inside the radius is resident unless the call says it could not fit
            pos = (focus[0], 0.9 * Y, focus[2])
            f, u, r = vx.GetDirections((-0.9, 0.3 * step, 0.0))
            fb = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            ctx.RenderScreen(W, H, fb, pos, f, u, r, vx.RenderOptions(shadow=True, bounce_samples=1, frame_number=step))
            want = _truncated(vxo, w, flags).render(
                vxo.make_params(W, H, pos, f, u, r, frame_number=step, shadow=1, bounce_samples=1),
                fb=np.zeros((H, W, 4), np.uint8), nthreads=16)["fb"]
            assert np.array_equal(fb.cpu().numpy(), want), step
        assert evicted > 0
    finally:
        ctx.close()
