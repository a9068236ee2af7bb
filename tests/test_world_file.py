"""Brickmap file (include/vxrt.h, "brickmap file"): header checks on the CPU, save -> load round trip and
rejection of damaged files on the GPU."""
import os
import struct

import numpy as np
import pytest

from tests import helpers

HEADER = "<8sIIiiiiQQQQQQQQQQQ"   # magic, version, header bytes, factor, cdims[3], ncells, nslots, 3 sizes, 3 sums, 3 sums of running sums
HEADER_BYTES = 120


def _header(factor=8, cdims=(8, 8, 8), nslots=3, version=2, magic=b"VXBRKMAP", sizes=None):
    ncells = cdims[0] * cdims[1] * cdims[2]
    bw = factor ** 3 // 32
    sizes = sizes or (((ncells + 31) // 32) * 4, ncells * 8, nslots * bw * 4)
    return struct.pack(HEADER, magic, version, struct.calcsize(HEADER), factor, *cdims, ncells, nslots, *sizes, 0, 0, 0, 0, 0, 0)


def test_header_is_120_bytes():
    assert struct.calcsize(HEADER) == HEADER_BYTES


def test_file_info_reads_a_header_and_rejects_bad_ones(tmp_path):
    import voxelengine_amd as vx
    good = tmp_path / "good.vxb"
    good.write_bytes(_header(factor=16, cdims=(16, 8, 24), nslots=5))
    info = vx.world_file_info(str(good))
    assert (info.factor, tuple(info.cdims), info.ncells, info.nslots) == (16, (16, 8, 24), 16 * 8 * 24, 5)
    assert info.hbm_bytes == 16 * 8 * 24 // 8 + 16 * 8 * 24 * 8 + 5 * 512
    cases = {
        "magic": _header(magic=b"NOTAMAP!"),
        "version": _header(version=1),             # the additive-checksum format of round 1 is not read any more
        "factor": _header(factor=12),
        "dims": _header(cdims=(8, 4, 8)),            # not a multiple of 8
        "sizes": _header(sizes=(64, 4096, 999)),     # pool size does not match nslots
        "slots": _header(nslots=8 * 8 * 8 + 1),      # more bricks than cells
        "short": _header()[:50],
    }
    for name, blob in cases.items():
        p = tmp_path / (name + ".vxb")
        p.write_bytes(blob)
        with pytest.raises(vx.VxrtError):
            vx.world_file_info(str(p))
    with pytest.raises(vx.VxrtError):
        vx.world_file_info(str(tmp_path / "missing.vxb"))


@pytest.fixture(scope="module")
def eng():
    import torch
    import voxelengine_amd as vx
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    ctx = vx.Context(0)
    yield vx, ctx, torch
    ctx.close()


def _frame(vx, ctx, torch, w, vxo):
    pos, f, u, r = helpers.camera("A", w.dims, vxo)
    fb = torch.zeros((90, 160, 4), dtype=torch.uint8, device="cuda")
    ctx.SetEnvironment((0.57735, 0.57735, 0.57735), (2, 2, 2), (0.5, 0.5, 0.5))
    ctx.RenderScreen(160, 90, fb, pos, f, u, r, vx.RenderOptions(shadow=True, bounce_samples=1, frame_number=2))
    return fb.cpu().numpy()


@pytest.mark.gpu
def test_save_load_round_trip(eng, vxo, tmp_path):
    vx, ctx, torch = eng
    w = vxo.World.generate(vxo.GEN_INT_TERRAIN, 256, 128, 256, 16)
    ctx.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
    want = _frame(vx, ctx, torch, w, vxo)
    path = str(tmp_path / "world.vxb")
    ctx.save_world(path)
    info = vx.world_file_info(path)
    assert (info.factor, tuple(info.cdims), info.nslots) == (16, tuple(w.cdims), w.pool.size // 128)
    assert os.path.getsize(path) == HEADER_BYTES + info.hbm_bytes

    other = vx.Context(0)
    try:
        got = other.load_world(path)
        assert (got.factor, tuple(got.cdims), got.nslots) == (info.factor, tuple(info.cdims), info.nslots)
        d = other.download_world()
        assert np.array_equal(d["coarse_bits"], w.coarse_bits)
        assert np.array_equal(d["brick_slot"], w.brick_slot)
        assert np.array_equal(d["bounds"], w.bounds.reshape(-1, 6))
        assert np.array_equal(d["pool"], w.pool)
        assert np.array_equal(_frame(vx, other, torch, w, vxo), want)
    finally:
        other.close()


@pytest.mark.gpu
def test_damaged_files_are_rejected_and_leave_no_world(eng, vxo, tmp_path):
    vx, ctx, torch = eng
    w = vxo.World.generate(vxo.GEN_HASH_HEIGHTFIELD, 128, 128, 128, 16)
    ctx.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
    path = str(tmp_path / "world.vxb")
    ctx.save_world(path)
    blob = bytearray(open(path, "rb").read())
    ncells = int(np.prod(w.cdims))
    meta_off = HEADER_BYTES + ((ncells + 31) // 32) * 4
    occupied = int(np.flatnonzero(w.brick_slot != 0xFFFFFFFF)[0])
    damaged = {}
    flipped = bytearray(blob)
    flipped[-5] ^= 0x40                                            # one bit of the pool
    damaged["checksum"] = bytes(flipped)
    damaged["truncated"] = bytes(blob[:len(blob) - 1000])
    wild = bytearray(blob)
    wild[meta_off + occupied * 8:meta_off + occupied * 8 + 4] = struct.pack("<I", 0x7FFFFFFF)   # slot outside the pool
    damaged["wild slot"] = bytes(wild)
    # two different pool words swapped: an additive checksum cannot see it, the sum of running sums does
    pool_off = meta_off + ncells * 8
    words = np.frombuffer(bytes(blob[pool_off:]), np.uint32).copy()
    i = int(np.flatnonzero(words != words[0])[0])
    words[0], words[i] = words[i], words[0]
    damaged["swapped words"] = bytes(blob[:pool_off]) + words.tobytes()
    # tight extents outside the brick / min above max in an occupied cell, with the checksums of the header recomputed
    # so that only the validation of the fields can catch them
    for label, ext in (("extent beyond brick", 31 << 15), ("min above max", (5 << 0) | (2 << 15))):
        bad = bytearray(blob)
        off = meta_off + occupied * 8 + 4
        bad[off:off + 4] = struct.pack("<I", ext)
        meta = np.frombuffer(bytes(bad[meta_off:pool_off]), np.uint32).astype(np.uint64)
        a = int(meta.sum(dtype=np.uint64))   # uint64 arithmetic wraps, as the file's sums do
        run = np.cumsum(meta, dtype=np.uint64)
        b = int(run.sum(dtype=np.uint64))
        hdr = list(struct.unpack(HEADER, bytes(bad[:HEADER_BYTES])))
        hdr[13], hdr[16] = a, b   # sum[1], sum2[1]
        bad[:HEADER_BYTES] = struct.pack(HEADER, *hdr)
        damaged[label] = bytes(bad)
    other = vx.Context(0)
    try:
        for name, data in damaged.items():
            p = str(tmp_path / (name.replace(" ", "_") + ".vxb"))
            open(p, "wb").write(data)
            with pytest.raises(vx.VxrtError):
                other.load_world(p)
            with pytest.raises(vx.VxrtError):   # nothing half-loaded stays resident
                other.world_info()
        other.load_world(path)                  # and the intact file still loads afterwards
        assert other.world_info().nslots == w.pool.size // 128
    finally:
        other.close()
