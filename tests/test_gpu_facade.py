"""The C++ facade (namespace GPUDDA, include/GPUDDA/*.h) driven the way VoxelApp/main.cu drives the reference:
CreateVoxels -> GenerateLowresVoxelBuffer -> Upload* -> SetEnvironment/SetFOV -> RenderScreen per frame -> D2H copy.
The frame the example writes must equal the oracle's frame for the same world, camera and frame numbers."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "voxelapp_headless")


@pytest.mark.gpu
@pytest.mark.parametrize("shaded", [0, 1])
def test_headless_voxelapp_matches_oracle(vxo, tmp_path, shaded):
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    W, H, edge = 160, 96, 256
    prefix = str(tmp_path / "frame")
    out = subprocess.run([EXE, str(edge), "2", prefix, str(W), str(H), str(shaded)], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Avg FPS" in out.stdout and "Raytracing time" in out.stdout
    got = np.fromfile(prefix + ".bgra", np.uint8).reshape(H, W, 4)

    w = vxo.World.generate(vxo.GEN_PERLIN_REF, edge, edge, edge, 32, nthreads=16)
    f, u, r = vxo.get_directions((-0.45, 0.7, 0.0))
    pos = (np.float32(edge) * np.float32(0.25), np.float32(edge) * np.float32(0.9), np.float32(edge) * np.float32(0.25))
    fb = np.full((H, W, 4), 255, np.uint8)
    for frame in (0, 1):  # RenderScreen copies FrameNumber, then increments it (Renderer.cu:310,322)
        if shaded:
            p = vxo.make_params(W, H, pos, f, u, r, frame_number=frame, mode=vxo.MODE_SHADED, checkerboard=1, shadow=1,
                                bounce_samples=1)
        else:
            p = vxo.make_params(W, H, pos, f, u, r, frame_number=frame, mode=vxo.MODE_DEBUG, checkerboard=1)
        fb = w.render(p, fb=fb)["fb"]
    assert np.array_equal(got, fb)
    # the batch rays the example prints: straight down from the camera hits, straight up misses
    lines = [l for l in out.stdout.splitlines() if l.startswith("ray ")]
    assert len(lines) == 4 and "valid=1" in lines[0] and "valid=0" in lines[2]
