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


@pytest.mark.gpu
def test_headless_voxelapp_replays_a_camera_path(vxo, tmp_path):
    """A fly-through read from a file (one pose per frame): GetDirections per frame, checkerboard frames that keep
    half of their predecessor, every frame dumped as PPM; each dump equals the oracle's frame sequence."""
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    W, H, edge = 160, 96, 256
    poses = [((64.0, 230.0, 64.0), (-0.45, 0.7, 0.0)), ((70.5, 228.0, 66.0), (-0.5, 0.8, 0.0)),
             ((80.0, 220.25, 72.0), (-0.6, 1.0, 0.0))]
    path = tmp_path / "path.txt"
    path.write_text("# x y z eulerX eulerY eulerZ\n" + "".join(
        "%r %r %r %r %r %r\n" % (*p, *e) for p, e in poses) + "\n# end\n")
    prefix = str(tmp_path / "fly")
    out = subprocess.run([EXE, str(edge), "0", prefix, str(W), str(H), "1", str(path), "1"], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    w = vxo.World.generate(vxo.GEN_PERLIN_REF, edge, edge, edge, 32, nthreads=16)
    fb = np.full((H, W, 4), 255, np.uint8)
    for frame, (pos, euler) in enumerate(poses):
        f, u, r = vxo.get_directions(euler)
        p = vxo.make_params(W, H, tuple(np.float32(v) for v in pos), f, u, r, frame_number=frame, mode=vxo.MODE_SHADED,
                            checkerboard=1, shadow=1, bounce_samples=1)
        fb = w.render(p, fb=fb)["fb"]
        raw = open("%s_%04d.ppm" % (prefix, frame), "rb").read()
        head = b"P6\n%d %d\n255\n" % (W, H)
        assert raw.startswith(head)
        rgb = np.frombuffer(raw[len(head):], np.uint8).reshape(H, W, 3)
        assert np.array_equal(rgb, fb[:, :, [2, 1, 0]]), frame   # memory order is b,g,r,a
    assert np.array_equal(np.fromfile(prefix + ".bgra", np.uint8).reshape(H, W, 4), fb)
    bad = subprocess.run([EXE, str(edge), "0", prefix, str(W), str(H), "1", str(tmp_path / "none.txt")],
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode == 2


@pytest.mark.gpu
def test_headless_voxelapp_renders_several_poses_per_launch(vxo, tmp_path):
    """Graphics::RenderScreens (this build's addition to the reference-shaped API): the camera path rendered two
    poses per launch; every dumped frame equals the oracle's frame for that pose and FrameNumber."""
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    W, H, edge = 160, 96, 256
    poses = [((64.0, 230.0, 64.0), (-0.45, 0.7, 0.0)), ((70.5, 228.0, 66.0), (-0.5, 0.8, 0.0)),
             ((80.0, 220.25, 72.0), (-0.6, 1.0, 0.0))]
    path = tmp_path / "path.txt"
    path.write_text("".join("%r %r %r %r %r %r\n" % (*p, *e) for p, e in poses))
    prefix = str(tmp_path / "multi")
    out = subprocess.run([EXE, str(edge), "0", prefix, str(W), str(H), "1", str(path), "1", "2"], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    w = vxo.World.generate(vxo.GEN_PERLIN_REF, edge, edge, edge, 32, nthreads=16)
    for frame, (pos, euler) in enumerate(poses):
        f, u, r = vxo.get_directions(euler)
        p = vxo.make_params(W, H, tuple(np.float32(v) for v in pos), f, u, r, frame_number=frame, mode=vxo.MODE_SHADED,
                            shadow=1, bounce_samples=1)
        want = w.render(p, fb=np.zeros((H, W, 4), np.uint8))["fb"]
        raw = open("%s_%04d.ppm" % (prefix, frame), "rb").read()
        head = b"P6\n%d %d\n255\n" % (W, H)
        rgb = np.frombuffer(raw[len(head):], np.uint8).reshape(H, W, 3)
        assert np.array_equal(rgb, want[:, :, [2, 1, 0]]), frame


@pytest.mark.gpu
def test_two_frames_in_flight_equal_the_synchronous_frames(vxo, tmp_path):
    """Graphics::RenderScreenAsync / WaitFrame (this build's addition for interactive callers): a five-pose fly-through
    rendered with two frames in flight gives, frame by frame, the bytes of the synchronous RenderScreen loop, and both
    equal the oracle.  The world is built on the device in one step (device_world argument)."""
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    W, H = 200, 120
    poses = [((64.0, 230.0, 64.0), (-0.45, 0.7, 0.0)), ((70.5, 228.0, 66.0), (-0.5, 0.8, 0.0)),
             ((80.0, 220.25, 72.0), (-0.6, 1.0, 0.0)), ((90.0, 215.0, 80.0), (-0.7, 1.2, 0.0)),
             ((100.0, 240.0, 90.0), (-1.2, 1.4, 0.0))]
    path = tmp_path / "path.txt"
    path.write_text("".join("%r %r %r %r %r %r\n" % (*p, *e) for p, e in poses))
    runs = {}
    for flight in (1, 2):
        prefix = str(tmp_path / ("f%d" % flight))
        out = subprocess.run([EXE, "256", "0", prefix, str(W), str(H), "2", str(path), "1", "1", str(flight), "256x256x256"],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        assert "World built on the device" in out.stdout and "Mrays/s" in out.stdout
        runs[flight] = [open("%s_%04d.ppm" % (prefix, i), "rb").read() for i in range(len(poses))]
        assert np.array_equal(np.fromfile(prefix + ".bgra", np.uint8), np.fromfile(str(tmp_path / "f1.bgra"), np.uint8))
    assert runs[1] == runs[2]
    w = vxo.World.generate(vxo.GEN_PERLIN_REF, 256, 256, 256, 32, nthreads=16)
    head = b"P6\n%d %d\n255\n" % (W, H)
    for frame, (pos, euler) in enumerate(poses):
        f, u, r = vxo.get_directions(euler)
        p = vxo.make_params(W, H, tuple(np.float32(v) for v in pos), f, u, r, frame_number=frame, mode=vxo.MODE_SHADED,
                            shadow=1, bounce_samples=1)
        want = w.render(p, fb=np.zeros((H, W, 4), np.uint8))["fb"]
        rgb = np.frombuffer(runs[2][frame][len(head):], np.uint8).reshape(H, W, 3)
        assert np.array_equal(rgb, want[:, :, [2, 1, 0]]), frame
