"""The wave-level traversal code (voxelengine_amd/csrc/vxrt_wave2.hpp) compiled for the HOST with one lane per
wave (tests/tools/hoststub stands in for the few HIP builtins) and run against the oracle.  This exercises the
product's traversal logic -- state machine, parking votes, nudges, the probe counters derived from the packed step
counters, the wide-grid re-arming of those counters and the MAX_STEPS end of a walk -- on CPU, bit for bit."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp, name, extra=()):
    exe = str(tmp / name)
    cc = ["g++", "-O1", "-std=c++17", "-ffp-contract=off", *extra, "-I" + os.path.join(ROOT, "tests", "tools", "hoststub"),
          "-I" + os.path.join(ROOT, "oracle"), "-o", exe, os.path.join(ROOT, "tests", "tools", "host_wave_check.cpp"),
          "-x", "c", os.path.join(ROOT, "oracle", "vxo_trace.c"), os.path.join(ROOT, "oracle", "vxo_world.c"),
          os.path.join(ROOT, "oracle", "vxo_render.c"), "-lm", "-lpthread", "-w"]
    subprocess.check_call(cc)
    return exe


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("hwc"), "host_wave_check")


@pytest.fixture(scope="module")
def harness_small_caps(tmp_path_factory):
    """The same code with the wide-grid field caps at 3 / 2 steps: a walk re-arms its packed counters every few cells."""
    return _build(tmp_path_factory.mktemp("hwc_cap"), "host_wave_check_caps", ("-DVXRT_FIELD_CAP_XZ=3u", "-DVXRT_FIELD_CAP_Y=2u"))


def _run(exe, *args):
    out = subprocess.run([exe, *[str(a) for a in args]], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "mismatches 0 of %d" % args[3] in out.stdout
    return out.stdout


@pytest.mark.parametrize("factor,edge,density,n", [(8, 64, 0.01, 20000), (8, 64, 0.3, 10000), (16, 128, 0.002, 10000),
                                                   (32, 256, 0.0005, 6000)])
def test_wave_tracer_single_lane_equals_oracle(harness, factor, edge, density, n):
    _run(harness, factor, edge, density, n)


@pytest.mark.parametrize("factor,sx,sy,sz,density,n", [(8, 32768, 64, 64, 0.000005, 20000), (8, 16384, 64, 128, 0.00002, 20000),
                                                       (16, 16384, 128, 128, 0.00001, 8000)])
def test_wide_grids_rearm_their_step_counters_and_end_walks_at_max_steps(harness, harness_small_caps, factor, sx, sy, sz, density, n):
    """Coarse grids of 4096 / 2048 / 1024 cells along x (beyond the 11-bit fields of the packed step counters, and long
    enough for one walk to reach DDARayTraversal's MAX_STEPS): results and probe counters equal the oracle's, and the run
    does contain walks that end by exhaustion."""
    out = _run(harness, factor, sx, density, n, sy, sz)
    exhausted = int(re.search(r"without a hit (\d+)", out).group(1))
    if sx // factor >= 2048:
        assert exhausted > 100, out
    _run(harness_small_caps, factor, sx, density, n, sy, sz)


@pytest.mark.parametrize("factor,edge,density,n", [(8, 64, 0.01, 20000), (8, 64, 0.3, 10000), (32, 256, 0.0005, 6000)])
def test_wide_grid_code_on_ordinary_grids(harness_small_caps, factor, edge, density, n):
    """The wide-grid path forced on small cubic grids, re-arming every 3 / 2 steps: the adversarial ray families (far-face
    starts, exact ties, denormal directions, single-step mode) through the re-armed history."""
    _run(harness_small_caps, factor, edge, density, n, edge, edge, 1)
