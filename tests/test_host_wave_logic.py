"""The wave-level traversal code (voxelengine_amd/csrc/vxrt_wave.hpp) compiled for the HOST with one lane per
wave (tests/tools/hoststub stands in for the few HIP builtins) and run against the oracle.  This exercises the
product's traversal logic -- state machine, parking votes, nudges, counters -- on CPU, bit for bit."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hwc") / "host_wave_check")
    cc = ["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "tests", "tools", "hoststub"),
          "-I" + os.path.join(ROOT, "oracle"), "-o", exe, os.path.join(ROOT, "tests", "tools", "host_wave_check.cpp"),
          "-x", "c", os.path.join(ROOT, "oracle", "vxo_trace.c"), os.path.join(ROOT, "oracle", "vxo_world.c"),
          os.path.join(ROOT, "oracle", "vxo_render.c"), "-lm", "-lpthread", "-w"]
    subprocess.check_call(cc)
    return exe


@pytest.mark.parametrize("factor,edge,density,n", [(8, 64, 0.01, 20000), (8, 64, 0.3, 10000), (16, 128, 0.002, 10000),
                                                   (32, 256, 0.0005, 6000)])
def test_wave_tracer_single_lane_equals_oracle(harness, factor, edge, density, n):
    out = subprocess.run([harness, str(factor), str(edge), str(density), str(n)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "mismatches 0 of %d" % n in out.stdout
