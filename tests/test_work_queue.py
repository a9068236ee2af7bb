"""The sharded work queue of the persistent kernels (csrc/vxrt_device.hpp, queue_take) on the host: the ticket arithmetic
partitions every queue, and waves taking tickets in any interleaving hand out every ticket exactly once and all leave.
The GPU side: tests/test_gpu_parity.py::test_tile_queue_hands_out_every_tile_once."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("shards", [8, 1, 2, 32])
def test_queue_arithmetic_and_take_loop(tmp_path, shards):
    exe = str(tmp_path / "queue_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-DVXRT_QUEUE_SHARDS=%d" % shards,
                           "-I" + os.path.join(ROOT, "tests", "tools", "hoststub"), "-o", exe,
                           os.path.join(ROOT, "tests", "tools", "queue_check.cpp"), "-w"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shards %d, 0 failure(s)" % shards in out.stdout
