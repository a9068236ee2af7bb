"""The short exact division of the render kernel's ray-finished phase (csrc/vxrt_device.hpp: div_rn) against the IEEE
quotient on the CPU, exhaustively over the operand families the kernel uses it on -- the tonemap c / (c + 1) for every
binary32 c of ordinary size, x / W for all 16-bit integers, the occlusion mean -- and a billion random pairs
(tests/tools/exact_div_check.c).  The reciprocal it starts from is the hardware's v_rcp_f32 + one Newton step on the GPU,
checked there against the IEEE reciprocal on every binary32 of ordinary size (tools/ubench/rcp_check.hip)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_markstein_division_is_the_ieee_quotient_on_the_kernels_operand_families(tmp_path):
    exe = str(tmp_path / "exact_div_check")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", "-o", exe,
                           os.path.join(ROOT, "tests", "tools", "exact_div_check.c"), "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "tonemap c/(c+1): 0 of 1677721601 differ; x/W: 0 of 4294901760; s/n: 0; random pairs: 0 of 1024000000" in out.stdout
