#!/usr/bin/env python3
"""isa_cost.py <file.s> <kernel-name-substring> [--blocks]

Prices the vector instructions of one kernel in hipcc's -S output with the per-SIMD issue costs measured by
tools/ubench/instr_cost{,2}.hip on gfx950 (profiles/r03_instr_cost.md): a SIMD retires a "fast" wave64 instruction every
~2.3 cycles, a "slow" one every ~4.15, a transcendental every ~8.1, beside one scalar instruction every ~4.15.
fast = v_add/sub/subrev/mul_f32, v_and/or/xor/not_b32, v_add/sub/subrev_u32, v_lshrrev_b32, v_ashrrev_i32, v_mov_b32 -- and
only while no operand is an SGPR (inline constants, literals and neg/abs/clamp modifiers are free).  Everything else that
executes on the vector ALU is slow (compares, v_cndmask, min/max, conversions, every three-operand form, v_lshlrev_b32,
packed, f64, DPP/SDWA, readlane/writelane)."""
import re
import sys

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32",
        "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32"}
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
C_FAST, C_SLOW, C_TRANS, C_SALU = 2.3, 4.15, 8.1, 4.15


def classify(line):
    """-> one of fast, slow, trans, salu, smem, vmem, lds, branch, wait, other, or None for non-instructions"""
    t = line.split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"):
        return None
    m = re.match(r"([a-z_0-9]+)", t)
    if not m:
        return None
    op = m.group(1)
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op == "s_branch" or op.startswith("s_endpgm") or op.startswith("s_setpc") or op.startswith("s_barrier"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime") or op.startswith("s_store"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_"):
        if base in TRANS:
            return "trans"
        if op.endswith("_sdwa") or op.endswith("_dpp"):
            return "slow"
        if base in FAST:
            ops = t[len(op):]
            # an SGPR / vcc / exec operand makes a fast instruction slow
            if re.search(r"(?<![a-z0-9_])(s\d+|s\[\d+:\d+\]|vcc(_lo|_hi)?|exec(_lo|_hi)?|m0|ttmp\d+)(?![a-z0-9_])", ops):
                return "slow"
            return "fast"
        return "slow"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and key in l and l.rstrip().split(";")[0].strip().endswith(":"):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    tot = {}
    cur, curname, curline = {}, "entry", start
    rows = []
    for i in range(start + 1, len(lines)):
        l = lines[i]
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m:
            rows.append((curname, curline - start, cur))
            cur, curname, curline = {}, m.group(1), i
            continue
        c = classify(l)
        if c:
            cur[c] = cur.get(c, 0) + 1
            tot[c] = tot.get(c, 0) + 1
    rows.append((curname, curline - start, cur))
    keys = ["fast", "slow", "trans", "salu", "smem", "vmem", "lds", "branch", "wait"]

    def cost(d):
        return d.get("fast", 0) * C_FAST + d.get("slow", 0) * C_SLOW + d.get("trans", 0) * C_TRANS

    if blocks:
        print("%-12s %6s " % ("block", "line") + " ".join("%6s" % k for k in keys) + "   valu-cycles  salu-cycles")
        for name, ln, d in rows:
            if sum(d.values()) == 0:
                continue
            print("%-12s %6d " % (name, ln) + " ".join("%6d" % d.get(k, 0) for k in keys) +
                  "   %10.0f  %10.0f" % (cost(d), d.get("salu", 0) * C_SALU))
    print("%-12s %6s " % ("TOTAL", "") + " ".join("%6d" % tot.get(k, 0) for k in keys) +
          "   %10.0f  %10.0f" % (cost(tot), tot.get("salu", 0) * C_SALU))
    nv = tot.get("fast", 0) + tot.get("slow", 0) + tot.get("trans", 0)
    print("static vector mix: %.0f %% fast, %.0f %% slow, %.1f %% transcendental; mean %.2f cycles per vector instruction" %
          (100.0 * tot.get("fast", 0) / nv, 100.0 * tot.get("slow", 0) / nv, 100.0 * tot.get("trans", 0) / nv, cost(tot) / nv))


if __name__ == "__main__":
    main()
