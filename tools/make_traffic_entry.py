"""Write the profile-derived entry of profiles/traffic.json that bench.py reports as roofline.traffic / issue_utilisation.

usage: make_traffic_entry.py <tools/prof.sh output dir> <workload> <views per launch> <kernel symbol> <profile tag>

Reads <dir>/summary.txt (tools/prof_summary.py: per-kernel means of the PMC passes), takes the block of <kernel symbol>
(e.g. "k_render_persist2<false, false, true, false>"), and stores next to the counters the source hash of the library the
profile was taken with (voxelengine_amd/csrc/libvxrt.so.srchash): bench.py emits the entry's numbers only when both the
kernel it launches and the library it loaded match, else null and "stale_profile": true.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir, workload, views, symbol, tag = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
norm = lambda s: re.sub(r"\s+", "", s)  # noqa: E731
vals, cur = {}, None
for line in open(os.path.join(out_dir, "summary.txt")):
    if line.startswith("-- "):
        cur = norm(line[3:])
    elif cur == norm(symbol):
        m = re.match(r"\s+(\w+)\s+n=(\d+) mean=([0-9.e+]+)", line)
        if m:
            vals[m.group(1)] = float(m.group(3))
if not vals:
    raise SystemExit("no block for %s in %s/summary.txt" % (symbol, out_dir))
srchash = open(os.path.join(ROOT, "voxelengine_amd", "csrc", "libvxrt.so.srchash")).read().strip()
waves_per_simd = round(vals.get("SQ_WAVES", 0) / 1024.0) if "SQ_WAVES" in vals else None
cycles = vals["GRBM_GUI_ACTIVE"] / 8.0
insts = sum(vals.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS"))
ent = {
    "kernel": "%s (%d view(s) per launch)" % (symbol, views), "kernel_symbol": norm(symbol), "lib_srchash": srchash, "profile": tag,
    "views_per_launch": views,
    "fetch_size_kb": vals.get("FETCH_SIZE"), "write_size_kb": vals.get("WRITE_SIZE"),
    "tcc_hit": vals.get("TCC_HIT_sum"), "tcc_miss": vals.get("TCC_MISS_sum"),
    "sq_insts_valu": vals.get("SQ_INSTS_VALU"), "sq_insts_salu": vals.get("SQ_INSTS_SALU"), "sq_insts_vmem": vals.get("SQ_INSTS_VMEM"),
    "sq_insts_lds": vals.get("SQ_INSTS_LDS"), "sq_wave_cycles_quad": vals.get("SQ_WAVE_CYCLES"),
    "sq_active_inst_any_quad": vals.get("SQ_ACTIVE_INST_ANY"), "sq_wait_inst_any_quad": vals.get("SQ_WAIT_INST_ANY"),
    "sq_wait_any_quad": vals.get("SQ_WAIT_ANY"), "grbm_gui_active": vals.get("GRBM_GUI_ACTIVE"),
    # (2 * FETCH_SIZE + WRITE_SIZE) KiB: the gfx950 correction of MI355X_MICROARCH.md (HBM section) for FETCH_SIZE
    "hbm_bytes_per_launch": int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024),
    "issue_utilisation": {
        "waves_per_simd": waves_per_simd,
        "cycles_per_instruction_per_simd": round(cycles * 1024 / insts, 3),
        "cycles_per_valu_per_simd": round(cycles * 1024 / vals["SQ_INSTS_VALU"], 3),
        "wave_time_issuing": round(vals["SQ_ACTIVE_INST_ANY"] / vals["SQ_WAVE_CYCLES"], 3),
        "wave_time_waiting_to_issue": round(vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"], 3),
        "wave_time_waiting_for_memory": round(vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"], 3),
        "note": "instructions / 1024 SIMDs against GRBM_GUI_ACTIVE / 8 cycles.  A SIMD retires a wave64 vector instruction in ~2.3 "
                "cycles (fast class) or ~4.15 (slow class), beside one scalar instruction per ~4.15 (profiles/r03_instr_cost.md); "
                "cycles_per_valu_per_simd near the kernel's mean instruction price (tools/isa_cost.py: ~3.7) means the vector pipe "
                "is busy",
    },
}
tj = os.path.join(ROOT, "profiles", "traffic.json")
tab = json.load(open(tj))
key = workload + ("_single_view_launch" if views == 1 else "")
tab[key] = ent
json.dump(tab, open(tj, "w"), indent=1)
print("profiles/traffic.json[%s] <- %s, library %s" % (key, symbol, srchash[:12]))
