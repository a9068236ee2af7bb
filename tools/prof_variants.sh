#!/bin/bash
# usage: tools/prof_variants.sh <tag> [variants]  -- PMC comparison of kernel variants via tools/ab.py (GPU box)
TAG=$1; VARS=${2:-0,1}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pv_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/tools/ab.py c3_8k_1080p_shadow_bounce 1 $VARS > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed: full log in $OUT/pmc_$N.log"; tail -20 $OUT/pmc_$N.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_render" not in k: continue
        name = "wave" if "k_render_wave" in k else ("persist" if "persist" in k else "direct")
        per[(name, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (name, did, cn), v in per.items():
        acc[cn][name].append(v)
for cn in sorted(acc):
    print("%-26s" % cn, "  ".join("%s mean=%.4g (n=%d)" % (n, sum(v)/len(v), len(v)) for n, v in sorted(acc[cn].items())))
PY
