#!/bin/bash
# usage: tools/batch_pmc.sh <tag>   (on the GPU box via gpurun)
# What bounds the batch API on incoherent rays: PMC passes of tools/batch_probe.py -- HBM bytes,
# L2 hit rate, instruction mix and the wave-time split of k_trace_batch_persist.
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/batchpmc_$TAG
mkdir -p $OUT
python3 -c 'import sys; sys.path.insert(0, "'$R'"); import voxelengine_amd as v; v.load()' || { echo "library build failed"; exit 1; }
export VXRT_SKIP_STALE_CHECK=1
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE" "FETCH_SIZE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/tools/batch_probe.py "$@" > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed: full log in $OUT/pmc_$N.log"; tail -20 $OUT/pmc_$N.log; exit 1; }
done
python3 - $OUT <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_trace_batch" not in k and "k_batch" not in k and "DeviceRadixSort" not in k and "k_ray_keys" not in k:
            continue
        per[(k.split("(")[0][:60], row["Dispatch_Id"], row["Counter_Name"], row.get("Grid_Size", ""))] += float(row["Counter_Value"])
    for (k, did, cn, gs), v in per.items():
        acc[(k, gs)][cn].append(v)
for (k, gs) in sorted(acc):
    print("-- %s  grid %s" % (k, gs))
    for cn, vals in sorted(acc[(k, gs)].items()):
        print("   %-22s n=%d mean=%.6g" % (cn, len(vals), sum(vals) / len(vals)))
PY
