#!/bin/bash
# usage: tools/pmc_write.sh <tag>   (on the GPU box via gpurun)
# WRITE_SIZE per bench step of the persistent kernel (every lane stores its pixel when its chain ends: single 4-byte stores at
# different times) against the straightforward kernel (variant 1: the 64 pixels of a tile are stored by one instruction, eight
# 32-byte rows) -- the calibration behind DESIGN.md's note on what the persistent kernel's WRITE_SIZE beyond the frames is.
set -o pipefail
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcw_$TAG
mkdir -p $OUT
python3 -c 'import sys; sys.path.insert(0, "'$R'"); import voxelengine_amd as v; v.load()' || { echo "library build failed"; exit 1; }
export VXRT_SKIP_STALE_CHECK=1
cd /tmp && export TMPDIR=/tmp
for V in 4 1; do
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/v$V -- python3 $R/bench.py --cpu-baseline off --steps 4 --warmup 1 --kernel-variant $V > $OUT/v$V.log 2>&1 || { echo "pmc variant $V failed"; tail -20 $OUT/v$V.log; exit 1; }
  python3 - $OUT/v$V $V <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_render" not in k or row["Counter_Name"] != "WRITE_SIZE":
            continue
        name = k.split("(")[0][-60:]
        tot[name] += float(row["Counter_Value"])
        if row["Dispatch_Id"] not in seen:
            seen.add(row["Dispatch_Id"]); n[name] += 1
for k in sorted(tot):
    print("variant %s  %-62s dispatches %4d  WRITE_SIZE total %.1f MiB = %.2f MiB per dispatch" % (sys.argv[2], k, n[k], tot[k] / 1024, tot[k] / 1024 / max(n[k], 1)))
PY
done
