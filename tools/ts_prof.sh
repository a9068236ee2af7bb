#!/bin/bash
# usage: tools/ts_prof.sh <tag> [views] [variants...]   (on the GPU box via gpurun)
# rocprofv3 kernel trace of the traversal/shading pipeline on the bench frames (full ray set, V views per step).
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/tsprof_$TAG
mkdir -p $OUT
python3 -c 'import sys; sys.path.insert(0, "'$R'"); import voxelengine_amd as v; v.load()' || { echo "library build failed"; exit 1; }
export VXRT_SKIP_STALE_CHECK=1
cd /tmp && export TMPDIR=/tmp
export TS_MODES=${TS_MODES:-2} TS_ONLY_V=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/tools/ts_ab.py "$@" > $OUT/kt.log 2>&1 || { tail -20 $OUT/kt.log; exit 1; }
cat $OUT/kt.log | tail -8
F=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls=%-5s avg_us=%10.1f total_ms=%9.2f pct=%s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
