#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/tpw
for w in c3_8k_1080p_shadow_bounce c2_1k_1080p_primary; do
for t in 17 6 9 12 24 34 17; do
  VXRT_LIB=$R/voxelengine_amd/csrc/libvxrt_exp.so VXRT_SKIP_STALE_CHECK=1 VXRT_TILES_PER_WAVE=$t python3 $R/bench.py --cpu-baseline off --workload $w --steps 4 --warmup 1 > $R/gpurun_out/tpw/${w}_$t.json 2> $R/gpurun_out/tpw/${w}_$t.err || { echo failed; tail -5 $R/gpurun_out/tpw/${w}_$t.err; exit 1; }
  python3 - $R/gpurun_out/tpw/${w}_$t.json "$w tpw=$t" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); o=d.get("one_view_per_launch",{}); t=d.get("one_view_two_in_flight",{})
print("%-40s %8.1f   one view per launch %8.1f   two in flight %8.1f"%(sys.argv[2], d["value"], o.get("value",0), t.get("value",0)), flush=True)
PY
done; done
