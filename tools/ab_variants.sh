#!/bin/bash
# usage: tools/ab_variants.sh <out-tag> <variant[:lib-tag[:waves_per_cu]]> ...   (on the GPU box via gpurun)
# A/B of render kernel variants (VXRT_VARIANT) and library builds on the bench workload in ONE GPU session, the default
# first and last.
set -o pipefail
# VXRT_VARIANT / VXRT_WAVES_PER_CU are read by the EXPERIMENTS build only: make -C voxelengine_amd/csrc libvxrt_exp.so first
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/abv_$1; shift
mkdir -p $OUT
run() {
  IFS=: read -r var tag wpc <<< "$1"
  local lib=$R/voxelengine_amd/csrc/libvxrt_exp.so
  [ -n "$tag" ] && [ "$tag" != base ] && lib=$R/voxelengine_amd/csrc/libvxrt_$tag.so
  VXRT_VARIANT=$var VXRT_LIB=$lib VXRT_WAVES_PER_CU=$wpc python3 $R/bench.py --cpu-baseline off ${BENCH_ARGS:-} > $OUT/$2.json 2> $OUT/$2.err || { echo "$1 failed"; tail -20 $OUT/$2.err; echo "stopping: no further GPU run behind a failed one (full log: $OUT/$2.err)"; exit 1; }
  python3 - "$OUT/$2.json" "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
o = d.get("one_view_per_launch", {}); t = d.get("one_view_two_in_flight", {})
print("%-24s %8.1f Mrays/s  (%.3f ms/step, roofline %.4f)   one view per launch %8.1f   two in flight %8.1f" % (
    sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"], o.get("value", 0.0), t.get("value", 0.0)), flush=True)
PY
}
run 2 base_first
n=0
for v in "$@"; do n=$((n+1)); run "$v" "v${n}_$(echo $v | tr ':' '_')"; done
run 2 base_last
