#!/bin/bash
# usage: tools/ab_libs.sh <out-tag> <lib-tag[:waves_per_cu]> ...   (on the GPU box via gpurun)
# A/B of library builds on the bench workload in ONE GPU session, the product library first and last.
# A lib-tag names voxelengine_amd/csrc/libvxrt_<tag>.so; "base" is the product library.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ab_$1; shift
mkdir -p $OUT
run() {
  local tag=${1%%:*} wpc=""
  [[ "$1" == *:* ]] && wpc=${1##*:}
  local lib=$R/voxelengine_amd/csrc/libvxrt.so
  [ "$tag" != base ] && lib=$R/voxelengine_amd/csrc/libvxrt_$tag.so
  # an A/B library must come from the sources the product library was built from (tools/build_variant.sh stamps it)
  if [ "$tag" != base ] && [ "$(head -1 $lib.srchash 2>/dev/null)" != "$(head -1 $R/voxelengine_amd/csrc/libvxrt.so.srchash)" ]; then
    echo "$lib is stale (or was not built by tools/build_variant.sh): rebuild it"; exit 1
  fi
  VXRT_LIB=$lib VXRT_WAVES_PER_CU=$wpc VXRT_SKIP_STALE_CHECK=1 python3 $R/bench.py --cpu-baseline off ${BENCH_ARGS:-} > $OUT/$2.json 2> $OUT/$2.err || { echo "$1 failed"; tail -20 $OUT/$2.err; echo "stopping: no further GPU run behind a failed one (full log: $OUT/$2.err)"; exit 1; }
  python3 - "$OUT/$2.json" "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
o = d.get("one_view_per_launch", {})
print("%-24s %8.1f Mrays/s  (%.3f ms/step, roofline %.4f)   one view per launch %8.1f" % (
    sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"], o.get("value", 0.0)), flush=True)
PY
}
run base base_first
for v in "$@"; do run "$v" "$(echo $v | tr ':' '_')"; done
run base base_last
