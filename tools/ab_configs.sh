#!/bin/bash
# usage: tools/ab_configs.sh <out-tag> <variant> ...   (on the GPU box via gpurun)
# the BASELINE configurations other than bench.py's default, each with the given render kernel variants
# (bench.py --kernel-variant: 4 = the product kernel, 1 = the straightforward loops)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/abc_$1; shift
mkdir -p $OUT
for w in ${WORKLOADS:-c2_1k_1080p_primary c3_f8_variant c4_8k_4k_shadow_bounce c5_16k_4k_shadow_bounce}; do
  for v in "$@"; do
    python3 $R/bench.py --cpu-baseline off --kernel-variant $v --workload $w --steps ${STEPS:-6} --warmup 1 > $OUT/${w}_v$v.json 2> $OUT/${w}_v$v.err || { echo "$w v$v failed"; tail -20 $OUT/${w}_v$v.err; echo "stopping: no further GPU run behind a failed one"; exit 1; }
    python3 - "$OUT/${w}_v$v.json" "$w v$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
o = d.get("one_view_per_launch", {}); t = d.get("one_view_two_in_flight", {})
print("%-34s %8.1f Mrays/s  (%.3f ms/step, roofline %.4f)   one view per launch %8.1f   two in flight %8.1f" % (
    sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"], o.get("value", 0.0), t.get("value", 0.0)), flush=True)
PY
  done
done
