#!/bin/bash
# usage: tools/pmc_quick.sh <tag> [env assignments...]   (on the GPU box via gpurun)
# Two PMC passes of the bench workload (instruction mix + wave time split; HBM write traffic) for a quick A/B of kernels.
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT
# build the library BEFORE rocprofv3 runs anything: the profiler's preload initialises the GPU before Python starts, so the
# profiled process must not start make/hipcc children (voxelengine_amd/build.py honours VXRT_SKIP_STALE_CHECK)
python3 -c 'import sys; sys.path.insert(0, "'$R'"); import voxelengine_amd as v; v.load()' || { echo "library build failed"; exit 1; }
export VXRT_SKIP_STALE_CHECK=1
cd /tmp && export TMPDIR=/tmp
for a in "$@"; do export "$a"; done
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "WRITE_SIZE" "FETCH_SIZE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py --cpu-baseline off --steps 4 --warmup 1 > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed: full log in $OUT/pmc_$N.log"; tail -20 $OUT/pmc_$N.log; exit 1; }
done
python3 $R/tools/prof_summary.py $OUT 2>&1 | grep -A14 "false, false, true"
