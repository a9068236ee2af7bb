#!/bin/bash
# usage: tools/pmc_quick.sh <tag> [env assignments...]   (on the GPU box via gpurun)
# Two PMC passes of the bench workload (instruction mix + wave time split; HBM write traffic) for a quick A/B of kernels.
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for a in "$@"; do export "$a"; done
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "WRITE_SIZE" "FETCH_SIZE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py --cpu-baseline off --steps 4 --warmup 1 > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed"; tail -3 $OUT/pmc_$N.log; }
done
python3 $R/tools/prof_summary.py $OUT 2>&1 | grep -A14 "false, false, true"
