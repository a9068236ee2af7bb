#!/bin/bash
# usage: tools/pmc_valu.sh <tag> <lib-tag>...   (on the GPU box via gpurun)
# One PMC pass per library build (base = the product library): vector / scalar instruction counts and the wave-time split
# of the bench step's render kernel -- the exact dynamic counts behind an A/B of kernel code.
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcv_$TAG
mkdir -p $OUT
python3 -c 'import sys; sys.path.insert(0, "'$R'"); import voxelengine_amd as v; v.load()' || { echo "library build failed"; exit 1; }
export VXRT_SKIP_STALE_CHECK=1
cd /tmp && export TMPDIR=/tmp
for L in base "$@"; do
  lib=$R/voxelengine_amd/csrc/libvxrt.so
  [ "$L" != base ] && lib=$R/voxelengine_amd/csrc/libvxrt_$L.so
  export VXRT_LIB=$lib
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/$L/pmc_a -- python3 $R/bench.py --cpu-baseline off --steps 4 --warmup 1 ${BENCH_ARGS:-} > $OUT/$L.log 2>&1 || { echo "pmc $L failed: full log in $OUT/$L.log"; tail -20 $OUT/$L.log; exit 1; }
  echo "== $L"
  python3 $R/tools/prof_summary.py $OUT/$L 2>&1 | grep -A9 "false, false, true" | grep -E "k_render|SQ_"
done
