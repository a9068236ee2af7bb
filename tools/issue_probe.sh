#!/bin/bash
# usage: tools/issue_probe.sh <tag>   (on the GPU box via gpurun)
# Settles what bounds the render kernel's issue rate: the pinned-stream microbenchmark (tools/ubench/issue_rate.hip) and one
# PMC pass of the scheduler counters gfx950 has (there is no SQ_INST_CYCLES_VALU on this chip: SQ_ACTIVE_INST_VALU is it).
set -o pipefail
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/issue_$TAG
mkdir -p $OUT
# build the library BEFORE rocprofv3 runs anything: the profiler's preload initialises the GPU before Python starts, so the
# profiled process must not start make/hipcc children (voxelengine_amd/build.py honours VXRT_SKIP_STALE_CHECK)
python3 -c 'import sys; sys.path.insert(0, "'$R'"); import voxelengine_amd as v; v.load()' || { echo "library build failed"; exit 1; }
export VXRT_SKIP_STALE_CHECK=1
cd /tmp && export TMPDIR=/tmp
$R/tools/ubench/issue_rate > $OUT/issue_rate.txt 2>&1 || { echo "ubench failed"; tail -3 $OUT/issue_rate.txt; exit 1; }
cat $OUT/issue_rate.txt
for C in "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py --cpu-baseline off --steps 6 --warmup 2 > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed: full log in $OUT/pmc_$N.log"; tail -20 $OUT/pmc_$N.log; exit 1; }
done
python3 $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
