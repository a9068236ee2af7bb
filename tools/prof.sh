#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]   (run on the GPU box via gpurun)
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# build the library BEFORE rocprofv3 runs anything: the profiler's preload initialises the GPU before Python starts, so the
# profiled process must not start make/hipcc children (voxelengine_amd/build.py honours VXRT_SKIP_STALE_CHECK)
python3 -c 'import sys; sys.path.insert(0, "'$R'"); import voxelengine_amd as v; v.load()' || { echo "library build failed"; exit 1; }
export VXRT_SKIP_STALE_CHECK=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --cpu-baseline off "$@" > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py --cpu-baseline off --steps 8 --warmup 2 > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed: full log in $OUT/pmc_$N.log"; tail -20 $OUT/pmc_$N.log; exit 1; }
done
python3 $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
