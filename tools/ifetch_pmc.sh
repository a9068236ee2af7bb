#!/bin/bash
# usage: tools/ifetch_pmc.sh <tag>   (on the GPU box via gpurun)
# Is instruction FETCH what bounds the render kernels' issue rate?  Instruction-cache and fetch counters of the bench step
# (fused kernel) and of the traversal/shading pipeline.
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ifetch_$TAG
mkdir -p $OUT
python3 -c 'import sys; sys.path.insert(0, "'$R'"); import voxelengine_amd as v; v.load()' || { echo "library build failed"; exit 1; }
export VXRT_SKIP_STALE_CHECK=1 TS_MODES=2 TS_ONLY_V=1
cd /tmp && export TMPDIR=/tmp
for C in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_TC_STALL" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/tools/ts_ab.py 16 5 6 > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed: full log in $OUT/pmc_$N.log"; tail -20 $OUT/pmc_$N.log; exit 1; }
done
python3 $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
grep -v "^==" $OUT/summary.txt
