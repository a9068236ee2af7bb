#!/bin/bash
# usage: tools/ts_pmc.sh <tag> <views> <variants...>   (on the GPU box via gpurun)
# PMC passes (instruction mix, wave-time split, HBM bytes) of tools/ts_ab.py: the traversal/shading pipeline on the bench frames.
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/tspmc_$TAG
mkdir -p $OUT
python3 -c 'import sys; sys.path.insert(0, "'$R'"); import voxelengine_amd as v; v.load()' || { echo "library build failed"; exit 1; }
export VXRT_SKIP_STALE_CHECK=1 TS_MODES=${TS_MODES:-2} TS_ONLY_V=1
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "WRITE_SIZE" "FETCH_SIZE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/tools/ts_ab.py "$@" > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed: full log in $OUT/pmc_$N.log"; tail -20 $OUT/pmc_$N.log; exit 1; }
done
python3 $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
