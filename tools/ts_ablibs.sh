#!/bin/bash
# usage: tools/ts_ablibs.sh <views> <lib-tag>...   (on the GPU box via gpurun)
# A/B of library builds (voxelengine_amd/csrc/libvxrt_<tag>.so; "base" = the product library) on tools/ts_ab.py: the
# traversal/shading pipeline (variant 6) on the bench frames, full ray set, in ONE GPU session, base first and last.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; shift
python3 -c 'import sys; sys.path.insert(0, "'$R'"); import voxelengine_amd as v; v.load()' || { echo "library build failed"; exit 1; }
export VXRT_SKIP_STALE_CHECK=1 TS_MODES=${TS_MODES:-2} TS_ONLY_V=${TS_ONLY_V:-1}
for tag in base "$@" base; do
  lib=$R/voxelengine_amd/csrc/libvxrt.so
  [ "$tag" != base ] && lib=$R/voxelengine_amd/csrc/libvxrt_$tag.so
  echo "--- $tag"
  VXRT_LIB=$lib python3 $R/tools/ts_ab.py $V ${TS_VARIANTS:-6} 2>&1 | grep -E "variant|==" || { echo "$tag failed; stopping"; exit 1; }
done
