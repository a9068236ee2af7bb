#!/bin/bash
# usage: tools/ab_batch.sh <out-tag> <lib-tag> ...   (on the GPU box via gpurun)
# A/B of library builds (tools/build_variant.sh) on the batch query workloads of tools/batch_probe.py in ONE GPU session, the
# product library first and last.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/abb_$1; shift
mkdir -p $OUT
for tag in base "$@" base; do
  lib=$R/voxelengine_amd/csrc/libvxrt.so
  [ "$tag" != base ] && lib=$R/voxelengine_amd/csrc/libvxrt_$tag.so
  if [ "$tag" != base ] && [ "$(head -1 $lib.srchash 2>/dev/null)" != "$(head -1 $R/voxelengine_amd/csrc/libvxrt.so.srchash)" ]; then
    echo "$lib is stale (or was not built by tools/build_variant.sh): rebuild it"; exit 1
  fi
  VXRT_LIB=$lib VXRT_SKIP_STALE_CHECK=1 python3 $R/tools/batch_probe.py > $OUT/$tag.txt 2> $OUT/$tag.err || { echo "$tag failed"; tail -5 $OUT/$tag.err; exit 1; }
  echo "== $tag"; grep "kernel variant 4" $OUT/$tag.txt
done
