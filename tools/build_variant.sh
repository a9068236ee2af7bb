#!/bin/bash
# usage: tools/build_variant.sh <tag> [-DVXRT_...=... ...]      (no GPU needed)
# Builds voxelengine_amd/csrc/libvxrt_<tag>.so from the current sources with extra compiler flags -- the A/B libraries of
# tools/ab_libs.sh and tools/pmc_valu.sh -- and stamps it with the content hash of the sources, so that an A/B run can refuse
# a library built from other sources than the product library it is compared with.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
make -C $R/voxelengine_amd/csrc -B OUT=libvxrt_$TAG.so EXTRA="$*" libvxrt_$TAG.so > /dev/null
python3 - "$R" "$TAG" "$*" <<'PY'
import sys
sys.path.insert(0, sys.argv[1])
from voxelengine_amd import build as vb
open(vb.CSRC + "/libvxrt_%s.so.srchash" % sys.argv[2], "w").write(vb._digest(vb.lib_sources()) + "\n" + sys.argv[3] + "\n")
print("built libvxrt_%s.so with %r" % (sys.argv[2], sys.argv[3]))
PY
