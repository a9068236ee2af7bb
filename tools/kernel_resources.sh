#!/bin/bash
# usage: tools/kernel_resources.sh [extra hipcc flags]   (no GPU needed: cross-compiles vxrt_kernels.hip for gfx950)
# Registers, spills, scratch, occupancy and LDS of every kernel of vxrt_kernels.hip, from the compiler's resource remarks.
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/voxelengine_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -mllvm -amdgpu-sched-strategy=max-ilp "$@" \
  -Rpass-analysis=kernel-resource-usage -c -o /dev/null vxrt_kernels.hip 2>&1 |
  grep "Function Name\|VGPRs:\|Spill\|Occupancy\|ScratchSize\|LDS Size" | sed 's/.*remark: //; s/\[-Rpass-analysis=kernel-resource-usage\]//' |
  paste - - - - - - - | sed 's/Function Name: //; s/ \+/ /g' | c++filt | cut -c1-260
