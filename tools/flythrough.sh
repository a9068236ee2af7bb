#!/bin/bash
# usage: tools/flythrough.sh <out-tag>   (on the GPU box via gpurun)
# The reference's call pattern at the headline configuration: examples/voxelapp_headless flies 480 poses through the
# 8192x512x8192 world at 1920x1080 (shaded: primary + shadow + 1 bounce sample, whole frames), one RenderScreen-shaped call
# per frame with the device->host copy of every frame, first synchronously (Graphics::RenderScreen, as VoxelApp/main.cu
# does), then with two frames in flight (Graphics::RenderScreenAsync / WaitFrame).  Prints the example's own Mrays/s.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/fly_$1
mkdir -p $OUT
python3 - > $OUT/path.txt <<'PY'
import math
X, Y, Z = 8192, 512, 8192
for i in range(480):
    t = i / 479.0
    a = 0.7 + 2.4 * t
    x = X * (0.5 + 0.3 * math.cos(6.0 * t)); z = Z * (0.5 + 0.3 * math.sin(6.0 * t))
    y = Y * (0.75 + 0.2 * math.sin(9.0 * t))
    print("%.3f %.3f %.3f %.4f %.4f 0.0" % (x, y, z, -0.25 - 0.3 * (0.5 + 0.5 * math.sin(5.0 * t)), a))
PY
for flight in 1 2 1 2; do
  $R/examples/voxelapp_headless 8192 0 $OUT/f$flight 1920 1080 2 $OUT/path.txt 0 1 $flight 8192x512x8192 > $OUT/run_f$flight.txt 2>&1 || { echo "flight $flight failed"; tail -3 $OUT/run_f$flight.txt; exit 1; }
  echo "frames in flight $flight: $(grep 'Frame loop' $OUT/run_f$flight.txt)"
done
cmp $OUT/f1.bgra $OUT/f2.bgra && echo "last frames byte-identical"
rm -f $OUT/*.bgra $OUT/*.ppm
