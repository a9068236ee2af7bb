"""Development: throughput of the bench workload with F frames in flight (F streams, F framebuffers): frame k+1's
waves fill the SIMD slots that frame k's last waves leave.  Checks every frame against the F=1 frames.

usage: overlap_probe.py [workload] [shard_count=1]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import voxelengine_amd as vx  # noqa: E402
from voxelengine_amd import sharding  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3_8k_1080p_shadow_bounce"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1
X, Y, Z, F, gen, W, H, shadow, bounce = bench.WORKLOADS[name]
ctx = vx.Context(0)
ctx.build_world(gen, X, Y, Z, F)
l = float(np.float32(1.0) / np.sqrt(np.float32(3.0), dtype=np.float32))
ctx.SetEnvironment((l, l, l), (2, 2, 2), (0.5, 0.5, 0.5))
cams = [(vx.GetDirections(e), (fr[0] * X, fr[1] * Y, fr[2] * Z)) for _, fr, e in bench.CAMERAS]
plan = sharding.ShardPlan(W, H, sharding.STRIP_ROWS, count, 0)
n = 64
ref = None
for flight in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(flight)]
    bufs = [torch.zeros(plan.shard_bytes if count > 1 else W * H * 4, dtype=torch.uint8, device="cuda") for _ in range(flight)]
    keep = []

    def frame(i, check):
        (f, u, r), pos = cams[i % 4]
        o = vx.RenderOptions(shadow=bool(shadow), bounce_samples=bounce, frame_number=i + 1, strip_rows=plan.strip_rows,
                             strip_count=count, strip_index=0, compact=count > 1)
        s = streams[i % flight]
        with torch.cuda.stream(s):
            ctx.RenderScreen(W, H, bufs[i % flight], pos, f, u, r, o)
            if check:
                keep.append(bufs[i % flight].clone())

    for i in range(8):
        frame(i, False)
    torch.cuda.synchronize()
    ctx.frame_stats()
    t0 = time.perf_counter()
    for i in range(n):
        frame(i, False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rays = ctx.frame_stats().total_rays()
    for i in range(8):
        frame(i, True)
    torch.cuda.synchronize()
    if ref is None:
        ref = keep
    ok = all(torch.equal(a, b) for a, b in zip(ref, keep))
    print("%s shard 1/%d, %d frame(s) in flight: %.3f ms per frame, %.0f Mrays/s, frames equal to the serial ones: %s" % (
        name, count, flight, 1e3 * dt / n, rays / dt / 1e6, ok), flush=True)
