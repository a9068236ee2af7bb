"""Development: rays/s of the bench workload with V views per launch (vxrt_render_views) against V single-view
launches, full frames and 1/8 shards.

usage: views_probe.py [workload]
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
X, Y, Z, F, gen, W, H, shadow, bounce = bench.WORKLOADS[name]
ctx = vx.Context(0)
ctx.build_world(gen, X, Y, Z, F)
l = float(np.float32(1.0) / np.sqrt(np.float32(3.0), dtype=np.float32))
ctx.SetEnvironment((l, l, l), (2, 2, 2), (0.5, 0.5, 0.5))
cams = [(vx.GetDirections(e), (fr[0] * X, fr[1] * Y, fr[2] * Z)) for _, fr, e in bench.CAMERAS]
for count in (1, 2, 4, 8):
    plan = sharding.ShardPlan(W, H, sharding.STRIP_ROWS, count, 0)
    nbytes = plan.shard_bytes if count > 1 else W * H * 4
    o = vx.RenderOptions(shadow=bool(shadow), bounce_samples=bounce, strip_rows=plan.strip_rows, strip_count=count,
                         strip_index=0, compact=count > 1)
    for nv in (1, 4, 16):
        bufs = [torch.zeros(nbytes, dtype=torch.uint8, device="cuda") for _ in range(nv)]

        def launch(k):
            views = [dict(fb=bufs[j], origin=cams[(k * nv + j) % 4][1], fwd=cams[(k * nv + j) % 4][0][0],
                          up=cams[(k * nv + j) % 4][0][1], right=cams[(k * nv + j) % 4][0][2], frame_number=k * nv + j + 1)
                     for j in range(nv)]
            ctx.RenderViews(W, H, views, o)

        reps = 64 // nv
        for k in range(2):
            launch(k)
        torch.cuda.synchronize()
        ctx.frame_stats()
        t0 = time.perf_counter()
        for k in range(reps):
            launch(k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rays = ctx.frame_stats().total_rays()
        print("%s shard 1/%d, %d view(s) per launch: %.3f ms per frame, %.0f Mrays/s" % (
            name, count, nv, 1e3 * dt / (reps * nv), rays / dt / 1e6), flush=True)
        if nv == 16:  # the same launches alternating between two streams: step k+1 fills the SIMD slots step k's tail leaves
            ext = [torch.cuda.Stream(), torch.cuda.Stream()]
            bufs2 = [[torch.zeros(nbytes, dtype=torch.uint8, device="cuda") for _ in range(nv)] for _ in range(2)]

            def launch2(k):
                views = [dict(fb=bufs2[k & 1][j], origin=cams[(k * nv + j) % 4][1], fwd=cams[(k * nv + j) % 4][0][0],
                              up=cams[(k * nv + j) % 4][0][1], right=cams[(k * nv + j) % 4][0][2], frame_number=k * nv + j + 1)
                         for j in range(nv)]
                ctx.RenderViews(W, H, views, o, stream=ext[k & 1].cuda_stream)

            reps2 = 8
            for k in range(2):
                launch2(k)
            torch.cuda.synchronize()
            ctx.frame_stats()
            t0 = time.perf_counter()
            for k in range(reps2):
                launch2(k)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            rays = ctx.frame_stats().total_rays()
            print("%s shard 1/%d, %d view(s) per launch, two launches in flight: %.3f ms per frame, %.0f Mrays/s" % (
                name, count, nv, 1e3 * dt / (reps2 * nv), rays / dt / 1e6), flush=True)
