"""Development: host-side cost of one RenderScreen call (Python + ctypes + launch) against the GPU time of a 1/8
shard of the bench frame -- what bounds the per-rank step rate of the 8-GPU strong-scaling run."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import voxelengine_amd as vx  # noqa: E402
from voxelengine_amd import sharding  # noqa: E402

X, Y, Z, F, gen, W, H, shadow, bounce = bench.WORKLOADS["c3_8k_1080p_shadow_bounce"]
ctx = vx.Context(0)
ctx.build_world(gen, X, Y, Z, F)
l = float(np.float32(1.0) / np.sqrt(np.float32(3.0), dtype=np.float32))
ctx.SetEnvironment((l, l, l), (2, 2, 2), (0.5, 0.5, 0.5))
cams = [(vx.GetDirections(e), (fr[0] * X, fr[1] * Y, fr[2] * Z)) for _, fr, e in bench.CAMERAS]
for count in (1, 2, 4, 8):
    plan = sharding.ShardPlan(W, H, sharding.STRIP_ROWS, count, 0)
    buf = torch.zeros(plan.shard_bytes, dtype=torch.uint8, device="cuda")
    o = vx.RenderOptions(shadow=True, bounce_samples=1, frame_number=1, strip_rows=plan.strip_rows, strip_count=count,
                         strip_index=0, compact=count > 1)
    n = 200
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            (f, u, r), pos = cams[i % 4]
            ctx.RenderScreen(W, H, buf, pos, f, u, r, o)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
    print("shard 1/%d: host enqueue %.1f us per call, GPU-bound step %.1f us per call" % (count, 1e6 * t_host / n, 1e6 * t_all / n),
          flush=True)
