"""Development: when does each pixel's ray chain start / end inside one launch of the persistent kernel?

Needs the library built with -DVXRT_TAIL_DEBUG (VXRT_LIB=.../libvxrt_taildbg.so): the probe-counting launch then
writes (start tick, end tick, primary steps) per pixel into the colour AOV (100 MHz ticks, low 24 bits).

usage: VXRT_LIB=... tail_map.py [workload] [schedule 0|1]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import voxelengine_amd as vx  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3_8k_1080p_shadow_bounce"
use_sched = len(sys.argv) > 2 and sys.argv[2] == "1"
X, Y, Z, F, gen, W, H, shadow, bounce = bench.WORKLOADS[name]
ctx = vx.Context(0)
ctx.build_world(gen, X, Y, Z, F)
l = float(np.float32(1.0) / np.sqrt(np.float32(3.0), dtype=np.float32))
ctx.SetEnvironment((l, l, l), (2, 2, 2), (0.5, 0.5, 0.5))
ctx.SetFOV(90.0)
fb = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
aov = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
for cname, frac, euler in bench.CAMERAS:
    f, u, r = vx.GetDirections(euler)
    pos = (frac[0] * X, frac[1] * Y, frac[2] * Z)
    order = None
    if use_sched:
        order = torch.from_numpy(vx.tile_schedule(W, range(H), f, u, r, 90.0, H).astype(np.int32)).cuda()
    o = vx.RenderOptions(shadow=bool(shadow), bounce_samples=bounce, frame_number=1, collect_stats=True)
    for _ in range(2):
        ctx.RenderScreen(W, H, fb, pos, f, u, r, o, color_aov=aov, tile_order=order)
    torch.cuda.synchronize()
    ctx.frame_stats()
    a = aov.cpu().numpy().astype(np.float64)
    t0, t1, steps = a[..., 0], a[..., 1], a[..., 2]
    base = t0.min()
    t0 = (t0 - base) / 100.0  # microseconds
    t1 = (t1 - base) / 100.0
    end = t1.max()
    dur = t1 - t0
    print("cam %s sched=%d: launch %.0f us | pixel chain duration us: median %.0f p90 %.0f p99 %.0f max %.0f | primary steps: "
          "median %.0f p99 %.0f max %.0f" % (cname, int(use_sched), end, np.median(dur), np.percentile(dur, 90),
                                             np.percentile(dur, 99), dur.max(), np.median(steps), np.percentile(steps, 99),
                                             steps.max()))
    print("   pixel end-time percentiles (%% of launch): p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f | last pixel to START at %.0f%%" % (
        100 * np.percentile(t1, 50) / end, 100 * np.percentile(t1, 90) / end, 100 * np.percentile(t1, 99) / end,
        100 * np.percentile(t1, 99.9) / end, 100 * t0.max() / end))
    late = t1 > 0.8 * end
    print("   pixels ending in the last 20%%: %d (%.2f%%); their start %% of launch: median %.0f min %.0f; their duration us median %.0f; "
          "their steps median %.0f; rows: %d..%d (median %d)" % (
              late.sum(), 100.0 * late.mean(), 100 * np.median(t0[late]) / end, 100 * t0[late].min() / end, np.median(dur[late]),
              np.median(steps[late]), np.where(late.any(axis=1))[0].min(), np.where(late.any(axis=1))[0].max(),
              int(np.median(np.where(late)[0]))))
    # correlation of chain duration with primary steps, and per-iteration time estimate
    big = dur > np.percentile(dur, 99.9)
    print("   top 0.1%% chains: duration %.0f..%.0f us, primary steps median %.0f, start %% median %.0f" % (
        dur[big].min(), dur[big].max(), np.median(steps[big]), 100 * np.median(t0[big]) / end), flush=True)
