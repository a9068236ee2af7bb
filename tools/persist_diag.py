"""Diagnostics of the persistent render kernel (k_render_persist2) on a bench workload: lane utilisation of the loop, average
wave lifetime against the launch duration (tail / imbalance) and the share of iterations run after the tile queue ran
dry.  Uses the probe-counting launch (collect_stats), which is slower than the timed kernel but has the same shape.

usage: persist_diag.py [workload] [schedule 0|1]
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
nwaves = torch.cuda.get_device_properties(0).multi_processor_count * 16
for cname, frac, euler in bench.CAMERAS:
    f, u, r = vx.GetDirections(euler)
    pos = (frac[0] * X, frac[1] * Y, frac[2] * Z)
    order = None
    if use_sched:
        order = torch.from_numpy(vx.tile_schedule(W, range(H), f, u, r, 90.0, H).astype(np.int32)).cuda()
    o = vx.RenderOptions(shadow=bool(shadow), bounce_samples=bounce, frame_number=1, collect_stats=True)
    ctx.RenderScreen(W, H, fb, pos, f, u, r, o, tile_order=order)  # warm
    ctx.frame_stats()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ctx.RenderScreen(W, H, fb, pos, f, u, r, o, tile_order=order)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    st = ctx.frame_stats()
    iters, walk, end_r, box_r, next_r, end_l, box_l, next_l, life, drain = [int(v) for v in st.dbg][:10]
    probes = st.coarse_probes + st.fine_probes
    print("%s cam %s sched=%d: %.3f ms (counting kernel) rays %.2fM | iters/wave %.0f  walking lanes/iter %.1f  probes/iter %.1f | "
          "mean wave lifetime %.3f ms = %.0f%% of launch | drained iterations %.1f%%" % (
              name, cname, int(use_sched), ms, st.total_rays() / 1e6, iters / nwaves, walk / max(iters, 1), probes / max(iters, 1),
              life / nwaves / 1e5, 100.0 * life / nwaves / 1e5 / ms, 100.0 * drain / max(iters, 1)), flush=True)
    print("    phase executions per 100 iterations (lanes served per execution): next %.1f (%.1f)  end %.1f (%.1f)  box %.1f (%.1f)" % (
        100.0 * next_r / max(iters, 1), next_l / max(next_r, 1), 100.0 * end_r / max(iters, 1), end_l / max(end_r, 1),
        100.0 * box_r / max(iters, 1), box_l / max(box_r, 1)), flush=True)
    print("    share of wave time: ray-finished phase %.1f%%  box+end phases %.1f%%  (rest: probes, votes, queue)" % (
        100.0 * int(st.dbg[10]) / max(life, 1), 100.0 * int(st.dbg[11]) / max(life, 1)), flush=True)
