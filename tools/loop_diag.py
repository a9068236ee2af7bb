"""Wave-loop diagnostics of the default render kernel on the bench world: iterations, walking-lane share, phase runs."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import voxelengine_amd as vx  # noqa: E402

X, Y, Z, F, gen, W, H, _, _ = bench.WORKLOADS["c3_8k_1080p_shadow_bounce"]
ctx = vx.Context(0)
ctx.build_world(gen, X, Y, Z, F)
l = float(np.float32(1.0) / np.sqrt(np.float32(3.0), dtype=np.float32))
ctx.SetEnvironment((l, l, l), (2, 2, 2), (0.5, 0.5, 0.5))
ctx.SetFOV(90.0)
fb = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
nwaves = ((W + 7) // 8) * ((H + 7) // 8)
for shadow, bounce in ((0, 0), (1, 0), (1, 1)):
    for cname, frac, euler in bench.CAMERAS:
        f, u, r = vx.GetDirections(euler)
        pos = (frac[0] * X, frac[1] * Y, frac[2] * Z)
        ctx.frame_stats()
        ctx.RenderScreen(W, H, fb, pos, f, u, r, vx.RenderOptions(shadow=bool(shadow), bounce_samples=bounce, frame_number=1,
                                                               collect_stats=True))
        st = ctx.frame_stats()
        iters, walk, endr, boxr = [int(v) for v in st.dbg][:4]
        probes = st.coarse_probes + st.fine_probes
        print("shadow=%d bounce=%d cam %s: rays %.2fM probes/ray %.1f | iters/wave %.0f  walking lanes/iter %.1f  useful probes/iter %.1f "
              "(%.0f%%)  end runs/wave %.1f box runs/wave %.1f | END events/ray %.2f BOX events/ray %.2f" % (
                  shadow, bounce, cname, st.total_rays() / 1e6, probes / st.total_rays(), iters / nwaves, walk / max(iters, 1),
                  probes / max(iters, 1), 100.0 * probes / max(iters, 1) / 64, endr / nwaves, boxr / nwaves,
                  (st.total_rays() + 2 * st.brick_entries) / st.total_rays(), st.brick_entries / st.total_rays()), flush=True)
