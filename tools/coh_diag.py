"""Development (round 3): wave-loop diagnostics of the render kernels per ray generation -- tile-coherent waves (variant 0:
one 8x8 tile per wave, no refill) against persistent waves with per-lane refill (variants 2, 5), on the bench workload's
frames with primary rays only / + shadow / + bounce.  Prints, per launch shape: time, iterations, walking lanes per
iteration, phase executions and the lanes they served.

usage: VXRT_LIB=voxelengine_amd/csrc/libvxrt_exp.so coh_diag.py [views]   (variant 0 lives in the experiments build)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voxelengine_amd as vx  # noqa: E402

X, Y, Z, F = 8192, 512, 8192, 32
W, H = 1920, 1080
CAMERAS = [((0.50, 0.90, 0.50), (-0.45, 0.70, 0.0)), ((0.10, 1.20, 0.10), (-0.60, 3.90, 0.0)),
           ((0.50, 1.50, 0.50), (-1.5707, 0.0, 0.0)), ((0.02, 0.55, 0.50), (-0.05, 1.5707, 0.0))]
V = int(sys.argv[1]) if len(sys.argv) > 1 else 16

ctx = vx.Context(0)
ctx.build_world(vx.GEN_PERLIN_REF, X, Y, Z, F)
light = float(np.float32(1.0) / np.sqrt(np.float32(3.0), dtype=np.float32))
ctx.SetEnvironment((light, light, light), (2, 2, 2), (0.5, 0.5, 0.5))
ctx.SetFOV(90.0)
dev = torch.device("cuda")
fbs = torch.zeros((V, H, W, 4), dtype=torch.uint8, device=dev)
views = []
for j in range(V):
    frac, euler = CAMERAS[j % 4]
    f, u, r = vx.GetDirections(euler)
    views.append(dict(fb=fbs[j], origin=(frac[0] * X, frac[1] * Y, frac[2] * Z), fwd=f, up=u, right=r, frame_number=j + 1))

for name, kw in (("primary only", {}), ("primary+shadow", dict(shadow=True)), ("primary+shadow+bounce", dict(shadow=True, bounce_samples=1))):
    print("== %s, %d views per step" % (name, V), flush=True)
    for variant in (0, 2, 5):
        ctx.set_kernel_variant(variant)
        o = vx.RenderOptions(**kw)
        for _ in range(2):
            ctx.RenderViews(W, H, views, o)
        torch.cuda.synchronize()
        ctx.frame_stats()
        t0 = time.perf_counter()
        for _ in range(4):
            ctx.RenderViews(W, H, views, o)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        n = ctx.frame_stats().total_rays() / 4
        ctx.RenderViews(W, H, views, vx.RenderOptions(collect_stats=True, **kw))
        torch.cuda.synchronize()
        st = ctx.frame_stats()
        g = [int(v) for v in st.dbg]
        it = max(g[0], 1)
        probes = st.coarse_probes + st.fine_probes
        print("  variant %d: %8.3f ms %6.0f Mrays/s | wave-iterations %.3e  walking lanes/iter %.1f  useful probes/iter %.1f | per 100 iter: "
              "end %.1f (%.1f lanes)  box %.1f (%.1f lanes)  next %.1f (%.1f lanes) | probes/ray %.1f  entries/ray %.2f" % (
                  variant, dt * 1e3, n / dt / 1e6, g[0], g[1] / it, probes / it, 100.0 * g[2] / it, g[5] / max(g[2], 1),
                  100.0 * g[3] / it, g[6] / max(g[3], 1), 100.0 * g[4] / it, g[7] / max(g[4], 1),
                  probes / max(st.total_rays(), 1), st.brick_entries / max(st.total_rays(), 1)), flush=True)
ctx.set_kernel_variant(4)
