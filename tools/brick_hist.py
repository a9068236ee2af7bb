"""Development (round 3): distinct bricks per wavefront iteration.

For every wave-loop iteration of a probe-counting launch: among the lanes that are walking INSIDE a brick, how many distinct
bricks are there?  (i) persistent waves with per-lane refill (the fused kernel, variant 5, and the traversal kernel T of
variant 6), (ii) tile-coherent waves (variant 0: one 8x8 pixel tile per wave, no refill).  Bench world and cameras, 1080p.

usage: VXRT_LIB=voxelengine_amd/csrc/libvxrt_exp.so brick_hist.py     (needs the experiments build)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voxelengine_amd as vx  # noqa: E402

X, Y, Z, F = 8192, 512, 8192, 32
W, H = 1920, 1080
CAMERAS = [("A", (0.50, 0.90, 0.50), (-0.45, 0.70, 0.0)), ("B", (0.10, 1.20, 0.10), (-0.60, 3.90, 0.0)),
           ("C", (0.50, 1.50, 0.50), (-1.5707, 0.0, 0.0)), ("D", (0.02, 0.55, 0.50), (-0.05, 1.5707, 0.0))]

ctx = vx.Context(0)
assert ctx.has_experiments(), "run with VXRT_LIB=.../libvxrt_exp.so (make -C voxelengine_amd/csrc libvxrt_exp.so)"
ctx.build_world(vx.GEN_PERLIN_REF, X, Y, Z, F)
light = float(np.float32(1.0) / np.sqrt(np.float32(3.0), dtype=np.float32))
ctx.SetEnvironment((light, light, light), (2, 2, 2), (0.5, 0.5, 0.5))
ctx.SetFOV(90.0)
fb = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
BINS = [(0, 0), (1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 64)]
print("bins (distinct bricks among the lanes walking inside a brick): " + "  ".join("%d-%d" % b if b[0] != b[1] else "%d" % b[0] for b in BINS))
for name, kw in (("primary only", {}), ("primary+shadow", dict(shadow=True)), ("primary+shadow+bounce", dict(shadow=True, bounce_samples=1))):
    for variant, label in ((0, "tile-coherent waves, no refill (variant 0)"), (5, "persistent waves, per-lane refill (variant 5)"),
                           (6, "traversal kernel T, per-lane refill from ray queues (variant 6)")):
        ctx.set_kernel_variant(variant)
        ctx.frame_stats()
        ctx.brick_histogram()
        for j, (cam, frac, euler) in enumerate(CAMERAS):
            f, u, r = vx.GetDirections(euler)
            ctx.RenderScreen(W, H, fb, (frac[0] * X, frac[1] * Y, frac[2] * Z), f, u, r,
                             vx.RenderOptions(frame_number=j + 1, collect_stats=True, **kw))
        torch.cuda.synchronize()
        st = ctx.frame_stats()
        hh = ctx.brick_histogram().astype(np.float64)
        h, hf = hh[:65], hh[65:]
        tot = h.sum()
        inb = h[1:].sum()
        mean = (h * np.arange(65)).sum() / max(inb, 1)
        cum = np.cumsum(h[1:]) / max(inb, 1)
        med = int(np.searchsorted(cum, 0.5)) + 1
        shares = [h[a:b + 1].sum() / max(tot, 1) for a, b in BINS]
        print("%-22s %-62s iterations %.3e  with lanes in bricks %.1f %%  mean %.1f  median %d  <=4: %.1f %%  <=8: %.1f %% | %s" % (
            name, label, tot, 100 * inb / max(tot, 1), mean, med, 100 * cum[3], 100 * cum[7],
            " ".join("%4.1f" % (100 * s) for s in shares)), flush=True)
        print("%-22s   iterations whose walking lanes are ALL inside bricks (no coarse walker): %.1f %%; of all iterations, in 1 brick %.1f %%, <=2 %.1f %%, <=4 %.1f %%" % (
            "", 100 * hf[1:].sum() / max(tot, 1), 100 * hf[1] / max(tot, 1), 100 * hf[1:3].sum() / max(tot, 1), 100 * hf[1:5].sum() / max(tot, 1)), flush=True)
ctx.set_kernel_variant(4)
