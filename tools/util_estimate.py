"""Estimate lock-step lane utilisation of primary rays: per 8x8 pixel tile, mean steps / max steps."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import voxelengine_amd as vx
X, Y, Z, F, gen, W, H, shadow, bounce = bench.WORKLOADS["c3_8k_1080p_shadow_bounce"]
ctx = vx.Context(0)
ctx.build_world(gen, X, Y, Z, F)
for cname, frac, euler in bench.CAMERAS:
    f, u, r = vx.GetDirections(euler)
    pos = np.array((frac[0] * X, frac[1] * Y, frac[2] * Z), np.float32)
    xs = (np.arange(W, dtype=np.float32) / W) * 2 - 1
    ys = (np.arange(H, dtype=np.float32) / H) * 2 - 1
    k = np.tan(np.float32(90 * 3.1415 / 180.0 / 2))
    d = f[None, None, :] + xs[None, :, None] * k * (W / H) * r[None, None, :] + ys[:, None, None] * k * u[None, None, :]
    o = np.broadcast_to(pos, d.shape)
    res = ctx.Raytrace(o.reshape(-1, 3), d.reshape(-1, 3).astype(np.float32), want_stats=True)
    steps = res["steps"].reshape(H, W).astype(np.float64) + 1
    hit = res["hit"].reshape(H, W)
    Hc, Wc = H // 8 * 8, W // 8 * 8
    t = steps[:Hc, :Wc].reshape(Hc // 8, 8, Wc // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    util = t.sum() / (64 * t.max(axis=1).sum())
    t16 = steps[:H // 16 * 16, :W // 16 * 16].reshape(H // 16, 16, W // 16, 16).transpose(0, 2, 1, 3).reshape(-1, 256)
    print(cname, "hit frac %.2f" % hit.mean(), "mean steps %.1f" % steps.mean(), "p50 %.0f p90 %.0f p99 %.0f max %.0f" % tuple(np.percentile(steps, [50, 90, 99, 100])),
          "lockstep util 8x8 = %.2f" % util, " per-tile mean of (mean/max) = %.2f" % (t.mean(axis=1) / t.max(axis=1)).mean(), flush=True)
