"""Development (round 3): the traversal/shading pipeline (variant 6) against the fused kernel (variant 5) on the bench
workload: time per step, frames compared byte for byte, and T's wave-loop diagnostics.

usage: ts_ab.py [views] [variants...]
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
variants = [int(v) for v in sys.argv[2:]] or [5, 6, 5, 6]

ctx = vx.Context(0)
ctx.build_world(vx.GEN_PERLIN_REF, X, Y, Z, F)
light = float(np.float32(1.0) / np.sqrt(np.float32(3.0), dtype=np.float32))
ctx.SetEnvironment((light, light, light), (2, 2, 2), (0.5, 0.5, 0.5))
ctx.SetFOV(90.0)
dev = torch.device("cuda")


def views_into(fbs, hits=None):
    out = []
    for j in range(fbs.shape[0]):
        frac, euler = CAMERAS[j % 4]
        f, u, r = vx.GetDirections(euler)
        v = dict(fb=fbs[j], origin=(frac[0] * X, frac[1] * Y, frac[2] * Z), fwd=f, up=u, right=r, frame_number=j + 1)
        if hits is not None:
            v["hit_aov"] = hits[j]
        out.append(v)
    return out


ref = {}
MODES = (("primary only", {}), ("primary+shadow", dict(shadow=True)), ("primary+shadow+bounce", dict(shadow=True, bounce_samples=1)))
if os.environ.get("TS_MODES"):  # e.g. TS_MODES=2: only the full ray set
    MODES = tuple(MODES[int(i)] for i in os.environ["TS_MODES"].split(","))
NVS = sorted({1, V}) if not os.environ.get("TS_ONLY_V") else [V]
for name, kw in MODES:
    for nv in NVS:
        print("== %s, %d view(s) per step" % (name, nv), flush=True)
        for variant in variants:
            ctx.set_kernel_variant(variant)
            fbs = torch.zeros((nv, H, W, 4), dtype=torch.uint8, device=dev)
            hits = torch.full((nv, H, W), -5, dtype=torch.int64, device=dev)
            views = views_into(fbs, hits)
            o = vx.RenderOptions(**kw)
            run = (lambda: ctx.RenderViews(W, H, views, o)) if nv > 1 else (
                lambda: ctx.RenderScreen(W, H, fbs[0], views[0]["origin"], views[0]["fwd"], views[0]["up"], views[0]["right"],
                                         vx.RenderOptions(frame_number=1, **kw), hit_aov=hits[0]))
            for _ in range(4):  # (variant 6 takes its workspace from a ring of three: every entry sized before the timed runs)
                run()
            torch.cuda.synchronize()
            ctx.frame_stats()
            reps = 5 if nv > 1 else 20
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            st = ctx.frame_stats()
            n = st.total_rays() / reps
            key = (name, nv)
            same = ""
            if key in ref:
                same = "  frames %s, hit AOV %s, rays %s" % ("equal" if torch.equal(ref[key][0], fbs) else "DIFFER",
                                                             "equal" if torch.equal(ref[key][1], hits) else "DIFFER",
                                                             "equal" if ref[key][2] == n else "DIFFER (%d vs %d)" % (ref[key][2], n))
            else:
                ref[key] = (fbs.clone(), hits.clone(), n)
            print("  variant %d: %8.3f ms %7.0f Mrays/s%s" % (variant, dt * 1e3, n / dt / 1e6, same), flush=True)
            if nv > 1:
                ov = views_into(fbs)
                ctx.RenderViews(W, H, ov, vx.RenderOptions(collect_stats=True, **kw))
                torch.cuda.synchronize()
                s2 = ctx.frame_stats()
                g = [int(v) for v in s2.dbg]
                it = max(g[0], 1)
                print("     diag: wave-iterations %.3e  walking lanes/iter %.1f | per 100 iter: end %.1f (%.1f lanes)  box %.1f (%.1f lanes)  "
                      "next %.1f (%.1f lanes) | time share: ray-finished %.1f %%  box+end (first vote) %.1f %% | probes %d/%d/%d" % (g[0], g[1] / it, 100.0 * g[2] / it, g[5] / max(g[2], 1), 100.0 * g[3] / it,
                                                                   g[6] / max(g[3], 1), 100.0 * g[4] / it, g[7] / max(g[4], 1),
                                                                   100.0 * g[10] / max(g[8], 1), 100.0 * g[11] / max(g[8], 1),
                                                                   s2.coarse_probes, s2.brick_entries, s2.fine_probes), flush=True)
ctx.set_kernel_variant(4)
