"""A/B timing of kernel variants on the bench workload: interleaved rounds in one process (median + min)."""
import sys

import numpy as np
import torch

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import voxelengine_amd as vx  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3_8k_1080p_shadow_bounce"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
variants = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "1"])]
X, Y, Z, F, gen, W, H, shadow, bounce = bench.WORKLOADS[name]
ctx = vx.Context(0)
info = ctx.build_world(gen, X, Y, Z, F)
l = float(np.float32(1.0) / np.sqrt(np.float32(3.0), dtype=np.float32))
ctx.SetEnvironment((l, l, l), (2, 2, 2), (0.5, 0.5, 0.5))
ctx.SetFOV(90.0)
fb = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
cams = []
for cname, frac, euler in bench.CAMERAS:
    f, u, r = vx.GetDirections(euler)
    cams.append((cname, (frac[0] * X, frac[1] * Y, frac[2] * Z), f, u, r))
opts = vx.RenderOptions(shadow=bool(shadow), bounce_samples=bounce, frame_number=1,
                        tile_schedule=not os.environ.get("AB_NOSCHED"))  # AB_SCHEDULE=1: host-made order instead
sched = {}
if os.environ.get('AB_SCHEDULE'):
    for cname, pos, f, u, r in cams:
        sched[cname] = torch.from_numpy(vx.tile_schedule(W, range(H), f, u, r, 90.0, H).astype(np.int32)).cuda()
times = {(v, c[0]): [] for v in variants for c in cams}
rays = {}
ref = {}
for rnd in range(rounds + 1):
    for v in variants:
        ctx.set_kernel_variant(v)
        for cname, pos, f, u, r in cams:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ctx.RenderScreen(W, H, fb, pos, f, u, r, opts, tile_order=sched.get(cname))
            b.record()
            torch.cuda.synchronize()
            if rnd == 0:
                rays[cname] = ctx.frame_stats().total_rays()
                h = fb.clone()
                if cname in ref:
                    assert torch.equal(ref[cname], h), "variants disagree on camera " + cname
                ref[cname] = h
            else:
                times[(v, cname)].append(a.elapsed_time(b))
ctx.frame_stats()
for v in variants:
    tot_med = 0
    tot_rays = 0
    for cname, *_ in cams:
        t = np.array(times[(v, cname)])
        tot_med += np.median(t)
        tot_rays += rays[cname]
        print("variant %d cam %s median %.3f ms min %.3f ms  %.0f Mrays/s" % (v, cname, np.median(t), t.min(),
                                                                              rays[cname] / np.median(t) / 1e3))
    print("variant %d ALL  %.3f ms per 4 frames  %.0f Mrays/s" % (v, tot_med, tot_rays / tot_med / 1e3), flush=True)
