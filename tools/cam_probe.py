"""Render the bench cameras on the bench world and dump thumbnails + per-camera stats (camera tuning aid)."""
import sys, time, json, numpy as np, torch
sys.path.insert(0, '.')
import voxelengine_amd as vx
import bench
from PIL import Image
name = sys.argv[1] if len(sys.argv) > 1 else "c3_8k_1080p_shadow_bounce"
X, Y, Z, F, gen, W, H, shadow, bounce = bench.WORKLOADS[name]
ctx = vx.Context(0)
t = time.time(); info = ctx.build_world(gen, X, Y, Z, F); ctx.synchronize(); print("build s", time.time() - t, "bricks", info.nslots, "of", info.ncells, "GiB", info.hbm_bytes / 2**30, flush=True)
l = float(np.float32(1.0) / np.sqrt(np.float32(3.0), dtype=np.float32))
ctx.SetEnvironment((l, l, l), (2, 2, 2), (0.5, 0.5, 0.5)); ctx.SetFOV(90.0)
fb = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
extra = [("E", (0.5, 0.6, 0.5), (-0.3, 2.0, 0.0)), ("F", (0.3, 0.99, 0.7), (-0.2, 5.0, 0.0)), ("G", (0.7, 0.75, 0.2), (-0.6, 1.0, 0.0))]
for cname, frac, euler in bench.CAMERAS + extra:
    f, u, r = vx.GetDirections(euler)
    pos = (frac[0] * X, frac[1] * Y, frac[2] * Z)
    ctx.frame_stats()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.RenderScreen(W, H, fb, pos, f, u, r, vx.RenderOptions(shadow=bool(shadow), bounce_samples=bounce, frame_number=1, collect_stats=True))
    torch.cuda.synchronize()
    a.record(); ctx.RenderScreen(W, H, fb, pos, f, u, r, vx.RenderOptions(shadow=bool(shadow), bounce_samples=bounce, frame_number=1)); b.record(); torch.cuda.synchronize()
    st = ctx.frame_stats()
    ms = a.elapsed_time(b)
    rays = st.total_rays() // 2
    print(cname, "ms %.3f" % ms, "Mrays/s %.1f" % (rays / ms / 1e3), "rays", rays, "hits", st.primary_hits // 2, "shadow", st.shadow_rays // 2, "bounce", st.bounce_rays // 2,
          "Nc/ray %.1f Nb/ray %.2f Nf/ray %.1f" % (st.coarse_probes / rays, st.brick_entries / rays, st.fine_probes / rays), flush=True)
    Image.fromarray(fb.cpu().numpy()[::4, ::4, [2, 1, 0]]).save("gpurun_out/cam_%s_%s.png" % (name[:2], cname))
