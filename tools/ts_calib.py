"""Development (round 3): what would a traversal-only kernel cost on the bench workload's rays?

The three ray generations of bench.py's frames (primary, shadow, bounce) are rebuilt with torch from the batch API's own
results and timed, generation by generation, on the existing persistent batch kernel (k_trace_batch_persist: a ray queue,
per-lane refill, no pixel chain) -- next to the fused render kernel on the same frames.  Directions of the bounce rays are
drawn with torch's RNG (same distribution as the hash, not the same values): timing only, not a parity check.

usage: ts_calib.py [views]
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
L = torch.tensor([light, light, light], device=dev)
Lu = L / L.norm()


def camera_rays(frac, euler):
    f, u, r = [torch.tensor(v, device=dev) for v in vx.GetDirections(euler)]
    pos = torch.tensor([frac[0] * X, frac[1] * Y, frac[2] * Z], dtype=torch.float32, device=dev)
    k = float(np.tan(np.float32(90.0 * 3.1415 / 180.0) / 2))
    xs = (torch.arange(W, device=dev, dtype=torch.float32) / W) * 2 - 1
    ys = (torch.arange(H, device=dev, dtype=torch.float32) / H) * 2 - 1
    # 8x8 pixel tiles in row-major tile order, pixels row-major inside a tile: the order the render kernel hands pixels out
    ty, tx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    key = ((ty // 8) * ((W + 7) // 8) + tx // 8) * 64 + (ty % 8) * 8 + tx % 8
    order = torch.argsort(key.reshape(-1))
    d = f[None, None, :] + (xs[None, :, None] * k * (W / H)) * r[None, None, :] + (ys[:, None, None] * k) * u[None, None, :]
    d = (d / d.norm(dim=2, keepdim=True)).reshape(-1, 3)[order].contiguous()
    o = pos[None, :].expand(d.shape[0], 3).contiguous()
    return o, d


def trace(o, d, max_steps=2048, reps=5, label=""):
    n = o.shape[0]
    pos = torch.empty((n, 3), dtype=torch.float32, device=dev)
    nrm = torch.empty((n, 3), dtype=torch.float32, device=dev)
    steps = torch.empty(n, dtype=torch.int32, device=dev)
    hit = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.set_batch_max_steps(max_steps)
    for _ in range(2):
        ctx.trace_batch_device(o, d, n, pos, nrm, steps, hit, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.trace_batch_device(o, d, n, pos, nrm, steps, hit, None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("  %-8s %9d rays  %7.3f ms  %7.0f Mrays/s  hits %.1f %%  mean steps %.1f" % (
        label, n, dt * 1e3, n / dt / 1e6, 100.0 * hit.float().mean().item(), steps.float().mean().item()), flush=True)
    return dt, pos, nrm, hit


def generations(views):
    o = torch.cat([v[0] for v in views])
    d = torch.cat([v[1] for v in views])
    total, rays = 0.0, 0
    t, pos, nrm, hit = trace(o, d, label="primary")
    total += t
    rays += o.shape[0]
    h = hit.bool()
    P, Nn = pos[h], -nrm[h]  # pixel normal = -step normal (Renderer.cu:212)
    so = P + Lu[None, :] * 0.01
    sd = Lu[None, :].expand(so.shape[0], 3).contiguous()
    t, _, _, shit = trace(so.contiguous(), sd, label="shadow")
    total += t
    rays += so.shape[0]
    ldot = torch.clamp((Nn * L[None, :]).sum(dim=1), min=0) * (~shit.bool()).float()
    g = ldot == 0
    P2, N2 = P[g], Nn[g]
    torch.manual_seed(1)
    bd = torch.rand((P2.shape[0], 3), device=dev) * 2 - 1
    bd = bd / bd.norm(dim=1, keepdim=True)
    flip = (bd * N2).sum(dim=1) < 0
    bd = torch.where(flip[:, None], bd - 2 * N2 * (N2 * bd).sum(dim=1, keepdim=True), bd)
    bo = P2 + N2 * 0.01
    t, _, _, _ = trace(bo.contiguous(), bd.contiguous(), max_steps=8, label="bounce")
    total += t
    rays += bo.shape[0]
    print("  traversal-only, three generations: %.3f ms for %d rays = %.0f Mrays/s" % (total * 1e3, rays, rays / total / 1e6), flush=True)
    return total, rays


cams = [camera_rays(*c) for c in CAMERAS]
for nv in sorted({1, V}):
    print("== %d view(s), batch kernel (ray queue), generation by generation" % nv, flush=True)
    for variant in (2, 0):
        ctx.set_kernel_variant(variant)
        print(" batch kernel variant %d (%s)" % (variant, "persistent queue" if variant == 2 else "one ray per lane"), flush=True)
        generations([cams[i % 4] for i in range(nv)])
    ctx.set_kernel_variant(4)
    # the fused render kernel on the same frames
    fbs = torch.zeros((nv, H, W, 4), dtype=torch.uint8, device=dev)
    views = []
    for j in range(nv):
        frac, euler = CAMERAS[j % 4]
        f, u, r = vx.GetDirections(euler)
        views.append(dict(fb=fbs[j], origin=(frac[0] * X, frac[1] * Y, frac[2] * Z), fwd=f, up=u, right=r, frame_number=j + 1))
    for name, o in (("primary only", vx.RenderOptions()), ("primary+shadow", vx.RenderOptions(shadow=True)),
                    ("primary+shadow+bounce", vx.RenderOptions(shadow=True, bounce_samples=1))):
        for _ in range(2):
            ctx.RenderViews(W, H, views, o) if nv > 1 else ctx.RenderScreen(W, H, fbs[0], views[0]["origin"], views[0]["fwd"], views[0]["up"], views[0]["right"], o)
        torch.cuda.synchronize()
        ctx.frame_stats()
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.RenderViews(W, H, views, o) if nv > 1 else ctx.RenderScreen(W, H, fbs[0], views[0]["origin"], views[0]["fwd"], views[0]["up"], views[0]["right"], o)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        st = ctx.frame_stats()
        n = st.total_rays() / 5
        print("  fused render kernel, %-22s %9d rays  %7.3f ms  %7.0f Mrays/s" % (name, n, dt * 1e3, n / dt / 1e6), flush=True)
