"""Development: throughput of the batch query API (vxrt_trace_batch, device buffers) on BASELINE configs[0]'s
million-ray fan (128^3 world) and on camera-like ray sets in the bench world.

usage: batch_probe.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voxelengine_amd as vx  # noqa: E402
from tests import helpers  # noqa: E402


def run(ctx, o, d, label, reps=20):
    n = len(o)
    do, dd = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    pos = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    nrm = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    steps = torch.empty(n, dtype=torch.int32, device="cuda")
    hit = torch.empty(n, dtype=torch.uint8, device="cuda")
    vox = torch.empty(n, dtype=torch.int64, device="cuda")
    for variant in (4, 1):   # the product kernels (queue or one ray per lane, by batch size), the straightforward loops
        ctx.set_kernel_variant(variant)
        for _ in range(3):
            ctx.trace_batch_device(do, dd, n, pos, nrm, steps, hit, vox)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.trace_batch_device(do, dd, n, pos, nrm, steps, hit, vox)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%s: %d rays, kernel variant %d: %.3f ms per batch, %.0f Mrays/s (hits %.1f %%, mean steps %.1f)" % (
            label, n, variant, dt * 1e3, n / dt / 1e6, 100.0 * hit.float().mean().item(), steps.float().mean().item()), flush=True)
    ctx.set_kernel_variant(4)


ctx = vx.Context(0)
ctx.build_world(vx.GEN_HASH_HEIGHTFIELD, 128, 128, 128, 8)
o, d = helpers.fibonacci_fan(1_000_000, (64.0, 100.0, 64.0))
run(ctx, o, d, "configs[0] fan, 128^3 f=8")
ctx.build_world(vx.GEN_PERLIN_REF, 8192, 512, 8192, 32)
rng = np.random.default_rng(1)
n = 4_000_000
o = np.empty((n, 3), np.float32)
o[:, 0] = rng.uniform(0, 8192, n)
o[:, 1] = rng.uniform(300, 600, n)
o[:, 2] = rng.uniform(0, 8192, n)
d = rng.normal(size=(n, 3)).astype(np.float32)
d[:, 1] = -np.abs(d[:, 1]) * 0.5
run(ctx, o, d, "random downward rays, 8192x512x8192 f=32")
