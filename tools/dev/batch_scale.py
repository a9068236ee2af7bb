"""Development: does the batch queue's time scale with the batch, or is there a fixed part?  k_trace_batch_persist on
n incoherent rays of tools/batch_probe.py's kind for several n (one library per process: VXRT_LIB).

usage: batch_scale.py [n ...]     (default 1, 2, 4, 8, 16 million)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import voxelengine_amd as vx  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1_000_000, 2_000_000, 4_000_000, 8_000_000, 16_000_000]
ctx = vx.Context(0)
ctx.build_world(vx.GEN_PERLIN_REF, 8192, 512, 8192, 32)
rng = np.random.default_rng(1)
nmax = max(sizes)
o = np.empty((nmax, 3), np.float32)
o[:, 0] = rng.uniform(0, 8192, nmax)
o[:, 1] = rng.uniform(300, 600, nmax)
o[:, 2] = rng.uniform(0, 8192, nmax)
d = rng.normal(size=(nmax, 3)).astype(np.float32)
d[:, 1] = -np.abs(d[:, 1]) * 0.5
do, dd = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
pos = torch.empty((nmax, 3), dtype=torch.float32, device="cuda")
nrm = torch.empty((nmax, 3), dtype=torch.float32, device="cuda")
steps = torch.empty(nmax, dtype=torch.int32, device="cuda")
hit = torch.empty(nmax, dtype=torch.uint8, device="cuda")
vox = torch.empty(nmax, dtype=torch.int64, device="cuda")
for n in sizes:
    for _ in range(3):
        ctx.trace_batch_device(do, dd, n, pos, nrm, steps, hit, vox)
    torch.cuda.synchronize()
    reps = max(4, 40_000_000 // n)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.trace_batch_device(do, dd, n, pos, nrm, steps, hit, vox)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%s: %9d rays  %.3f ms  %.0f Mrays/s" % (os.path.basename(os.environ.get("VXRT_LIB", "libvxrt.so")), n, dt * 1e3, n / dt / 1e6), flush=True)
