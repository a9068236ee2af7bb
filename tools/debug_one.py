import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vxo
from tests import helpers
import voxelengine_amd as vx
ctx = vx.Context(0)
w = helpers.random_voxel_world(vxo, (128,128,128), 16, 0.002, 3)
ctx.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
o = np.array([[12000.0, 12000.0, 6000.0]], np.float32); d = np.array([[-11936.0, -11936.0, -5936.0]], np.float32)
c = w.trace_batch(o, d)
print("cpu", c['hit'][0], c['steps'][0], c['pos'][0].tolist(), c['normal'][0].tolist(), (c['stats'].coarse_probes, c['stats'].brick_entries, c['stats'].fine_probes))
for v in (0,):
    ctx.set_kernel_variant(v)
    g = ctx.Raytrace(o, d, want_stats=True)
    print("gpu variant", v, g['hit'][0], g['steps'][0], g['hitPoint'][0].tolist(), g['normal'][0].tolist(), (g['stats'].coarse_probes, g['stats'].brick_entries, g['stats'].fine_probes))
