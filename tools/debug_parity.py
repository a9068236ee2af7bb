import sys, numpy as np
sys.path.insert(0, '.')
from oracle import vxo
from tests import helpers
import voxelengine_amd as vx
ctx = vx.Context(0)
import ast
cfg = ast.literal_eval(sys.argv[1]) if len(sys.argv) > 1 else (8, (64, 64, 64), 0.01, 1)
w = helpers.random_voxel_world(vxo, cfg[1], cfg[0], cfg[2], cfg[3])
ctx.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
o, d = helpers.mixed_rays(w.dims, 30000, cfg[3])
cpu = w.trace_batch(o, d); gpu = ctx.Raytrace(o, d, want_stats=True)
bad = np.nonzero((gpu['steps'] != cpu['steps']) | (gpu['hit'] != cpu['hit']) | (gpu['voxel'] != cpu['voxel']))[0]
print("mismatches", len(bad), "of", len(o))
for i in bad[:12]:
    print(i, "o", o[i].tolist(), "d", d[i].tolist())
    print("   cpu", cpu['hit'][i], cpu['steps'][i], cpu['pos'][i].tolist(), cpu['normal'][i].tolist(), cpu['voxel'][i])
    print("   gpu", gpu['hit'][i], gpu['steps'][i], gpu['hitPoint'][i].tolist(), gpu['normal'][i].tolist(), gpu['voxel'][i])
pb = np.nonzero((gpu['hitPoint'].view(np.uint32) != cpu['pos'].view(np.uint32)).any(axis=1))[0]
print("pos mismatches", len(pb), pb[:10])
