"""Development: the scenario of tests/test_gpu_parity.py::test_trace_batch_persistent_queue, printing the rays that differ."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import voxelengine_amd as vx
from oracle import vxo
from tests import helpers

ctx = vx.Context(0)
ctx.set_persistent_waves_per_cu(1)
w = helpers.random_voxel_world(vxo, (128, 128, 128), 16, 0.004, 21)
ctx.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
for n in (524288 + 63,):
    o, d = helpers.mixed_rays(w.dims, n, 5)
    cpu = w.trace_batch(o, d)
    for stats in (True, False):
        g = ctx.Raytrace(o, d, want_stats=stats)
        bad = np.flatnonzero((g["hitPoint"].view(np.uint32) != cpu["pos"].view(np.uint32)).any(axis=1) | (g["hit"] != cpu["hit"]) |
                             (g["steps"] != cpu["steps"]) | (g["voxel"] != cpu["voxel"]) | (g["normal"] != cpu["normal"]).any(axis=1))
        print("n=%d stats=%d: %d rays differ" % (n, stats, len(bad)), flush=True)
        for i in bad[:12]:
            print("  pos bits gpu %s cpu %s" % (g["hitPoint"][i].view(np.uint32).tolist(), cpu["pos"][i].view(np.uint32).tolist()))
            print("  ray %d (family %d of 6, ticket %d lane %d) o=%r d=%r\n     gpu hit=%d steps=%d vox=%d pos=%r n=%r\n     cpu hit=%d steps=%d vox=%d pos=%r n=%r" % (
                i, i // max(n // 6, 1), i // 64, i % 64, o[i].tolist(), d[i].tolist(), g["hit"][i], g["steps"][i], g["voxel"][i], g["hitPoint"][i].tolist(),
                g["normal"][i].tolist(), cpu["hit"][i], cpu["steps"][i], cpu["voxel"][i], cpu["pos"][i].tolist(), cpu["normal"][i].tolist()))
