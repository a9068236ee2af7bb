import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import vxo
from tests import helpers
import voxelengine_amd as vx
w = vxo.World.generate(vxo.GEN_HASH_HEIGHTFIELD, 128, 128, 128, 16)
c2 = vx.Context(0)
c2.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
pos, f, u, r = helpers.camera("A", w.dims, vxo)
p0 = vxo.make_params(160, 90, pos, f, u, r)
c2.SetEnvironment(list(p0.env.light_dir), list(p0.env.light_color), list(p0.env.ambient))
for n in range(3):
    d_fb = torch.full((90, 160, 4), 255, dtype=torch.uint8, device="cuda")
    c2.RenderScreen(160, 90, d_fb, pos, f, u, r, vx.RenderOptions(checkerboard=True, bounce_samples=1))
    st = c2.frame_stats()
    p = vxo.make_params(160, 90, pos, f, u, r, frame_number=n, checkerboard=1, bounce_samples=1)
    o = w.render(p)
    g = d_fb.cpu().numpy()
    diff = (g != o['fb']).any(axis=2)
    print(n, "diff px", diff.sum(), "gpu rays", st.primary_rays, st.bounce_rays, "cpu", o['stats'].primary_rays, o['stats'].bounce_rays)
    ys, xs = np.nonzero(diff)
    for y, x in list(zip(ys, xs))[:5]:
        print("  ", x, y, g[y, x], o['fb'][y, x])
