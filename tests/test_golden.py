"""Committed golden vectors (tests/golden/*.npz, generator: tests/golden/make_golden.py).

They were produced by this repository's CPU oracle and pin the oracle AND the HIP path against drift between rounds:
both must reproduce them bit for bit.  They do not pin parity with the reference itself."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
G = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(G)


def _load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


@pytest.mark.parametrize("name", sorted(G.TRACES))
def test_oracle_reproduces_golden_traces(vxo, name):
    w, o, d = G.trace_case(name)
    g = _load(name)
    assert str(g["inputs"]) == G.digest(o, d, w.coarse_bits, w.pool), "the seeded inputs changed"
    r = w.trace_batch(o, d)
    assert np.array_equal(r["hit"], g["hit"]) and np.array_equal(r["steps"], g["steps"])
    assert np.array_equal(r["voxel"], g["voxel"])
    assert np.array_equal(r["pos"].view(np.uint32), g["pos_bits"])
    assert np.array_equal(r["normal"].astype(np.int8), g["normal"])
    assert 0 < int(g["hit"].sum()) < len(g["hit"])   # the case has hits and misses


@pytest.mark.parametrize("name", sorted(G.FRAMES))
def test_oracle_reproduces_golden_frames(vxo, name):
    w, W, H, (pos, f, u, r), kw, stale = G.frame_case(name)
    g = _load(name)
    assert str(g["inputs"]) == G.digest(stale, w.coarse_bits, w.pool), "the seeded inputs changed"
    out = w.render(vxo.make_params(W, H, pos, f, u, r, **kw), fb=stale.copy(), want_hit=True)
    st = out["stats"]
    assert np.array_equal(out["fb"], g["fb"]) and np.array_equal(out["hit"], g["hit"])
    assert [st.primary_rays, st.shadow_rays, st.bounce_rays, st.primary_hits] == g["rays"].tolist()


def test_oracle_reproduces_golden_worlds(vxo):
    g = _load("worlds")
    for name in G.WORLDS:
        w = G.world_tables(name)
        assert G.digest(w.coarse_bits, w.brick_slot, w.bounds, w.pool) == str(g[name]), name


@pytest.fixture(scope="module")
def eng():
    import torch
    import voxelengine_amd as vx
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    ctx = vx.Context(0)
    yield vx, ctx, torch
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(G.TRACES))
def test_hip_reproduces_golden_traces(eng, vxo, name):
    vx, ctx, _ = eng
    w, o, d = G.trace_case(name)
    g = _load(name)
    ctx.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
    for variant in (4, 1):
        ctx.set_kernel_variant(variant)
        r = ctx.Raytrace(o, d)
        assert np.array_equal(r["hit"], g["hit"]) and np.array_equal(r["steps"], g["steps"])
        assert np.array_equal(r["voxel"], g["voxel"])
        assert np.array_equal(r["hitPoint"].view(np.uint32), g["pos_bits"])
        assert np.array_equal(r["normal"].astype(np.int8), g["normal"])
    ctx.set_kernel_variant(4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(G.FRAMES))
def test_hip_reproduces_golden_frames(eng, vxo, name):
    vx, ctx, torch = eng
    w, W, H, (pos, f, u, r), kw, stale = G.frame_case(name)
    g = _load(name)
    ctx.upload_world(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
    p = vxo.make_params(W, H, pos, f, u, r, **kw)
    ctx.SetEnvironment(list(p.env.light_dir), list(p.env.light_color), list(p.env.ambient))
    ctx.SetFOV(p.fov_deg)
    opts = vx.RenderOptions(mode=kw.get("mode", 0), checkerboard=bool(kw.get("checkerboard", 0)), shadow=bool(kw.get("shadow", 0)),
                            bounce_samples=kw.get("bounce_samples", 0), bounce_all_hits=bool(kw.get("bounce_all_hits", 0)),
                            bounce_depth=kw.get("bounce_depth", 1), frame_number=kw["frame_number"])
    for variant in (4, 1):
        ctx.set_kernel_variant(variant)
        ctx.frame_stats()
        fb = torch.from_numpy(stale.copy()).cuda()
        hit = torch.full((H, W), -1, dtype=torch.int64, device="cuda")
        ctx.RenderScreen(W, H, fb, pos, f, u, r, opts, hit_aov=hit)
        st = ctx.frame_stats()
        assert np.array_equal(fb.cpu().numpy(), g["fb"]) and np.array_equal(hit.cpu().numpy(), g["hit"])
        assert [st.primary_rays, st.shadow_rays, st.bounce_rays, st.primary_hits] == g["rays"].tolist()
    ctx.set_kernel_variant(4)


@pytest.mark.gpu
def test_hip_builder_reproduces_golden_worlds(eng, vxo):
    vx, ctx, _ = eng
    g = _load("worlds")
    for name, (gen, dims, f) in G.WORLDS.items():
        ctx.build_world(gen, dims[0], dims[1], dims[2], f)
        d = ctx.download_world()
        assert G.digest(d["coarse_bits"], d["brick_slot"], d["bounds"].reshape(-1, 6) if d["bounds"].ndim == 1 else d["bounds"],
                        d["pool"]) == str(g[name]), name
