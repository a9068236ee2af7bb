"""Hand-derived known-answer tests for the renderer half of the CPU oracle (Renderer.cu semantics).

Expected pixels are computed here in numpy binary32 straight from the reference's formulas (file:line in the
docstrings), independently of the oracle's C code: camera mapping with pi = 3.1415, miss colour = ray direction,
clamp * 255 with truncation, b,g,r,a byte order, crosshair, checkerboard row mapping with preserved stale pixels,
lighting / ambient lerp / specular pow 32 / Reinhard tonemap, shadow ray, the `lDot == 0` gate and the binary
occlusion sample.  Parity of the oracle with the reference itself stays unpinned."""
import math

import numpy as np

f32 = np.float32


def _empty_world(vxo, size=64, factor=8):
    return vxo.World.from_voxels(np.zeros((size, size, size), bool), factor)


def _ray_dirs(W, H, fwd, up, right, ys=None, fov_deg=90.0):
    """getRayDirection, Renderer.cu:44-59: fov = FOV * 3.1415 / 180.0 in double, narrowed to float; tanf(fov/2);
    dir = normalize(fwd + (2u-1)*tan*aspect*right + (2v-1)*tan*up), u = x/W, v = y/H, all in binary32."""
    fov = f32(float(fov_deg) * 3.1415 / 180.0)
    t = f32(math.tan(float(fov / f32(2.0))))          # tanf, correctly rounded here
    aspect = f32(W) / f32(H)
    kx, ky = f32(t * aspect), t
    xs = np.arange(W, dtype=np.float32)
    ys = np.arange(H, dtype=np.float32) if ys is None else np.asarray(ys, np.float32)
    su = (xs / f32(W)) * f32(2) - f32(1)
    sv = (ys / f32(H)) * f32(2) - f32(1)
    su, sv = np.meshgrid(su, sv)                      # [row, col]
    comp = []
    for a in range(3):
        comp.append((f32(fwd[a]) + (su * kx) * f32(right[a])) + (sv * ky) * f32(up[a]))
    x, y, z = comp
    inv = f32(1.0) / np.sqrt((x * x + y * y) + z * z, dtype=np.float32)
    return np.stack([x * inv, y * inv, z * inv], axis=-1).astype(np.float32)


def _to_bgra(rgb):
    """setPixelColor, Renderer.cu:72-87: clamp to [0,1], * 255, truncate; bytes b,g,r,a (SDLRenderer.h:8-11)."""
    c = np.clip(rgb.astype(np.float32), f32(0), f32(1))
    q = (c * f32(255)).astype(np.uint8)               # truncation of a value in [0, 255]
    out = np.empty(rgb.shape[:-1] + (4,), np.uint8)
    out[..., 0], out[..., 1], out[..., 2], out[..., 3] = q[..., 2], q[..., 1], q[..., 0], 255
    return out


FWD, UP, RIGHT = (0.0, 0.0, -1.0), (0.0, -1.0, 0.0), (1.0, 0.0, 0.0)   # GetDirections(0,0,0): Renderer.cu:27-42


def test_get_directions_of_zero_euler(vxo):
    f, u, r = vxo.get_directions((0.0, 0.0, 0.0))
    assert tuple(np.asarray(f) + 0.0) == FWD and tuple(np.asarray(u) + 0.0) == UP and tuple(np.asarray(r) + 0.0) == RIGHT


def test_miss_colour_camera_mapping_and_pixel_format(vxo):
    """An empty world: every pixel is a miss and shows its ray direction (Renderer.cu:254-258); the pixel whose
    launch coordinates are (W>>1, H>>1) carries the white crosshair (:261-268)."""
    w = _empty_world(vxo)
    W, H = 40, 24
    p = vxo.make_params(W, H, (32.0, 32.0, 32.0), FWD, UP, RIGHT, frame_number=1)
    fb = w.render(p, fb=np.zeros((H, W, 4), np.uint8))["fb"]
    want = _to_bgra(_ray_dirs(W, H, FWD, UP, RIGHT))
    want[H >> 1, W >> 1] = (255, 255, 255, 255)
    assert np.array_equal(fb, want)
    assert (fb[..., 3] == 255).all()
    # direction components below zero clamp to 0: looking down -z, the red channel (x) is 0 on the left half
    assert (fb[:, :W // 2, 2] == 0).all() and (fb[:, W // 2 + 1:, 2] > 0).all()


def test_checkerboard_rows_and_stale_pixels(vxo):
    """ENABLE_CHECKERBOARD_RENDER, Renderer.cu:186-196,311-316: the launch has H/2 rows; thread (x, y') writes row
    y = 2y' + (x even ? 1 : 0) + (FrameNumber even ? 1 : 0) if y < H; everything else keeps its old contents."""
    w = _empty_world(vxo)
    W, H = 16, 10
    stale = np.random.default_rng(2).integers(0, 255, size=(H, W, 4), dtype=np.uint8)
    for frame in (0, 1, 2, 3):
        p = vxo.make_params(W, H, (32.0, 32.0, 32.0), FWD, UP, RIGHT, frame_number=frame, checkerboard=1)
        fb = w.render(p, fb=stale.copy())["fb"]
        want = stale.copy()
        colours = _to_bgra(_ray_dirs(W, H, FWD, UP, RIGHT))   # indexed by the frame row the thread ends up writing
        for yp in range(H // 2):
            for x in range(W):
                y = 2 * yp + (1 if x % 2 == 0 else 0) + (1 if frame % 2 == 0 else 0)
                if y < H:
                    want[y, x] = colours[y, x]
        # the crosshair needs launch coordinates (W>>1, H>>1); the launch only has H/2 rows, so it is never drawn
        assert np.array_equal(fb, want), frame
        written = (fb != stale).any(axis=2).sum()
        assert written <= W * (H // 2)


def _floor_world(vxo, size=64, factor=8, floor=8, block=None):
    v = np.zeros((size, size, size), bool)
    v[:, :floor, :] = True
    if block is not None:
        (x0, x1), (y0, y1), (z0, z1) = block
        v[x0:x1, y0:y1, z0:z1] = True
    return vxo.World.from_voxels(v, factor)


def _shade(l_dot, n_y, spec_base, light_color, ambient):
    """calculateColor, Renderer.cu:104-118 + Tonemap :170-177 for one pixel, binary32 throughout (the ambient blend
    factor is a double expression narrowed once)."""
    t = f32(float(n_y) * 0.5 + 0.5)
    c = np.array(light_color, np.float32) * f32(l_dot) + np.array(ambient, np.float32) * (f32(0.25) + t * (f32(1.0) - f32(0.25)))
    if spec_base is not None:
        spec = f32(float(spec_base) ** 32)            # five exact squarings in double, rounded once
        c = c + spec * np.array(light_color, np.float32)
    return (c / (c + f32(1.0))).astype(np.float32)


def test_flat_floor_lit_from_above_orthographic(vxo):
    """ORTHO (Renderer.cu:61-70): every ray is `fwd`.  Looking straight down on a floor with the light straight up:
    N = (0,1,0), lDot = 1, view = (0,-1,0), reflect(L,N) = (0,-1,0), spec = 1 -> colour = 2*Lc + Amb = 4.5 per
    channel -> Reinhard 4.5/5.5 -> 208."""
    w = _floor_world(vxo)
    W, H = 24, 16
    p = vxo.make_params(W, H, (32.0, 60.0, 32.0), (0.0, -1.0, 0.0), (0.0, 0.0, 1.0), (1.0, 0.0, 0.0), frame_number=1, ortho=1,
                        ortho_size=(8.0, 8.0), light_dir=(0.0, 1.0, 0.0), shadow=1)
    out = w.render(p, fb=np.zeros((H, W, 4), np.uint8), want_hit=True)
    val = _to_bgra(_shade(1.0, 1.0, 1.0, (2, 2, 2), (0.5, 0.5, 0.5))[None, None, :])[0, 0]
    assert tuple(val) == (208, 208, 208, 255)
    want = np.broadcast_to(val, (H, W, 4)).copy()
    want[H >> 1, W >> 1] = (255, 255, 255, 255)
    assert np.array_equal(out["fb"], want)
    # every pixel hits the top layer of the floor: voxel y = 7 (index x + X*(y + Y*z))
    ys = (out["hit"] // 64) % 64
    assert (out["hit"] >= 0).all() and (ys == 7).all()
    st = out["stats"]
    assert st.primary_rays == W * H and st.primary_hits == W * H and st.shadow_rays == W * H and st.bounce_rays == 0


def test_shadow_gate_and_binary_occlusion(vxo):
    """A block hovering over the floor, light along (1,1,0)/sqrt2, camera looking straight down (ortho).  Floor pixels
    in the block's shadow: lDot = 0 -> no diffuse, no specular: colour = Amb -> 0.5/1.5 -> 85.  Lit floor pixels:
    lDot = 0.70710677, specular = (-view . reflect)^32 with reflect(L,N) = L - 2N(N.L).  The occlusion sample is
    taken only where lDot == 0 (Renderer.cu:121) and is binary (a hit adds 0, a miss 1, :146-157): with one sample a
    shadowed pixel is either unchanged or black; lit pixels never change."""
    w = _floor_world(vxo, block=((8, 24), (24, 32), (8, 56)))
    W, H = 48, 32
    inv = f32(1.0) / np.sqrt(f32(2.0), dtype=np.float32)
    L = (float(inv), float(inv), 0.0)
    base = dict(frame_number=1, ortho=1, ortho_size=(18.0, 18.0), light_dir=L, shadow=1)
    cam = ((32.0, 60.0, 32.0), (0.0, -1.0, 0.0), (0.0, 0.0, 1.0), (1.0, 0.0, 0.0))
    out = w.render(vxo.make_params(W, H, *cam, **base), fb=np.zeros((H, W, 4), np.uint8), want_hit=True)
    fb, hit = out["fb"], out["hit"]
    hy = (hit // 64) % 64
    floor = hy == 7
    top = hy == 31
    assert (floor | top).all() and floor.any() and top.any()
    n_dot_l = f32(inv)                                                   # N = (0,1,0)
    refl_y = f32(inv) - f32(2.0) * n_dot_l                              # reflect(L,N).y = L.y - 2*(N.L); view.y = -1
    lit = tuple(_to_bgra(_shade(n_dot_l, 1.0, max(float(-refl_y), 0.0), (2, 2, 2), (0.5, 0.5, 0.5))[None, None, :])[0, 0])
    dark = tuple(_to_bgra(_shade(0.0, 1.0, None, (2, 2, 2), (0.5, 0.5, 0.5))[None, None, :])[0, 0])
    assert dark == (85, 85, 85, 255)
    cross = np.zeros((H, W), bool)
    cross[H >> 1, W >> 1] = True
    px = [tuple(v) for v in fb[floor & ~cross]]
    assert set(px) == {lit, dark}                                        # the floor is either lit or in the block's shadow
    assert all(tuple(v) == lit for v in fb[top & ~cross])                # nothing shadows the block's top
    shadowed = floor & (fb == np.array(dark, np.uint8)).all(axis=2)
    assert shadowed.sum() > 20
    # the shadow lies on the -x side of... the light comes from +x+y, so the shadow falls toward -x of the block
    xs = np.where(shadowed.any(axis=0))[0]
    assert xs.max() < np.where(top.any(axis=0))[0].max()
    # one binary occlusion sample, reference gate
    out1 = w.render(vxo.make_params(W, H, *cam, bounce_samples=1, **base), fb=np.zeros((H, W, 4), np.uint8))
    fb1 = out1["fb"]
    assert np.array_equal(fb1[~shadowed], fb[~shadowed])                 # gate closed where lDot > 0
    vals = {tuple(v) for v in fb1[shadowed]}
    assert vals <= {dark, (0, 0, 0, 255)} and out1["stats"].bounce_rays == int(shadowed.sum())
    # all-hits gate (this build's switch): every hit pixel takes a sample
    out2 = w.render(vxo.make_params(W, H, *cam, bounce_samples=1, bounce_all_hits=1, **base), fb=np.zeros((H, W, 4), np.uint8))
    assert out2["stats"].bounce_rays == W * H
