"""Shared scene/ray builders for the parity tests (seeded, small enough for the oracle to finish in seconds)."""
import numpy as np

CAMERAS = {  # position as a fraction of the world size, euler angles
    "A": ((0.5, 0.9, 0.5), (-0.45, 0.7, 0.0)),
    "B": ((0.1, 1.2, 0.1), (-0.6, 3.9, 0.0)),      # outside the grid
    "C": ((0.5, 1.5, 0.5), (-1.5707, 0.0, 0.0)),   # top-down
    "D": ((0.02, 0.55, 0.5), (-0.05, 1.5707, 0.0)),  # grazing
}


def random_voxel_world(vxo, size=(64, 64, 64), factor=8, density=0.01, seed=0):
    rng = np.random.default_rng(seed)
    v = rng.random(size) < density
    return vxo.World.from_voxels(v, factor)


def mixed_rays(dims, n, seed=0):
    """Origins inside / outside / on cell edges, directions random, axis-aligned and with zero components."""
    rng = np.random.default_rng(seed)
    X, Y, Z = dims
    o = np.empty((n, 3), np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    k = n // 6
    o[:] = rng.random((n, 3)).astype(np.float32) * np.array([X, Y, Z], np.float32)          # inside
    o[k:2 * k] = (rng.random((k, 3)).astype(np.float32) * 3 - 1) * np.array([X, Y, Z], np.float32)  # in/outside
    o[2 * k:3 * k] = np.floor(o[2 * k:3 * k])                                                  # exact cell corners
    o[3 * k:4 * k, 0] = np.float32(X)                                                          # on the +x face
    d[3 * k:4 * k, 0] = -np.abs(d[3 * k:4 * k, 0])
    ax = rng.integers(0, 3, size=k)                                                            # axis-aligned
    d[4 * k:5 * k] = 0
    d[np.arange(4 * k, 5 * k), ax] = rng.choice([-1.0, 1.0], size=k).astype(np.float32)
    d[5 * k:6 * k, rng.integers(0, 3)] = 0                                                     # one zero component
    # aim a share of the outside rays at the grid so they enter it
    c = np.array([X, Y, Z], np.float32) / 2
    aim = slice(k, k + k // 2)
    d[aim] = (c + rng.normal(size=(k // 2, 3)).astype(np.float32) * c * 0.5) - o[aim]
    # adversarial families (every 13th/17th/... ray): tiny and denormal direction components, far-away origins
    # aimed at the grid, axis-aligned rays running exactly along cell boundaries
    idx = np.arange(n)
    m = idx % 13 == 0
    d[m, (idx[m] // 13) % 3] *= np.float32(1e-30)
    m = idx % 17 == 0
    d[m, (idx[m] // 17) % 3] = np.float32(1e-42)
    m = idx % 23 == 0
    o[m] = o[m] * np.float32(1000.0)
    d[m] = c - o[m]
    m = idx % 31 == 0                                 # exact x/y ties entering through a grid corner / far faces
    o[m, 0] = o[m, 1] = np.float32(X) * (1.5 + (idx[m] % 7)).astype(np.float32)
    d[m, 0] = d[m, 1] = -np.abs(d[m, 0]) - np.float32(0.1)
    m = idx % 37 == 0
    o[m, 1] = o[m, 2] = np.float32(-0.5 * Y)
    d[m, 1] = d[m, 2] = np.abs(d[m, 1]) + np.float32(0.1)
    m = idx % 29 == 0
    d[m] = 0
    d[m, 0] = np.where(idx[m] & 1, 1.0, -1.0).astype(np.float32)
    o[m, 1:] = np.floor(o[m, 1:])
    # a direction whose squared length underflows normalises to inf/NaN: not a ray (and float->int of NaN is
    # where host C and GPUs legitimately differ), so keep at least one ordinary component
    bad = (np.abs(d).max(axis=1) < 1e-10)
    d[bad] = (1, 0, 0)
    return o, d


def camera(name, dims, vxo_or_engine):
    (fx, fy, fz), euler = CAMERAS[name]
    pos = (fx * dims[0], fy * dims[1], fz * dims[2])
    f, u, r = vxo_or_engine.get_directions(euler) if hasattr(vxo_or_engine, "get_directions") \
        else vxo_or_engine.GetDirections(euler)
    return pos, f, u, r


def fibonacci_fan(n, origin):
    """n rays from one origin with Fibonacci-sphere directions: the 3-D analogue of the angular fan of the
    reference's 2-D tester (DDATestCpp/DDATestCpp.cpp:443-448, RAYS = 1000000 at :21).  Ray i points along
    (r cos(i*ga), 1 - (2i+1)/n, r sin(i*ga)), ga = the golden angle; evaluated in binary64, stored as binary32."""
    i = np.arange(n, dtype=np.float64)
    y = 1.0 - (2.0 * i + 1.0) / n
    r = np.sqrt(1.0 - y * y)
    phi = i * (np.pi * (3.0 - np.sqrt(5.0)))
    d = np.stack([r * np.cos(phi), y, r * np.sin(phi)], 1).astype(np.float32)
    o = np.tile(np.asarray(origin, np.float32), (n, 1))
    return o, d


def float_bits(a):
    """binary32 array -> its bits as uint32, every NaN as the one canonical quiet NaN"""
    a = np.ascontiguousarray(a, np.float32)
    b = a.view(np.uint32).copy()
    b[np.isnan(a)] = 0x7FC00000
    return b
