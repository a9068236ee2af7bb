"""ctypes binding of the CPU ORACLE (oracle/libvxo.so).

TEST INFRASTRUCTURE ONLY -- see oracle/vxo.h.  Importers: tests/, bench.py's
cpu_baseline leg, __graft_entry__.smoke().  Never imported by voxelengine_amd.
Parity status: UNPINNED (no reference fixtures exist; reference not buildable here).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EMPTY_SLOT = 0xFFFFFFFF
MAX_STEPS = 2048
GEN_HASH_HEIGHTFIELD, GEN_PERLIN_REF, GEN_INT_TERRAIN = 0, 1, 2
MODE_SHADED, MODE_DEBUG = 0, 1


class _World(C.Structure):
    _fields_ = [
        ("factor", C.c_int),
        ("cdims", C.c_int * 3),
        ("ncells", C.c_uint64),
        ("coarse_bits", C.POINTER(C.c_uint32)),
        ("brick_slot", C.POINTER(C.c_uint32)),
        ("bounds", C.POINTER(C.c_float)),
        ("nslots", C.c_uint64),
        ("pool", C.POINTER(C.c_uint32)),
        ("owns", C.c_int),
    ]


class RayStats(C.Structure):
    _fields_ = [("coarse_probes", C.c_uint64), ("brick_entries", C.c_uint64), ("fine_probes", C.c_uint64)]


class Env(C.Structure):
    _fields_ = [("light_dir", C.c_float * 3), ("light_color", C.c_float * 3), ("ambient", C.c_float * 3)]


class RenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("frame_number", C.c_uint32),
        ("fov_deg", C.c_float), ("ortho_size", C.c_float * 2), ("ortho", C.c_int), ("mode", C.c_int),
        ("checkerboard", C.c_int), ("shadow", C.c_int), ("bounce_samples", C.c_int),
        ("bounce_all_hits", C.c_int), ("bounce_depth", C.c_int),
        ("origin", C.c_float * 3), ("fwd", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3),
        ("env", Env), ("row_begin", C.c_uint32), ("row_end", C.c_uint32),
    ]


class FrameStats(C.Structure):
    _fields_ = [
        ("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("bounce_rays", C.c_uint64),
        ("primary_hits", C.c_uint64), ("probes", RayStats), ("pixels_written", C.c_uint64),
    ]

    def total_rays(self) -> int:
        return int(self.primary_rays + self.shadow_rays + self.bounce_rays)


class DDAParams(C.Structure):
    _fields_ = [
        ("bits", C.POINTER(C.c_uint32)), ("nbits", C.c_uint64), ("dims", C.c_int * 3),
        ("start", C.c_float * 3), ("dir", C.c_float * 3), ("has_bounds", C.c_int),
        ("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3), ("max_steps", C.c_int),
        ("cell_bounds", C.POINTER(C.c_float)), ("cell_bounds_scale", C.c_int), ("take_initial_step", C.c_int),
    ]


class DDAResult(C.Structure):
    _fields_ = [
        ("hit", C.c_int), ("out_of_bounds", C.c_int), ("hit_cell", C.c_float * 3), ("point", C.c_float * 3),
        ("next_cell", C.c_float * 3), ("normal", C.c_float * 3), ("steps", C.c_int), ("probes", C.c_int),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (IEEE semantics flags live in oracle/Makefile)."""
    so = os.path.join(_HERE, "libvxo.so")
    srcs = [os.path.join(_HERE, f) for f in ("vxo_trace.c", "vxo_world.c", "vxo_render.c", "vxo.h")]
    stale = not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libvxo.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u32p, f32p = C.POINTER(C.c_uint32), C.POINTER(C.c_float)
        L.vxo_sample_index.restype = C.c_uint32
        L.vxo_sample_index.argtypes = [C.c_uint32] * 5
        L.vxo_sample_index64.restype = C.c_uint64
        L.vxo_sample_index64.argtypes = [C.c_uint64] * 5
        L.vxo_position_from_index.argtypes = [C.c_uint32] * 3 + [u32p] * 3
        L.vxo_hash32.restype = C.c_uint32
        L.vxo_hash32.argtypes = [C.c_uint32]
        L.vxo_random_float.restype = C.c_float
        L.vxo_random_float.argtypes = [C.c_uint32]
        L.vxo_fbm_perlin.restype = C.c_float
        L.vxo_fbm_perlin.argtypes = [C.c_float] * 3
        L.vxo_gen_solid.restype = C.c_int
        L.vxo_gen_solid.argtypes = [C.c_int] * 7
        L.vxo_gen_dense.restype = C.c_void_p
        L.vxo_gen_dense.argtypes = [C.c_int] * 5
        L.vxo_build_brickmap.restype = C.POINTER(_World)
        L.vxo_build_brickmap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.vxo_gen_brickmap.restype = C.POINTER(_World)
        L.vxo_gen_brickmap.argtypes = [C.c_int] * 6
        L.vxo_world_wrap.restype = C.POINTER(_World)
        L.vxo_world_wrap.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_uint64, C.c_void_p]
        L.vxo_world_free.argtypes = [C.POINTER(_World)]
        L.vxo_ray_aabb.restype = C.c_int
        L.vxo_ray_aabb.argtypes = [f32p] * 6
        L.vxo_dda.argtypes = [C.POINTER(DDAParams), C.POINTER(DDAResult)]
        L.vxo_raytrace.restype = C.c_int
        L.vxo_raytrace.argtypes = [C.POINTER(_World), C.c_int, f32p, f32p, C.POINTER(C.c_int), f32p, f32p,
                                   C.POINTER(C.c_int), C.POINTER(RayStats)]
        L.vxo_trace_batch.argtypes = [C.POINTER(_World), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RayStats),
                                      C.c_int]
        L.vxo_get_directions.argtypes = [f32p] * 4
        L.vxo_render.argtypes = [C.POINTER(_World), C.POINTER(RenderParams), C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.POINTER(FrameStats), C.c_int]
        L.vxo_render_accum.argtypes = [C.POINTER(_World), C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.POINTER(FrameStats), C.c_int]
        L.free = C.CDLL(None).free
        L.free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def sample_index(x, y, z, w, h) -> int:
    return int(lib().vxo_sample_index(x, y, z, w, h))


def position_from_index(i, w, h):
    x, y, z = C.c_uint32(), C.c_uint32(), C.c_uint32()
    lib().vxo_position_from_index(i, w, h, C.byref(x), C.byref(y), C.byref(z))
    return x.value, y.value, z.value


def hash32(s: int) -> int:
    return int(lib().vxo_hash32(s & 0xFFFFFFFF))


def random_float(s: int) -> float:
    return float(lib().vxo_random_float(s & 0xFFFFFFFF))


def get_directions(euler):
    f, u, r = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    lib().vxo_get_directions(_f3(euler), f, u, r)
    return (np.array(f[:], np.float32), np.array(u[:], np.float32), np.array(r[:], np.float32))


def ray_aabb(start, d, bmin, bmax):
    p, n = (C.c_float * 3)(), (C.c_float * 3)()
    h = lib().vxo_ray_aabb(_f3(start), _f3(d), _f3(bmin), _f3(bmax), p, n)
    return bool(h), np.array(p[:], np.float32), np.array(n[:], np.float32)


class World:
    """Brickmap in the oracle's (reference-shaped) tables; arrays exposed as numpy views."""

    def __init__(self, ptr, keep=None):
        self._p = ptr
        self._keep = keep  # numpy arrays backing a wrapped world
        w = ptr.contents
        self.factor = int(w.factor)
        self.cdims = tuple(int(v) for v in w.cdims)
        self.ncells = int(w.ncells)
        self.nslots = int(w.nslots)
        self.dims = tuple(c * self.factor for c in self.cdims)
        bw = self.factor ** 3 // 32
        self.coarse_bits = np.ctypeslib.as_array(w.coarse_bits, ((self.ncells + 31) // 32,))
        self.brick_slot = np.ctypeslib.as_array(w.brick_slot, (self.ncells,))
        self.bounds = np.ctypeslib.as_array(w.bounds, (self.ncells, 6))
        self.pool = np.ctypeslib.as_array(w.pool, (max(self.nslots, 1) * bw,))[: self.nslots * bw]

    def __del__(self):
        try:
            if self._p is not None:
                lib().vxo_world_free(self._p)
        except Exception:
            pass
        self._p = None

    # -- constructors
    @staticmethod
    def generate(gen: int, X: int, Y: int, Z: int, factor: int, nthreads: int = 8) -> "World":
        p = lib().vxo_gen_brickmap(gen, X, Y, Z, factor, nthreads)
        if not p:
            raise ValueError("invalid world shape for the tiled-linear layout")
        return World(p)

    @staticmethod
    def from_dense(dense_words: np.ndarray, X: int, Y: int, Z: int, factor: int) -> "World":
        dense_words = np.ascontiguousarray(dense_words, np.uint32)
        p = lib().vxo_build_brickmap(dense_words.ctypes.data, X, Y, Z, factor)
        if not p:
            raise ValueError("invalid world shape for the tiled-linear layout")
        return World(p)

    @staticmethod
    def from_voxels(vox: np.ndarray, factor: int) -> "World":
        """vox: bool array indexed [x, y, z]."""
        X, Y, Z = vox.shape
        return World.from_dense(dense_from_voxels(vox), X, Y, Z, factor)

    @staticmethod
    def wrap(factor, cdims, coarse_bits, brick_slot, bounds, pool) -> "World":
        arrs = [np.ascontiguousarray(coarse_bits, np.uint32), np.ascontiguousarray(brick_slot, np.uint32),
                np.ascontiguousarray(bounds, np.float32), np.ascontiguousarray(pool, np.uint32)]
        bw = factor ** 3 // 32
        cd = (C.c_int * 3)(*cdims)
        p = lib().vxo_world_wrap(factor, cd, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data,
                                 arrs[3].size // bw, arrs[3].ctypes.data)
        return World(p, keep=arrs)

    # -- tracing
    def raytrace(self, origin, ray, max_steps=MAX_STEPS):
        steps = C.c_int()
        n, pos = (C.c_float * 3)(), (C.c_float * 3)()
        vox = (C.c_int * 3)()
        st = RayStats()
        h = lib().vxo_raytrace(self._p, max_steps, _f3(origin), _f3(ray), C.byref(steps), n, pos, vox,
                               C.byref(st))
        return dict(hit=bool(h), steps=steps.value, normal=np.array(n[:], np.float32),
                    pos=np.array(pos[:], np.float32), voxel=tuple(vox[:]) if h else None,
                    stats=(st.coarse_probes, st.brick_entries, st.fine_probes))

    def trace_batch(self, origins, dirs, nthreads: int = 8):
        origins = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = origins.shape[0]
        pos = np.empty((n, 3), np.float32)
        nrm = np.empty((n, 3), np.float32)
        steps = np.empty(n, np.int32)
        hit = np.empty(n, np.uint8)
        vox = np.empty(n, np.int64)
        st = RayStats()
        lib().vxo_trace_batch(self._p, origins.ctypes.data, dirs.ctypes.data, n, pos.ctypes.data,
                              nrm.ctypes.data, steps.ctypes.data, hit.ctypes.data, vox.ctypes.data,
                              C.byref(st), nthreads)
        return dict(pos=pos, normal=nrm, steps=steps, hit=hit, voxel=vox, stats=st)

    def render(self, params: RenderParams, fb: np.ndarray | None = None, want_color=False, want_hit=False,
               nthreads: int = 8, accum: np.ndarray | None = None, accum_reset: bool = False):
        W, H = params.width, params.height
        if params.row_end == 0:
            params.row_end = H
        if fb is None:
            fb = np.full((H, W, 4), 255, np.uint8)
        col = np.zeros((H, W, 3), np.float32) if want_color else None
        hit = np.full((H, W), -1, np.int64) if want_hit else None
        st = FrameStats()
        if accum is not None:
            assert accum.dtype == np.float32 and accum.shape == (H, W, 4) and accum.flags.c_contiguous
        lib().vxo_render_accum(self._p, C.byref(params), fb.ctypes.data, col.ctypes.data if want_color else None,
                               hit.ctypes.data if want_hit else None, accum.ctypes.data if accum is not None else None,
                               int(bool(accum_reset)), C.byref(st), nthreads)
        return dict(fb=fb, color=col, hit=hit, stats=st)


def dense_from_voxels(vox: np.ndarray) -> np.ndarray:
    """bool [x,y,z] -> tiled-linear bit words (VolumeRaytracer.cuh:107-131 order), in numpy."""
    X, Y, Z = vox.shape
    assert X % 8 == 0 and Y % 8 == 0 and Z % 8 == 0
    v = vox.astype(bool).reshape(X // 8, 8, Y // 8, 8, Z // 8, 8)
    # order: tile z, tile y, tile x, in-z, in-y, in-x  (x fastest)
    v = v.transpose(4, 2, 0, 5, 3, 1).reshape(-1)
    bits = np.packbits(v.astype(np.uint8), bitorder="little")
    return bits.view(np.uint32).copy()


def make_params(width, height, origin, fwd, up, right, *, frame_number=1, fov=90.0, mode=MODE_SHADED,
                checkerboard=0, shadow=0, bounce_samples=0, bounce_all_hits=0, ortho=0,
                ortho_size=(10.0, 10.0), light_dir=None, light_color=(2, 2, 2), ambient=(0.5, 0.5, 0.5),
                row_begin=0, row_end=0, bounce_depth=1) -> RenderParams:
    p = RenderParams()
    p.width, p.height, p.frame_number = width, height, frame_number
    p.fov_deg = fov
    p.ortho_size = (C.c_float * 2)(*ortho_size)
    p.ortho, p.mode, p.checkerboard = ortho, mode, checkerboard
    p.shadow, p.bounce_samples, p.bounce_all_hits = shadow, bounce_samples, bounce_all_hits
    p.bounce_depth = bounce_depth
    p.origin, p.fwd, p.up, p.right = _f3(origin), _f3(fwd), _f3(up), _f3(right)
    if light_dir is None:  # VoxelApp/main.cu:59-60: normalize((1,1,1)) in float
        inv = np.float32(1.0) / np.sqrt(np.float32(3.0), dtype=np.float32)
        light_dir = (inv, inv, inv)
    p.env.light_dir, p.env.light_color, p.env.ambient = _f3(light_dir), _f3(light_color), _f3(ambient)
    p.row_begin, p.row_end = row_begin, row_end if row_end else height
    return p
