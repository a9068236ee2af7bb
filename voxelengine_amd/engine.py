"""Host-side mirror of the reference's VoxelRT interface over the C ABI (include/vxrt.h).

Reference surface (VoxelRT/VolumeRaytracer.cuh:291-377, VoxelRT/Renderer.cuh:39-55):
``VoxelRaytracer3D`` {UploadVoxelBuffer, UploadVoxelBufferDatas, UploadVoxelBufferDataBounds, SetFactor,
Raytrace} and ``Graphics`` {GetDirections, SetEnvironment, SetFOV, SetOrthoWindowSize, RenderScreen}.
Here one :class:`Context` per GPU carries what the reference keeps in process globals, so eight GPUs can
run from eight ranks.  PyTorch only supplies device memory and streams; every pixel and ray is produced by
the HIP kernels in ``csrc/``.  There is no CPU fallback: without libvxrt.so and a GPU these calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

from . import _native as N


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def GetDirections(euler):
    """Graphics::GetDirections (VoxelRT/Renderer.cu:27-42). Host math only."""
    f, u, r = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    N.load().vxrt_get_directions(_f3(euler), f, u, r)
    return (np.array(f[:], np.float32), np.array(u[:], np.float32), np.array(r[:], np.float32))


def tile_schedule(width: int, frame_rows, fwd, up, right, fov_deg: float = 90.0, height: int | None = None):
    """Hand-out order of the 8x8 pixel tiles for the persistent render kernel: expected-longest ray chains first,
    so the wave-level tail at the end of a frame is made of cheap tiles.  A function of the camera only: the cost
    proxy is the elevation of the tile's centre ray (rays pointing up leave the grid at once; rays just below the
    horizon travel farthest).  ``frame_rows``: frame row of every launch-grid row (``range(height)`` for an
    unsharded frame).  Returns uint32 numpy array; scheduling only -- results never depend on it."""
    frame_rows = np.asarray(list(frame_rows), np.int64)
    H = int(height if height is not None else (frame_rows.max() + 1 if frame_rows.size else 1))
    ntx, nty = (width + 7) // 8, (len(frame_rows) + 7) // 8
    t = np.tan(np.float64(fov_deg) * 3.1415 / 180.0 / 2.0)
    cx = (np.arange(ntx) * 8 + 4) / width * 2 - 1
    rows_c = frame_rows[np.minimum(np.arange(nty) * 8 + 4, len(frame_rows) - 1)] if len(frame_rows) else np.zeros(0)
    cy = rows_c / H * 2 - 1
    f, u, r = [np.asarray(v, np.float64) for v in (fwd, up, right)]
    d = f[None, None, :] + (cx[None, :, None] * t * (width / H)) * r[None, None, :] + (cy[:, None, None] * t) * u[None, None, :]
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    dy = d[..., 1]
    cost = np.where(dy >= 0, 0.0, 1.0 / np.maximum(-dy, 0.02))  # up-pointing: cheap; grazing: expensive (capped)
    return np.argsort(-cost.reshape(-1), kind="stable").astype(np.uint32)


def world_file_info(path: str) -> "N.WorldInfo":
    """Header of a brickmap file written by :meth:`Context.save_world` (needs no GPU)."""
    info = N.WorldInfo()
    N.check(N.load().vxrt_world_file_info(os.fsencode(path), C.byref(info)))
    return info


def compact_rows(height: int, strip_rows: int, strip_count: int, strip_index: int) -> int:
    return int(N.load().vxrt_compact_rows(height, strip_rows, strip_count, strip_index))


@dataclass
class RenderOptions:
    """Run-time forms of the reference's compile-time switches (VoxelRT/Renderer.cu:4-5,102,123)."""
    mode: int = N.MODE_SHADED
    checkerboard: bool = False
    shadow: bool = False
    bounce_samples: int = 0
    bounce_all_hits: bool = False
    bounce_depth: int = 1           # 2: extension beyond the reference (second bounce, include/vxrt.h)
    ortho: bool = False
    frame_number: int = -1          # < 0: context counter with the reference's post-copy increment
    strip_rows: int = 16
    strip_count: int = 1
    strip_index: int = 0
    compact: bool = False
    collect_stats: bool = False
    tile_schedule: bool = True      # persistent kernel: expected-longest tiles first (scheduling only)
    extra: dict = field(default_factory=dict)


class Context:
    """One MI355X: resident brickmap + camera/lighting state + launches."""

    def __init__(self, device: int = 0):
        self._L = N.load()
        h = C.c_void_p()
        N.check(self._L.vxrt_create(device, C.byref(h)))
        self._h = h
        self.device = device
        self.kernel_variant = 4  # the library's default (= 7: persistent wavefronts on the wave-level tracer)

    def close(self):
        if getattr(self, "_h", None):
            self._L.vxrt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- world ----------------------------------------------------------------------------------
    def upload_world(self, factor: int, cdims, coarse_bits, brick_slot, bounds, pool) -> None:
        """UploadVoxelBuffer + UploadVoxelBufferDatas + UploadVoxelBufferDataBounds + SetFactor
        (VoxelRT/VolumeRaytracer.cu:527-572) with the tables as flat host arrays."""
        cb = np.ascontiguousarray(coarse_bits, np.uint32)
        bs = np.ascontiguousarray(brick_slot, np.uint32)
        bd = np.ascontiguousarray(bounds, np.float32)
        pl = np.ascontiguousarray(pool, np.uint32)
        d = N.WorldDesc()
        d.struct_size = C.sizeof(N.WorldDesc)
        d.factor = factor
        d.cdims = (C.c_int32 * 3)(*[int(c) for c in cdims])
        d.nslots = pl.size // (factor ** 3 // 32)
        d.coarse_bits, d.brick_slot, d.bounds = cb.ctypes.data, bs.ctypes.data, bd.ctypes.data
        d.pool = pl.ctypes.data if pl.size else None
        N.check(self._L.vxrt_upload_world(self._h, C.byref(d)))

    def build_world(self, generator: int, X: int, Y: int, Z: int, factor: int) -> "N.WorldInfo":
        """CreateVoxels + GenerateLowresVoxelBuffer on the device, brick by brick."""
        N.check(self._L.vxrt_build_world_procedural(self._h, generator, X, Y, Z, factor))
        return self.world_info()

    def world_info(self) -> "N.WorldInfo":
        info = N.WorldInfo()
        N.check(self._L.vxrt_world_info_get(self._h, C.byref(info)))
        return info

    def save_world(self, path: str) -> None:
        """Write the resident brickmap to a file (format: include/vxrt.h, "brickmap file")."""
        N.check(self._L.vxrt_save_world(self._h, os.fsencode(path)))

    def load_world(self, path: str) -> "N.WorldInfo":
        """Replace the resident brickmap by the one in ``path`` (validated while it streams into HBM)."""
        N.check(self._L.vxrt_load_world(self._h, os.fsencode(path)))
        return self.world_info()

    # ---- chunk streaming (extension, include/vxrt.h) -------------------------------------------------
    def stream_open(self, path: str, pool_capacity_bricks: int) -> "N.WorldInfo":
        """A world whose bricks are read from the brickmap file ``path`` only for the chunks (8x8x8 tiles of coarse
        cells) near a focus point; the pool is a cache of ``pool_capacity_bricks`` bricks.  Empty until ``stream_focus``."""
        N.check(self._L.vxrt_stream_open(self._h, os.fsencode(path), int(pool_capacity_bricks)))
        return self.world_info()

    def stream_focus(self, focus, radius: float) -> "N.StreamStats":
        st = N.StreamStats()
        N.check(self._L.vxrt_stream_focus(self._h, _f3(focus), float(radius), C.byref(st)))
        return st

    def stream_resident(self) -> np.ndarray:
        """One flag per chunk (tile index of the coarse grid): 1 = its bricks are resident."""
        info = self.world_info()
        flags = np.zeros(int(info.ncells) // 512, np.uint8)
        N.check(self._L.vxrt_stream_resident(self._h, flags.ctypes.data, flags.size))
        return flags

    def stream_close(self) -> None:
        N.check(self._L.vxrt_stream_close(self._h))

    def download_world(self, with_pool: bool = True):
        info = self.world_info()
        n = int(info.ncells)
        cb = np.empty((n + 31) // 32, np.uint32)
        bs = np.empty(n, np.uint32)
        bd = np.empty((n, 6), np.float32)
        bw = info.factor ** 3 // 32
        pl = np.empty(int(info.nslots) * bw, np.uint32) if with_pool else None
        N.check(self._L.vxrt_download_world(self._h, cb.ctypes.data, bs.ctypes.data, bd.ctypes.data,
                                            pl.ctypes.data if (with_pool and pl.size) else None))
        return dict(factor=int(info.factor), cdims=tuple(info.cdims), coarse_bits=cb, brick_slot=bs, bounds=bd,
                    pool=pl)

    # ---- state ----------------------------------------------------------------------------------
    def SetEnvironment(self, light_dir, light_color, ambient) -> None:
        N.check(self._L.vxrt_set_environment(self._h, _f3(light_dir), _f3(light_color), _f3(ambient)))

    def SetFOV(self, fov: float) -> None:
        N.check(self._L.vxrt_set_fov(self._h, float(fov)))

    def SetOrthoWindowSize(self, sx: float, sy: float) -> None:
        N.check(self._L.vxrt_set_ortho_window_size(self._h, float(sx), float(sy)))

    # ---- per frame -------------------------------------------------------------------------------
    def RenderScreen(self, width: int, height: int, d_fb, origin, fwd, up, right, opts: RenderOptions | None = None,
                     color_aov=None, hit_aov=None, stream: int | None = None, tile_order=None, accum=None,
                     accum_reset: bool = False) -> None:
        """Graphics::RenderScreen (VoxelRT/Renderer.cu:305-328).  ``d_fb``/AOVs: torch CUDA tensors or raw
        device addresses.  Asynchronous on ``stream`` (default: torch's current stream)."""
        fl = self._flags(opts, stream)
        fl.d_color_aov = _ptr(color_aov)
        fl.d_hit_aov = _ptr(hit_aov)
        fl.d_tile_order = _ptr(tile_order)
        fl.d_accum = _ptr(accum)  # temporal accumulation history, (H, W, 4) float32 (extension, include/vxrt.h)
        fl.accum_reset = int(bool(accum_reset))
        N.check(self._L.vxrt_render(self._h, width, height, _ptr(d_fb), _f3(origin), _f3(fwd), _f3(up), _f3(right),
                                    C.byref(fl)))

    def _flags(self, opts: RenderOptions | None, stream: int | None) -> "N.RenderFlags":
        o = opts or RenderOptions()
        fl = N.RenderFlags()
        self._L.vxrt_render_flags_default(C.byref(fl))
        fl.mode, fl.checkerboard, fl.shadow = int(o.mode), int(o.checkerboard), int(o.shadow)
        fl.bounce_samples, fl.bounce_all_hits, fl.ortho = int(o.bounce_samples), int(o.bounce_all_hits), int(o.ortho)
        fl.frame_number = int(o.frame_number)
        fl.strip_rows, fl.strip_count, fl.strip_index = int(o.strip_rows), int(o.strip_count), int(o.strip_index)
        fl.compact, fl.collect_stats = int(o.compact), int(o.collect_stats)
        fl.tile_schedule = int(o.tile_schedule)
        fl.bounce_depth = int(o.bounce_depth)
        fl.stream = _stream(stream)
        return fl

    def RenderViews(self, width: int, height: int, views, opts: RenderOptions | None = None,
                    stream: int | None = None) -> None:
        """Several views of the resident world in ONE launch (vxrt_render_views): the next view's first tiles fill
        the lanes the previous view's last rays leave.  ``views``: sequence of dicts with keys ``fb, origin, fwd, up,
        right`` and optionally ``frame_number`` (default: ``opts.frame_number``, or the context counter), ``color_aov``,
        ``hit_aov``.  Every view equals what :meth:`RenderScreen` produces for it."""
        fl = self._flags(opts, stream)
        if opts is not None and opts.extra.get("accum") is not None:
            fl.d_accum = _ptr(opts.extra["accum"])  # rejected by the library: a multi-view launch has no per-view history
        arr = (N.View * len(views))()
        for dst, v in zip(arr, views):
            dst.d_fb = _ptr(v["fb"])
            dst.origin, dst.fwd, dst.up, dst.right = _f3(v["origin"]), _f3(v["fwd"]), _f3(v["up"]), _f3(v["right"])
            dst.frame_number = int(v.get("frame_number", fl.frame_number))
            dst.d_color_aov = _ptr(v.get("color_aov"))
            dst.d_hit_aov = _ptr(v.get("hit_aov"))
        N.check(self._L.vxrt_render_views(self._h, width, height, len(views), arr, C.byref(fl)))

    def frame_stats(self) -> "N.FrameStats":
        st = N.FrameStats()
        N.check(self._L.vxrt_frame_stats_get(self._h, C.byref(st)))
        return st

    def deinterleave_strips(self, width, height, strip_rows, strip_count, d_shards, shard_stride_bytes, d_fb,
                            stream: int | None = None) -> None:
        N.check(self._L.vxrt_deinterleave_strips(self._h, width, height, strip_rows, strip_count, _ptr(d_shards),
                                                 int(shard_stride_bytes), _ptr(d_fb), _stream(stream)))

    def deinterleave_views(self, width, height, strip_rows, strip_count, d_shards, shard_stride_bytes, view_stride_bytes,
                           n_views, d_fb, fb_stride_bytes, stream: int | None = None) -> None:
        """The strips of all `n_views` views of one multi-view step in one launch (vxrt_deinterleave_views)."""
        N.check(self._L.vxrt_deinterleave_views(self._h, width, height, strip_rows, strip_count, _ptr(d_shards),
                                                int(shard_stride_bytes), int(view_stride_bytes), int(n_views), _ptr(d_fb),
                                                int(fb_stride_bytes), _stream(stream)))

    # ---- batch -----------------------------------------------------------------------------------
    def set_batch_max_steps(self, max_steps: int) -> None:
        """``maxSteps`` of the device function ``Raytrace`` for the following batch calls (default 2048; the
        reference's secondary rays use 8, VoxelRT/Renderer.cu:141)."""
        N.check(self._L.vxrt_set_batch_max_steps(self._h, int(max_steps)))

    def Raytrace(self, origins, dirs, want_stats: bool = False):
        """VoxelRaytracer3D::Raytrace (VoxelRT/VolumeRaytracer.cu:574-618) on host arrays: copy in, trace,
        copy out.  Returns the reference's fields (hitPoint=+inf on a miss, normal, steps, valid, distance)
        plus this build's hit voxel index."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = o.shape[0]
        pos = np.empty((n, 3), np.float32)
        nrm = np.empty((n, 3), np.float32)
        steps = np.empty(n, np.int32)
        hit = np.empty(n, np.uint8)
        vox = np.empty(n, np.int64)
        st = N.FrameStats()
        N.check(self._L.vxrt_trace_batch_host(self._h, o.ctypes.data, d.ctypes.data, n, pos.ctypes.data,
                                              nrm.ctypes.data, steps.ctypes.data, hit.ctypes.data, vox.ctypes.data,
                                              C.byref(st) if want_stats else None))
        valid = np.isfinite(pos).all(axis=1)
        with np.errstate(invalid="ignore"):
            dist = np.sqrt(((o - pos) ** 2).sum(axis=1, dtype=np.float32))
        return dict(hitPoint=pos, normal=nrm, steps=steps, valid=valid, distance=dist, hit=hit, voxel=vox,
                    stats=st if want_stats else None)

    def trace_batch_device(self, d_origins, d_dirs, n, d_pos, d_normal, d_steps, d_hit=None, d_voxel=None,
                           want_stats=False, stream: int | None = None):
        st = N.FrameStats()
        N.check(self._L.vxrt_trace_batch(self._h, _ptr(d_origins), _ptr(d_dirs), int(n), _ptr(d_pos), _ptr(d_normal),
                                         _ptr(d_steps), _ptr(d_hit), _ptr(d_voxel),
                                         C.byref(st) if want_stats else None, _stream(stream)))
        return st if want_stats else None

    def set_kernel_variant(self, variant: int) -> None:
        """7 = the product kernels (persistent wavefronts on the wave-level tracer of csrc/vxrt_wave2.hpp), 4 = the default
        (= 7), 1 = straightforward per-lane loops (cross-check).  kernel_for_launch tells which kernel a launch runs."""
        N.check(self._L.vxrt_set_kernel_variant(self._h, int(variant)))
        self.kernel_variant = int(variant)

    def set_persistent_waves_per_cu(self, waves_per_cu: int) -> None:
        """Grid of the persistent kernels in wavefronts per CU at 4 waves per SIMD (0 = default 16); tests use a small
        grid so that modest batches take the queue kernel."""
        N.check(self._L.vxrt_set_persistent_waves_per_cu(self._h, int(waves_per_cu)))

    def guard_pretend_no_slack(self, on: bool) -> None:
        """Test hook (vxrt_debug_guard_pretend_no_slack): the load guard of collect_stats launches treats the bit tables as
        allocated without slack."""
        N.check(self._L.vxrt_debug_guard_pretend_no_slack(self._h, int(bool(on))))

    def has_experiments(self) -> bool:
        """True for the A/B build of the library (development knobs read from the environment)."""
        return bool(self._L.vxrt_has_experiments())

    KERNEL_NAMES = {1: "k_render", 7: "k_render_persist2"}

    def kernel_for_launch(self, width: int, height: int, opts: "RenderOptions | None" = None, nviews: int = 0) -> int:
        """The kernel (7 or 1) a RenderScreen (nviews = 0) or RenderViews launch of this shape runs under the
        current variant (vxrt_kernel_for_launch); KERNEL_NAMES maps it to the kernel's name in a profile."""
        fl = self._flags(opts, None)
        k = int(self._L.vxrt_kernel_for_launch(self._h, int(width), int(height), C.byref(fl), int(nviews)))
        if k < 0:
            raise N.VxrtError("vxrt_kernel_for_launch: bad arguments")
        return k

    def synchronize(self) -> None:
        N.check(self._L.vxrt_synchronize(self._h))


def grid_is_wide(cdims) -> bool:
    """The rule of csrc/vxrt_device.hpp (grid_is_wide): a coarse grid beyond the tracer's packed step counters -- more than
    1020 / 508 / 1020 cells, or one a single walk could cross in MAX_STEPS iterations -- runs the WIDE instantiation."""
    cx, cy, cz = (int(c) for c in cdims)
    return cx > 1020 or cz > 1020 or cy > 508 or cx + cy + cz + 4 >= 2048


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        if not x.is_cuda:
            raise ValueError("device tensor expected")
        return x.data_ptr()
    raise TypeError(f"cannot take a device pointer from {type(x)}")


def _stream(s):
    if s is not None:
        return s
    import torch
    return torch.cuda.current_stream().cuda_stream
