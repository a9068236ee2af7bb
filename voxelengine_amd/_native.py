"""ctypes view of the C ABI in include/vxrt.h.  The library is required: nothing here falls back to a
CPU path -- a missing or unloadable libvxrt.so raises."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

EMPTY_SLOT = 0xFFFFFFFF
MAX_STEPS = 2048
GEN_HASH_HEIGHTFIELD, GEN_PERLIN_REF, GEN_INT_TERRAIN = 0, 1, 2
MODE_SHADED, MODE_DEBUG = 0, 1

# every symbol include/vxrt.h declares
EXPORTS = [
    "vxrt_abi_version", "vxrt_create", "vxrt_destroy", "vxrt_last_error", "vxrt_synchronize", "vxrt_set_kernel_variant", "vxrt_kernel_for_launch",
    "vxrt_has_experiments", "vxrt_set_persistent_waves_per_cu", "vxrt_debug_guard_pretend_no_slack",
    "vxrt_upload_world", "vxrt_build_world_procedural", "vxrt_world_info_get", "vxrt_download_world",
    "vxrt_save_world", "vxrt_load_world", "vxrt_world_file_info",
    "vxrt_set_environment", "vxrt_set_fov", "vxrt_set_ortho_window_size", "vxrt_get_directions",
    "vxrt_render_flags_default", "vxrt_render", "vxrt_render_views", "vxrt_compact_rows", "vxrt_frame_stats_get",
    "vxrt_deinterleave_strips", "vxrt_deinterleave_views", "vxrt_trace_batch", "vxrt_trace_batch_host",
    "vxrt_set_batch_max_steps", "vxrt_stream_open", "vxrt_stream_focus", "vxrt_stream_resident", "vxrt_stream_close",
]


class WorldDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("factor", C.c_int32), ("cdims", C.c_int32 * 3), ("nslots", C.c_uint64),
        ("coarse_bits", C.c_void_p), ("brick_slot", C.c_void_p), ("bounds", C.c_void_p), ("pool", C.c_void_p),
    ]


class WorldInfo(C.Structure):
    _fields_ = [("factor", C.c_int32), ("cdims", C.c_int32 * 3), ("ncells", C.c_uint64), ("nslots", C.c_uint64),
                ("hbm_bytes", C.c_uint64)]


class FrameStats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("bounce_rays", C.c_uint64),
                ("primary_hits", C.c_uint64), ("coarse_probes", C.c_uint64), ("brick_entries", C.c_uint64),
                ("fine_probes", C.c_uint64), ("dbg", C.c_uint64 * 12), ("guard_slack_loads", C.c_uint64),
                ("guard_stray_loads", C.c_uint64)]

    def total_rays(self) -> int:
        return int(self.primary_rays + self.shadow_rays + self.bounce_rays)

    def algorithmic_bytes(self, batch: bool = False) -> int:
        """B = Nc*(4+24) + Nb*24 + Nf*4 + P, on the reference's data layout at its load
        granularity (4-B bit word + 24-B Bounds3Df per coarse probe, 24-B VoxelBuffer3D descriptor per brick
        entry, 4-B word per brick probe); P = 4-B pixel store per primary ray (batch: 24 B in + 28 B out)."""
        per_primary = 52 if batch else 4
        return int(self.coarse_probes * 28 + self.brick_entries * 24 + self.fine_probes * 4
                   + self.primary_rays * per_primary)


class StreamStats(C.Structure):
    _fields_ = [("chunks_total", C.c_uint64), ("chunks_occupied", C.c_uint64), ("chunks_resident", C.c_uint64),
                ("bricks_resident", C.c_uint64), ("chunks_loaded", C.c_uint64), ("chunks_evicted", C.c_uint64),
                ("chunks_missing", C.c_uint64), ("bytes_read", C.c_uint64)]


class RenderFlags(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("mode", C.c_int32), ("checkerboard", C.c_int32), ("shadow", C.c_int32),
        ("bounce_samples", C.c_int32), ("bounce_all_hits", C.c_int32), ("bounce_depth", C.c_int32), ("ortho", C.c_int32),
        ("frame_number", C.c_int64),
        ("strip_rows", C.c_int32), ("strip_count", C.c_int32), ("strip_index", C.c_int32), ("compact", C.c_int32),
        ("collect_stats", C.c_int32), ("tile_schedule", C.c_int32),
        ("d_color_aov", C.c_void_p), ("d_hit_aov", C.c_void_p), ("d_tile_order", C.c_void_p), ("stream", C.c_void_p),
        ("d_accum", C.c_void_p), ("accum_reset", C.c_int32), ("reserved_", C.c_int32),
    ]


class View(C.Structure):
    _fields_ = [
        ("d_fb", C.c_void_p), ("origin", C.c_float * 3), ("fwd", C.c_float * 3), ("up", C.c_float * 3),
        ("right", C.c_float * 3), ("frame_number", C.c_int64), ("d_color_aov", C.c_void_p), ("d_hit_aov", C.c_void_p),
    ]


_LIB = None


def lib_path() -> str:
    return _build.LIB_PATH


def load() -> C.CDLL:
    """Load libvxrt.so (building it first if sources are newer).  Raises if that is impossible."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # PyTorch ships its own libamdhip64; it must be the HIP runtime of the process before libvxrt.so binds to
    # one, or the two runtimes fight over the device (seen as "no ROCm-capable device is detected").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = os.environ.get("VXRT_LIB", _build.LIB_PATH)  # VXRT_LIB: A/B builds of the same library (tools/)
    if path == _build.LIB_PATH:
        # Content-hash staleness check against every source of the library; make/hipcc run only when it is stale, under a
        # lock, into a temporary file renamed into place (build.build_lib).  Normally this process has not touched the GPU
        # yet, so the make child is harmless; under a profiler (whose preload initialises the GPU before Python starts)
        # build beforehand and set VXRT_SKIP_STALE_CHECK=1 so that no child is started here.
        _build.build_lib(force=bool(os.environ.get("VXRT_REBUILD")))
    if not os.path.exists(path):
        raise RuntimeError(f"libvxrt.so not found at {path}; run __graft_entry__.build()")
    L = C.CDLL(path)
    f3 = C.POINTER(C.c_float)
    L.vxrt_abi_version.restype = C.c_int
    L.vxrt_last_error.restype = C.c_char_p
    L.vxrt_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.vxrt_destroy.argtypes = [C.c_void_p]
    L.vxrt_synchronize.argtypes = [C.c_void_p]
    L.vxrt_set_kernel_variant.argtypes = [C.c_void_p, C.c_int]
    L.vxrt_has_experiments.restype = C.c_int
    L.vxrt_debug_guard_pretend_no_slack.argtypes = [C.c_void_p, C.c_int]
    L.vxrt_set_persistent_waves_per_cu.argtypes = [C.c_void_p, C.c_int]
    L.vxrt_kernel_for_launch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
    L.vxrt_kernel_for_launch.restype = C.c_int
    L.vxrt_upload_world.argtypes = [C.c_void_p, C.POINTER(WorldDesc)]
    L.vxrt_build_world_procedural.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.vxrt_world_info_get.argtypes = [C.c_void_p, C.POINTER(WorldInfo)]
    L.vxrt_download_world.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.vxrt_save_world.argtypes = [C.c_void_p, C.c_char_p]
    L.vxrt_load_world.argtypes = [C.c_void_p, C.c_char_p]
    L.vxrt_world_file_info.argtypes = [C.c_char_p, C.POINTER(WorldInfo)]
    L.vxrt_set_environment.argtypes = [C.c_void_p, f3, f3, f3]
    L.vxrt_set_fov.argtypes = [C.c_void_p, C.c_float]
    L.vxrt_set_ortho_window_size.argtypes = [C.c_void_p, C.c_float, C.c_float]
    L.vxrt_get_directions.argtypes = [f3, f3, f3, f3]
    L.vxrt_get_directions.restype = None
    L.vxrt_render_flags_default.argtypes = [C.POINTER(RenderFlags)]
    L.vxrt_render_flags_default.restype = None
    L.vxrt_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, f3, f3, f3, f3, C.POINTER(RenderFlags)]
    L.vxrt_render_views.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(View), C.POINTER(RenderFlags)]
    L.vxrt_compact_rows.argtypes = [C.c_uint32, C.c_int32, C.c_int32, C.c_int32]
    L.vxrt_compact_rows.restype = C.c_uint32
    L.vxrt_frame_stats_get.argtypes = [C.c_void_p, C.POINTER(FrameStats)]
    L.vxrt_deinterleave_strips.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_void_p,
                                           C.c_uint64, C.c_void_p, C.c_void_p]
    L.vxrt_deinterleave_views.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64,
                                          C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    L.vxrt_trace_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.POINTER(FrameStats), C.c_void_p]
    L.vxrt_set_batch_max_steps.argtypes = [C.c_void_p, C.c_int32]
    L.vxrt_stream_open.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
    L.vxrt_stream_focus.argtypes = [C.c_void_p, f3, C.c_float, C.POINTER(StreamStats)]
    L.vxrt_stream_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.vxrt_stream_close.argtypes = [C.c_void_p]
    L.vxrt_trace_batch_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(FrameStats)]
    for name in EXPORTS:  # every symbol the header declares must resolve
        getattr(L, name)
    _LIB = L
    return L


class VxrtError(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc != 0:
        raise VxrtError(f"vxrt error {rc}: {load().vxrt_last_error().decode()}")
