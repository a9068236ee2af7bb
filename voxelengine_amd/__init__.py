"""voxelengine_amd -- MI355X-native voxel brickmap ray tracing behind the reference's VoxelRT interface.

Layout: ``csrc/`` hand-written HIP for gfx950 + the C ABI (include/vxrt.h); ``engine`` the host mirror of
``GPUDDA::VoxelRaytracer3D`` / ``GPUDDA::Graphics``; ``sharding`` the screen-strip split and the RCCL gather.
"""
from ._native import (EMPTY_SLOT, GEN_HASH_HEIGHTFIELD, GEN_INT_TERRAIN, GEN_PERLIN_REF, MAX_STEPS, MODE_DEBUG,
                      MODE_SHADED, EXPORTS, FrameStats, VxrtError, lib_path, load)
from .engine import Context, GetDirections, RenderOptions, compact_rows, grid_is_wide, tile_schedule, world_file_info

__all__ = ["Context", "RenderOptions", "GetDirections", "compact_rows", "grid_is_wide", "tile_schedule", "world_file_info", "FrameStats",
           "VxrtError", "load",
           "lib_path", "EXPORTS", "EMPTY_SLOT", "MAX_STEPS", "MODE_SHADED", "MODE_DEBUG",
           "GEN_HASH_HEIGHTFIELD", "GEN_PERLIN_REF", "GEN_INT_TERRAIN"]
