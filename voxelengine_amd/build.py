"""Build helpers: compile libvxrt.so (hipcc, gfx950) and the C++ facade/example in-tree."""
from __future__ import annotations

import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
REPO = os.path.dirname(PKG_DIR)
LIB_PATH = os.path.join(CSRC, "libvxrt.so")


def _stale(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def lib_sources() -> list[str]:
    names = ["vxrt_api.hip", "vxrt_kernels.hip", "vxrt_worldgen.hip", "vxrt_device.hpp", "vxrt_kernels.hpp",
             "vxrt_wave.hpp", "Makefile"]
    return [os.path.join(CSRC, n) for n in names] + [os.path.join(REPO, "include", "vxrt.h")]


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 cross-compiles without a GPU."""
    if force or _stale(LIB_PATH, lib_sources()):
        cmd = ["make", "-C", CSRC] + (["-B"] if force else []) + ["libvxrt.so"]
        subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def build_facade(force: bool = False, verbose: bool = False) -> str:
    """C++ facade (namespace GPUDDA) + headless VoxelApp example, linked against libvxrt.so."""
    out = os.path.join(REPO, "examples", "voxelapp_headless")
    mk = os.path.join(REPO, "examples", "Makefile")
    if os.path.exists(mk):
        cmd = ["make", "-C", os.path.join(REPO, "examples")] + (["-B"] if force else [])
        subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return out
