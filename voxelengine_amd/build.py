"""Build helpers: compile libvxrt.so (hipcc, gfx950) and the C++ facade/example in-tree."""
from __future__ import annotations

import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
REPO = os.path.dirname(PKG_DIR)
LIB_PATH = os.path.join(CSRC, "libvxrt.so")


def _digest(sources: list[str]) -> str:
    import hashlib

    h = hashlib.sha256()
    for s in sources:
        h.update(os.path.basename(s).encode() + b"\0")
        with open(s, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target: str, sources: list[str]) -> bool:
    """Content-based: the library carries a stamp file with the digest of the sources it was built from.  (File times
    do not survive the snapshot copy to a GPU box in a dependable order; contents do.)"""
    stamp = target + ".srchash"
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    return open(stamp).read().strip() != _digest(sources)


def lib_sources() -> list[str]:
    """Everything libvxrt.so is built from: every .hip / .hpp of csrc/ (a glob, so that a new kernel header cannot be
    forgotten here), the Makefile and the C ABI header."""
    import glob

    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")))
    return files + [os.path.join(CSRC, "Makefile"), os.path.join(REPO, "include", "vxrt.h")]


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 cross-compiles without a GPU."""
    if force or _stale(LIB_PATH, lib_sources()):
        subprocess.check_call(["make", "-C", CSRC, "-B", "libvxrt.so"], stdout=None if verbose else subprocess.DEVNULL)
        with open(LIB_PATH + ".srchash", "w") as f:
            f.write(_digest(lib_sources()) + "\n")
    return LIB_PATH


def build_facade(force: bool = False, verbose: bool = False) -> str:
    """C++ facade (namespace GPUDDA) + headless VoxelApp example, linked against libvxrt.so."""
    out = os.path.join(REPO, "examples", "voxelapp_headless")
    mk = os.path.join(REPO, "examples", "Makefile")
    if os.path.exists(mk):
        cmd = ["make", "-C", os.path.join(REPO, "examples")] + (["-B"] if force else [])
        subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return out
