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
    """hipcc --offload-arch=gfx950 cross-compiles without a GPU.

    Safe when several ranks import the package at once on a stale tree (a fresh snapshot, or after a source edit): the
    build runs under an exclusive lock on csrc/.build.lock, into a temporary name that is renamed over libvxrt.so when it
    is complete, and the stamp is written last, still under the lock -- a sibling either finds the finished library or
    waits for the lock and then finds it.  VXRT_SKIP_STALE_CHECK=1: never build (for processes started under a profiler,
    whose preload has initialised the GPU before Python runs: no make/hipcc children there; build beforehand)."""
    import fcntl

    if os.environ.get("VXRT_SKIP_STALE_CHECK") and os.path.exists(LIB_PATH) and not force:
        return LIB_PATH
    if not force and not _stale(LIB_PATH, lib_sources()):
        return LIB_PATH
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or _stale(LIB_PATH, lib_sources()):  # (again: a sibling may have built it while this one waited)
                tmp = "libvxrt.so.tmp.%d" % os.getpid()
                try:
                    subprocess.check_call(["make", "-C", CSRC, "-B", tmp, "OUT=" + tmp],
                                          stdout=None if verbose else subprocess.DEVNULL)
                    os.replace(os.path.join(CSRC, tmp), LIB_PATH)
                finally:
                    if os.path.exists(os.path.join(CSRC, tmp)):
                        os.unlink(os.path.join(CSRC, tmp))
                stamp = LIB_PATH + ".srchash"
                with open(stamp + ".tmp", "w") as f:
                    f.write(_digest(lib_sources()) + "\n")
                os.replace(stamp + ".tmp", stamp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def build_facade(force: bool = False, verbose: bool = False) -> str:
    """C++ facade (namespace GPUDDA) + headless VoxelApp example, linked against libvxrt.so."""
    out = os.path.join(REPO, "examples", "voxelapp_headless")
    mk = os.path.join(REPO, "examples", "Makefile")
    if os.path.exists(mk):
        cmd = ["make", "-C", os.path.join(REPO, "examples")] + (["-B"] if force else [])
        subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return out
