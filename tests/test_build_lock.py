"""voxelengine_amd.build.build_lib under concurrency: N ranks that import the package on a stale tree must
end up with ONE build, a complete library and a stamp that matches it -- never a half-written file.  The compiler is
replaced by a slow stand-in that writes the output in two halves; the real lock, temporary name, rename and stamp logic run."""
import os
import threading
import time

import pytest


def test_concurrent_build_lib_builds_once_and_atomically(tmp_path, monkeypatch):
    from voxelengine_amd import build as vb

    csrc = tmp_path / "csrc"
    csrc.mkdir()
    (csrc / "a.hip").write_text("kernel source 1\n")
    lib = csrc / "libvxrt.so"
    monkeypatch.setattr(vb, "CSRC", str(csrc))
    monkeypatch.setattr(vb, "LIB_PATH", str(lib))
    monkeypatch.setattr(vb, "lib_sources", lambda: [str(csrc / "a.hip")])
    calls = []
    seen_partial = []

    def fake_make(cmd, **kw):
        out = [a for a in cmd if a.startswith("OUT=")][0][4:]
        calls.append(out)
        assert out != "libvxrt.so"          # never straight into the library's own name
        with open(csrc / out, "w") as f:
            f.write("first half;")
            f.flush()
            time.sleep(0.3)
            f.write("second half")

    monkeypatch.setattr(vb.subprocess, "check_call", fake_make)

    def watcher(stop):
        while not stop.is_set():
            if lib.exists() and lib.read_text() != "first half;second half":
                seen_partial.append(lib.read_text())
            time.sleep(0.01)

    stop = threading.Event()
    w = threading.Thread(target=watcher, args=(stop,))
    w.start()
    errors = []

    def rank():
        try:
            assert vb.build_lib() == str(lib)
            assert lib.read_text() == "first half;second half"
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ranks = [threading.Thread(target=rank) for _ in range(6)]
    for t in ranks:
        t.start()
    for t in ranks:
        t.join()
    stop.set()
    w.join()
    assert not errors, errors
    assert len(calls) == 1, calls                       # one rank built, the others waited on the lock and found it fresh
    assert not seen_partial                              # the library's name never held a partial file
    assert not vb._stale(str(lib), vb.lib_sources())
    assert not [p for p in os.listdir(csrc) if ".tmp." in p]
    # an edit makes it stale again; VXRT_SKIP_STALE_CHECK keeps a profiled process from building
    (csrc / "a.hip").write_text("kernel source 2\n")
    monkeypatch.setenv("VXRT_SKIP_STALE_CHECK", "1")
    vb.build_lib()
    assert len(calls) == 1
    monkeypatch.delenv("VXRT_SKIP_STALE_CHECK")
    vb.build_lib()
    assert len(calls) == 2 and not vb._stale(str(lib), vb.lib_sources())
