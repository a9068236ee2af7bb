"""Self-launch of a one-process-per-GPU job on ONE node, for callers that start `bench.py --gpus N` as a plain program.

The parent never imports torch or touches HIP: it starts N fresh children of the same script (RANK, LOCAL_RANK,
WORLD_SIZE, MASTER_ADDR, MASTER_PORT in their environment, exactly what torch.distributed.run would set), relays the
single stdout line rank 0 prints and fails if any rank fails.  No process that has initialised the GPU is ever replaced
by another program (that takes the machine down on this pool): children are ordinary subprocesses.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading


def under_launcher(environ=None) -> bool:
    """True when this process already is one rank of a job (torch.distributed.run or launch_ranks started it)."""
    env = os.environ if environ is None else environ
    return "RANK" in env and "WORLD_SIZE" in env


def free_port() -> int:
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(nproc: int, argv: list[str], timeout: float | None = None, extra_env: dict | None = None) -> tuple[int, str]:
    """Run `argv` (a full command line) as `nproc` ranks; returns (exit code, rank 0's stdout).

    Rank 0's stdout is captured and returned (it is the job's result line); every other rank's stdout is sent to this
    process's stderr so that it cannot pollute the result; stderr of all ranks is inherited.  The exit code is 0 only if
    every rank exited 0; when one rank dies the others are terminated (a collective would otherwise wait for ever).
    """
    if nproc < 1:
        raise ValueError("nproc must be at least 1")
    port = free_port()
    procs = []
    for rank in range(nproc):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(nproc), "LOCAL_WORLD_SIZE": str(nproc),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
        if extra_env:
            env.update(extra_env)
        out = subprocess.PIPE if rank == 0 else sys.stderr
        procs.append(subprocess.Popen(argv, env=env, stdout=out, stderr=None, text=(rank == 0)))
    captured: list[str] = []
    reader = threading.Thread(target=lambda: captured.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    import time
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    pending = set(range(nproc))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
        if rc != 0 or (deadline is not None and time.monotonic() > deadline):
            if rc == 0:
                rc = 124  # timed out
            for r in pending:  # the exact processes started above, nothing else
                procs[r].terminate()
            for r in pending:
                try:
                    procs[r].wait(timeout=10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            pending.clear()
            break
        if pending:
            time.sleep(0.05)
    reader.join(timeout=10)
    return rc, (captured[0] if captured else "")
