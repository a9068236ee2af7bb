"""bench.py's self-launch for --gpus N > 1 (voxelengine_amd/launcher.py), on CPU: N fresh children with the
torch.distributed environment, rank 0's single stdout line relayed, a failing rank fails the job, and the parent never
needs a GPU.  The end-to-end case runs a real 2-rank gloo all-reduce in the children."""
import json
import os
import subprocess
import sys
import textwrap

from voxelengine_amd import launcher

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(tmp_path, body):
    p = tmp_path / "child.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_children_get_the_distributed_environment_and_rank0_line_is_relayed(tmp_path):
    argv = _script(tmp_path, """
        import json, os
        env = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        print(json.dumps(env))       # rank 0: the result line; other ranks: must not reach the parent's stdout
    """)
    rc, out = launcher.launch_ranks(3, argv, timeout=60)
    assert rc == 0
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1
    env = json.loads(lines[0])
    assert env["RANK"] == "0" and env["LOCAL_RANK"] == "0" and env["WORLD_SIZE"] == "3"
    assert env["MASTER_ADDR"] == "127.0.0.1" and int(env["MASTER_PORT"]) > 0


def test_a_failing_rank_fails_the_job_and_the_others_are_stopped(tmp_path):
    argv = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(120)              # a rank stuck in a collective whose peer died
    """)
    rc, out = launcher.launch_ranks(2, argv, timeout=60)
    assert rc == 7


def test_timeout_stops_every_rank(tmp_path):
    argv = _script(tmp_path, "import time; time.sleep(120)\n")
    rc, _ = launcher.launch_ranks(2, argv, timeout=1.0)
    assert rc == 124


def test_under_launcher_detection():
    assert launcher.under_launcher({"RANK": "0", "WORLD_SIZE": "2"})
    assert not launcher.under_launcher({})
    assert not launcher.under_launcher({"WORLD_SIZE": "2"})


def test_two_ranks_rendezvous_over_gloo(tmp_path):
    argv = _script(tmp_path, """
        import os, json
        import torch, torch.distributed as dist
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        t = torch.tensor([float(os.environ["RANK"]) + 1.0])
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            print(json.dumps({"sum": t.item()}))
        dist.destroy_process_group()
    """)
    rc, out = launcher.launch_ranks(2, argv, timeout=240)
    assert rc == 0
    assert json.loads(out.strip().splitlines()[-1]) == {"sum": 3.0}


def test_bench_started_bare_with_two_gpus_becomes_the_launcher():
    """Without a GPU the children stop with bench.py's own message; what matters here is that a bare
    `python bench.py --gpus 2` no longer refuses to start: it spawns its ranks and propagates their failure."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["HIP_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "launch with torch.distributed.run" not in p.stderr
    assert "needs a GPU" in p.stderr
