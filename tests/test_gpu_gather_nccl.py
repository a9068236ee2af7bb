"""The RCCL leg of the N > 1 path on the GPU box: `sharding.GatherPipeline` with the asynchronous
`dist.gather` on DEVICE tensors over an `nccl` (= RCCL) process group, packed strip shards rendered by the HIP kernels,
`vxrt_deinterleave_views` on the root -- what `bench.py --force-gather` does, as a test.  The box has one GPU, so the group
has one rank: the code path (communicator, stream ordering between the render stream and RCCL's, the two-deep buffer
rotation, the de-interleave launch) is the N > 1 one, the wire is not.  Frames must equal direct single-view renders."""
import os
import socket

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gather_pipeline_over_a_one_rank_nccl_group(vxo):
    import torch
    import torch.distributed as dist

    import voxelengine_amd as vx
    from voxelengine_amd import sharding

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    # RCCL prints its version banner on fd 1 when the communicator is created; pytest captures it, nothing to do here
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    ctx = vx.Context(0)
    try:
        X = Y = Z = 256
        ctx.build_world(vx.GEN_INT_TERRAIN, X, Y, Z, 32)
        ctx.SetEnvironment((0.5, 0.7, 0.3), (2, 2, 2), (0.5, 0.5, 0.5))
        ctx.SetFOV(90.0)
        W, H, V = 324, 200, 3   # 200 rows: 12 whole strips of 16 + a ragged one
        world = 1
        plan = sharding.ShardPlan(W, H, sharding.STRIP_ROWS, world, 0)
        step_bytes = V * plan.shard_bytes
        frames = torch.zeros((V, H, W, 4), dtype=torch.uint8, device=dev)
        seen = []

        def deinterleave(sh, fr):
            ctx.deinterleave_views(W, H, plan.strip_rows, world, sh, step_bytes, plan.shard_bytes, V, fr, W * H * 4)
            seen.append(fr.clone())

        # as bench.py does for N > 1: consecutive steps on two render streams, frame k de-interleaved into buffer k mod 2
        frames2 = [frames, torch.zeros_like(frames)]
        pipe = sharding.GatherPipeline(plan, lambda n: torch.zeros(n, dtype=torch.uint8, device=dev), frames2, deinterleave,
                                       nbytes=step_bytes, streams=[torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)], depth=3)
        cams = ["A", "B", "D", "C"]

        def views_of(step, target):
            out = []
            for j in range(V):
                g = step * V + j
                pos, f, u, r = helpers.camera(cams[g % 4], (X, Y, Z), vxo)
                fb = target[j] if target.dim() == 4 else target[j * plan.shard_bytes:(j + 1) * plan.shard_bytes]
                out.append(dict(fb=fb, origin=pos, fwd=f, up=u, right=r, frame_number=g + 1))
            return out

        opts = dict(shadow=True, bounce_samples=1)
        nsteps = 7   # more steps than pipeline slots: all three buffers are reused, on both streams
        for k in range(nsteps):
            with pipe.stream(k):
                ctx.RenderViews(W, H, views_of(k, pipe.local(k)),
                                vx.RenderOptions(strip_rows=plan.strip_rows, strip_count=world, strip_index=0, compact=True, **opts))
                pipe.submit(k)
        pipe.flush()
        torch.cuda.synchronize()
        assert len(seen) == nsteps
        for k in range(nsteps):
            direct = torch.zeros_like(frames)
            for v in views_of(k, direct):
                ctx.RenderScreen(W, H, v["fb"], v["origin"], v["fwd"], v["up"], v["right"],
                                 vx.RenderOptions(frame_number=v["frame_number"], **opts))
            torch.cuda.synchronize()
            assert torch.equal(seen[k], direct), "gathered frames of step %d differ from direct renders" % k
        assert int(np.count_nonzero(seen[0].cpu().numpy())) > 0
        dist.barrier()
    finally:
        ctx.close()
        dist.destroy_process_group()
