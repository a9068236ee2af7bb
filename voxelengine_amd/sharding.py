"""Screen-strip sharding of one frame across the GPUs of a node and the framebuffer gather.

The reference is single-GPU; this is the MI355X-native addition named by the north star:
the brickmap is replicated in every GPU's HBM, the frame is cut into strips of ``strip_rows`` rows, strip
``s`` belongs to rank ``s % world_size`` (interleaved, so sky-heavy and terrain-heavy rows spread evenly),
every rank renders its strips into a packed local buffer, and the packed buffers are gathered to rank 0
over RCCL/xGMI (a gather, not a reduction: each peer's bytes cross its own direct link into the root once)
where a small HIP kernel scatters the strips back into frame order.  No other collective is on the data
path; rays are independent.
"""
from __future__ import annotations

from dataclasses import dataclass

STRIP_ROWS = 16  # one 256-thread workgroup is 16 rows tall


@dataclass(frozen=True)
class ShardPlan:
    width: int
    height: int
    strip_rows: int
    world_size: int
    rank: int

    @property
    def n_strips(self) -> int:
        return (self.height + self.strip_rows - 1) // self.strip_rows

    def strips_of(self, rank: int) -> list[int]:
        return list(range(rank, self.n_strips, self.world_size))

    def rows_of(self, rank: int) -> int:
        rows = 0
        for s in self.strips_of(rank):
            begin = s * self.strip_rows
            rows += min(begin + self.strip_rows, self.height) - begin
        return rows

    @property
    def local_rows(self) -> int:
        return self.rows_of(self.rank)

    @property
    def max_rows(self) -> int:
        """Every rank's packed buffer is padded to the largest shard so the gather is uniform."""
        return max(self.rows_of(r) for r in range(self.world_size))

    @property
    def shard_bytes(self) -> int:
        return self.max_rows * self.width * 4

    def frame_row(self, rank: int, local_row: int) -> int:
        """Frame row of packed row ``local_row`` of ``rank`` (inverse of the kernel's out_row mapping)."""
        return (local_row // self.strip_rows * self.world_size + rank) * self.strip_rows + local_row % self.strip_rows


class GatherPipeline:
    """Pipeline of frames: the gather of frame k runs (on RCCL's stream) while the next frames are rendered.

    Per frame: ``buf = pipe.local(k)`` -> render into it -> ``pipe.submit(k)``.  The root de-interleaves frame k-1
    inside ``submit(k)``; ``pipe.flush()`` completes the last frame.  Each rank owns ``depth`` packed shard buffers,
    the root ``depth`` gather buffers, so a buffer is reused only after the collective that read it has completed
    (``work.wait()`` orders the stream, it does not block the host for NCCL).

    Two render streams (``with pipe.stream(k): ...`` around the three calls of frame k, device tensors only): frame k's
    render, the wait for the gather that last read its buffer, the issue of its own gather and the de-interleave of frame
    k-1 all sit on stream k mod 2, so the render of frame k+1 waits for nothing of frame k and fills the SIMD slots its
    last waves leave -- a strip shard is a small launch whose tail is a fifth of it (tools/views_probe.py: 1/8 shards of
    16 views 6.7 -> 7.9 Grays/s per GPU).  ``frame_on_root`` may then be a pair of buffers (frame k lands in buffer
    k mod 2): two de-interleaves on two streams must not write one buffer.  With streams, ``depth`` should be 3: the
    persistent render kernel leaves RCCL's kernel no room on the GPU until its waves start to leave, so the gather of
    frame k runs in the tail of frame k+1 -- exactly when frame k+2 should start, which with two buffers would have to
    wait for that gather.
    """

    def __init__(self, plan: ShardPlan, make_buffer, frame_on_root, deinterleave, group=None, nbytes: int | None = None,
                 streams=None, depth: int = 2):
        """``nbytes``: bytes each rank contributes per step (default one packed shard; several views per launch
        contribute several shards back to back -- one larger collective instead of several small ones).
        ``streams``: None, or two ``torch.cuda.Stream`` objects (see above).  ``depth``: buffers per rank (>= 2)."""
        self.plan, self.frame, self.deinterleave, self.group = plan, frame_on_root, deinterleave, group
        nbytes = plan.shard_bytes if nbytes is None else int(nbytes)
        assert depth >= 2
        self.depth = int(depth)
        self.locals = [make_buffer(nbytes) for _ in range(self.depth)]
        self.shards = [make_buffer(plan.world_size * nbytes).view(plan.world_size, nbytes)
                       for _ in range(self.depth)] if plan.rank == 0 else [None] * self.depth
        self.works = [None] * self.depth
        self.pending = None  # frame index gathered but not yet de-interleaved on the root
        self.streams = streams
        self.read = [None] * self.depth  # with streams: events behind the de-interleave that read the root's gather buffer b
        if streams is not None:
            assert len(streams) == 2

    def stream(self, k: int):
        """Context manager: the stream frame k's work is issued on (a no-op without render streams)."""
        if self.streams is None:
            import contextlib
            return contextlib.nullcontext()
        import torch
        return torch.cuda.stream(self.streams[k & 1])

    def frame_of(self, k: int):
        """The root's buffer frame k is (or will be) de-interleaved into."""
        return self.frame[k & 1] if isinstance(self.frame, (list, tuple)) else self.frame

    def local(self, k: int):
        b = k % self.depth
        if self.works[b] is not None:  # the gather issued `depth` frames ago read this buffer
            self.works[b].wait()
            self.works[b] = None
        return self.locals[b]

    def submit(self, k: int):
        import torch.distributed as dist

        b = k % self.depth
        p = self.plan
        if self.read[b] is not None:  # frame k-depth's de-interleave may have run on the other stream: this gather overwrites what it read
            import torch
            torch.cuda.current_stream().wait_event(self.read[b])
            self.read[b] = None
        if p.rank == 0:
            self.works[b] = dist.gather(self.locals[b], [self.shards[b][r] for r in range(p.world_size)], dst=0,
                                        group=self.group, async_op=True)
        else:
            self.works[b] = dist.gather(self.locals[b], None, dst=0, group=self.group, async_op=True)
        self._finish(k - 1)
        self.pending = k

    def _finish(self, k: int):
        if self.pending is None or self.pending != k:
            return
        b = k % self.depth
        if self.works[b] is not None:
            self.works[b].wait()
            self.works[b] = None
        if self.plan.rank == 0:
            self.deinterleave(self.shards[b], self.frame_of(k))
            if self.streams is not None:
                import torch
                self.read[b] = torch.cuda.Event()
                self.read[b].record()
        self.pending = None

    def flush(self):
        if self.pending is not None:
            with self.stream(self.pending):
                self._finish(self.pending)
        for b in range(self.depth):
            if self.works[b] is not None:
                self.works[b].wait()
                self.works[b] = None


def gather_frame(plan: ShardPlan, local_shard, shards_on_root, frame_on_root, deinterleave, group=None):
    """Gather the packed shard buffers to rank 0 and rebuild the frame there.

    ``local_shard``: uint8 tensor of ``plan.shard_bytes`` on this rank's device.
    ``shards_on_root``: (world_size, shard_bytes) uint8 tensor on rank 0 (None elsewhere).
    ``deinterleave(shards, frame)``: scatters packed strips into frame order (the HIP kernel behind
    ``Context.deinterleave_strips`` in production).
    """
    import torch.distributed as dist

    if plan.world_size == 1:
        deinterleave(local_shard.view(1, -1), frame_on_root)
        return frame_on_root
    if plan.rank == 0:
        dist.gather(local_shard, [shards_on_root[r] for r in range(plan.world_size)], dst=0, group=group)
        deinterleave(shards_on_root, frame_on_root)
        return frame_on_root
    dist.gather(local_shard, None, dst=0, group=group)
    return None
