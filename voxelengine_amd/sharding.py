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
