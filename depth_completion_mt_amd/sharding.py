"""Per-frame sharding across the GPUs of one node.

Frames are independent (the reference function is stateless: src/DC_lidar_only/img_completion.cpp
takes one Mat and returns one Mat), so the multi-GPU story is a partition of the frame index
range with NO collective on the data path: rank r owns a contiguous block of frames, has its
own dcmt context and stream, and never talks to the other ranks about pixels.
torch.distributed (RCCL when the ranks own GPUs, gloo in the CPU tests) is used only for
barriers and for the max-over-ranks of the elapsed time.
"""
from __future__ import annotations

from typing import List, Tuple


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [begin, end) of frames owned by `rank`: the first n % world ranks get one
    extra frame (1024 frames on 8 GPUs -> 128 each, the BASELINE configs[4] sharding)."""
    if world < 1 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad shard request")
    q, r = divmod(n_frames, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def all_shards(n_frames: int, world: int) -> List[Tuple[int, int]]:
    return [shard_range(n_frames, r, world) for r in range(world)]


def job_throughput(frames_per_rank: List[int], seconds_per_rank: List[float]) -> float:
    """Whole-job frames/s the way bench.py reports it: all frames / the slowest rank's time."""
    return sum(frames_per_rank) / max(seconds_per_rank)


class ShardedCompleter:
    """This rank's slice of a frame batch on this rank's GPU (one process per GPU).

    complete(frames) takes the GLOBAL batch (numpy [n][rows][cols]) or only indexes into it, runs
    this rank's frames through the HIP path and returns (begin, end, dense_block)."""

    def __init__(self, rank: int, world: int, device: int, rows: int, cols: int, max_frames_per_rank: int):
        from .api import Context
        self.rank, self.world = rank, world
        self.ctx = Context(device, rows, cols, max_frames_per_rank)

    def complete(self, frames, params=None):
        b, e = shard_range(len(frames), self.rank, self.world)
        return b, e, self.ctx.complete(frames[b:e], params) if e > b else None

    def close(self):
        self.ctx.close()
