"""Multi-GPU partition of the hot path: rays are independent and the BVH is read-only, so
each rank holds a replica of the tree (deterministic GPU build from the same mesh — no broadcast needed) and
traces a tile of the image; the only exchange is one gather of the 16-byte hit records (RCCL over xGMI when
the backend is "nccl").

Tiles are interleaved rows: rank r of `world` owns rows r, r+world, r+2*world, ... — every rank sees the same
mix of the scene, so the slowest rank is not the one looking at the dense part.
"""
import numpy as np


def rank_rows(height, rank, world):
    """Rows of a `height`-row image owned by `rank` (interleaved)."""
    return np.arange(rank, height, world)


def rows_per_rank(height, rank, world):
    return (height - rank + world - 1) // world


def gather_buffer(local_hits, world, rank, dst=0):
    """The root's receive buffer for gather_hit_records (None elsewhere): world equal slices, rank-major."""
    import torch

    if rank != dst:
        return None
    return torch.empty(world * local_hits.numel(), dtype=local_hits.dtype, device=local_hits.device)


def gather_hit_records(local_hits, world, rank, dist, out=None, dst=0, async_op=False):
    """Gather equal-sized per-rank hit buffers (torch uint8 tensors) to rank `dst`: the root receives every rank's
    records over its direct links at once, the others only send (an all-gather would move (world-1)x the data into
    every GPU for nothing).  `out`: the root's buffer from gather_buffer() (allocated here when None).
    Returns (out or None, work)."""
    if out is None:
        out = gather_buffer(local_hits, world, rank, dst)
    work = dist.gather(local_hits, gather_list=list(out.chunk(world)) if rank == dst else None, dst=dst, async_op=async_op)
    return out, work


def assemble_image(gathered, width, height, world, record_dtype):
    """Undo the interleaving: gathered = [rank0 rows | rank1 rows | ...] -> row-major image records.
    Requires height % world == 0 (equal tiles)."""
    rec = np.frombuffer(gathered, dtype=record_dtype) if not isinstance(gathered, np.ndarray) else gathered.view(record_dtype)
    per = height // world
    img = np.empty((height, width), dtype=record_dtype)
    for r in range(world):
        img[r::world] = rec[r * per * width:(r + 1) * per * width].reshape(per, width)
    return img.reshape(-1)
