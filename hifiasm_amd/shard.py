"""Launcher-side helpers of the sharded mode (one process per GPU): what travels over the launcher's
torch.distributed process group before the engine's own RCCL communicator exists.

Reads are partitioned by query read into contiguous global id ranges in rank order - the engine relies on
that (the all-gathered minimizer records are then already in global read order)."""
from __future__ import annotations

import numpy as np


def shard_range(n_total: int, rank: int, world: int):
    """contiguous, near-equal ranges in rank order: [lo, hi)"""
    return n_total * rank // world, n_total * (rank + 1) // world


def gather_lengths(dist, lengths: np.ndarray, device="cpu") -> np.ndarray:
    """read_length[] of ALL reads, replicated (4 B/read): all-gather of unequal pieces, padded to the longest."""
    import torch
    world = dist.get_world_size()
    n = torch.tensor([lengths.size], dtype=torch.int64, device=device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    ns = [int(x.item()) for x in ns]
    m = max(ns)
    buf = torch.zeros(m, dtype=torch.int32, device=device)
    buf[: lengths.size] = torch.from_numpy(lengths.astype(np.int32)).to(device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return np.concatenate([o[:k].cpu().numpy() for o, k in zip(out, ns)]).astype(np.uint32), ns


def share_unique_id(dist, make_id):
    """rank 0 creates the RCCL unique id (hao_dist_unique_id), everybody receives it"""
    box = [make_id() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]
