"""Multi-GPU layout of the hot path: channels (tracking) / PRNs (acquisition) are independent units that all read the
same IF sample stream (gnss_flowgraph.cc:1227-1231 connects one conditioner to every channel), so they shard
across ranks with no reduction; the only exchange step is getting each sample block to every GPU once.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU tests).
The ingest rank owns the block and broadcasts it; results stay on the rank that owns the channel.
"""
from __future__ import annotations

from typing import List, Sequence


def shard_range(n_units: int, world: int, rank: int) -> range:
    """Contiguous, balanced partition: the first n_units % world ranks get one extra unit."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    q, r = divmod(n_units, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def owner_of(unit: int, n_units: int, world: int) -> int:
    q, r = divmod(n_units, world)
    cut = r * (q + 1)
    if unit < cut:
        return unit // (q + 1)
    return r + (unit - cut) // max(q, 1)


def broadcast_block(block, src: int = 0, async_op: bool = False):
    """Replicate one IF sample block (a real-view torch tensor, any device) from the ingest rank.
    A single large message per block: the broadcast is latency-bound at real-time rates (25 Msps * 8 B = 0.2 GB/s
    against ~153 GB/s per xGMI link), so blocks should hold >= 10 ms of samples."""
    import torch.distributed as dist
    return dist.broadcast(block, src=src, async_op=async_op)


def epoch_major_jobs(channel_ids: Sequence[int], per_channel: dict, epochs: int, n_samples: int, shifts: Sequence[float]) -> List[dict]:
    """Job table for one rank's channels: epoch-major, channel-minor, so that jobs reading the same samples are
    adjacent and land on the same XCD (multicorrelator.hip remaps blockIdx accordingly).
    per_channel[c] = (first_sample_offset, dict of NCO fields)."""
    rows = []
    for e in range(epochs):
        for slot, c in enumerate(channel_ids):
            off, p = per_channel[c]
            rows.append(dict(sample_offset=off + e * n_samples, n_samples=n_samples, code_slot=slot, shifts_chips=list(shifts), **p))
    return rows
