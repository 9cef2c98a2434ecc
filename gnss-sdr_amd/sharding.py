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


class BlockDistributor:
    """Replicates raw IF sample blocks from the ingest rank to every rank, shaped for xGMI rather than for a switch.

    xGMI is point-to-point: a GPU has one link to each of its 7 peers (~153 GB/s each).  A ring broadcast pushes the whole
    block through ONE link per hop; here the ingest rank instead SCATTERS 1/world of the block to every peer over all of its
    links at once, then an ALL-GATHER lets every rank collect the other pieces over all of its own links -- every link
    carries block/world bytes twice instead of one link carrying the whole block.  Blocks travel in the front-end's raw
    item type (ibyte: 2 bytes per sample instead of 8 for complex64); each rank converts on its own GPU
    (gsh_convert_samples_device).  mode="broadcast" keeps the single dist.broadcast as a fallback.

    Buffers are flat uint8/int8 tensors of the same length on every rank; `nbytes` is padded to a multiple of world."""

    def __init__(self, nbytes: int, world: int, rank: int, src: int = 0, mode: str = "scatter_allgather"):
        if mode not in ("scatter_allgather", "broadcast"):
            raise ValueError(mode)
        self.world, self.rank, self.src, self.mode = world, rank, src, mode
        self.chunk = (nbytes + world - 1) // world
        self.padded = self.chunk * world

    def start(self, dst, src_block=None, piece=None):
        """Queue the distribution of one block.  dst: flat tensor of `padded` bytes on every rank (receives the block);
        src_block: the block on the ingest rank (may be `dst` itself); piece: a `chunk`-byte scratch tensor per rank.
        Returns a list of work handles; call finish() before reading dst."""
        import torch.distributed as dist
        if self.world == 1:
            if src_block is not None and src_block.data_ptr() != dst.data_ptr():
                dst.copy_(src_block)
            return []
        if self.mode == "broadcast":
            if self.rank == self.src and src_block is not None and src_block.data_ptr() != dst.data_ptr():
                dst.copy_(src_block)
            return [dist.broadcast(dst, src=self.src, async_op=True)]
        pieces = list(src_block.view(self.world, self.chunk).unbind(0)) if self.rank == self.src else None
        w1 = dist.scatter(piece, scatter_list=pieces, src=self.src, async_op=True)
        if dist.get_backend() != "nccl":
            w1.wait()  # RCCL runs both on the communicator's stream, in order; gloo's asynchronous ops are unordered
        w2 = dist.all_gather_into_tensor(dst, piece, async_op=True)
        return [w1, w2]

    @staticmethod
    def finish(works) -> None:
        for w in works:
            w.wait()


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
