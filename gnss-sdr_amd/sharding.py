"""Host-side control plane of the multi-GPU layout, one process per GPU -- what bench.py --gpus N runs on every rank.

Channels (tracking) and PRNs (acquisition) are independent units that all read the same IF sample stream (gnss_flowgraph.cc:1227-1231 connects one
conditioner to every channel), so they shard over the ranks with no reduction: channel c -> GPU c mod G, PRN p -> GPU (p - 1) mod G.  The one exchange
step -- every sample block reaching every GPU once -- is NOT here: it happens inside the engine (gsh_stream_group_*, csrc/stream_group.hip: RCCL over
xGMI).  torch.distributed ("nccl" = RCCL on ROCm; "gloo" in the CPU tests and the shared-GPU self-test) carries only the 128-byte communicator id, the
barrier and the MAX / SUM reductions of the timing contract.  The reference's whole multi-GPU logic is a random device pick
(cuda_multicorrelator.cu:136-155); the layout here is north_star's."""
from __future__ import annotations

import os
from typing import List


def channel_owner(channel: int, world: int) -> int:
    """channel c -> GPU c mod G."""
    return channel % world


def channels_of(rank: int, world: int, n_channels: int) -> List[int]:
    """Strong scaling: the channels of a fixed set that `rank` tracks."""
    _check(rank, world)
    return [c for c in range(n_channels) if channel_owner(c, world) == rank]


def weak_channel_prn(rank: int, channels_per_gpu: int, slot: int) -> int:
    """Weak scaling (bench.py): every GPU tracks channels_per_gpu channels; slot s of rank r is global channel r * C + s and tracks PRN (r * C + s) mod 32 + 1."""
    return (rank * channels_per_gpu + slot) % 32 + 1


def prn_owner(prn: int, world: int) -> int:
    """PRN p (1-based) -> GPU (p - 1) mod G."""
    return (prn - 1) % world


def prns_of(rank: int, world: int, n_prn: int = 32) -> List[int]:
    _check(rank, world)
    return [p for p in range(1, n_prn + 1) if prn_owner(p, world) == rank]


def _check(rank: int, world: int) -> None:
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")


class ControlPlane:
    """The launcher's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) turned into the few collective services a rank needs around the engine.
    world == 1 needs no torch.distributed at all."""

    def __init__(self, backend: str = "nccl", device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        _check(self.rank, self.world)
        self.backend = backend
        self.dist = None
        self._dev = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=device)
                self._dev = device
            else:
                dist.init_process_group(backend, rank=self.rank, world_size=self.world)
                self._dev = torch.device("cpu")   # (gloo reduces host tensors)
            self.dist = dist

    def barrier(self) -> None:
        if self.dist is not None:
            self.dist.barrier()

    def _reduce(self, v: float, op: str) -> float:
        if self.dist is None:
            return float(v)
        import torch
        t = torch.tensor([v], dtype=torch.float64, device=self._dev)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return float(t.item())

    def reduce_max(self, v: float) -> float:
        return self._reduce(v, "MAX")

    def reduce_min(self, v: float) -> float:
        return self._reduce(v, "MIN")

    def reduce_sum(self, v: float) -> float:
        return self._reduce(v, "SUM")

    def same_everywhere(self, v: float) -> bool:
        """True on every rank iff all ranks hold the same value (a checksum of checksums)."""
        return self.reduce_max(v) == self.reduce_min(v)

    def communicator_id(self) -> bytes | None:
        """The engine's 128-byte communicator id: made by gsh_comm_unique_id on rank 0, handed to every rank.  None for a world of one."""
        if self.world == 1:
            return None
        from .sample_stream import StreamGroup
        box = [StreamGroup.unique_id() if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def close(self) -> None:
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
