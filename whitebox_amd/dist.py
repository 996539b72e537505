"""Multi-GPU sharding of the mix path: one process per GPU, tracks partitioned in contiguous ranges, one
collective per render — the sum of the un-clamped partial masters onto the root (RCCL reduce over xGMI
with backend "nccl"; gloo on CPU in the tests) — then the master clamp on the root only.

The reference has no distributed code: this is the one exchange step the path has.  Tracks
are independent (no sends / side-chains in the reference), per-track peaks never leave the GPU that owns
the track, clamping a partial would be wrong, so the clamp (engine.cpp:1627-1636) runs after the reduce.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_tracks(n_tracks: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous range [first, first+count) of global track indices owned by `rank`; ranges are in rank
    order so that in-GPU summation order equals the reference's track order within a shard."""
    base, rem = divmod(n_tracks, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


class MasterReducer:
    """Sum partial masters [K][C][F] onto `root` and finalize (clamp) there.

    reduce() is asynchronous with respect to the caller's stream: the collective is enqueued with
    async_op=True, so the next render can be issued at once; finish(slot) makes the finalize stream wait
    for it and runs `finalize(buffer)` (the clamp kernel on the root).  Buffers are rotated by the caller
    (slot = step % n_slots) so that a render never overwrites a buffer a reduce is still reading.
    """

    def __init__(self, finalize: Callable[[torch.Tensor], None], root: int = 0, group=None):
        self.finalize = finalize
        self.root = root
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._work = {}             # slot -> outstanding collective

    def reduce(self, partial: torch.Tensor, slot: int = 0) -> None:
        if self.world > 1 and partial.is_cuda and dist.get_backend(self.group) == "gloo":
            # debugging aid (several ranks sharing one GPU, where RCCL cannot be used): through the host, synchronously
            torch.cuda.current_stream().synchronize()
            host = partial.cpu()
            dist.reduce(host, dst=self.root, op=dist.ReduceOp.SUM, group=self.group)
            if self.rank == self.root:
                partial.copy_(host)
            self._work[slot] = None
        elif self.world > 1:
            self._work[slot] = dist.reduce(partial, dst=self.root, op=dist.ReduceOp.SUM, group=self.group,
                                           async_op=True)
        else:
            self._work[slot] = None

    def finish(self, partial: torch.Tensor, slot: int = 0) -> None:
        w = self._work.pop(slot, None)
        if w is not None:
            w.wait()            # orders the CURRENT stream after the collective (no host block for nccl)
        if self.rank == self.root:
            self.finalize(partial)
