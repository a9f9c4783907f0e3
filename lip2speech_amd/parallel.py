"""Clip sharding for multi-GPU inference / evaluation.

The inference path has no exchange step: clips are independent, so N ranks (one process per GPU, launched with
``torch.distributed.run``) each take a contiguous shard of the clips, keep a full replica of the weights and run the
same kernels.  The only communication is the host-side gather of per-clip results (metrics, lengths), done with
``all_gather_object`` over whatever backend the job uses ("nccl" = RCCL on the GPU box, "gloo" in the CPU tests).

Padding note: the model ignores lengths, so zero-padded frames change results.  ``shard_batches``
therefore never re-pads: it cuts a list of already collated batches, so every batch keeps the composition the
single-process run would have used and results are identical to it.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n: int, rank: int, size: int) -> range:
    """Contiguous, balanced shard of ``range(n)``: the first ``n % size`` ranks get one extra item."""
    base, extra = divmod(n, size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard_batches(batches: Sequence, rank: int = None, size: int = None) -> List:
    r, s = world()
    rank = r if rank is None else rank
    size = s if size is None else size
    return [batches[i] for i in shard_range(len(batches), rank, size)]


def run_sharded(batches: Sequence, fn: Callable, gather: bool = True) -> List:
    """Apply ``fn`` to this rank's batches; with ``gather`` every rank gets the results of ALL batches in order."""
    rank, size = world()
    mine = [fn(b) for b in shard_batches(batches, rank, size)]
    if size == 1 or not gather:
        return mine
    parts: List = [None] * size
    dist.all_gather_object(parts, mine)
    return [x for part in parts for x in part]
