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

import os
import queue
import threading
import time
import warnings
from typing import Callable, Dict, List, Sequence, Tuple

import torch
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


class InflightPool:
    """Several independent batches in flight on ONE GPU.

    A pass is a chain of ~1 500 dependent launches in which every kernel has idle phases (the 1 us boundary, ramp, parameter fetch,
    store drain): one chain keeps the chip busy for about 60 % of the time.  Batches are independent, so `n_inflight` host threads,
    each with its own HIP stream and its own packed-weight handle, keep that many chains interleaved on the chip: 1.43x the throughput
    with two, 1.66x with three (B=32, T=29, S=300; tools/two_batches.py).  Every batch is computed exactly as it would be alone
    (same kernels, same order within its stream) - results are bit-identical to the one-at-a-time run.

    Hardware queues: every stream needs a HIP hardware queue of its own, and the runtime's default of 4 includes the null stream's.
    Up to three in flight work out of the box; for four (1.81x) export GPU_MAX_HW_QUEUES=8 before the process makes its first HIP
    call - with the default, four streams share queues and the throughput DROPS below three-in-flight (1.42x).  Five or more active
    queues collapse (0.9x) whatever the setting: the chip runs four compute pipes.

    `stagger=True` starts worker w a fraction w/n of a cycle late (workers that start together move in lockstep - all in their encoder,
    then all in their decode loops - and drift apart only over tens of passes).  Measured at 20 passes: four in flight 1.38 -> 1.44 M
    mel-frames/s, three in flight 1.46 -> 1.39 M (the ramp costs more than the lockstep); no difference from 60 passes on.  Off by default.

    `tensors` / `keys`: the checkpoint tensors as for `NativeModel.load`.  `map(batches)` takes a list of (video, emb, gumbel) and
    returns the (mel_post, lengths, attn) tuples in order."""

    def __init__(self, tensors: Dict[str, torch.Tensor], keys=None, n_inflight: int = 3, device=None, stagger: bool = False):
        from . import native
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        keys = list(tensors.keys()) if keys is None else list(keys)
        if n_inflight > 4:
            raise ValueError("more than four batches in flight oversubscribe the GPU's compute pipes (measured 0.9x of ONE at a time)")
        if n_inflight == 4 and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 5:
            warnings.warn("InflightPool(n_inflight=4) needs GPU_MAX_HW_QUEUES >= 5 set before the first HIP call; with the default the "
                          "four streams share hardware queues and run slower than three in flight", RuntimeWarning)
        self.models, self.streams = [], []
        self.stagger = stagger
        self._cycle_s = {}            # per kind of work: GPU time of one pass of one worker with all workers busy (first full-load call)
        for _ in range(max(1, n_inflight)):
            nm = native.NativeModel()
            nm.load(tensors, keys)
            self.models.append(nm)
            self.streams.append(torch.cuda.Stream(device=self.device))

    @property
    def n_inflight(self) -> int:
        return len(self.models)

    def map(self, batches: Sequence, S: int = 300, want_attn: bool = False, fn: Callable = None) -> List:
        """Run `inference` on every (video, emb, gumbel) of `batches`; worker i takes the next unclaimed batch (dynamic schedule).
        `fn(model, batch)` replaces the default `model.inference(*batch, S=S, want_attn=want_attn)`."""
        todo: "queue.Queue[int]" = queue.Queue()
        for i in range(len(batches)):
            todo.put(i)
        out: List = [None] * len(batches)
        errors: List[BaseException] = []
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))          # inputs produced on the caller's stream

        # Workers that start together run in lockstep - all in their encoder at once, then all in their decode loops - and drift apart
        # only over tens of passes; the gain comes from one batch's dense kernels overlapping the others' decode chains.  So worker w
        # starts w/n of a cycle late (the cycle = one pass of one worker under full load, measured on this pool's first call).
        kind = fn if fn is not None else ("inference", S, want_attn)
        cycle = self._cycle_s.get(kind)
        delay = (cycle / self.n_inflight) if (self.stagger and cycle and len(batches) > self.n_inflight) else 0.0
        t_first = [None]

        def worker(w: int):
            try:
                torch.cuda.set_device(self.device)
                if delay > 0.0 and w > 0:
                    time.sleep(w * delay)
                with torch.cuda.stream(self.streams[w]):
                    self.streams[w].wait_event(ready)
                    n_done = 0
                    while True:
                        try:
                            i = todo.get_nowait()
                        except queue.Empty:
                            break
                        probe = w == 0 and n_done == 0 and cycle is None and len(batches) >= self.n_inflight
                        if probe:       # this pool's first full-load pass: its GPU time is the cycle the stagger is derived from
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(self.streams[w])
                        if fn is not None:
                            out[i] = fn(self.models[w], batches[i])
                        else:
                            video, emb, gumbel = batches[i]
                            out[i] = self.models[w].inference(video, emb, gumbel, S=S, want_attn=want_attn)
                        n_done += 1
                        if probe:
                            e1.record(self.streams[w])
                            e1.synchronize()
                            t_first[0] = e0.elapsed_time(e1) * 1e-3
            except BaseException as e:      # noqa: BLE001 - re-raised on the caller's thread
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(w,)) for w in range(min(self.n_inflight, max(1, len(batches))))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        if t_first[0] is not None:
            self._cycle_s[kind] = t_first[0]
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:                                     # results are consumed on the caller's stream
            cur.wait_stream(st)
        return out
