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
    """Independent batches on ONE GPU, G per launch chain and several chains in flight, all on ONE packed weight blob.

    A pass over one B=32 batch is a chain of ~1 500 dependent launches whose step kernels are latency-bound: each of the 1 200 step launches
    moves the same 10-12 MB of step weights for 32 rows of work, and every kernel has idle phases (the ~1 us boundary, ramp, parameter
    fetch, store drain).  Batches are independent, so two things recover the idle chip:

    * `group` = G: the next G batches run as rows g*B..g*B+B-1 of ONE chain (`l2s_inference_multi`): one set of step launches and one pass
      over the step weights per G batches, register-blocked step kernels from 64 rows on, and the dense kernels (front-end, GEMMs) run at
      G times the rows, where their tiles fill the chip better.  Every kernel is row-independent: results are bit-identical to running
      the batches one by one.
    * `n_inflight` chains run concurrently, each issued by its own host thread on its own HIP stream (the C-ABI call releases the GIL): the
      dense kernels of one chain fill the holes of the other's decode loop.  All threads use the SAME `NativeModel` - one weight blob, so
      concurrent step launches hit the same lines in L2 / Infinity Cache - with a workspace per thread.

    Hardware queues: every stream needs a HIP hardware queue of its own, and the runtime's default of 4 includes the null stream's.  Up to
    three chains work out of the box; for four export GPU_MAX_HW_QUEUES=8 before the process makes its first HIP call.  Five or more
    active queues collapse (0.9x of ONE chain): the chip runs four compute pipes.

    `tensors` / `keys`: the checkpoint tensors as for `NativeModel.load`.  `map(batches)` takes a list of (video, emb, gumbel) and returns
    the (mel_post, lengths, attn) tuples in order."""

    def __init__(self, tensors: Dict[str, torch.Tensor] = None, keys=None, n_inflight: int = 2, device=None, group: int = 1, model=None):
        """`model`: an already packed `NativeModel` to share (then `tensors` is not needed) - e.g. a second pool with another shape of
        concurrency over the same weight blob."""
        from . import native
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if n_inflight > 4:
            raise ValueError("more than four chains in flight oversubscribe the GPU's compute pipes (measured 0.9x of ONE at a time)")
        if not 1 <= group <= 8:
            raise ValueError("group must be in 1..8 (L2S_MAX_GROUP)")
        if n_inflight == 4 and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 5:
            warnings.warn("InflightPool(n_inflight=4) needs GPU_MAX_HW_QUEUES >= 5 set before the first HIP call; with the default the "
                          "four streams share hardware queues and run slower than three in flight", RuntimeWarning)
        self.group = group
        if model is None:
            model = native.NativeModel()              # ONE packed blob for every chain
            model.load(tensors, list(tensors.keys()) if keys is None else list(keys))
        self.model = model
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(max(1, n_inflight))]
        self.copy_stream = None                       # created on first use of map(prepare=...): ONE stream for every chain's input staging

    @property
    def n_inflight(self) -> int:
        return len(self.streams)

    @property
    def models(self):
        """The per-worker model handles - all the same object (one weight blob)."""
        return [self.model] * len(self.streams)

    def map(self, batches: Sequence, S: int = 300, want_attn: bool = False, fn: Callable = None, prepare: Callable = None, shape_of: Callable = None) -> List:
        """Run `inference` on every (video, emb, gumbel) of `batches`.  Consecutive batches form groups of `self.group` (the last one may be
        smaller; batches of a group must share one shape, so a shape change also closes a group); worker i takes the next unclaimed group
        (dynamic schedule).  `fn(model, batch)` replaces the default call and is applied batch by batch (no grouping).
        `prepare(batch) -> (video, emb, gumbel)` stages a batch on the device - e.g. the host-to-device copy of packed uint8 frames and their
        normalisation (`datasets.device.PackedFrames.to_device`).  It runs on the pool's copy stream ONE GROUP AHEAD: a worker stages its next
        group right before it launches the current one, so the copies and the normalise kernel travel under that chain's own compute as well
        as the others' (staged on the compute stream they cost the chain 4 ms per group of eight).  With it `shape_of(batch)` must give the
        (B,3,T,H,W) shape the batch will have (groups are formed before preparation)."""
        shape = (lambda b: tuple(shape_of(b))) if shape_of is not None else (lambda b: tuple(b[0].shape))
        items: List[List[int]] = []
        if fn is not None or self.group == 1:
            items = [[i] for i in range(len(batches))]
        else:
            cur: List[int] = []
            for i, b in enumerate(batches):
                if cur and (len(cur) == self.group or shape(batches[cur[0]]) != shape(b)):
                    items.append(cur)
                    cur = []
                cur.append(i)
            if cur:
                items.append(cur)
        todo: "queue.Queue[List[int]]" = queue.Queue()
        for it in items:
            todo.put(it)
        out: List = [None] * len(batches)
        errors: List[BaseException] = []
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))          # inputs produced on the caller's stream

        if prepare is not None and self.copy_stream is None:
            self.copy_stream = torch.cuda.Stream(device=self.device)
        copy_lock = threading.Lock()                                   # one group is staged at a time (the copies share the DMA engine anyway)

        def claim():
            try:
                return todo.get_nowait()
            except queue.Empty:
                return None

        def stage(idx, compute_stream):
            """prepare() the batches of a group on the copy stream; returns (tensors, event).  The tensors are handed to the compute stream
            (record_stream: the caching allocator must not recycle them while that stream still reads them)."""
            with copy_lock, torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(ready)
                grp = [prepare(batches[i]) for i in idx]
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            for b in grp:
                for t in b:
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(compute_stream)
            return grp, ev

        def worker(w: int):
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(self.streams[w]):
                    self.streams[w].wait_event(ready)
                    staged = None
                    if prepare is not None and fn is None:
                        idx = claim()
                        staged = (idx, *stage(idx, self.streams[w])) if idx is not None else None
                    while True:
                        if prepare is not None and fn is None:
                            if staged is None:
                                break
                            idx, grp, ev = staged
                            nxt = claim()                              # the next group's inputs travel under this group's compute
                            staged = (nxt, *stage(nxt, self.streams[w])) if nxt is not None else None
                            self.streams[w].wait_event(ev)
                        else:
                            idx = claim()
                            if idx is None:
                                break
                            if fn is not None:
                                out[idx[0]] = fn(self.model, batches[idx[0]])
                                continue
                            grp = [batches[i] for i in idx]
                        if len(idx) == 1:
                            out[idx[0]] = self.model.inference(*grp[0], S=S, want_attn=want_attn)
                        else:
                            for i, r in zip(idx, self.model.inference_multi(grp, S=S, want_attn=want_attn)):
                                out[i] = r
            except BaseException as e:      # noqa: BLE001 - re-raised on the caller's thread
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(w,)) for w in range(min(self.n_inflight, max(1, len(items))))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        cur_stream = torch.cuda.current_stream(self.device)
        for st in self.streams:                                     # results are consumed on the caller's stream
            cur_stream.wait_stream(st)
        return out
