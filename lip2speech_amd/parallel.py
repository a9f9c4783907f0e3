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

import collections
import os
import queue
import threading
import warnings
from typing import Callable, Dict, Iterable, Iterator, List, Sequence, Tuple

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
        self._stats_lock = threading.Lock()
        self.stats = collections.Counter()            # imap(): groups / batches issued, by entry point ("inference_groups", "forward_batches", ...)

    @staticmethod
    def balanced_groups(run: List[int], group: int, chains: int) -> List[List[int]]:
        """Cut `run` into consecutive groups of at most `group` items whose count is a multiple of `chains` (when the run is long enough)
        and whose sizes differ by at most one."""
        k = len(run)
        if k == 0:
            return []
        n = min(k, chains * -(-k // (chains * group)))
        base, extra = divmod(k, n)
        out, pos = [], 0
        for g in range(n):
            size = base + (1 if g < extra else 0)
            out.append(run[pos:pos + size])
            pos += size
        return out

    @staticmethod
    def chains_for(k: int, group: int, max_chains: int = 3) -> int:
        """How many chains to keep in flight for a run of `k` same-shape batches: the count in 2..max_chains whose balanced groups carry the
        most rows per launch (the decode loop's launches cost nearly the same at 160 rows as at 256), the LARGER count on a tie: since the step
        kernels' blocks take half a compute unit (options "lstm_x3" = 3, "flat_half"), kernels of different chains run side by side on the
        CUs, and a third chain is +6 % over two (3.52 against 3.30 M mel-frames/s at 192 batches; four: 3.43).  20 batches in groups of up to
        8: three chains (7 + 7 + 6 rows of batches per launch) rather than two (four groups of 5)."""
        if k <= 0:
            return 1
        best, best_rows = 1, 0.0
        for c in range(2, max(2, max_chains) + 1):
            n = len(InflightPool.balanced_groups(list(range(k)), group, c))
            rows = k / max(1, n)
            if rows >= best_rows - 1e-9:
                best, best_rows = c, max(rows, best_rows)
        return min(best, max(1, k))

    @property
    def n_inflight(self) -> int:
        return len(self.streams)

    @property
    def models(self):
        """The per-worker model handles - all the same object (one weight blob)."""
        return [self.model] * len(self.streams)

    def map(self, batches: Sequence, S: int = 300, want_attn: bool = False, fn: Callable = None, prepare: Callable = None, shape_of: Callable = None) -> List:
        """Run `inference` on every (video, emb, gumbel) of `batches`.  Consecutive batches of one shape form groups of at most `self.group`,
        balanced over the chains in flight (`balanced_groups`; a shape change closes a group); worker i takes the next unclaimed group
        (dynamic schedule).  `fn(model, batch)` replaces the default call and is applied batch by batch (no grouping).
        `prepare(batch) -> (video, emb, gumbel)` stages a batch on the device - e.g. the host-to-device copy of packed uint8 frames and their
        normalisation (`datasets.device.PackedFrames.to_device`).  It runs on the pool's copy stream ONE GROUP AHEAD: a worker stages its next
        group right before it launches the current one, so the copies and the normalise kernel travel under that chain's own compute as well
        as the others' (staged on the compute stream they cost the chain 4 ms per group of eight).  With it `shape_of(batch)` must give the
        (B,3,T,H,W) shape the batch will have (groups are formed before preparation)."""
        from . import native
        shape = (lambda b: tuple(shape_of(b))) if shape_of is not None else (lambda b: tuple(b[0].shape))
        items: List[List[int]] = []
        if fn is not None or self.group == 1:
            items = [[i] for i in range(len(batches))]
        else:
            # runs of same-shape batches, each cut into groups of at most `self.group` - a multiple of the chains in flight, sizes differing
            # by at most one, so every chain gets the same rows per round and no chain idles behind a short tail group (20 batches on two
            # chains of up to 8: four groups of 5, not 8 + 8 + 4)
            runs: List[List[int]] = []
            for i, b in enumerate(batches):
                if runs and shape(batches[runs[-1][0]]) == shape(b):
                    runs[-1].append(i)
                else:
                    runs.append([i])
            for run in runs:
                items.extend(self.balanced_groups(run, self.group, self.n_inflight))
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

        n_workers = min(self.n_inflight, max(1, len(items)))

        def worker(w: int):
            try:
                torch.cuda.set_device(self.device)
                native.set_thread_chains(n_workers)        # this thread's chain shares the chip with n_workers - 1 others: half-CU block forms
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
                        # ONE arithmetic whatever the group size: a one-batch tail group takes the grouped entry too (l2s_inference_multi never
                        # takes the persistent latency form, which sums in another order), so a batch's bits do not depend on how the stream was cut
                        for i, r in zip(idx, self.model.inference_multi(grp, S=S, want_attn=want_attn)):
                            out[i] = r
            except BaseException as e:      # noqa: BLE001 - re-raised on the caller's thread
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(w,)) for w in range(n_workers)]
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

    # ------------------------------------------------------------------------------------------------ streaming form
    @staticmethod
    def _job_key(job: dict):
        mask = job.get("mask")
        return (job["entry"], tuple(job["video"].shape), int(job["S"]), bool(job.get("want_attn", False)),
                bytes(bytearray(mask)) if mask is not None else None)

    def _run_group(self, jobs: List[dict]) -> List[tuple]:
        j0 = jobs[0]
        with self._stats_lock:                                   # several worker threads run groups at once
            self.stats[j0["entry"] + "_groups"] += 1
            self.stats[j0["entry"] + "_batches"] += len(jobs)
            self.stats["max_group"] = max(self.stats["max_group"], len(jobs))
        if j0["entry"] == "inference":      # also for a one-job group: the pool's results do not depend on the group size (see map)
            return self.model.inference_multi([(j["video"], j["emb"], j["gumbel"]) for j in jobs], S=j0["S"], want_attn=j0.get("want_attn", False))
        if j0["entry"] == "forward":
            return self.model.forward_eval_multi([(j["video"], j["emb"], j["gumbel"], j.get("teacher")) for j in jobs], j0["S"], teacher_mask=j0.get("mask"))
        raise ValueError(f"unknown entry {j0['entry']!r}")

    def imap(self, items: Iterable, prepare: Callable, depth: int = 2) -> Iterator:
        """Streaming `map` for loader-driven callers (the reference's demo.py:60-90 / evaluate.py:22-51 loops take one batch per iteration):
        `items` is any iterable (a DataLoader), `prepare(item)` turns an item into a job - a dict with `entry` ("inference" =
        `Lip2Speech.inference`, "forward" = eval-mode `Lip2Speech.forward`), device tensors `video`, `emb`, `gumbel`, the step count `S`, and
        optionally `want_attn` (inference), `teacher` + `mask` (forward: scheduled sampling) and `finish(result) -> value`.  Yields one value per
        item, IN ORDER, each bit-identical to the single-batch call.

        Consecutive jobs with the same entry / shape / S / mask run `self.group` per launch chain (`l2s_inference_multi` /
        `l2s_forward_eval_multi`), `n_inflight` chains at once.  The iterator and `prepare` are only ever touched under one lock, in item order
        (host RNG draws inside `prepare` - the scheduled-sampling `torch.rand(1)`s - are consumed exactly as a sequential loop consumes them);
        `prepare` runs on the pool's copy stream, so its host-to-device copies and the voice tower travel under the chains' compute.  A worker
        keeps at most `depth` groups enqueued and the pool runs at most `depth + 1` rounds ahead of the consumer."""
        from . import native
        it = iter(items)
        caller_stream = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(caller_stream)
        if self.copy_stream is None:
            self.copy_stream = torch.cuda.Stream(device=self.device)
        self.copy_stream.wait_event(ready)
        lock = threading.Lock()
        cond = threading.Condition()
        results: Dict[int, tuple] = {}
        st = {"next": 0, "pending": None, "exhausted": False, "errors": [], "live": 0, "consumed": 0, "stop": False}
        window = self.group * self.n_inflight * (depth + 1)

        def claim():
            with lock:
                if st["stop"] or (st["exhausted"] and st["pending"] is None):
                    return None
                group = []
                with torch.cuda.stream(self.copy_stream):
                    while len(group) < self.group:
                        if st["pending"] is not None:
                            idx, job = st["pending"]
                            st["pending"] = None
                        else:
                            try:
                                item = next(it)
                            except StopIteration:
                                st["exhausted"] = True
                                break
                            job = prepare(item)
                            idx = st["next"]
                            st["next"] += 1
                        if group and self._job_key(job) != self._job_key(group[0][1]):
                            st["pending"] = (idx, job)          # another shape / S / mask closes the running group
                            break
                        group.append((idx, job))
                    ev = torch.cuda.Event()
                    ev.record(self.copy_stream)
                return (group, ev) if group else None

        def worker(w: int):
            try:
                torch.cuda.set_device(self.device)
                native.set_thread_chains(self.n_inflight)
                stream = self.streams[w]
                enq = collections.deque()
                with torch.cuda.stream(stream):
                    stream.wait_event(ready)
                    while True:
                        while len(enq) >= depth:
                            enq.popleft().synchronize()
                        with cond:                               # do not run further ahead of the consumer than `window` items
                            while not st["stop"] and st["next"] - st["consumed"] >= window:
                                cond.wait(0.05)
                        got = claim()
                        if got is None:
                            break
                        group, ev = got
                        stream.wait_event(ev)
                        for _, job in group:                     # staged on the copy stream, read on this one
                            for t in job.values():
                                if isinstance(t, torch.Tensor) and t.is_cuda:
                                    t.record_stream(stream)
                        outs = self._run_group([job for _, job in group])
                        done = torch.cuda.Event()
                        done.record(stream)
                        enq.append(done)
                        with cond:
                            for (idx, job), r in zip(group, outs):
                                results[idx] = (job, r, done)
                            cond.notify_all()
            except BaseException as e:      # noqa: BLE001 - re-raised on the consumer's thread
                with cond:
                    st["errors"].append(e)
                    cond.notify_all()
            finally:
                with cond:
                    st["live"] -= 1
                    cond.notify_all()

        threads = [threading.Thread(target=worker, args=(w,), daemon=True) for w in range(self.n_inflight)]
        st["live"] = len(threads)
        for t in threads:
            t.start()
        i = 0
        try:
            while True:
                with cond:
                    while i not in results and not st["errors"] and st["live"] > 0:
                        cond.wait(0.05)
                    if st["errors"]:
                        raise st["errors"][0]
                    if i not in results:
                        break                                    # every worker has finished: the iterable is exhausted
                    job, r, done = results.pop(i)
                caller_stream.wait_event(done)
                # results come from a worker stream, the job's own tensors (what `prepare` staged: a `finish` closure may hand them on, e.g. the
                # speaker embedding in forward_many's list of 7) from the copy stream: both are the caller's to read from here on, so their
                # blocks must not be recycled under a kernel the caller still has queued
                for t in list(r) + list(job.values()):
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(caller_stream)
                finish = job.get("finish")
                yield finish(r) if finish is not None else r
                i += 1
                with cond:
                    st["consumed"] = i
                    cond.notify_all()
        finally:
            with cond:
                st["stop"] = True
                cond.notify_all()
            for t in threads:
                t.join()
