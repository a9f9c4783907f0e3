#!/usr/bin/env python
"""Where does the driver's short run (`bench.py --steps 20 --warmup 5`: 7 + 7 + 6 batches on three chains) spend its 57 ms?  Two modes:
  run (default):   the bench's pool and grouping, REP bursts of the K-step map separated by 0.4 s of idle - under `rocprofv3 --kernel-trace`
  analyse <db>:    the LAST burst of the rocpd kernel trace: span, share of it with 0 / 1 / 2 / 3+ kernels in flight, busy time per kernel class (dense
                   kernels / decode-step kernels / small glue), and a coarse timeline (per 2 ms: mean kernels in flight, dense and step share)
-> profiles/r06_steps20_timeline.txt"""
import os, sys
if len(sys.argv) > 1 and sys.argv[1] == "analyse":
    import sqlite3
    import numpy as np
    c = sqlite3.connect(sys.argv[2])
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    nm = np.array([r[0] for r in rows]); st = np.array([r[1] for r in rows], dtype=np.int64); en = np.array([r[2] for r in rows], dtype=np.int64)
    gaps = np.flatnonzero(st[1:] - np.maximum.accumulate(en)[:-1] > 150_000_000) + 1          # > 0.15 s without a kernel: between bursts
    lo = gaps[-1] if len(gaps) else 0
    nm, st, en = nm[lo:], st[lo:], en[lo:]
    t0, t1 = st.min(), en.max()
    span = (t1 - t0) / 1e6
    cls = np.array(["step" if ("skinny" in n or "step_attn" in n) else "dense" if any(k in n for k in ("frontend3d", "shuffle_s", "gemm_x3", "gemm_nt", "gemm_")) else "glue" for n in nm])
    ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    t = ev[:, 0]; k = np.cumsum(ev[:, 1]); dt = np.diff(t); kk = k[:-1]
    print(f"last burst: {len(nm)} kernels, span {span:.2f} ms, sum of kernel durations {(en - st).sum() / 1e6:.2f} ms = {(en - st).sum() / (t1 - t0):.2f} in flight on average")
    for n in range(0, 5):
        s = dt[kk == n].sum() if n < 4 else dt[kk >= 4].sum()
        print(f"  {n}{'+' if n == 4 else ' '} kernel(s) in flight: {100.0 * s / (t1 - t0):5.1f} % of the span ({s / 1e6:6.2f} ms)")
    for cl in ("dense", "step", "glue"):
        m = cls == cl
        print(f"  {cl:5s}: {int(m.sum()):6d} launches, sum of durations {(en[m] - st[m]).sum() / 1e6:7.2f} ms")
    print("timeline (2-ms bins): mean kernels in flight | dense busy share | step busy share")
    nb = int(np.ceil((t1 - t0) / 2e6))
    for b in range(nb):
        a, e = t0 + b * 2_000_000, min(t0 + (b + 1) * 2_000_000, t1)
        def busy(mask):
            s_, e_ = np.clip(st[mask], a, e), np.clip(en[mask], a, e)
            return (e_ - s_).clip(min=0).sum() / (e - a)
        print(f"  {2 * b:5.0f} ms: {busy(np.ones(len(nm), bool)):4.2f} | {busy(cls == 'dense'):4.2f} | {busy(cls == 'step'):4.2f}")
    sys.exit(0)

import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_amd import native, synth
from lip2speech_amd.parallel import InflightPool
K = int(os.environ.get("K", "20")); G = 8; REP = int(os.environ.get("REP", "3"))
B, T, S = 32, 29, 300
sd = synth.synth_state_dict()
NI = InflightPool.chains_for(K, G)
if os.environ.get("GROUPS"):      # experiment: cut the K steps into these group sizes instead of the pool's balanced groups (e.g. GROUPS=8,8,4)
    sizes = [int(x) for x in os.environ["GROUPS"].split(",")]
    assert sum(sizes) == K
    def cut(run, group, chains):
        out, pos = [], 0
        for n in sizes:
            out.append(run[pos:pos + n]); pos += n
        return out if len(run) == K else InflightPool._balanced(run, group, chains)
    InflightPool._balanced = staticmethod(InflightPool.balanced_groups)
    InflightPool.balanced_groups = staticmethod(cut)
    NI = int(os.environ.get("NI", NI))
pool = InflightPool({k: v.cuda() for k, v in sd.items()}, list(sd.keys()), n_inflight=NI, group=G)
batches = [(synth.synth_video(B, T, tag=f"b{i}").cuda(), synth.synth_speaker_embedding(B, tag=f"b{i}").cuda(), synth.synth_gumbel(B * 4, tag=f"b{i}").cuda()) for i in range(G * NI)]
work = lambda n: [batches[i % len(batches)] for i in range(n)]      # noqa: E731
pool.map(work(G * NI), S=S); pool.map(work(K), S=S)
for _ in range(REP):
    torch.cuda.synchronize(); time.sleep(0.4)
    t0 = time.perf_counter()
    pool.map(work(K), S=S)
    torch.cuda.synchronize()
    print(f"K={K} steps on {NI} chains: {(time.perf_counter() - t0) * 1e3:.2f} ms = {(time.perf_counter() - t0) * 1e3 / K:.3f} ms per batch")
