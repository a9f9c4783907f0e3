#!/usr/bin/env python
"""How many kernels of a rocprofv3 kernel trace run at once: share of the traced window with 0 / 1 / 2 / 3+ kernels in flight, per-kernel mean
concurrency while it runs, and the window's length per step launch.  Input: a rocpd database of `rocprofv3 --kernel-trace`, optionally a substring
that selects the window (first to last kernel whose name contains it).
-> profiles/rNN_concurrency_*.txt"""
import sqlite3, sys
import numpy as np
db = sys.argv[1]
sel = sys.argv[2] if len(sys.argv) > 2 else "skinny_rc"
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
names = np.array([r[0] for r in rows]); st = np.array([r[1] for r in rows], dtype=np.int64); en = np.array([r[2] for r in rows], dtype=np.int64)
m = np.array([sel in n for n in names])
# the LAST contiguous run of the window kernels (the timed repetition), bounded by dense kernels
idx = np.flatnonzero(m)
lo, hi = st[idx[len(idx) // 2]], en[idx[-1]]          # second half of the selected launches: steady state
w = (en > lo) & (st < hi)
ev = np.concatenate([np.stack([np.maximum(st[w], lo), np.ones(w.sum(), dtype=np.int64)], 1), np.stack([np.minimum(en[w], hi), -np.ones(w.sum(), dtype=np.int64)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
t = ev[:, 0]; k = np.cumsum(ev[:, 1])
dt = np.diff(t); kk = k[:-1]
tot = dt.sum()
print(f"window {tot / 1e3:.1f} us, {int(w.sum())} kernels, sum of durations {(np.minimum(en[w], hi) - np.maximum(st[w], lo)).sum() / 1e3:.1f} us = {(np.minimum(en[w], hi) - np.maximum(st[w], lo)).sum() / tot:.2f} kernels in flight on average")
for n in range(0, 6):
    s = dt[kk == n].sum()
    if s: print(f"  {n} kernel(s) in flight: {100.0 * s / tot:5.1f} % of the window")
print("per kernel name inside the window: launches, mean duration us, window us per launch")
for nm in sorted(set(names[w])):
    q = w & (names == nm)
    d = (en[q] - st[q]) / 1e3
    print(f"  {nm[:80]:80s} {int(q.sum()):6d} {d.mean():7.2f} {tot / 1e3 / q.sum():7.2f}")
