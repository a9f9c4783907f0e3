#!/usr/bin/env python
"""Timeline of ONE training step out of a rocprofv3 --kernel-trace database of tools/time_train.py (or bench.py --mode train):
span of the step, sum of kernel durations, idle time between kernels, and per kernel name: launches, busy time, and the idle
time that FOLLOWS its launches (the dependency / launch gap it leaves) - the figure the step's wall time is made of.
usage: train_timeline.py <db> [step_index_from_end=1]"""
import sqlite3, sys, collections
db = sys.argv[1]; back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
# a step starts at the forward front-end kernel (first launch of the step); statistics passes of train mode come first
starts = [i for i, r in enumerate(rows) if "frontend" in r[0] and "bwd" not in r[0] and "dw" not in r[0]]
# keep the first front-end launch of each step (launches of one step sit within a few ms)
firsts = [starts[0]]
for i in starts[1:]:
    if rows[i][1] - rows[firsts[-1]][1] > 8e6: firsts.append(i)
lo = firsts[-1 - back]; hi = firsts[-back] if back > 0 else len(rows)
step = rows[lo:hi]
span = (step[-1][2] - step[0][1]) / 1e6
busy = collections.defaultdict(float); gap = collections.defaultdict(float); cnt = collections.Counter()
tb = tg = 0.0; last_end = step[0][1]
for k, (name, s, e) in enumerate(step):
    short = name.split("(")[0].replace("void ", "").replace("l2s::", "")
    d = (e - max(s, last_end)) / 1e3 if e > last_end else 0.0
    g = max(0.0, (s - last_end) / 1e3)
    busy[short] += d; cnt[short] += 1; tb += d
    if k: gap[prev] += g; tg += g
    prev = short; last_end = max(last_end, e)
print(f"step of {len(step)} launches: span {span:.2f} ms, kernels busy {tb/1e3:.2f} ms, idle between kernels {tg/1e3:.2f} ms")
print(f"{'kernel':48s} {'launches':>8s} {'busy ms':>8s} {'idle after ms':>13s} {'avg us':>7s} {'gap us':>7s}")
for n in sorted(busy, key=lambda n: -(busy[n] + gap[n]))[:40]:
    print(f"{n[:48]:48s} {cnt[n]:8d} {busy[n]/1e3:8.3f} {gap[n]/1e3:13.3f} {busy[n]/cnt[n]:7.1f} {gap[n]/cnt[n]:7.1f}")
