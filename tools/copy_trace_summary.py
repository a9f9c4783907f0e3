"""Summarise a rocprofv3 --memory-copy-trace / --hip-trace CSV pair: copies by (direction, bytes) and HIP API call counts."""
import csv, collections, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(f"{d}_memory_copy_trace.csv")))
print(len(rows), "copies; columns:", list(rows[0].keys()))
c = collections.Counter()
for r in rows:
    size = next((int(r[k]) for k in r if k.lower() in ("bytes", "size")), -1)
    direction = next((r[k] for k in r if k.lower() == "direction"), "?")
    c[(direction, size)] += 1
for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:20]:
    print(f"{v:6d} x {k}")
for line in open(f"{d}_hip_api_stats.csv").read().splitlines()[:14]:
    print(line)
