#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database: per kernel and per launch shape (name, grid) statistics -> markdown.
-> profiles/rNN_kernel_stats*.md"""
import sqlite3, sys
db, title = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
c = sqlite3.connect(db)
print(f"# {title}\n\n| kernel | grid (threads) | calls | avg us | min us | max us | total ms |\n|---|---|---|---|---|---|---|")
q = """select name, grid_x, grid_y, grid_z, count(*), avg(duration), min(duration), max(duration), sum(duration) from kernels
       group by name, grid_x, grid_y, grid_z order by sum(duration) desc limit 40"""
for r in c.execute(q):
    print(f"| `{r[0][:70]}` | {r[1]}x{r[2]}x{r[3]} | {r[4]} | {r[5]/1e3:.1f} | {r[6]/1e3:.1f} | {r[7]/1e3:.1f} | {r[8]/1e6:.3f} |")
