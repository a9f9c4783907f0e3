import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
t = [x for x in tabs if x.startswith("counters_collection")][0]
q = f"select kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name, avg(value), count(*) from {t} group by kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name"
try:
    rows = list(c.execute(q))
except Exception as e:
    cols = [r[1] for r in c.execute(f"pragma table_info({t})")]; print(cols); sys.exit()
for r in rows:
    if r[6] >= int(sys.argv[2]) if len(sys.argv) > 2 else r[6] > 100: print(f"{r[0][:34]:34s} {r[1]}x{r[2]}x{r[3]:<3} {r[4]:30s} avg {r[5]:14.1f} (n={r[6]})")
