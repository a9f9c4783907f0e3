"""Print per-kernel averages of the counters in a rocprofv3 rocpd database.
-> the counter rows of profiles/rNN_pmc_dense_kernels.txt / rNN_pmc_step_kernels.txt"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
pv = [t for t in tabs if t.startswith("counters_collection") or t == "counters_collection"]
t = pv[0] if pv else None
if t is None: print("tables:", tabs); sys.exit(0)
cols = [r[1] for r in c.execute(f"pragma table_info({t})")]
q = f"select kernel_name, counter_name, avg(value), count(*) from {t} where kernel_name like ? group by kernel_name, counter_name"
for r in c.execute(q, (sys.argv[2] if len(sys.argv) > 2 else "%gemm_nt%",)):
    print(f"{r[0][:50]:50s} {r[1]:32s} avg {r[2]:16.1f} (n={r[3]})")
