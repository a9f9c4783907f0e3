#!/bin/bash
# Round-3 profiles (run on the GPU box from the repo root): rocprofv3 kernel-trace summary of the default bench command, kernel-trace + PMC
# passes of the decode loop at 256 rows per launch, and the attention-alone L2 probe.  Everything lands under gpurun_out/prof_r3/.
set -u
R=$PWD; O=$R/gpurun_out/prof_r3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { timeout 500 "$@" < /dev/null > /tmp/prof.log 2>&1 || tail -3 /tmp/prof.log; }
rm -rf /tmp/p_bench; run rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o b -- python $R/bench.py --steps 64 --warmup 16 --skip-cpu-baseline
python $R/tools/rocprof_summary.py $(find /tmp/p_bench -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py --steps 64 --warmup 16 --skip-cpu-baseline (round 3: 8 batches per launch chain, 2 chains in flight, B=32, T=29, S=300)" > $O/r03_kernel_stats.md
rm -rf /tmp/p_tr; run rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o t -- python $R/tools/prof_decode.py
python $R/tools/rocprof_summary.py $(find /tmp/p_tr -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python tools/prof_decode.py (decode loop only, 256 rows per launch, 3 x 300 steps)" > $O/r03_kernel_stats_decode256.md
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do n=$(echo $set | cut -d' ' -f1); rm -rf /tmp/p_$n; run rocprofv3 --pmc $set -d /tmp/p_$n -o c -- python $R/tools/prof_decode.py; done
python $R/tools/pmc_decode_json.py $(find /tmp/p_tr -name "*.db" | head -1) $(find /tmp/p_TCC_HIT_sum -name "*.db" | head -1) $(find /tmp/p_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/p_WRITE_SIZE -name "*.db" | head -1) 256 > $O/r03_pmc_decode.json
# attention kernel alone, back to back: L2 hit rate without the weight streams in between
rm -rf /tmp/p_attn; run rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/p_attn -o a -- python $R/tools/attn_l2_probe.py
python - <<PY > $O/r03_attn_l2_probe.txt
import sqlite3, glob
db = glob.glob("/tmp/p_attn/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
t = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')") if r[0].startswith("counters_collection")][0]
rows = list(c.execute(f"select kernel_name, counter_name, avg(value), count(*) from {t} where kernel_name like '%step_attn%' group by 1,2"))
d = {r[1]: r[2] for r in rows}
print("rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -- python tools/attn_l2_probe.py  (300 back-to-back launches of step_attn_kernel alone, 256 rows)")
for r in rows: print(f"  {r[0][:50]:50s} {r[1]:14s} avg {r[2]:12.1f} over {r[3]} launches")
print(f"  L2 hit rate with nothing between the launches: {d['TCC_HIT_sum'] / (d['TCC_HIT_sum'] + d['TCC_MISS_sum']):.3f}")
PY
python $R/tools/attn_l2_probe.py 2>/dev/null | tail -1 >> $O/r03_attn_l2_probe.txt
head -12 $O/r03_kernel_stats.md; cat $O/r03_pmc_decode.json | grep -E "l2_hit|traffic_bytes|avg_us|step_" ; cat $O/r03_attn_l2_probe.txt
