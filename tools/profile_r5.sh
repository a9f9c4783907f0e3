#!/bin/bash
# Round-5 profiles (run on the GPU box from the repo root): rocprofv3 kernel-trace summary of the default bench command, kernel-trace + PMC
# passes of the decode loop at 256 rows per launch.  Everything lands under gpurun_out/prof_r5/.
set -u
R=$PWD; O=$R/gpurun_out/prof_r5; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { timeout 500 "$@" < /dev/null > /tmp/prof.log 2>&1 || tail -3 /tmp/prof.log; }
rm -rf /tmp/p_bench; run rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o b -- python $R/bench.py --steps 64 --warmup 16 --skip-cpu-baseline
python $R/tools/rocprof_summary.py $(find /tmp/p_bench -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py --steps 64 --warmup 16 --skip-cpu-baseline (round 5: 8 batches per launch chain, 3 chains in flight, B=32, T=29, S=300)" > $O/r05_kernel_stats.md
rm -rf /tmp/p_tr; run rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o t -- python $R/tools/prof_decode.py
python $R/tools/rocprof_summary.py $(find /tmp/p_tr -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python tools/prof_decode.py (decode loop only, 256 rows per launch, 3 x 300 steps)" > $O/r05_kernel_stats_decode256.md
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do n=$(echo $set | cut -d' ' -f1); rm -rf /tmp/p_$n; run rocprofv3 --pmc $set -d /tmp/p_$n -o c -- python $R/tools/prof_decode.py; done
python $R/tools/pmc_decode_json.py $(find /tmp/p_tr -name "*.db" | head -1) $(find /tmp/p_TCC_HIT_sum -name "*.db" | head -1) $(find /tmp/p_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/p_WRITE_SIZE -name "*.db" | head -1) 256 > $O/r05_pmc_decode.json
head -12 $O/r05_kernel_stats.md; cat $O/r05_pmc_decode.json | grep -E "l2_hit|traffic_bytes|avg_us|step_" 
