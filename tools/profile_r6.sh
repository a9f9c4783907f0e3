#!/bin/bash
# Round-6 counter evidence for the block forms the timed region of bench.py really launches (three chains in flight, half-CU forms:
# skinny_rc4h<4,2,{2,6},3>, skinny_flat<21,22,2>, step_attn<true>): rocprofv3 kernel trace + one --pmc pass per counter set over
# tools/coresident_probe.py CHAINS=3 MODE=decode FORMS=half (three decode loops of 256 rows at once on three streams).
# -> profiles/r06_kernel_stats.md (the default bench command), r06_overlap_stamps.txt, r06_kernel_stats_decode256_3chains.md, r06_kernel_stats_decode256_1chain.md, r06_concurrency_3chains.txt, r06_pmc_decode_half3.json, r06_pmc_step_sq_half3.txt
# (run on the GPU box from the repo root; everything lands under gpurun_out/prof_r6/)
set -u
R=$PWD; O=$R/gpurun_out/prof_r6; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
P="env CHAINS=3 MODE=decode FORMS=half REP=1 python $R/tools/coresident_probe.py 8"
run() { timeout 400 "$@" < /dev/null > /tmp/prof.log 2>&1 || tail -3 /tmp/prof.log; }
rm -rf /tmp/p_bench; run rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o b -- python $R/bench.py --steps 64 --warmup 16 --skip-cpu-baseline --skip-train-leg
python $R/tools/rocprof_summary.py $(find /tmp/p_bench -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py --steps 64 --warmup 16 --skip-cpu-baseline --skip-train-leg (round 6: 8 batches per launch chain, 3 chains in flight, B=32, T=29, S=300)" > $O/r06_kernel_stats.md
rm -rf /tmp/p3; run rocprofv3 --kernel-trace --stats -d /tmp/p3 -o t -- $P
TR=$(find /tmp/p3 -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $TR "rocprofv3 --kernel-trace --stats -- CHAINS=3 MODE=decode FORMS=half REP=1 python tools/coresident_probe.py 8 (three decode loops at once, 256 rows each, the half-CU block forms of bench.py's timed region)" > $O/r06_kernel_stats_decode256_3chains.md
python $R/tools/rocprof_concurrency.py $TR > $O/r06_concurrency_3chains.txt
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do n=$(echo $set | cut -d' ' -f1); rm -rf /tmp/p_$n; run rocprofv3 --pmc $set -d /tmp/p_$n -o c -- $P; done
rm -rf /tmp/p1; run rocprofv3 --kernel-trace --stats -d /tmp/p1 -o t -- env CHAINS=1 MODE=decode FORMS=half REP=1 python $R/tools/coresident_probe.py 8
python $R/tools/rocprof_summary.py $(find /tmp/p1 -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- CHAINS=1 MODE=decode FORMS=half REP=1 python tools/coresident_probe.py 8 (ONE decode loop, 256 rows, the same half-CU block forms: a launch that has the chip to itself)" > $O/r06_kernel_stats_decode256_1chain.md
FORMS=half python $R/tools/pmc_decode_json.py $TR $(find /tmp/p_TCC_HIT_sum -name "*.db" | head -1) $(find /tmp/p_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/p_WRITE_SIZE -name "*.db" | head -1) 256 $(find /tmp/p1 -name "*.db" | head -1) > $O/r06_pmc_decode_half3.json
rm -rf /tmp/p_sq; run rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/p_sq -o c -- $P
{ for pat in "%skinny_rc4h%" "%skinny_flat%" "%step_attn%"; do python $R/tools/pmc_read.py $(find /tmp/p_sq -name "*.db" | head -1) "$pat"; done; } > $O/r06_pmc_step_sq_half3.txt
(cd $R && CHAINS=1,3 python tools/overlap_stamps.py 8 2>&1 | grep -v amdgpu.ids) > $O/r06_overlap_stamps.txt
head -14 $O/r06_kernel_stats.md; head -14 $O/r06_kernel_stats_decode256_3chains.md; cat $O/r06_overlap_stamps.txt; cat $O/r06_concurrency_3chains.txt; grep -E "l2_hit|traffic_bytes|avg_us|step_|symbols|skinny|attn" $O/r06_pmc_decode_half3.json; cat $O/r06_pmc_step_sq_half3.txt
