set -u
R=$PWD; O=$R/gpurun_out/prof_r5; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p3; CHAINS=3 MODE=decode FORMS=half REP=2 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p3 -o t -- python $R/tools/coresident_probe.py 8 > /tmp/p3.log 2>&1; tail -2 /tmp/p3.log
python $R/tools/rocprof_summary.py $(find /tmp/p3 -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- CHAINS=3 MODE=decode FORMS=half python tools/coresident_probe.py 8 (three decode loops at once, 256 rows each, half-CU block forms)" > $O/r05_kernel_stats_decode256_3chains.md
python $R/tools/rocprof_concurrency.py $(find /tmp/p3 -name "*.db" | head -1) > $O/r05_concurrency_3chains.txt; cat $O/r05_concurrency_3chains.txt
rm -rf /tmp/p1; CHAINS=1 MODE=decode FORMS=half REP=2 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o t -- python $R/tools/coresident_probe.py 8 > /tmp/p1.log 2>&1; tail -2 /tmp/p1.log
python $R/tools/rocprof_summary.py $(find /tmp/p1 -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- CHAINS=1 MODE=decode FORMS=half python tools/coresident_probe.py 8 (one decode loop, 256 rows)" > $O/r05_kernel_stats_decode256_1chain.md

