#!/bin/bash
# L2 hits / misses of the attention launch alone, back to back, at 32..256 clips per launch (rocprofv3 --pmc, one pass per row count)
R=$PWD; cd /tmp; export TMPDIR=/tmp
for rows in 32 64 128 192 256; do
  rm -rf /tmp/p_l2; ROWS=$rows timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/p_l2 -o c -- python $R/tools/attn_l2_probe.py > /tmp/pmc.log 2>&1 || tail -3 /tmp/pmc.log
  grep "rows:" /tmp/pmc.log
  python $R/tools/pmc_read.py $(find /tmp/p_l2 -name "*.db" | head -1) "%step_attn_kernel%"
done
