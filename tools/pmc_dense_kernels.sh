#!/bin/bash
# SQ / L2 counters of the dense kernels of a 256-clip pass (rocprofv3 --pmc, one counter set per pass; tools/pmc_dense.py), then per kernel:
# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), LDS busy = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES-equivalent, L2 hit rate.
# usage: bash tools/pmc_dense_kernels.sh > gpurun_out/pmc_dense.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE"; do
  rm -rf /tmp/p_dn; timeout 400 rocprofv3 --pmc $set -d /tmp/p_dn -o c -- python $R/tools/pmc_dense.py > /tmp/pmc.log 2>&1 || tail -3 /tmp/pmc.log
  db=$(find /tmp/p_dn -name "*.db" | head -1)
  for k in "%frontend3d%" "%gemm_x3w%" "%gemm_x3_kernel%" "%shuffle_s1%" "%shuffle_s2%"; do python $R/tools/pmc_read.py $db "$k"; done
done
rm -rf /tmp/p_dn; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p_dn -o c -- python $R/tools/pmc_dense.py > /tmp/pmc.log 2>&1 || tail -3 /tmp/pmc.log
python $R/tools/rocprof_summary.py $(find /tmp/p_dn -name "*.db" | head -1) 2>/dev/null | head -30 || find /tmp/p_dn -name "*stats*" | head
