#!/bin/bash
# SQ counters of the decode-step kernels at 256 rows (rocprofv3 --pmc, one pass): matrix-pipe busy cycles, wave cycles and what the waves wait on.
# usage: bash tools/pmc_step_kernels.sh [ENV=VALUE ...]   (e.g. L2S_OPT=skinny_rc_jb=2)
R=$PWD; cd /tmp; export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/p_sq; env "$@" timeout 300 rocprofv3 --pmc $set -d /tmp/p_sq -o c -- python $R/tools/prof_decode.py > /tmp/pmc.log 2>&1 || tail -3 /tmp/pmc.log
  python $R/tools/pmc_read.py $(find /tmp/p_sq -name "*.db" | head -1) "%skinny_rc%4, 2, %"
  python $R/tools/pmc_read.py $(find /tmp/p_sq -name "*.db" | head -1) "%skinny_flat%"
done
