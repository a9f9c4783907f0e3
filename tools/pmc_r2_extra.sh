#!/bin/bash
# Round-2 PMC evidence for the split-bf16 kernels (run on the GPU box from the repo root): one rocprofv3 --pmc pass per counter set.
set -u
R=$PWD; O=$R/gpurun_out/prof_r2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
pass() { rm -rf /tmp/pmcx; timeout 300 rocprofv3 --pmc $1 -d /tmp/pmcx -o x -- python $2 > /dev/null 2>&1 < /dev/null; timeout 60 python $R/tools/pmc_read.py $(find /tmp/pmcx -name "*.db" | head -1) "$3" < /dev/null; }
{
echo "# rocprofv3 --pmc passes, one counter set per pass (MI355X).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES counts cycles;"
echo "# GRBM_GUI_ACTIVE is summed over the 8 XCDs.  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)."
echo; echo "## tools/pmc_gemm_both.py: 38400 x 512 x 2560 GEMM (post-net layer, 4 batches per chain): split-bf16 kernel and f32 MFMA kernel"
for c in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do pass "$c" $R/tools/pmc_gemm_both.py "%gemm_%"; done
echo; echo "## tools/pmc_frontend.py: front-end conv over 128 clips (4 batches per chain), split-bf16 kernel"
for c in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "FETCH_SIZE"; do pass "$c" $R/tools/pmc_frontend.py "%frontend3d%"; done
echo; echo "## the same with X3=0: f32 MFMA front-end kernel"
export X3=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do pass "$c" $R/tools/pmc_frontend.py "%frontend3d%"; done
} > $O/r02_pmc_x3_kernels.txt 2>&1
wc -l $O/r02_pmc_x3_kernels.txt; tail -12 $O/r02_pmc_x3_kernels.txt
