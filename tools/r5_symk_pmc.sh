#!/bin/bash
# PMC passes over the wide (T = 11) and thin (T = 1) packed-cache products at the C4 shape (tools/r5_symk_pmc_run.py); run on
# the GPU box from the repo root.   -> gpurun_out/<tag>_symk_pmc_summary.txt
L=${1:-r5b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
P="python3 $R/tools/r5_symk_pmc_run.py"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${L}_sa -o p -- $P > /dev/null 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC --output-format csv -d $R/gpurun_out/${L}_sb -o p -- $P > /dev/null 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${L}_sc -o p -- $P > /dev/null 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${L}_sd -o p -- $P > /dev/null 2>&1 < /dev/null
cd $R
for x in a b c d; do python3 tools/pmc_kernels.py gpurun_out/${L}_s$x symk_mvm mvm_reduce ; done > gpurun_out/${L}_symk_pmc_summary.txt
rm -rf gpurun_out/${L}_sa gpurun_out/${L}_sb gpurun_out/${L}_sc gpurun_out/${L}_sd
cat gpurun_out/${L}_symk_pmc_summary.txt
