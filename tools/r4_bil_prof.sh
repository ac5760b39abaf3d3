#!/bin/bash
# bilinear derivative at C4 (T = 11): kernel stats + PMC of the hand-scheduled kernel and (RPGP_BIL_ASM=0) the compiler's
L=${1:-r4_bil}
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for A in 1 0; do
  export RPGP_BIL_ASM=$A
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${L}_trace_$A -o b -- python3 $R/tools/bil_only.py 8 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${L}_pmc_$A -o p -- python3 $R/tools/bil_only.py 4 > /dev/null 2>&1
done
cd $R
for A in 1 0; do
  echo "== RPGP_BIL_ASM=$A"
  find gpurun_out/${L}_trace_$A -name "*kernel_stats.csv" -exec head -5 {} \; | cut -c1-200
  python3 tools/pmc_kernels.py gpurun_out/${L}_pmc_$A bilinear
done > gpurun_out/${L}_summary.txt 2>&1
rm -rf gpurun_out/${L}_trace_1 gpurun_out/${L}_trace_0
cat gpurun_out/${L}_summary.txt
