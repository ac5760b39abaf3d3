#!/bin/bash
# PMC passes over tools/ski_bench.py (C5 + C4S shapes, planned SKI product); run on the GPU box from the repo root
L=${1:-r3_skipmc}
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
SKI_REPS=5 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${L}_a -o p -- python3 $R/tools/ski_bench.py > /dev/null 2>&1
SKI_REPS=5 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/${L}_b -o p -- python3 $R/tools/ski_bench.py > /dev/null 2>&1
SKI_REPS=5 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${L}_c -o p -- python3 $R/tools/ski_bench.py > /dev/null 2>&1
SKI_REPS=5 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${L}_d -o p -- python3 $R/tools/ski_bench.py > /dev/null 2>&1
cd $R
for x in a b c d; do python3 tools/pmc_kernels.py gpurun_out/${L}_$x ski_ ; done > gpurun_out/${L}_summary.txt
rm -rf gpurun_out/${L}_a gpurun_out/${L}_b gpurun_out/${L}_c gpurun_out/${L}_d
cat gpurun_out/${L}_summary.txt
