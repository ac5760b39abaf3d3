#!/bin/bash
# PMC passes over tools/c5_iter_bench.py (one native mBCG solve at the C5 shape); run on the GPU box from the repo root
L=${1:-r3_iterpmc}; SHAPE=${2:-C5}; T=${3:-11}
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${L}_a -o p -- python3 $R/tools/c5_iter_bench.py $SHAPE $T 10 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC --output-format csv -d $R/gpurun_out/${L}_b -o p -- python3 $R/tools/c5_iter_bench.py $SHAPE $T 10 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${L}_c -o p -- python3 $R/tools/c5_iter_bench.py $SHAPE $T 10 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${L}_d -o p -- python3 $R/tools/c5_iter_bench.py $SHAPE $T 10 > /dev/null 2>&1
cd $R
for x in a b c d; do python3 tools/pmc_kernels.py gpurun_out/${L}_$x k_pass k_reduce ski_ ; done > gpurun_out/${L}_${SHAPE}_T${T}_summary.txt
rm -rf gpurun_out/${L}_a gpurun_out/${L}_b gpurun_out/${L}_c gpurun_out/${L}_d
cat gpurun_out/${L}_${SHAPE}_T${T}_summary.txt
