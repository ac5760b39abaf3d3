#!/bin/bash
# Round-3 measurement set (run on the GPU box from the repo root).  Summaries land in gpurun_out/<label>_*; the ones to be
# judged are copied into profiles/ afterwards.
L=${1:-r3_final}
R=$GRAFT_REPO_ROOT
python3 $R/bench.py > $R/gpurun_out/${L}_bench.json 2> $R/gpurun_out/${L}_bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${L}_trace -o bench -- python3 $R/bench.py --no-extras --cpu-budget 0 > $R/gpurun_out/${L}_bench_profiled.json 2>/dev/null
cp $R/gpurun_out/${L}_trace/bench_kernel_stats.csv $R/gpurun_out/${L}_bench_kernel_stats.csv; rm -rf $R/gpurun_out/${L}_trace
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/${L}_pmc_$C -o p -- python3 $R/bench.py --no-extras --cpu-budget 0 --steps 10 --warmup 2 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${L}_pmc_SQ -o p -- python3 $R/bench.py --no-extras --cpu-budget 0 --steps 10 --warmup 2 > /dev/null 2>&1
cd $R
python3 tools/collect_pmc.py gpurun_out/${L}_pmc_counters.json gpurun_out/${L}_pmc_FETCH_SIZE gpurun_out/${L}_pmc_WRITE_SIZE gpurun_out/${L}_pmc_SQ > /dev/null
rm -rf gpurun_out/${L}_pmc_FETCH_SIZE gpurun_out/${L}_pmc_WRITE_SIZE gpurun_out/${L}_pmc_SQ
# bilinear derivative: times + PMC of the symmetric kernel
python3 tools/time_bilinear.py > gpurun_out/${L}_bilinear.txt 2>/dev/null
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${L}_pmc_bil -o p -- python3 $R/tools/time_bilinear.py > /dev/null 2>&1
cd $R
python3 tools/pmc_kernels.py gpurun_out/${L}_pmc_bil bilinear_sym > gpurun_out/${L}_bilinear_pmc.txt; rm -rf gpurun_out/${L}_pmc_bil
# end-to-end solves + kernel stats of the C5 / C4 training loops
bash tools/r3_prof_solve.sh ${L} > /dev/null 2>&1
# SKI product: bench + per-kernel times + PMC
bash tools/r3_prof_ski.sh ${L}_ski > /dev/null 2>&1
bash tools/r3_pmc_ski.sh ${L}_skipmc > /dev/null 2>&1
# one mBCG iteration at C5 (T = 11 and T = 1) and C4 (cached)
for T in 11 1; do bash tools/r3_prof_iter.sh ${L}_iter C5 $T > /dev/null 2>&1; done
bash tools/r3_prof_iter.sh ${L}_iter C4cache 11 > /dev/null 2>&1
bash tools/r3_pmc_iter.sh ${L}_iterpmc C5 11 > /dev/null 2>&1
cat gpurun_out/${L}_bench.json
