#!/bin/bash
# Round-end measurement set (run on the GPU box from the repo root): bench, rocprofv3 kernel stats of the same command,
# three separate --pmc passes for the headline kernel, solve bench, SKI bench.  Summaries land in gpurun_out/<label>_*.
L=${1:-r2_final}
R=$GRAFT_REPO_ROOT
python3 $R/bench.py > $R/gpurun_out/${L}_bench.json 2> $R/gpurun_out/${L}_bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${L}_trace -o bench -- python3 $R/bench.py --no-extras --cpu-budget 0 > $R/gpurun_out/${L}_bench_profiled.json 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/${L}_pmc_$C -o p -- python3 $R/bench.py --no-extras --cpu-budget 0 --steps 10 --warmup 2 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${L}_pmc_SQ -o p -- python3 $R/bench.py --no-extras --cpu-budget 0 --steps 10 --warmup 2 > /dev/null 2>&1
# backward (bilinear derivative) and T = 11 block under the profiler
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${L}_trace_bil -o bil -- python3 $R/tools/time_bilinear.py > $R/gpurun_out/${L}_bilinear.txt 2>/dev/null
cd $R
python3 tools/collect_pmc.py gpurun_out/${L}_pmc_counters.json gpurun_out/${L}_pmc_FETCH_SIZE gpurun_out/${L}_pmc_WRITE_SIZE gpurun_out/${L}_pmc_SQ > /dev/null
python3 tools/solve_bench.py --configs C2,C3,C4,C5,C3S,C4S > gpurun_out/${L}_solve.log 2>&1; grep "^{" gpurun_out/${L}_solve.log > gpurun_out/${L}_solve.jsonl
python3 tools/ski_bench.py > gpurun_out/${L}_ski_bench.jsonl 2>/dev/null
# packed symmetric cache: both layouts, sizes of the BASELINE configs, plus kernel stats of the wide T = 11 product
python3 tools/symk_check.py 7372 14939 30000 50000 > gpurun_out/${L}_symcache_thin.txt 2>&1
SYMK_WIDE=1 python3 tools/symk_check.py 7372 14939 30000 50000 > gpurun_out/${L}_symcache_wide.txt 2>&1
python3 tools/time_dense.py > gpurun_out/${L}_dense_stream.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${L}_trace_symk -o symk -- python3 $R/tools/symk_only.py 50000 11 wide > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/${L}_pmc_symk -o p -- python3 $R/tools/symk_only.py 50000 11 wide > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/${L}_pmc_symk_$C -o p -- python3 $R/tools/symk_only.py 50000 11 wide > /dev/null 2>&1
done
cd $R
cat gpurun_out/${L}_bench.json
