#!/bin/bash
# Headline kernel measurement set (run on the GPU box from the repo root): bench, rocprofv3 kernel stats of the same
# command, three separate --pmc passes.  Summaries land in gpurun_out/<label>_*.
L=${1:-r4_hl}
R=$GRAFT_REPO_ROOT
python3 $R/bench.py > $R/gpurun_out/${L}_bench.json 2> $R/gpurun_out/${L}_bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${L}_trace -o bench -- python3 $R/bench.py --no-extras --cpu-budget 0 > $R/gpurun_out/${L}_bench_profiled.json 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/${L}_pmc_$C -o p -- python3 $R/bench.py --no-extras --cpu-budget 0 --steps 10 --warmup 2 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${L}_pmc_SQ -o p -- python3 $R/bench.py --no-extras --cpu-budget 0 --steps 10 --warmup 2 > /dev/null 2>&1
cd $R
python3 tools/collect_pmc.py gpurun_out/${L}_pmc_counters.json gpurun_out/${L}_pmc_FETCH_SIZE gpurun_out/${L}_pmc_WRITE_SIZE gpurun_out/${L}_pmc_SQ > /dev/null
find gpurun_out/${L}_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${L}_kernel_stats.csv \;
rm -rf gpurun_out/${L}_trace
cat gpurun_out/${L}_bench.json | head -c 1500; echo; cat gpurun_out/${L}_pmc_counters.json | tail -12; head -5 gpurun_out/${L}_kernel_stats.csv
