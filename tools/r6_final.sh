#!/bin/bash
# Round-6 closing measurement set: full GPU suite, bench, rocprofv3 kernel stats of the same command, the three --pmc passes of
# the headline kernel (separate runs, no trace domains), smoke.  Summaries land in gpurun_out/r6z/.
R=$GRAFT_REPO_ROOT
L=r6z/r6_final
mkdir -p $R/gpurun_out/r6z
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q --durations=15 > gpurun_out/r6z/pytest_gpu.txt 2>&1; echo "rc $?" >> gpurun_out/r6z/pytest_gpu.txt
python3 $R/bench.py > $R/gpurun_out/${L}_bench_before_pmc.json 2> $R/gpurun_out/${L}_bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${L}_trace -o bench -- python3 $R/bench.py --no-extras --cpu-budget 0 > $R/gpurun_out/${L}_bench_profiled.json 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/${L}_pmc_$C -o p -- python3 $R/bench.py --no-extras --cpu-budget 0 --steps 10 --warmup 2 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${L}_pmc_SQ -o p -- python3 $R/bench.py --no-extras --cpu-budget 0 --steps 10 --warmup 2 > /dev/null 2>&1
cd $R
python3 tools/collect_pmc.py gpurun_out/${L}_pmc_counters.json gpurun_out/${L}_pmc_FETCH_SIZE gpurun_out/${L}_pmc_WRITE_SIZE gpurun_out/${L}_pmc_SQ > /dev/null
cp gpurun_out/${L}_pmc_counters.json profiles/pmc_counters_current.json
cp gpurun_out/${L}_trace/*/bench_kernel_stats.csv gpurun_out/${L}_kernel_stats.csv 2>/dev/null || cp $(find gpurun_out/${L}_trace -name "*kernel_stats.csv" | head -1) gpurun_out/${L}_kernel_stats.csv
rm -rf gpurun_out/${L}_pmc_FETCH_SIZE gpurun_out/${L}_pmc_WRITE_SIZE gpurun_out/${L}_pmc_SQ gpurun_out/${L}_trace
python3 $R/bench.py > $R/gpurun_out/${L}_bench.json 2>> $R/gpurun_out/${L}_bench.err
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6z/smoke.txt 2>&1
