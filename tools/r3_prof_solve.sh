#!/bin/bash
# solve bench (all configs) + rocprofv3 kernel stats of the C5 and C4 training loops; run on the GPU box from the repo root
L=${1:-r3_exec}
R=$GRAFT_REPO_ROOT
python3 $R/tools/solve_bench.py --configs C2,C3,C4,C5,C3S,C4S > $R/gpurun_out/${L}_solve.log 2>&1; grep "^{" $R/gpurun_out/${L}_solve.log > $R/gpurun_out/${L}_solve.jsonl
cd /tmp; export TMPDIR=/tmp
for C in C5 C4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${L}_trace_$C -o t -- python3 $R/tools/solve_bench.py --configs $C > /dev/null 2>&1
  cp $R/gpurun_out/${L}_trace_$C/t_kernel_stats.csv $R/gpurun_out/${L}_${C}_train_kernel_stats.csv 2>/dev/null
  rm -rf $R/gpurun_out/${L}_trace_$C
done
cd $R
head -25 gpurun_out/${L}_C5_train_kernel_stats.csv
cat gpurun_out/${L}_solve.jsonl
