#!/bin/bash
# Round-4 optimiser-step evidence in one call: robust step times (tools/r4_step_time.py), kernel traces with idle gaps
# (tools/r4_step_gaps.sh) and the end-to-end solve bench at the BASELINE shapes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for c in C2 C3 C4 C5; do python3 $R/tools/r4_step_time.py $c 2>/dev/null | tail -1; done > $O/r4_step_times_final.jsonl
for c in C2 C3 C5; do RPGP_STEP_KERNELS=0 python3 $R/tools/r4_step_time.py $c 2>/dev/null | tail -1; done >> $O/r4_step_times_final.jsonl
for c in C2 C3 C5; do bash $R/tools/r4_step_gaps.sh $c r4_final_step > /dev/null 2>&1; done
cd $R && python3 tools/solve_bench.py --configs C2,C3,C4,C5 2>/dev/null > $O/r4_solve_bench_final.jsonl
cat $O/r4_step_times_final.jsonl; head -1 $O/r4_final_step_C*_step_gaps.txt
