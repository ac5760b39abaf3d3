cd /root/repo
mkdir -p gpurun_out/r6j
python tools/solve_bench.py --configs C2 > /dev/null 2>&1   # (not the first GPU process of the box for what follows)
python tools/solve_bench.py --configs C2,C3,C4,C5 > gpurun_out/r6j/solve_bench.jsonl 2> gpurun_out/r6j/solve_bench.err
bash tools/r4_step_gaps.sh C2 r6j/r6_final > gpurun_out/r6j/gaps_C2.log 2>&1
bash tools/r4_step_gaps.sh C3 r6j/r6_final > gpurun_out/r6j/gaps_C3.log 2>&1
bash tools/r4_step_gaps.sh C5 r6j/r6_final > gpurun_out/r6j/gaps_C5.log 2>&1
