cd /root/repo
mkdir -p gpurun_out/r6b
# cold-start first: the C4 prediction as the FIRST GPU process of the box
python tools/solve_bench.py --configs C4 > gpurun_out/r6b/solve_bench_C4_first_process.jsonl 2> gpurun_out/r6b/solve_bench_C4_first.err
python tools/solve_bench.py --configs C2,C3,C4,C5 > gpurun_out/r6b/solve_bench.jsonl 2> gpurun_out/r6b/solve_bench.err
python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r6b/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6b/pytest_gpu.txt
python bench.py > gpurun_out/r6b/bench.json 2> gpurun_out/r6b/bench.err
bash tools/r6_marker_trace.sh C3 r6b/r6_step 20 > gpurun_out/r6b/marker_C3.log 2>&1
bash tools/r4_step_gaps.sh C5 r6b/r6_step > gpurun_out/r6b/gaps_C5.log 2>&1
bash tools/r4_step_gaps.sh C2 r6b/r6_step > gpurun_out/r6b/gaps_C2.log 2>&1
