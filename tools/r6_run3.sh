cd /root/repo
mkdir -p gpurun_out/r6c
# C4 as the first GPU process of the box with a fit of realistic length (the warm-up of the factorisation runs beside training)
python tools/solve_bench.py --configs C4 --steps 40 > gpurun_out/r6c/solve_bench_C4_first_process_40_steps.jsonl 2> gpurun_out/r6c/sb1.err
python tools/solve_bench.py --configs C4 > gpurun_out/r6c/solve_bench_C4_second_process.jsonl 2> gpurun_out/r6c/sb2.err
python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r6c/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6c/pytest_gpu.txt
