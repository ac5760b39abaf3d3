cd /root/repo
mkdir -p gpurun_out/r6a
python -m pytest tests -m gpu -x -q --durations=40 > gpurun_out/r6a/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6a/pytest_gpu.txt
python bench.py > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err
python tools/r6_c4_pred_first.py > gpurun_out/r6a/c4_pred_blocked.json 2> gpurun_out/r6a/c4_pred_blocked.err
RPGP_BLOCKED_CHOL=0 python tools/r6_c4_pred_first.py > gpurun_out/r6a/c4_pred_library.json 2> gpurun_out/r6a/c4_pred_library.err
python tools/solve_bench.py --configs C2,C3,C4,C5 > gpurun_out/r6a/solve_bench.jsonl 2> gpurun_out/r6a/solve_bench.err
python tools/time_shards.py > gpurun_out/r6a/time_shards.txt 2>&1
