cd /root/repo
mkdir -p gpurun_out/r6e
python -m pytest tests -m gpu -q --durations=90 > gpurun_out/r6e/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6e/pytest_gpu.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r6e/trace -o bench -- python3 /root/repo/bench.py --no-extras --cpu-budget 0 > /root/repo/gpurun_out/r6e/bench_profiled.json 2>/dev/null
cd /root/repo
find gpurun_out/r6e/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r6e/kernel_stats.csv \;
rm -rf gpurun_out/r6e/trace
