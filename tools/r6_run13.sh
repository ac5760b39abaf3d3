cd /root/repo
mkdir -p gpurun_out/r6n
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_ski_gpu.py tests/test_gp_gpu.py tests/test_native_cg_gpu.py -m gpu -q -x > gpurun_out/r6n/pytest.txt 2>&1; echo "rc $?" >> gpurun_out/r6n/pytest.txt
python tools/step_only.py C5 30 > gpurun_out/r6n/step_C5.json 2>/dev/null
bash tools/r4_step_gaps.sh C5 r6n/r6_tuned > gpurun_out/r6n/gaps_C5.log 2>&1
python bench.py --steps 5 --warmup 2 --cpu-budget 0 > gpurun_out/r6n/bench_short.json 2> gpurun_out/r6n/bench.err
