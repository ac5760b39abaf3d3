cd /root/repo
mkdir -p gpurun_out/r6m
timeout 900 python -m pytest tests/test_gp_gpu.py tests/test_lib_abi.py -m gpu -q -x > gpurun_out/r6m/pytest.txt 2>&1; echo "rc $?" >> gpurun_out/r6m/pytest.txt
python bench.py --steps 5 --warmup 2 --cpu-budget 0 > gpurun_out/r6m/bench_short.json 2> gpurun_out/r6m/bench.err
python tools/step_only.py C2 40 > gpurun_out/r6m/step_C2.json 2>/dev/null
python tools/step_only.py C5 30 > gpurun_out/r6m/step_C5.json 2>/dev/null
