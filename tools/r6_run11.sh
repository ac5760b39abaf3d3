cd /root/repo
mkdir -p gpurun_out/r6k
timeout 900 python -m pytest tests/test_ski_gpu.py tests/test_lib_abi.py tests/test_kernels_gpu.py -m gpu -q -x -k "pivoted or abi or ski" > gpurun_out/r6k/pytest.txt 2>&1; echo "rc $?" >> gpurun_out/r6k/pytest.txt
python tools/step_only.py C5 30 > gpurun_out/r6k/step_C5.json 2>/dev/null
python tools/step_only.py C5 30 > gpurun_out/r6k/step_C5_b.json 2>/dev/null
bash tools/r4_step_gaps.sh C5 r6k/r6_colmajor > gpurun_out/r6k/gaps_C5.log 2>&1
