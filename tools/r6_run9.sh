cd /root/repo
mkdir -p gpurun_out/r6i
timeout 900 python -m pytest tests/test_family_gpu.py tests/test_double_gpu.py -m gpu -q -x > gpurun_out/r6i/pytest_family.txt 2>&1; echo "rc $?" >> gpurun_out/r6i/pytest_family.txt
