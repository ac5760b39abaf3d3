cd /root/repo
mkdir -p gpurun_out/r6g
timeout 600 python -m pytest tests/test_fact_asm_gpu.py -m gpu -q -x > gpurun_out/r6g/pytest_fact_asm.txt 2>&1; echo "rc $?" >> gpurun_out/r6g/pytest_fact_asm.txt
timeout 300 python tools/time_shards.py > gpurun_out/r6g/time_shards.txt 2>&1
