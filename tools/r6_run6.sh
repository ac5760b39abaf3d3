cd /root/repo
mkdir -p gpurun_out/r6f
timeout 600 python -m pytest tests/test_fact_asm_gpu.py -m gpu -q -x > gpurun_out/r6f/pytest_fact_asm.txt 2>&1; echo "rc $?" >> gpurun_out/r6f/pytest_fact_asm.txt
timeout 300 python tools/time_shards.py > gpurun_out/r6f/time_shards.txt 2>&1
RPGP_FACT_ASM=0 timeout 300 python tools/time_shards.py > gpurun_out/r6f/time_shards_compiler_kernels.txt 2>&1
