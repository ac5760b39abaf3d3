cd /root/repo
mkdir -p gpurun_out/r6r
timeout 900 python -m pytest tests/test_gp_gpu.py -m gpu -q -x -k "training_loop or eager" > gpurun_out/r6r/pytest.txt 2>&1; echo "rc $?" >> gpurun_out/r6r/pytest.txt
