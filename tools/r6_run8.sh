cd /root/repo
mkdir -p gpurun_out/r6h
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r6h/pytest_gpu.txt 2>&1; echo "rc $?" >> gpurun_out/r6h/pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/r6h/bench.json 2> gpurun_out/r6h/bench.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6h/smoke.txt 2>&1
