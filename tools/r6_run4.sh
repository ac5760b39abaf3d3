cd /root/repo
mkdir -p gpurun_out/r6d
python tools/experiments/r6_ski_padded_rows_ab.py > gpurun_out/r6d/ski_padded_rows_ab.json 2> gpurun_out/r6d/ski_padded.err
python -m pytest tests/test_ski_gpu.py -m gpu -q -k "other_sub_kernels or another_sub_kernel" > gpurun_out/r6d/pytest_ski_kinds.txt 2>&1; echo "rc $?" >> gpurun_out/r6d/pytest_ski_kinds.txt
bash tools/r4_headline_prof.sh r6d/r6_headline > gpurun_out/r6d/headline_prof.log 2>&1
