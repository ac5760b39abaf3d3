cd /root/repo
mkdir -p gpurun_out/r6p
python tools/step_only.py C2 40 > /dev/null 2>&1
for c in C2 C3 C5; do python tools/step_only.py $c 40 > gpurun_out/r6p/step_$c.json 2>/dev/null; done
bash tools/r4_step_gaps.sh C2 r6p/r6_loop > gpurun_out/r6p/gaps_C2.log 2>&1
bash tools/r4_step_gaps.sh C3 r6p/r6_loop > gpurun_out/r6p/gaps_C3.log 2>&1
bash tools/r4_step_gaps.sh C5 r6p/r6_loop > gpurun_out/r6p/gaps_C5.log 2>&1
