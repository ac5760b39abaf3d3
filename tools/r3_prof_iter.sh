#!/bin/bash
# per-kernel times of one mBCG iteration (rocprofv3 --kernel-trace of tools/c5_iter_bench.py); run on the GPU box
L=${1:-r3_iter}; SHAPE=${2:-C5}; T=${3:-11}
R=$GRAFT_REPO_ROOT
python3 $R/tools/c5_iter_bench.py $SHAPE $T 40 | tee $R/gpurun_out/${L}_${SHAPE}_T${T}.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${L}_trace -o t -- python3 $R/tools/c5_iter_bench.py $SHAPE $T 40 > /dev/null 2>&1
python3 - <<PY | tee $R/gpurun_out/${L}_${SHAPE}_T${T}_kernels.txt
import csv,collections
rows=list(csv.DictReader(open("$R/gpurun_out/${L}_trace/t_kernel_trace.csv")))
agg=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")
    agg[(n.split("(")[0][:60],r["Grid_Size_X"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    if len(v) >= 20:
        v2=sorted(v)
        print("%-62s grid=%-8s launches=%4d avg_us=%7.1f median_us=%7.1f" % (k[0],k[1],len(v),sum(v)/len(v),v2[len(v2)//2]))
PY
rm -rf $R/gpurun_out/${L}_trace
