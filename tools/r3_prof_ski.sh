#!/bin/bash
# per-kernel times of the planned SKI MVM (tools/ski_bench.py with SKI_PLAN=1 under rocprofv3 --kernel-trace); GPU box
L=${1:-r3_ski}
R=$GRAFT_REPO_ROOT
SKI_PLAN=1 python3 $R/tools/ski_bench.py | tee $R/gpurun_out/${L}_bench.jsonl
cd /tmp; export TMPDIR=/tmp
SKI_PLAN=1 SKI_REPS=20 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${L}_trace -o t -- python3 $R/tools/ski_bench.py > /dev/null 2>&1
cp $R/gpurun_out/${L}_trace/t_kernel_stats.csv $R/gpurun_out/${L}_kernel_stats.csv
python3 - <<PY | tee $R/gpurun_out/${L}_kernels.txt
import csv,os,collections
rows=list(csv.DictReader(open("$R/gpurun_out/${L}_trace/t_kernel_trace.csv")))
agg=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")
    if "ski" in n or "plan" in n or "radix" in n or "onesweep" in n:
        agg[(n.split("(")[0][:56],r["Grid_Size_X"],r["Grid_Size_Y"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(agg.items()):
    print("%-58s grid=(%s,%s) launches=%d avg_us=%.1f" % (k[0],k[1],k[2],len(v),sum(v)/len(v)))
PY
rm -rf $R/gpurun_out/${L}_trace
