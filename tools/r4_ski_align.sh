#!/bin/bash
# scatter / gather kernel times of the planned SKI product at C5 for several block widths T (row = 4 T bytes of V)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for T in 4 8 11 12; do
SKI_SHAPES=1 SKI_TS=$T SKI_PLAN=1 SKI_REPS=20 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/skialign_$T -o t -- python3 $R/tools/ski_bench.py > /dev/null 2>&1
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("$R/gpurun_out/skialign_$T/t_kernel_trace.csv")))
agg=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")
    if n.startswith("ski_"):
        agg[n.split("(")[0][:40]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("T=$T", {k: round(sorted(v)[len(v)//2],1) for k,v in agg.items()})
PY
rm -rf $R/gpurun_out/skialign_$T
done
