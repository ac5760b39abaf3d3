#!/bin/bash
# Kernel medians of the planned SKI product, chunked and cell-sorted forms, at the C5 shape (locality order, T = 11 and 1).
# Usage (on the GPU box): bash tools/r5_ski_prof.sh [tag]   -> gpurun_out/<tag>_ski_kernels.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r5}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_ski_kernels.txt
: > $OUT
ORDERS=${ORDERS:-locality} timeout 240 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_skiprof -o t -- python3 $R/tools/r5_ski_chunk_ab.py > $R/gpurun_out/${TAG}_ski_ab.jsonl 2>/dev/null < /dev/null
timeout 120 python3 - >> $OUT <<PY
import csv, collections, glob
fs = glob.glob("$R/gpurun_out/${TAG}_skiprof/**/t_kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0]))) if fs else []
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("ski_") or n.startswith("chunk_") or n.startswith("plan_") or "radix" in n:
        agg[n.split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    v = sorted(v)
    print("%-62s n=%5d  median %8.1f us  p10 %8.1f  p90 %8.1f" % (k, len(v), v[len(v) // 2], v[len(v) // 10], v[(9 * len(v)) // 10]))
PY
rm -rf $R/gpurun_out/${TAG}_skiprof
cat $OUT
cat $R/gpurun_out/${TAG}_ski_ab.jsonl
