#!/bin/bash
# kernel medians of one C5 solve with and without the folded pass A
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/foldprof -o t -- python3 $R/tools/r5_fold_ab.py > /dev/null 2>&1 < /dev/null
timeout 60 python3 - <<PY
import csv, collections, glob
fs = glob.glob("$R/gpurun_out/foldprof/**/t_kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0]))) if fs else []
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    agg[n.split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v = sorted(v)
    print("%-62s n=%6d  median %8.1f us  total %9.1f ms" % (k, len(v), v[len(v) // 2], sum(v) / 1e3))
PY
rm -rf $R/gpurun_out/foldprof
