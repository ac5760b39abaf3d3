#!/bin/bash
# Kernel medians of the wide packed-cache product and its slab reduce per (N, reduce width, product kernel).
# Usage (on the GPU box): bash tools/r5_symk_red_prof.sh [tag]   -> gpurun_out/<tag>_symk_red_kernels.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r5}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_symk_red_kernels.txt
: > $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_symkprof -o t -- python3 $R/tools/r5_symk_red_prof.py > /dev/null 2>&1 < /dev/null
timeout 120 python3 - >> $OUT <<PY
import csv, collections, glob
fs = glob.glob("$R/gpurun_out/${TAG}_symkprof/**/t_kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0]))) if fs else []
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# consecutive runs of (product kernel, reduce kernel) pairs: group by the sequence of distinct (name pair) blocks of 12
seq = []
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if n.startswith("symk_mvm_tile") or n.startswith("mvm_reduce"):
        seq.append((n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", "")))
blocks = []
for i in range(0, len(seq) - 1, 2):
    key = (seq[i][0], seq[i + 1][0], seq[i][2])
    if not blocks or blocks[-1][0] != key:
        blocks.append((key, [], []))
    blocks[-1][1].append(seq[i][1]); blocks[-1][2].append(seq[i + 1][1])
for key, a, b in blocks:
    a = sorted(a); b = sorted(b)
    print("%-40s %-28s grid %-8s n=%3d  product median %8.1f us   reduce median %7.1f us   sum %8.1f" % (key[0][:40], key[1][:28], key[2], len(a), a[len(a) // 2], b[len(b) // 2], a[len(a) // 2] + b[len(b) // 2]))
PY
rm -rf $R/gpurun_out/${TAG}_symkprof
cat $OUT
