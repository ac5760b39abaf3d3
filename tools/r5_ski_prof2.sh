#!/bin/bash
# kernel medians of the chunked product only (T = 11, locality order), optionally with RPGP_SKI_DBG variants
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for DBG in ${DBGS:-0}; do
RPGP_SKI_DBG=$DBG ORDERS=locality timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/p2_$DBG -o t -- python3 $R/tools/r5_ski_chunk_ab.py > /dev/null 2>&1 < /dev/null
timeout 60 python3 - <<PY
import csv, collections, glob
fs = glob.glob("$R/gpurun_out/p2_$DBG/**/t_kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0]))) if fs else []
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("ski_chunk") or n.startswith("ski_gather_lds") or n.startswith("ski_scatter_cell") or n.startswith("ski_cellsum4") or n.startswith("ski_toep"):
        agg[n.split("(")[0][:50]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("DBG=$DBG", {k: round(sorted(v)[len(v)//2], 1) for k, v in sorted(agg.items())})
PY
rm -rf $R/gpurun_out/p2_$DBG
done
