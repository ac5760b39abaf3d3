#!/bin/bash
# scatter forms of the planned SKI product at C5 (same box): kernel medians per form and block width, then the step time
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for F in cell item; do for T in 1 11; do
RPGP_SKI_SCATTER=$F SKI_SHAPES=1 SKI_TS=$T SKI_PLAN=1 SKI_REPS=20 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/skiab_$F$T -o t -- python3 $R/tools/ski_bench.py > /dev/null 2>&1
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("$R/gpurun_out/skiab_$F$T/t_kernel_trace.csv")))
agg=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")
    if n.startswith("ski_") and "minmax" not in n and "grid_finish" not in n:
        agg[n.split("(")[0][:40]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
med={k: round(sorted(v)[len(v)//2],1) for k,v in agg.items()}
print("$F T=$T", med, "sum", round(sum(med.values()),1))
PY
rm -rf $R/gpurun_out/skiab_$F$T
done; done
cd $R
for F in cell item cell item; do RPGP_SKI_SCATTER=$F python3 tools/r4_step_time.py C5 2>/dev/null | tail -1; done
