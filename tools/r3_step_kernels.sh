#!/bin/bash
# kernel launches per optimiser step at a BASELINE shape (rocprofv3 --kernel-trace of tools/solve_bench.py); GPU box
C=${1:-C2}; L=${2:-r3_step}
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${L}_trace_$C -o t -- python3 $R/tools/step_only.py $C 20 > $R/gpurun_out/${L}_${C}_step.json 2>/dev/null
python3 - <<PY | tee $R/gpurun_out/${L}_${C}_step_kernels.txt
import csv,collections,json
rows=list(csv.DictReader(open("$R/gpurun_out/${L}_trace_$C/t_kernel_trace.csv")))
info=json.load(open("$R/gpurun_out/${L}_${C}_step.json"))
t0=info["t_begin_ns"]; t1=info["t_end_ns"]; steps=info["steps"]
agg=collections.defaultdict(lambda:[0,0.0]); tot=0; n=0
sel=[r for r in rows]
# the timed steps are the LAST `steps` of (warm + steps): select by launch order fraction
k0=int(len(sel)*info["warm"]/(info["warm"]+steps))
for r in sel[k0:]:
    nm=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").replace("at::native::","")
    nm=(nm if nm.startswith(("elementwise","vectorized","reduce","unrolled")) else nm.split("(")[0])[:150]
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    agg[nm][0]+=1; agg[nm][1]+=d; tot+=d; n+=1
print("per step: %d launches, %.1f us of kernel time; wall %.1f us" % (n/steps, tot/steps, info["step_ms"]*1e3))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:45]:
    print("%-150s %6.1f launches/step %8.1f us/step" % (k, v[0]/steps, v[1]/steps))
PY
rm -rf $R/gpurun_out/${L}_trace_$C
