#!/bin/bash
# One optimiser step under rocprofv3 --kernel-trace at a BASELINE shape: launches, kernel time AND the idle gaps of the
# device — per kernel name, how long the GPU sat idle before that kernel started (who is the host / a sync holding up?).
C=${1:-C2}; L=${2:-r4_step}
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${L}_trace_$C -o t -- python3 $R/tools/step_only.py $C 20 > $R/gpurun_out/${L}_${C}_step.json 2>/dev/null
python3 - <<PY | tee $R/gpurun_out/${L}_${C}_step_gaps.txt
import csv,collections,json
rows=list(csv.DictReader(open("$R/gpurun_out/${L}_trace_$C/t_kernel_trace.csv")))
info=json.load(open("$R/gpurun_out/${L}_${C}_step.json"))
steps=info["steps"]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
k0=int(len(rows)*info["warm"]/(info["warm"]+steps))
sel=rows[k0:]
def nm(r):
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").replace("at::native::","")
    return (n if n.startswith(("elementwise","vectorized","reduce","unrolled")) else n.split("(")[0])[:110]
agg=collections.defaultdict(lambda:[0,0.0,0.0]); tot=0; gaps=0; prev_end=None
for r in sel:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    g=0.0 if prev_end is None else max(0.0,(s-prev_end)/1e3)
    prev_end=e if prev_end is None else max(prev_end,e)
    a=agg[nm(r)]; a[0]+=1; a[1]+=(e-s)/1e3; a[2]+=g; tot+=(e-s)/1e3; gaps+=g
span=(int(sel[-1]["End_Timestamp"])-int(sel[0]["Start_Timestamp"]))/1e3
print("per step: %d launches, %.1f us kernel time, %.1f us idle gaps, span %.1f us; wall %.1f us (under the profiler)" % (len(sel)/steps, tot/steps, gaps/steps, span/steps, info["step_ms"]*1e3))
print("%-110s %8s %10s %12s" % ("kernel", "n/step", "us/step", "gap-before us/step"))
for k,v in sorted(agg.items(), key=lambda kv:-(kv[1][1]+kv[1][2]))[:40]:
    print("%-110s %8.1f %10.1f %12.1f" % (k, v[0]/steps, v[1]/steps, v[2]/steps))
# the last step as a timeline: gap before, duration, kernel
n1=int(round(len(sel)/steps))
with open("$R/gpurun_out/${L}_${C}_last_step_timeline.txt","w") as f:
    pe=int(sel[-n1-1]["End_Timestamp"])
    for r in sel[-n1:]:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        f.write("%8.1f %8.1f  %s\n" % (max(0,(s-pe))/1e3,(e-s)/1e3,nm(r)[:90])); pe=max(pe,e)
PY
rm -rf $R/gpurun_out/${L}_trace_$C
