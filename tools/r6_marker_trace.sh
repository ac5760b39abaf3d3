#!/bin/bash
# One optimiser step under rocprofv3 --marker-trace --kernel-trace, summarised BY PHASE (roctx range): per range name the calls
# per step, the mean host duration, and the device time of the kernels that started inside the range's window.
# (No --pmc here: gpurun refuses counters together with marker tracing.)
C=${1:-C3}; L=${2:-r6_step}; STEPS=${3:-20}
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --marker-trace --kernel-trace --output-format csv -d $R/gpurun_out/${L}_mtrace_$C -o t -- python3 $R/tools/r6_marker_step.py $C $STEPS > $R/gpurun_out/${L}_${C}_marker_step.json 2>$R/gpurun_out/${L}_${C}_marker_step.err
python3 $R/tools/r6_marker_step.py $C $STEPS > $R/gpurun_out/${L}_${C}_unprofiled_step.json 2>/dev/null
python3 - <<PY | tee $R/gpurun_out/${L}_${C}_marker_trace_summary.txt
import csv, collections, glob, json, bisect
d="$R/gpurun_out/${L}_mtrace_$C"
info=json.load(open("$R/gpurun_out/${L}_${C}_marker_step.json"))
plain=json.load(open("$R/gpurun_out/${L}_${C}_unprofiled_step.json"))
mk=glob.glob(d+"/**/*marker_api_trace.csv", recursive=True); kt=glob.glob(d+"/**/*kernel_trace.csv", recursive=True)
print("shape $C: %d steps; step %.3f ms under rocprofv3 (ranges live: %d), %.3f ms unprofiled (ranges live: %d)" % (info["steps"], info["step_ms"], info["ranges_live"], plain["step_ms"], plain["ranges_live"]))
if not mk or not kt:
    print("no marker / kernel trace found in", d, glob.glob(d+"/**/*", recursive=True)[:20]); raise SystemExit
marks=list(csv.DictReader(open(mk[0]))); kern=list(csv.DictReader(open(kt[0])))
namecol=[c for c in marks[0].keys() if c.lower() in ("function","name","message")][0]
kern.sort(key=lambda r:int(r["Start_Timestamp"]))
kstart=[int(r["Start_Timestamp"]) for r in kern]
steps=info["steps"]+info["warm"]
agg=collections.defaultdict(lambda:[0,0.0,0.0,0,collections.Counter()])
for m in marks:
    s,e=int(m["Start_Timestamp"]),int(m["End_Timestamp"])
    a=agg[m[namecol]]; a[0]+=1; a[1]+=(e-s)/1e3
    i0,i1=bisect.bisect_left(kstart,s),bisect.bisect_right(kstart,e)
    for r in kern[i0:i1]:
        a[2]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; a[3]+=1
        a[4][r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:60]]+=1
print("%-28s %9s %14s %22s %10s   %s" % ("range", "n/step", "host us/call", "kernel us inside/call", "launches", "most frequent kernels started inside"))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    top=", ".join("%s x%.1f" % (n, c/v[0]) for n,c in v[4].most_common(3))
    print("%-28s %9.2f %14.1f %22.1f %10.1f   %s" % (k, v[0]/steps, v[1]/v[0], v[2]/v[0], v[3]/v[0], top))
PY
rm -rf $R/gpurun_out/${L}_mtrace_$C
