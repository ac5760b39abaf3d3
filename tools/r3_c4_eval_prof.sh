#!/bin/bash
# kernel-time breakdown of the C4 CLI run (2 epochs + evaluate-on-train + test evaluation); GPU box
R=$GRAFT_REPO_ROOT
cat > /tmp/c4run.py <<PY
import json,sys,os,time
sys.path.insert(0, "$R")
from rpgp_amd import runner, specs
spec = specs.get("additive_rp_prescale_J20"); spec["train_kwargs"]["max_iter"] = 2
json.dump(spec, open("/tmp/c4.json","w"))
t0=time.time()
df = runner.main(["-m","/tmp/c4.json","-d","synthetic:synthetic50k","-o","/tmp/c4.csv","--no_cv","--device","cuda:0","--skip_random_restart"])
print("C4_CLI_SECONDS", round(time.time()-t0,2))
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c4eval_trace -o t -- python3 /tmp/c4run.py 2>&1 | grep C4_CLI
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("$R/gpurun_out/c4eval_trace/t_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f s" % (tot/1e9))
for r in rows[:18]:
    print("%-90s calls %6s total %8.1f ms  (%4.1f %%)" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, 100*float(r["TotalDurationNs"])/tot))
PY
rm -rf $R/gpurun_out/c4eval_trace
