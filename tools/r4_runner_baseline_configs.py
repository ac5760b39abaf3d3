"""BASELINE.json configs 2-5 end to end through the runner CLI on the GPU (synthetic stand-ins of the same shapes, at most
40 epochs, fold 0).  Writes gpurun_out/r4_runner_baseline_configs.jsonl."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpgp_amd import runner, specs, linear_cg as lcg
CASES = [("C2", "additive_rp_prescale_J20", "synthetic:kin8nm", []),
         ("C3", "additive_spread_prescale_J20", "synthetic:elevators", []),
         ("C4", "additive_rp_prescale_J20", "synthetic:synthetic50k", []),
         ("C4-fast_pred", "additive_rp_prescale_J20", "synthetic:synthetic50k", ["--fast_pred"]),
         ("C5", "additive_spread_prescale_Jd_ski", "synthetic:3droad", ["--skip_posterior_variances"])]
out = open(os.path.join("gpurun_out", "r4_runner_baseline_configs.jsonl"), "w")
for tag, name, data, flags in CASES:
    spec = specs.get(name)
    spec["train_kwargs"]["max_iter"] = 40
    f = "/tmp/bc_%s.json" % tag
    json.dump(spec, open(f, "w"))
    n0 = lcg.stats.get("native_calls", 0)
    t0 = time.time()
    df = runner.main(["-m", f, "-d", data, "-o", "/tmp/bc_%s.csv" % tag, "--no_cv", "--device", "cuda:0", "--skip_random_restart"] + flags)
    r = df.iloc[0]
    rec = {"config": tag, "spec": name, "data": data, "flags": flags, "seconds_total": round(time.time() - t0, 2),
           "train_time_s": float(r.get("train_time", float("nan"))), "trained_epochs": int(r.get("trained_epochs", -1)),
           "rmse": float(r["rmse"]), "test_nll": float(r.get("test_nll", float("nan"))),
           "native_solves": lcg.stats.get("native_calls", 0) - n0}
    out.write(json.dumps(rec) + "\n"); out.flush()
    print(json.dumps(rec), flush=True)
