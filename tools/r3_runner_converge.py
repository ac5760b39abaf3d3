"""Fits to convergence with the specs' own train_kwargs (max_iter 1000, patience 20, restarts as specified) through the runner on
the GPU; checks finite results and reports epochs / time.  Writes gpurun_out/r3_runner_converge.jsonl."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpgp_amd import runner, specs
out = open(os.path.join("gpurun_out", "r3_runner_converge.jsonl"), "w")
for name, data in (("additive_rp_prescale_J20", "synthetic:kin8nm"), ("additive_spread_prescale_J20", "synthetic:elevators"),
                   ("additive_spread_prescale_Jd_ski", "synthetic:kin8nm"), ("GAM_spec", "synthetic:kin8nm"),
                   ("additive_rp_J20_K1", "synthetic:kin8nm"), ("polynomial_rp_smaller", "synthetic:kin8nm"),
                   ("additive_spread_projections_RO", "synthetic:kin8nm"), ("additive_spread_prescale_Jd_ski", "synthetic:3droad")):
    flags = ["--skip_posterior_variances"] if data.endswith("3droad") else []
    t0 = time.time()
    df = runner.main(["-m", name, "-d", data, "-o", "/tmp/conv_%s.csv" % name, "--no_cv", "--device", "cuda:0"] + flags)
    r = df.iloc[0]
    rec = {"spec": name, "data": data, "seconds": round(time.time() - t0, 2), "trained_epochs": int(r.get("trained_epochs", -1)),
           "rmse": float(r["rmse"]), "prior_train_nmll": float(r.get("prior_train_nmll", float("nan"))),
           "training_warnings": int(r.get("training_warnings", 0)), "rows": len(df)}
    out.write(json.dumps(rec) + "\n"); out.flush()
    print(json.dumps(rec), flush=True)
