"""Every served model specification end to end through the runner CLI on the GPU (kin8nm-shaped synthetic data, fold 0,
at most 60 epochs): fit + train / test evaluation with predictive covariances.  Writes gpurun_out/r4_runner_all_specs.jsonl."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpgp_amd import runner, specs, linear_cg as lcg

out = open(os.path.join("gpurun_out", "r4_runner_all_specs.jsonl"), "w")
bad = 0
for name in specs.names():
    spec = specs.get(name)
    tk = spec["base_model_kwargs"]["train_kwargs"] if spec["kind"] == "model_average" else spec["train_kwargs"]
    tk["max_iter"] = min(tk.get("max_iter", 1000), 60)
    if "random_restarts" in tk:
        tk["random_restarts"] = 2
    if "init_iters" in tk:
        tk["init_iters"] = min(tk["init_iters"], 20)
    if spec["kind"] == "model_average":
        spec["varying_params"] = {"J": [1, 3, 8]}
    f = "/tmp/spec_%s.json" % name
    json.dump(spec, open(f, "w"))
    n0 = lcg.stats.get("native_calls", 0)
    t0 = time.time()
    try:
        df = runner.main(["-m", f, "-d", "synthetic:kin8nm", "-o", "/tmp/out_%s.csv" % name, "--no_cv", "--device", "cuda:0"])
        r = df.iloc[0]
        rec = {"spec": name, "kind": spec["kind"], "seconds": round(time.time() - t0, 2), "rmse": float(r["rmse"]),
               "test_nll": float(r.get("test_nll", float("nan"))), "native_solves": lcg.stats.get("native_calls", 0) - n0}
        ok = np.isfinite(rec["rmse"])
    except Exception as e:                                        # noqa
        rec = {"spec": name, "kind": spec["kind"], "error": repr(e)[:300]}
        ok = False
    bad += 0 if ok else 1
    out.write(json.dumps(rec) + "\n"); out.flush()
    print(json.dumps(rec), flush=True)
print("SPECS_FAILED=%d" % bad)
