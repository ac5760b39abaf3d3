"""Runner CLI flag combinations end to end on the GPU (60 epochs at most).  Writes gpurun_out/r4_runner_flags_soak.jsonl."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpgp_amd import runner, specs

CASES = [
    ("additive_rp_prescale_J20", "synthetic:kin8nm", ["--fast_pred"]),
    ("additive_rp_prescale_J20", "synthetic:kin8nm", ["--use_chol"]),
    ("additive_rp_prescale_J20", "synthetic:kin8nm", ["--skip_posterior_variances", "--skip_evaluate_on_train"]),
    ("additive_rp_prescale_J20", "synthetic:kin8nm", ["--no_cache_kernel"]),
    ("additive_rp_prescale_J20", "synthetic:kin8nm", ["--cache_kernel", "--record_pred_unc"]),
    ("additive_rp_prescale_J20", "synthetic:kin8nm", ["--skip_log_det_forward"]),
    ("additive_rp_prescale_J20", "synthetic:kin8nm", ["--double"]),
    ("additive_rp_prescale_J20", "synthetic:kin8nm", ["--checkpoint_kernel", "1500", "--memory_efficient"]),
    ("additive_rp_prescale_J20", "synthetic:kin8nm", ["--ablation", "--J", "2", "5"]),
    ("additive_rp_prescale_J20", "synthetic:kin8nm", ["--cg_tol", "0.01", "--eval_cg_tol", "0.001", "--max_cg_iterations", "500"]),
    ("additive_rp_prescale_J20", "synthetic:yacht", []),
    ("additive_rp_prescale_J20", "synthetic:elevators", ["--fast_pred"]),
    ("additive_spread_prescale_J20", "synthetic:elevators", []),
    ("additive_spread_prescale_Jd_ski", "synthetic:kin8nm", ["--no_toeplitz"]),
    ("additive_spread_prescale_Jd_ski", "synthetic:elevators", ["--fast_pred"]),
    ("additive_spread_prescale_Jd", "synthetic:elevators", []),
    ("GAM_spec", "synthetic:elevators", []),
    ("RBF_model_spec", "synthetic:yacht", ["--device-cpu"]),
    ("additive_rp_J20_K1_ski", "synthetic:elevators", []),
    ("additive_rp_prescale_J20", "synthetic:kin8nm", ["--cv", "--fold", "3"]),
    # round 4: --double for family members / the memory-efficient GAM, the k ablation with sizes that are not instantiated
    ("additive_rp_J20_K1", "synthetic:kin8nm", ["--double"]),
    ("GAM_spec", "synthetic:kin8nm", ["--double"]),
    ("additive_rp_prescale_J1_K20", "synthetic:kin8nm", ["--ablation", "--k", "3", "6", "7"]),
    ("additive_spread_prescale_Jd", "synthetic:kin8nm", ["--double"]),
    # ... and for the grid-interpolation operator (rpgp_ski_f64.hip)
    ("additive_spread_prescale_Jd_ski", "synthetic:kin8nm", ["--double"]),
    ("additive_rp_J20_K1_ski", "synthetic:kin8nm", ["--double"]),
]
out = open(os.path.join("gpurun_out", "r4_runner_flags_soak.jsonl"), "w")
bad = 0
for i, (name, data, flags) in enumerate(CASES):
    spec = specs.get(name)
    spec["train_kwargs"]["max_iter"] = min(spec["train_kwargs"].get("max_iter", 1000), 40)
    f = "/tmp/flag_spec_%d.json" % i
    json.dump(spec, open(f, "w"))
    dev = "cpu" if "--device-cpu" in flags else "cuda:0"
    fl = [x for x in flags if x != "--device-cpu"]
    cv = "--cv" in fl
    fl = [x for x in fl if x != "--cv"]
    argv = ["-m", f, "-d", data, "-o", "/tmp/flag_out_%d.csv" % i, "--device", dev, "--skip_random_restart"] + \
           ([] if cv else ["--no_cv"]) + fl
    t0 = time.time()
    try:
        df = runner.main(argv)
        rm = [float(v) for v in df["rmse"]]
        rec = {"case": i, "spec": name, "data": data, "flags": flags, "rows": len(df), "seconds": round(time.time() - t0, 2),
               "rmse": rm, "errors": int(df["error"].notna().sum()) if "error" in df else 0}
        ok = all(np.isfinite(rm)) and rec["errors"] == 0
    except Exception as e:                                        # noqa
        rec = {"case": i, "spec": name, "data": data, "flags": flags, "exception": repr(e)[:400]}
        ok = False
    bad += 0 if ok else 1
    out.write(json.dumps(rec) + "\n"); out.flush()
    print(json.dumps(rec), flush=True)
print("CASES_FAILED=%d" % bad)
