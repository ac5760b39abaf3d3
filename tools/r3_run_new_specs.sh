cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import json, time, sys, os
sys.path.insert(0, os.getcwd())
from rpgp_amd import runner, specs
import numpy as np
out = {}
for name, extra in (("polynomial_rp_smaller", {}), ("polynomial_rp", {}), ("additive_spread_projections", {}),
                    ("additive_rp_prescale_on_sphere", {}), ("additive_deterministic_spec_unweighted", {}),
                    ("ma_dpa_gp_ard", {"vp": [1, 3, 8]})):
    spec = specs.get(name)
    if "vp" in extra:
        spec["varying_params"] = {"J": extra["vp"]}
        spec["base_model_kwargs"]["train_kwargs"]["max_iter"] = 30
    else:
        spec["train_kwargs"]["max_iter"] = 30
    f = "/tmp/%s.json" % name
    json.dump(spec, open(f, "w"))
    t0 = time.time()
    df = runner.main(["-m", f, "-d", "synthetic:kin8nm", "-o", "/tmp/%s.csv" % name, "--no_cv", "--skip_random_restart",
                      "--device", "cuda:0"])
    r = df.iloc[0]
    out[name] = {"seconds": round(time.time() - t0, 2), "rmse": float(r["rmse"]),
                 "test_nll": float(r.get("test_nll", float("nan")))}
    print(name, out[name], flush=True)
json.dump(out, open("gpurun_out/r3_runner_new_specs.json", "w"), indent=1)
PY
