"""User-journey check on the GPU: the experiment runner CLI end to end (fit to convergence + evaluation) for several
model specs on a synthetic stand-in of a BASELINE dataset shape; prints one JSON line per spec."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpgp_amd import runner
dataset = sys.argv[1] if len(sys.argv) > 1 else "synthetic:kin8nm"
specs = sys.argv[2:] or ["additive_rp_prescale_J20", "additive_spread_prescale_J20", "additive_rp_prescale_J20_matern",
                         "GAM_spec", "additive_rp_J20_K1"]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in specs:
    t0 = time.perf_counter()
    df = runner.main(["-m", spec, "-d", dataset, "-o",
                      os.path.join(root, "gpurun_out", "run_%s.csv" % spec), "--device", "cuda:0", "--no_cv",
                      "--skip_random_restart"])
    dt = time.perf_counter() - t0
    row = df.iloc[0].to_dict()
    keep = {k: row[k] for k in ("rmse", "trained_epochs", "prior_train_nmll", "test_nll", "train_time", "n", "d",
                                "training_warnings", "error") if k in row}
    print(json.dumps({"spec": spec, "dataset": dataset, "wall_s": round(dt, 2), **{k: (float(v) if isinstance(v, (int, float)) else str(v)) for k, v in keep.items()}}), flush=True)
