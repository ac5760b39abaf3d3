"""C4-size (N = 50 000, N_test = 2 000) prediction timings with and without the mixed-precision refinement, and the pieces
the refinement is made of (float64 twin MVM, float64 GEMM residual, float32 factorisation).  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import settings
from tests.test_baseline_sizes_gpu import _data, N4, D4, J4, NTEST4
from tests.test_host_stack import _build_model
import math

dev = torch.device("cuda:0")
X, y = _data(N4 + NTEST4, D4, seed=4)
torch.manual_seed(104)
from rpgp_amd import rp
P = torch.cat([rp.gen_rp(D4, 1, "gaussian") for _ in range(J4)], dim=1).contiguous()
ls = math.sqrt(D4) * (1.0 + 0.3 * torch.rand(D4, generator=torch.Generator().manual_seed(9)))
model, lik, mll = _build_model(X[:N4].to(dev), y[:N4].to(dev), P, ls, 0.1, 0.9)
model = model.to(dev)
Xs, ys = X[N4:].to(dev), y[N4:].to(dev)
res = {}


def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
    return time.perf_counter() - t0, r


for rounds in (0, 1, 0, 1):
    for tol in (0.01,):
        model.train(); model.eval()
        with torch.no_grad(), settings.eval_cg_tolerance(tol), settings.solve_refinement(rounds):
            with settings.skip_posterior_variances(True):
                t_mean, out = timed(lambda: model(Xs).mean)
            t_full, out = timed(lambda: model(Xs))
            t_nll, nll = timed(lambda: -mll(out, ys).item())
        st = model.prediction_strategy
        res["refine=%d tol=%g" % (rounds, tol)] = {"mean_pred_s": t_mean, "full_cov_s": t_full, "nll_s": t_nll, "test_nll": nll,
                                                   "thin_hist": getattr(st, "refinement_residuals", None),
                                                   "wide_hist": getattr(st, "wide_refinement_residuals", None)}
# evaluate-on-train (training_routines.py:551-556,567-569): posterior at the 50 000 training inputs + its NLL
model.train(); model.eval()
with torch.no_grad(), settings.eval_cg_tolerance(0.01):
    t_tr, tr_out = timed(lambda: model(model.train_inputs))
    t_mse, mse = timed(lambda: float(((tr_out.mean - model.train_targets) ** 2).mean()))
    t_tnll, tnll = timed(lambda: -mll(tr_out, model.train_targets).item())
res["evaluate_on_train"] = {"posterior_s": t_tr, "train_mse": mse, "train_nll_s": t_tnll, "train_nll": tnll,
                            "kind": type(tr_out).__name__}
f64 = model.covar_module.float64_operator(model.train_inputs)
v = torch.randn(N4, 1, dtype=torch.float64, device=dev)
f64._matmul(v, 0.1)
res["f64_twin_mvm_T1_s"] = timed(lambda: f64._matmul(v, 0.1))[0]
print(json.dumps(res))
