"""Where does the FIRST full prediction of a fresh process go at C4 (N = 50 000, N* = 2 000)?  VERDICT r5 weak #6: 3.70 s in
tools/solve_bench.py's fresh process against ~1.0 s for the second call of the same process.  Times the stages of the
prediction (library calls wrapped with a synchronisation) for call 1 and call 2 of one process; run once per setting of
RPGP_BLOCKED_CHOL in its own process."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from rpgp_amd import settings, precond
from rpgp_amd.training import create_exact_gp, make_optimizer
from rpgp_amd.models import ExactMarginalLogLikelihood

dev = torch.device("cuda:0")
acc = {}


def wrap(mod, name, key=None):
    f = getattr(mod, name)
    key = key or name

    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0
        acc[key + "_n"] = acc.get(key + "_n", 0) + 1
        return r
    setattr(mod, name, g)


def build(N, d, J, ntest, steps=2):
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N + ntest, d, generator=g)
    y = torch.sin(X).sum(1) + 0.05 * torch.randn(N + ntest, generator=g)
    y = (y - y.mean()) / y.std()
    Xtr, ytr, Xte, yte = X[:N].to(dev), y[:N].to(dev), X[N:].to(dev), y[N:].to(dev)
    torch.manual_seed(0); np.random.seed(0)
    model, lik = create_exact_gp(Xtr, ytr, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False,
                                 prescale=True, space_proj=False, ski=False)
    model = model.to(dev)
    mll = ExactMarginalLogLikelihood(lik, model)
    opt = make_optimizer(torch.optim.Adam, [p for p in model.parameters() if p.requires_grad], 0.1)
    model.train()
    for _ in range(steps):
        opt.zero_grad(); loss = -mll(model(Xtr), ytr); loss.backward(); opt.step()
    return model, mll, Xte, yte


with settings.cg_tolerance(0.05), settings.eval_cg_tolerance(0.01), settings.max_cg_iterations(10000):
    if "--no-warm" not in sys.argv:
        m, mll, Xte, yte = build(3000, 8, 20, 300, 1)       # the warm-up of tools/solve_bench.py
        m.eval()
        with torch.no_grad():
            -mll(m(Xte), yte).item()
        del m, mll
    model, mll, Xte, yte = build(50000, 20, 20, 2000)
    wrap(precond, "blocked_cholesky")
    wrap(torch.linalg, "cholesky_ex", "lib_cholesky_ex")
    wrap(torch.linalg, "solve_triangular", "lib_trsm")
    wrap(torch, "cholesky_solve", "lib_cholesky_solve")
    wrap(torch, "mm", "mm")
    wrap(torch, "addmm", "addmm")
    out = {"RPGP_BLOCKED_CHOL": os.environ.get("RPGP_BLOCKED_CHOL", "1")}
    for call in (1, 2, 3):
        acc.clear()
        model.train(); model.eval()
        with torch.no_grad():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with settings.skip_posterior_variances(True):
                model(Xte).mean.sum().item()
            torch.cuda.synchronize(); t_mean = time.perf_counter() - t0
            model.train(); model.eval()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            nll = -mll(model(Xte), yte).item()
            torch.cuda.synchronize(); t_full = time.perf_counter() - t0
        out["call%d" % call] = dict({"mean_pred_s": round(t_mean, 3), "full_pred_s": round(t_full, 3), "test_nll": nll},
                                    **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in acc.items()})
    print(json.dumps(out))
