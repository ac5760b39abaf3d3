import sys, time, json, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import numpy as np
import rpgp_amd
from rpgp_amd import settings, precond, linear_cg as lcg
from rpgp_amd.training import create_exact_gp
from rpgp_amd.models import ExactMarginalLogLikelihood
orig = precond.blocked_cholesky
def timed_bc(K, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = orig(K, *a, **k); torch.cuda.synchronize()
    print("  blocked_cholesky n=%d: %.3f s" % (K.shape[0], time.perf_counter() - t0), flush=True); return r
precond.blocked_cholesky = timed_bc
oc = torch.linalg.cholesky_ex
def timed_c(K, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = oc(K, *a, **k); torch.cuda.synchronize()
    if K.shape[-1] > 4000: print("  cholesky_ex n=%d: %.3f s" % (K.shape[-1], time.perf_counter() - t0), flush=True)
    return r
torch.linalg.cholesky_ex = timed_c
dev = torch.device("cuda:0")
N, d, J, nt = 50000, 20, 20, 2000
g = torch.Generator().manual_seed(0)
X = torch.randn(N + nt, d, generator=g); y = torch.sin(X).sum(1) + 0.05 * torch.randn(N + nt, generator=g); y = (y - y.mean()) / y.std()
Xtr, ytr, Xte, yte = X[:N].to(dev), y[:N].to(dev), X[N:].to(dev), y[N:].to(dev)
torch.manual_seed(0); np.random.seed(0)
model, lik = create_exact_gp(Xtr, ytr, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True, space_proj=False)
model = model.to(dev); mll = ExactMarginalLogLikelihood(lik, model)
print("solve_refinement default:", settings.solve_refinement.value())
with settings.cg_tolerance(0.05), settings.eval_cg_tolerance(0.01), settings.max_cg_iterations(10000):
    from rpgp_amd.training import make_optimizer
    opt = make_optimizer(torch.optim.Adam, [p for p in model.parameters() if p.requires_grad], 0.1)
    model.train()
    for it in range(6):
        opt.zero_grad(); loss = -mll(model(Xtr), ytr); loss.backward(); opt.step()
    torch.cuda.synchronize(); print("trained; free GB", round(torch.cuda.mem_get_info()[0] / 1e9, 1), "reserved", round(torch.cuda.memory_reserved() / 1e9, 1), "allocated", round(torch.cuda.memory_allocated() / 1e9, 1), flush=True)
    model.eval()
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with settings.skip_posterior_variances(True):
            out = model(Xte)
        torch.cuda.synchronize(); print("mean_pred", round(time.perf_counter() - t0, 3), flush=True)
        import os
        for rep in range(8):
            os.environ["RPGP_BLOCKED_CHOL"] = "1" if rep % 2 == 0 else "0"
            model.train(); model.eval()
            torch.cuda.synchronize(); t0 = time.perf_counter(); lcg.stats["iterations"] = 0
            out = model(Xte)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            nll = -mll(out, yte).item()
            torch.cuda.synchronize(); print("  model(Xte)", round(t1 - t0, 3), " mll", round(time.perf_counter() - t1, 3), "mp:", type(getattr(model.prediction_strategy, "_mp", None)).__name__ if hasattr(model, "prediction_strategy") else "?", flush=True)
            print("full_pred rep", rep, "blocked" if rep % 2 == 0 else "library", round(time.perf_counter() - t0, 3), "cg iters", lcg.stats["iterations"], flush=True)
