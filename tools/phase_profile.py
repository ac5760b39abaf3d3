"""Host-side phase timing of one training step (synchronising between phases) to see where small-N steps spend time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from rpgp_amd import settings, linear_cg as lcg
from rpgp_amd.training import create_exact_gp
from rpgp_amd.models import ExactMarginalLogLikelihood
from rpgp_amd import inv_quad_logdet as iql, precond
import rpgp_amd.precond as P

dev = torch.device("cuda:0")
N, d, J = 7372, 8, 20
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, generator=g); y = torch.sin(X).sum(1); y = (y - y.mean()) / y.std()
X, y = X.to(dev), y.to(dev)
torch.manual_seed(0)
model, lik = create_exact_gp(X, y, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True)
model = model.to(dev); mll = ExactMarginalLogLikelihood(lik, model)
T = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(*a, **k); torch.cuda.synchronize()
        T[name] = T.get(name, 0) + time.perf_counter() - t0; return r
    return w
iql.build_preconditioner = timed("preconditioner", P.build_preconditioner)
iql.linear_cg = timed("cg", lcg.linear_cg)
iql.slq_logdet = timed("slq", iql.slq_logdet)
orig_bil = type(model.covar_module(X))._bilinear_derivative
with settings.cg_tolerance(0.05):
    for it in range(6):
        if it == 1: T.clear(); torch.cuda.synchronize(); t_all = time.perf_counter()
        model.zero_grad()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model(X); torch.cuda.synchronize(); T["model_fwd(project)"] = T.get("model_fwd(project)", 0) + time.perf_counter() - t0
        t0 = time.perf_counter(); loss = -mll(out, y); torch.cuda.synchronize(); T["mll_fwd_total"] = T.get("mll_fwd_total", 0) + time.perf_counter() - t0
        t0 = time.perf_counter(); loss.backward(); torch.cuda.synchronize(); T["backward"] = T.get("backward", 0) + time.perf_counter() - t0
    torch.cuda.synchronize(); tot = time.perf_counter() - t_all
print("per-step ms:", {k: round(v / 5 * 1e3, 3) for k, v in T.items()}, "total", round(tot / 5 * 1e3, 3))
