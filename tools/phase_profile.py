"""Host-side phase timing of one training step (synchronising between phases) to see where small-N steps spend time.
usage: phase_profile.py [N] [d] [ski]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from rpgp_amd import settings, linear_cg as lcg, ops, operators, kernels
from rpgp_amd.training import create_exact_gp
from rpgp_amd.models import ExactMarginalLogLikelihood
from rpgp_amd import inv_quad_logdet as iql, precond
import rpgp_amd.precond as P

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 7372
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
J = 20
SKI = len(sys.argv) > 3 and sys.argv[3] == "ski"
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, generator=g); y = torch.sin(X).sum(1); y = (y - y.mean()) / y.std()
X, y = X.to(dev), y.to(dev)
torch.manual_seed(0)
model, lik = create_exact_gp(X, y, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True,
                             ski=SKI, ski_options={"grid_size": 1024, "num_dims": 1} if SKI else None)
model = model.to(dev); mll = ExactMarginalLogLikelihood(lik, model)
T = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(*a, **k); torch.cuda.synchronize()
        T[name] = T.get(name, 0) + time.perf_counter() - t0; return r
    return w
def wrap(obj, attr, name=None):
    setattr(obj, attr, timed(name or attr, getattr(obj, attr)))
wrap(iql, "build_preconditioner", "preconditioner(total)")
wrap(iql, "linear_cg", "cg(total)")
wrap(iql, "slq_logdet")
wrap(ops, "mbcg_solve", " cg:native_call")
wrap(P.WoodburyPreconditioner, "cinv", " cg:cinv")
wrap(P.WoodburyPreconditioner, "sample", " probes:sample")
wrap(lcg, "_tridiag_from_history", " cg:tridiag")
wrap(ops, "pivoted_cholesky", " pre:pivchol_kernel")
wrap(ops, "bilinear_grad", " bwd:bilinear_grad")
wrap(ops, "project_grad", " bwd:project_grad")
wrap(ops, "project", " fwd:project")
wrap(ops.Prepared, "__init__", " prepare")
wrap(operators.AdditiveRPOperator, "to_dense_cached", " cachedK:build")
wrap(settings, "use_cached_kernel", " cachedK:policy")
_orig_cg = iql.linear_cg
ITERS = []
def _cg_count(*a, **k):
    r = _orig_cg(*a, **k); ITERS.append(lcg.stats["last_iterations"]); return r
iql.linear_cg = _cg_count
with settings.cg_tolerance(0.05):
    for it in range(6):
        if it == 1: T.clear(); torch.cuda.synchronize(); t_all = time.perf_counter()
        model.zero_grad()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model(X); torch.cuda.synchronize(); T["model_fwd"] = T.get("model_fwd", 0) + time.perf_counter() - t0
        t0 = time.perf_counter(); loss = -mll(out, y); torch.cuda.synchronize(); T["mll_fwd(total)"] = T.get("mll_fwd(total)", 0) + time.perf_counter() - t0
        t0 = time.perf_counter(); loss.backward(); torch.cuda.synchronize(); T["backward(total)"] = T.get("backward(total)", 0) + time.perf_counter() - t0
    torch.cuda.synchronize(); tot = time.perf_counter() - t_all
for k, v in T.items(): print("%-28s %8.3f ms" % (k, v / 5 * 1e3))
print("step total %.3f ms" % (tot / 5 * 1e3), "cg iterations", ITERS)
