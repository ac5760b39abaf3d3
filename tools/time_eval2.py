import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import settings, linear_cg as lcg
from rpgp_amd.training import create_exact_gp
from rpgp_amd.models import PredictionStrategy
dev = torch.device("cuda:0")
N, d, J = (int(sys.argv[1]) if len(sys.argv) > 1 else 14939), 18, 20
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, generator=g); y = torch.sin(X).sum(1) + 0.01 * torch.randn(N, generator=g); y = (y - y.mean()) / y.std()
X, y = X.to(dev), y.to(dev)
torch.manual_seed(0)
model, lik = create_exact_gp(X, y, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True)
model, lik = model.to(dev), lik.to(dev)
lik.noise = 0.05
def tm(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print("%-40s %8.3f s" % (name, time.perf_counter() - t0), flush=True); return r
with settings.cg_tolerance(0.05), settings.eval_cg_tolerance(0.01), settings.max_cg_iterations(10000), torch.no_grad():
    model.eval(); lik.eval()
    ps = tm("PredictionStrategy (mean cache)", lambda: PredictionStrategy(model))
    cross = model.covar_module(X, model.train_inputs)
    cov = tm("K** to_dense", lambda: model.covar_module(X).to_dense())
    idx = torch.arange(N, device=dev)
    Kx = tm("cross rows -> K(X, X*) transpose copy", lambda: cross._get_rows(idx).t().contiguous())
    sol = tm("wide solve (CG on dense Khat)", lambda: ps.solve(Kx))
    print("    cg", lcg.stats["last_iterations"], "iterations")
    KS = tm("Khat @ S (fused operator, wide)", lambda: ps.khat._matmul(sol))
    BtS = tm("K(X*,X) @ S (rect fused, wide)", lambda: cross._matmul(sol))
    tm("cov -= BtS + BtS^T - S^T KS", lambda: cov.sub_(BtS + BtS.t() - sol.t() @ KS))
