"""Where the evaluation block of train_exact_gp goes (predict on the train set with full covariance + log_prob)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import settings, linear_cg as lcg
from rpgp_amd.training import create_exact_gp
from rpgp_amd.models import ExactMarginalLogLikelihood
dev = torch.device("cuda:0")
N, d, J = (int(sys.argv[1]) if len(sys.argv) > 1 else 14939), 18, 20
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, generator=g); y = torch.sin(X).sum(1) + 0.01 * torch.randn(N, generator=g); y = (y - y.mean()) / y.std()
X, y = X.to(dev), y.to(dev)
torch.manual_seed(0)
model, lik = create_exact_gp(X, y, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True)
model, lik = model.to(dev), lik.to(dev)
lik.noise = 0.05
mll = ExactMarginalLogLikelihood(lik, model)
def tm(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print("%-34s %8.3f s" % (name, time.perf_counter() - t0), flush=True); return r
with settings.cg_tolerance(0.05), settings.eval_cg_tolerance(0.01), settings.max_cg_iterations(10000), torch.no_grad():
    model.eval(); lik.eval()
    if len(sys.argv) > 2 and sys.argv[2] == "cprofile":
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        out = model(X); torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)
        sys.exit(0)
    out = tm("predict(train X) full covariance", lambda: model(X))
    print("   cg stats", lcg.stats)
    model.train(); model.eval()          # drop the prediction strategy: time a second, warm evaluation
    out = tm("predict(train X) again (warm)", lambda: model(X))
    pred = tm("likelihood(out) (+noise)", lambda: lik(out))
    tm("log_prob (float64 Cholesky)", lambda: pred.log_prob(y))
    tm("confidence_region/variance", lambda: pred.variance.sqrt())
