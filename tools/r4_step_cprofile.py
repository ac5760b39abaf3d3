"""Host-side profile (cProfile) of the optimiser step at a BASELINE shape: where the PYTHON time of one step goes."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["C2"])
shape = sys.argv[1]
import torch, numpy as np
from rpgp_amd import settings
from rpgp_amd.training import create_exact_gp, make_optimizer
from rpgp_amd.models import ExactMarginalLogLikelihood
SHAPES = {"C2": (7372, 8, 20, False, False), "C3": (14939, 18, 20, True, False), "C5": (391386, 3, 3, True, True)}
N, d, J, sp, ski = SHAPES[shape]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, generator=g); y = torch.sin(X).sum(1) + 0.05 * torch.randn(N, generator=g); y = (y - y.mean()) / y.std()
X, y = X.to(dev), y.to(dev)
torch.manual_seed(0); np.random.seed(0)
model, lik = create_exact_gp(X, y, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True,
                             space_proj=sp, ski=ski, ski_options={"grid_size": 1024, "num_dims": 1} if ski else None)
model = model.to(dev); mll = ExactMarginalLogLikelihood(lik, model)
opt = make_optimizer(torch.optim.Adam, [p for p in model.parameters() if p.requires_grad], 0.0)
steps = 200
def run(n):
    for it in range(n):
        opt.zero_grad(); loss = -mll(model(X), y); loss.backward(); opt.step()
with settings.cg_tolerance(0.05), settings.max_cg_iterations(10000):
    model.train(); run(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("step %.1f us (no profiler)" % ((t1 - t0) / steps * 1e6))
    pr = cProfile.Profile(); pr.enable(); run(steps); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(int(os.environ.get('TOPN', '45')))
txt = s.getvalue()
# per-step microseconds
print("(tottime / cumtime below are totals over %d steps: divide by %d; 1 s total = %.0f us per step)" % (steps, steps, 1e6 / steps))
print(txt)
