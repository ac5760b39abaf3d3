"""Where the HOST spends a training step (cProfile over N steps at a BASELINE shape): the step is a chain of ~25 Python-level
operations around the native solve; this lists them by cumulative host time per step."""
import cProfile, io, json, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import fused_mll, settings
from rpgp_amd.training import create_exact_gp, make_optimizer
from rpgp_amd.models import ExactMarginalLogLikelihood
SHAPES = {"C2": (7372, 8, 20, False, False), "C3": (14939, 18, 20, True, False), "C5": (391386, 3, 3, True, True)}
shape = sys.argv[1]; steps = int(sys.argv[2]); warm = 5
N, d, J, sp, ski = SHAPES[shape]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, generator=g); y = torch.sin(X).sum(1) + 0.05 * torch.randn(N, generator=g); y = (y - y.mean()) / y.std()
X, y = X.to(dev), y.to(dev)
torch.manual_seed(0)
import numpy as np
np.random.seed(0)
model, lik = create_exact_gp(X, y, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True,
                             space_proj=sp, ski=ski, ski_options={"grid_size": 1024, "num_dims": 1} if ski else None)
model = model.to(dev); mll = ExactMarginalLogLikelihood(lik, model)
opt = make_optimizer(torch.optim.Adam, [p for p in model.parameters() if p.requires_grad], 0.0)
with settings.cg_tolerance(0.05), settings.max_cg_iterations(10000):
    model.train()
    for it in range(warm):
        opt.zero_grad(); loss = mll.negative_and_backward(model(X), y); opt.step(); fused_mll.loss_value(loss)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for it in range(steps):
        opt.zero_grad(); loss = mll.negative_and_backward(model(X), y); opt.step(); fused_mll.loss_value(loss)
    torch.cuda.synchronize()
    pr.disable()
    t1 = time.perf_counter()
print("step (under cProfile) %.3f ms" % ((t1 - t0) / steps * 1e3))
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("cumulative")
ps.print_stats(70)
out = s.getvalue()
# per-step numbers
print(out)
