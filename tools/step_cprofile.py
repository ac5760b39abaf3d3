"""cProfile of the host side of optimiser steps at a BASELINE shape (GPU box): python3 tools/step_cprofile.py C2 30"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import settings
from rpgp_amd.training import create_exact_gp
from rpgp_amd.models import ExactMarginalLogLikelihood
SHAPES = {"C2": (7372, 8, 20, False, False), "C3": (14939, 18, 20, True, False), "C5": (391386, 3, 3, True, True)}
shape = sys.argv[1]; steps = int(sys.argv[2])
N, d, J, sp, ski = SHAPES[shape]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, generator=g); y = torch.sin(X).sum(1); y = (y - y.mean()) / y.std()
X, y = X.to(dev), y.to(dev)
torch.manual_seed(0)
import numpy as np
np.random.seed(0)
model, lik = create_exact_gp(X, y, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True,
                             space_proj=sp, ski=ski, ski_options={"grid_size": 1024, "num_dims": 1} if ski else None)
model = model.to(dev); mll = ExactMarginalLogLikelihood(lik, model)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=0.0)
with settings.cg_tolerance(0.05), settings.max_cg_iterations(10000):
    model.train()
    for it in range(5):
        opt.zero_grad(); loss = -mll(model(X), y); loss.backward(); opt.step()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for it in range(steps):
        opt.zero_grad(); loss = -mll(model(X), y); loss.backward(); opt.step()
    torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats(sys.argv[3] if len(sys.argv) > 3 else "cumulative"); ps.print_stats(45)
txt = s.getvalue().replace("/root/repo/", "")
print("\n".join(l[:190] for l in txt.splitlines()))
