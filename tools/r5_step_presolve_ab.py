"""Optimiser-step time at a BASELINE shape with M^-1 [probes] (the first two launches of the backward pass) queued before the host-side SLQ quadrature
(default) against at the head of the backward pass (RPGP_EARLY_PRESOLVE=0); same process, alternating rounds.
python tools/r5_step_presolve_ab.py C2 [rounds] [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from rpgp_amd import settings
from rpgp_amd.training import create_exact_gp, make_optimizer
from rpgp_amd.models import ExactMarginalLogLikelihood
SHAPES = {"C2": (7372, 8, 20, False, False), "C3": (14939, 18, 20, True, False), "C4": (50000, 20, 20, False, False), "C5": (391386, 3, 3, True, True)}
shape = sys.argv[1]; rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
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
def run(n):
    v = None
    for it in range(n):
        opt.zero_grad(); loss = mll.negative(model(X), y); loss.backward(); opt.step(); v = loss.item()
    return v
ts = {"1": [], "0": []}
with settings.cg_tolerance(0.05), settings.max_cg_iterations(10000):
    model.train()
    for m in ("1", "0"):
        os.environ["RPGP_EARLY_PRESOLVE"] = m; run(10)
    torch.cuda.synchronize()
    for r in range(rounds):
        for m in ("0", "1"):
            os.environ["RPGP_EARLY_PRESOLVE"] = m
            t0 = time.perf_counter(); v = run(steps); torch.cuda.synchronize()
            ts[m].append((time.perf_counter() - t0) / steps * 1e6)
rec = {"shape": shape}
for m, name in (("0", "presolve_in_backward"), ("1", "presolve_queued_before_slq")):
    t = sorted(ts[m]); rec[name + "_us_min"] = round(t[0], 1); rec[name + "_us_median"] = round(t[len(t) // 2], 1)
rec["loss"] = v
print(json.dumps(rec))
