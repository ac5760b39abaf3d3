"""Optimiser step time with the fused objective (settings.fused_training) on and off, alternating in one process, at the
C2 / C3 / C5 shapes (tools/step_only.py's models; lr 0: every step is the same problem)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rpgp_amd import settings
from rpgp_amd.training import create_exact_gp
from rpgp_amd.models import ExactMarginalLogLikelihood
SHAPES = {"C2": (7372, 8, 20, False, False), "C3": (14939, 18, 20, True, False), "C5": (391386, 3, 3, True, True)}
dev = torch.device("cuda:0")
for shape in sys.argv[1:] or ["C2", "C3", "C5"]:
    N, d, J, sp, ski = SHAPES[shape]
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, d, generator=g); y = torch.sin(X).sum(1) + 0.05 * torch.randn(N, generator=g); y = (y - y.mean()) / y.std()
    X, y = X.to(dev), y.to(dev)
    torch.manual_seed(0); np.random.seed(0)
    model, lik = create_exact_gp(X, y, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True,
                                 space_proj=sp, ski=ski, ski_options={"grid_size": 1024, "num_dims": 1} if ski else None)
    model = model.to(dev); mll = ExactMarginalLogLikelihood(lik, model)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=0.0)
    res = {"shape": shape, "pairs": []}
    with settings.cg_tolerance(0.05), settings.max_cg_iterations(10000), settings.deterministic_probes(True):
        model.train()
        for rep in range(4):
            pair = {}
            for fused in (True, False):
                with settings.fused_training(fused):
                    for it in range(3):
                        opt.zero_grad(); loss = -mll(model(X), y); loss.backward(); opt.step()
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for it in range(20):
                        opt.zero_grad(); loss = -mll(model(X), y); loss.backward(); opt.step()
                    torch.cuda.synchronize()
                    pair["fused_ms" if fused else "generic_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
                    pair["loss_fused" if fused else "loss_generic"] = loss.item()
                    pair["g_fused" if fused else "g_generic"] = [float(p.grad.reshape(-1)[0]) for p in model.parameters() if p.requires_grad]
            res["pairs"].append(pair)
    print(json.dumps(res), flush=True)
