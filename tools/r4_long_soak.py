"""Long run of the flagship specification through the runner: 10-fold CV at the C2 shape (every fold rebuilds the model), then
400 optimiser steps in one process with the device / pinned memory watched for growth.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from rpgp_amd import runner, specs, settings
from rpgp_amd.training import create_exact_gp, make_optimizer
from rpgp_amd.models import ExactMarginalLogLikelihood
res = {}
spec = specs.get("additive_rp_prescale_J20")
spec["train_kwargs"]["max_iter"] = 60
json.dump(spec, open("/tmp/long_spec.json", "w"))
t0 = time.time()
df = runner.main(["-m", "/tmp/long_spec.json", "-d", "synthetic:kin8nm", "-o", "/tmp/long_out.csv", "--device", "cuda:0"])
res["cv_rows"], res["cv_seconds"], res["cv_rmse_mean"] = int(len(df)), round(time.time() - t0, 2), float(df["rmse"].mean())
res["cv_nan_rows"] = int(df["rmse"].isna().sum())
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, d, J = 7372, 8, 20
X = torch.randn(N, d, generator=g); y = torch.sin(X).sum(1) + 0.05 * torch.randn(N, generator=g); y = (y - y.mean()) / y.std()
X, y = X.to(dev), y.to(dev)
model, lik = create_exact_gp(X, y, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True)
model = model.to(dev); mll = ExactMarginalLogLikelihood(lik, model)
opt = make_optimizer(torch.optim.Adam, [p for p in model.parameters() if p.requires_grad], 0.05)
mem = []
model.train()
with settings.cg_tolerance(0.05), settings.max_cg_iterations(10000):
    for it in range(400):
        opt.zero_grad(); loss = -mll(model(X), y); loss.backward(); opt.step(); v = loss.item()
        if it in (20, 399):
            torch.cuda.synchronize(); mem.append((torch.cuda.memory_allocated(), torch.cuda.memory_reserved()))
res["loss_final"], res["mem_step20"], res["mem_step399"] = v, mem[0], mem[1]
res["allocated_growth_bytes"] = mem[1][0] - mem[0][0]
print(json.dumps(res))
