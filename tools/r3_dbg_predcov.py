import os, sys
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from rpgp_amd import specs, training, settings
name = sys.argv[1]
for seed in range(int(sys.argv[2])):
    g = torch.Generator().manual_seed(100 + seed)
    X = torch.randn(8192, 8, generator=g); y = torch.sin(X).sum(1) + 0.05 * torch.randn(8192, generator=g); y = (y - y.mean()) / y.std()
    spec = specs.get(name); spec["train_kwargs"]["max_iter"] = 60
    torch.manual_seed(seed)
    with settings.cg_tolerance(0.05), settings.eval_cg_tolerance(0.01):
        m, pred, model = training.train_exact_gp(X[:7372], y[:7372], X[7372:], y[7372:], spec["kind"], spec["model_kwargs"],
                                                 spec["train_kwargs"], devices=["cuda:0"], skip_random_restart=True,
                                                 skip_posterior_variances=True)
        model.eval(); model.likelihood.eval()
        with torch.no_grad():
            out = model(X[7372:].cuda())
            cov = out.covariance.double()
            ev = torch.linalg.eigvalsh(cov)
            print(name, seed, "noise %.2e outputscale %.3g diag mean %.3g  min eig %.3e max eig %.3e" % (
                float(model.likelihood.noise), float(model.covar_module.outputscale), float(cov.diagonal().mean()),
                float(ev[0]), float(ev[-1])), flush=True)
