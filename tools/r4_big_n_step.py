"""One optimiser step beyond the BASELINE sizes: N = 120 000 (packed cache, 29 GB) and N = 200 000 (no cache fits a quarter of
HBM: the fused T = 11 sweep inside the native executor), step kernels on / off — same objective, gradients, and the time."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rpgp_amd import settings, linear_cg as lcg
from rpgp_amd.training import create_exact_gp
from rpgp_amd.models import ExactMarginalLogLikelihood
dev = torch.device("cuda:0")
for N in (120000, 200000):
    g = torch.Generator().manual_seed(0)
    d, J = 10, 20
    X = torch.randn(N, d, generator=g); y = torch.sin(X).sum(1) + 0.05 * torch.randn(N, generator=g); y = (y - y.mean()) / y.std()
    X, y = X.to(dev), y.to(dev)
    out = {"N": N}
    for mode in (True, False):
        torch.manual_seed(0); np.random.seed(0)
        model, lik = create_exact_gp(X, y, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True)
        model = model.to(dev); mll = ExactMarginalLogLikelihood(lik, model); model.train()
        with settings.step_kernels(mode), settings.deterministic_probes(True), settings.cg_tolerance(0.05), \
                settings.max_cg_iterations(10000):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            loss = -mll(model(X), y); loss.backward(); v = loss.item()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        grads = torch.cat([p.grad.detach().double().reshape(-1).cpu() for p in model.parameters() if p.grad is not None])
        out["kernels" if mode else "torch"] = {"loss": v, "seconds": round(dt, 3), "iters": lcg.stats["last_iterations"],
                                               "grad_norm": float(grads.norm())}
        out["g_" + ("k" if mode else "t")] = grads
        del model, lik, mll
        torch.cuda.empty_cache()
    out["grad_rel_diff"] = float((out["g_k"] - out["g_t"]).norm() / out["g_t"].norm()); del out["g_k"], out["g_t"]
    out["loss_rel_diff"] = abs(out["kernels"]["loss"] - out["torch"]["loss"]) / abs(out["torch"]["loss"])
    print(json.dumps(out), flush=True)
