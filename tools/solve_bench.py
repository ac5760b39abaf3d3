"""End-to-end timing of the exact-GP solve path on one MI355X: optimiser steps (mBCG + SLQ forward, fused derivative
backward) and the evaluation block (mean cache, predictive mean/variance) on synthetic stand-ins of the BASELINE
configs.  Prints one JSON line per configuration."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import settings, linear_cg as lcg
from rpgp_amd.training import create_exact_gp, make_optimizer
from rpgp_amd.models import ExactMarginalLogLikelihood


def run(name, N, d, J, ntest, steps, space_proj, cg_tol, eval_tol, ski=False, full_cov=True, quiet=False):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N + ntest, d, generator=g)
    y = torch.sin(X).sum(1) + 0.05 * torch.randn(N + ntest, generator=g)
    y = (y - y.mean()) / y.std()
    Xtr, ytr, Xte, yte = X[:N].to(dev), y[:N].to(dev), X[N:].to(dev), y[N:].to(dev)
    torch.manual_seed(0)
    import numpy as np
    np.random.seed(0)            # space_equally draws its orthonormal start from the NumPy RNG (rp.py:228)
    model, lik = create_exact_gp(Xtr, ytr, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False,
                                 prescale=True, space_proj=space_proj, ski=ski,
                                 ski_options={"grid_size": 1024, "num_dims": 1} if ski else None)
    model = model.to(dev)
    mll = ExactMarginalLogLikelihood(lik, model)
    opt = make_optimizer(torch.optim.Adam, [p for p in model.parameters() if p.requires_grad], 0.1)
    res = {"config": name, "N": N, "d": d, "J": J, "N_test": ntest, "ski": ski}
    with settings.cg_tolerance(cg_tol), settings.eval_cg_tolerance(eval_tol), settings.max_cg_iterations(10000):
        model.train()
        times, iters, losses = [], [], []
        for it in range(steps + 1):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            lcg.stats["iterations"] = 0
            opt.zero_grad()
            loss = mll.negative(model(Xtr), ytr)
            loss.backward()
            opt.step()
            torch.cuda.synchronize()
            if it > 0:
                times.append(time.perf_counter() - t0); iters.append(lcg.stats["iterations"])
            losses.append(loss.item())
        st = sorted(times)
        res.update({"train_step_s": sum(times) / len(times), "train_step_median_s": st[len(st) // 2],
                    "train_step_max_s": st[-1], "cg_iters_per_step": sum(iters) / len(iters),
                    "loss_first": losses[0], "loss_last": losses[-1]})
        model.eval()
        with torch.no_grad():
            torch.cuda.synchronize(); t0 = time.perf_counter(); lcg.stats["iterations"] = 0
            with settings.skip_posterior_variances(True):
                out = model(Xte)
                rmse = float(((out.mean - yte) ** 2).mean().sqrt())
            torch.cuda.synchronize(); res["mean_pred_s"] = time.perf_counter() - t0
            res["mean_cache_cg_iters"] = lcg.stats["iterations"]; res["test_rmse"] = rmse
            if full_cov:
                model.train(); model.eval()
                torch.cuda.synchronize(); t0 = time.perf_counter(); lcg.stats["iterations"] = 0
                out = model(Xte)
                nll = -mll(out, yte).item()
                torch.cuda.synchronize(); res["full_pred_s"] = time.perf_counter() - t0
                res["full_pred_cg_iters"] = lcg.stats["iterations"]; res["test_nll"] = nll
    if not quiet:
        print(json.dumps(res))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C2,C3")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cache_kernel", choices=["auto", "on", "off"], default="auto")
    ap.add_argument("--no_warmup", action="store_true", help="time the first configuration cold (library first-use costs included)")
    a = ap.parse_args()
    table = {"C2": ("C2 kin8nm-shaped RPA-GP", 7372, 8, 20, 820, False), "C3": ("C3 elevators-shaped DPA-GP", 14939, 18, 20, 1660, True),
             "C4": ("C4 synthetic 50k RPA-GP", 50000, 20, 20, 2000, False), "S": ("small", 3000, 8, 20, 300, False),
             "C5": ("C5 3droad-shaped DPA-GP + SKI (J=d=3, grid 1024)", 391386, 3, 3, 43488, True),
             # the reference's J = 20 SKI specs (additive_spread_prescale_J20_ski.json) at the C3 / C4 shapes
             "C3S": ("C3 elevators-shaped DPA-GP + SKI (J=20, grid 1024)", 14939, 18, 20, 1660, True),
             "C4S": ("C4 synthetic 50k RPA-GP + SKI (J=20, grid 1024)", 50000, 20, 20, 2000, False)}
    if not a.no_warmup:
        # one small untimed fit + prediction (exact and SKI): the first use of every library in a fresh process (rocBLAS /
        # rocSOLVER / hipBLASLt kernel loading, code-object upload) is paid here, not inside the first timed configuration
        run("warm-up", 3000, 8, 20, 300, 1, False, 0.05, 0.01, quiet=True)
        run("warm-up ski", 3000, 3, 3, 300, 1, True, 0.05, 0.01, ski=True, quiet=True)
    for c in a.configs.split(","):
        name, N, d, J, nt, sp = table[c]
        with settings.cache_kernel({"auto": "auto", "on": True, "off": False}[a.cache_kernel]):
            run(name + " [cache_kernel=%s]" % a.cache_kernel, N, d, J, nt, a.steps, sp, 0.05, 0.01, ski=(c in ("C5", "C3S", "C4S")),
                full_cov=(c != "C5"))
