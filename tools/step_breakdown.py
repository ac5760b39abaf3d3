"""Where one optimiser step goes: wall-clock per phase (host-synchronised) for a training step at a BASELINE shape.
Run on the GPU box: python3 tools/step_breakdown.py C2 [steps].  Phases are timed by wrapping the stack's own entry points
(no change to the product path): kernel evaluation, preconditioner, mBCG solve, SLQ, backward, optimiser."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import settings, linear_cg as lcg, inv_quad_logdet as iql, precond
from rpgp_amd.training import create_exact_gp
from rpgp_amd.models import ExactMarginalLogLikelihood

SHAPES = {"C2": (7372, 8, 20, False, False), "C3": (14939, 18, 20, True, False), "C4": (50000, 20, 20, False, False),
          "C5": (391386, 3, 3, True, True)}
acc = {}


def timed(name, fn):
    def wrap(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return out
    return wrap


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "C2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    N, d, J, sp, ski = SHAPES[shape]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, d, generator=g)
    y = torch.sin(X).sum(1) + 0.05 * torch.randn(N, generator=g)
    y = (y - y.mean()) / y.std()
    X, y = X.to(dev), y.to(dev)
    torch.manual_seed(0)
    import numpy as np
    np.random.seed(0)
    model, lik = create_exact_gp(X, y, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False,
                                 prescale=True, space_proj=sp, ski=ski,
                                 ski_options={"grid_size": 1024, "num_dims": 1} if ski else None)
    model = model.to(dev)
    mll = ExactMarginalLogLikelihood(lik, model)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=0.1)
    from rpgp_amd import operators as opm
    patch = [(iql, "linear_cg"), (iql, "build_preconditioner"), (iql, "slq_logdet"),
             (opm.AdditiveRPOperator, "to_symcache"), (opm.AdditiveRPOperator, "_bilinear_derivative"),
             (opm.SKIAdditiveOperator, "_bilinear_derivative")]
    with settings.cg_tolerance(0.05), settings.max_cg_iterations(10000):
        model.train()
        for it in range(3):
            opt.zero_grad(); loss = -mll(model(X), y); loss.backward(); opt.step()
        torch.cuda.synchronize()
        # unsynchronised wall time first
        t0 = time.perf_counter()
        for it in range(steps):
            opt.zero_grad(); loss = -mll(model(X), y); loss.backward(); opt.step()
        torch.cuda.synchronize()
        free = (time.perf_counter() - t0) / steps
        for owner, n in patch:
            setattr(owner, n, timed(getattr(owner, "__name__", "") .split(".")[-1] + "." + n, getattr(owner, n)))
        ph = {}
        for it in range(steps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            opt.zero_grad()
            out = model(X)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            loss = -mll(out, y)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            loss.backward()
            torch.cuda.synchronize(); t3 = time.perf_counter()
            opt.step()
            torch.cuda.synchronize(); t4 = time.perf_counter()
            for k, v in (("model_call", t1 - t0), ("mll_forward", t2 - t1), ("backward", t3 - t2), ("optimizer", t4 - t3)):
                ph[k] = ph.get(k, 0.0) + v
    res = {"shape": shape, "N": N, "step_ms_unsynchronised": free * 1e3,
           "phases_ms": {k: v / steps * 1e3 for k, v in ph.items()},
           "inside_ms": {k: v / steps * 1e3 for k, v in acc.items()}, "cg_iterations_last": lcg.stats.get("iterations")}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
