"""Generates tests/golden/*.npz by importing the REFERENCE implementation in the build container.

Run (only where /root/reference exists; never on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden.py

What is captured:
  gam.npz          GAMFunction forward/backward (gp_models/kernels/memory_efficient_gam_kernel.py:5-59) on the
                   test.py:641-647 inputs and on seeded random float64 cases (ARD and single lengthscale).
  gen_rp.npz       rp.gen_rp(d, k, dist) (rp.py:10-32) for every `dist` under torch.manual_seed.
  space_equally.npz rp.space_equally (rp.py:220-268) for (J,d) in {(20,18),(20,8),(3,3),(4,2)} under torch+numpy seeds.
The reference's source is NOT copied: only inputs and outputs are stored.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("RPGP_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def _load_reference():
    # stub of the only gpytorch symbol memory_efficient_gam_kernel.py needs at import time
    gpytorch = types.ModuleType("gpytorch")
    kernels = types.ModuleType("gpytorch.kernels")

    class Kernel(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    kernels.Kernel = Kernel
    gpytorch.kernels = kernels
    sys.modules.setdefault("gpytorch", gpytorch)
    sys.modules.setdefault("gpytorch.kernels", kernels)

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    gam = load("ref_gam", "gp_models/kernels/memory_efficient_gam_kernel.py")
    rp = load("ref_rp", "rp.py")
    return gam, rp


def make_gam(gam):
    out = {}
    cases = []
    # the reference's own test inputs (test.py:641-647)
    x1 = torch.tensor([[0., 2., 4.], [3., 4.3, 2.], [6.2, 1.2, 2.2]], dtype=torch.double)
    x2 = torch.tensor([[3., 2., 1.], [5.3, 2.1, 7.1]], dtype=torch.double)
    raw = torch.tensor([1., 3., 2.], dtype=torch.double)
    cases.append(("testpy", x1, x2, torch.nn.functional.softplus(raw)))
    g = torch.Generator().manual_seed(1234)
    for idx, (n, m, d, ard) in enumerate([(5, 7, 3, True), (16, 16, 8, True), (33, 12, 6, False), (64, 64, 8, True)]):
        a = torch.randn(n, d, generator=g, dtype=torch.double) * 1.5
        b = torch.randn(m, d, generator=g, dtype=torch.double) * 1.5
        ls = torch.rand(d if ard else 1, generator=g, dtype=torch.double) * 2 + 0.3
        cases.append(("rand%d" % idx, a, b, ls))
    for name, a, b, ls in cases:
        a = a.clone().requires_grad_(True)
        b = b.clone().requires_grad_(True)
        ls = ls.clone().requires_grad_(True)
        K = gam.GAMFunction.apply(a, b, ls)
        go = torch.randn(K.shape, generator=torch.Generator().manual_seed(7), dtype=torch.double)
        (K * go).sum().backward()
        out[name + "_x1"] = a.detach().numpy()
        out[name + "_x2"] = b.detach().numpy()
        out[name + "_ls"] = ls.detach().numpy()
        out[name + "_K"] = K.detach().numpy()
        out[name + "_go"] = go.numpy()
        out[name + "_gx1"] = a.grad.numpy()
        out[name + "_gx2"] = b.grad.numpy()
        out[name + "_gls"] = ls.grad.numpy()
    out["names"] = np.array([c[0] for c in cases])
    np.savez(os.path.join(OUT, "gam.npz"), **out)


def make_gen_rp(rp):
    out = {}
    for dist in ["gaussian", "sphere", "very-sparse", "bernoulli", "uniform"]:
        for (d, k, seed) in [(8, 1, 0), (20, 1, 1), (18, 3, 2)]:
            torch.manual_seed(seed)
            W = rp.gen_rp(d, k, dist)
            out["%s_d%d_k%d_s%d" % (dist, d, k, seed)] = W.numpy()
    np.savez(os.path.join(OUT, "gen_rp.npz"), **out)


def make_space_equally(rp):
    out = {}
    for (J, d, seed) in [(20, 18, 0), (20, 8, 1), (3, 3, 2), (4, 2, 3)]:
        torch.manual_seed(seed)
        np.random.seed(seed)
        P0 = torch.cat([rp.gen_rp(d, 1, "gaussian") for _ in range(J)], dim=1).t().contiguous()
        out["J%d_d%d_s%d_in" % (J, d, seed)] = P0.clone().numpy()
        niter = 5000 if d < J else 10
        newP, loss = rp.space_equally(P0.clone(), 0.1, niter)
        out["J%d_d%d_s%d_out" % (J, d, seed)] = newP.detach().numpy()
        out["J%d_d%d_s%d_loss" % (J, d, seed)] = np.array(float(loss.item()) if loss is not None else np.nan)
    np.savez(os.path.join(OUT, "space_equally.npz"), **out)


if __name__ == "__main__":
    gam, rp = _load_reference()
    make_gam(gam)
    make_gen_rp(rp)
    make_space_equally(rp)
    print("golden fixtures written to", OUT)
