"""`--double` (training_routines.py:481): the float64 HIP kernels through the C-ABI against the float64 oracle — with
both sides in double the parity gates tighten from 1e-4 to ~1e-9."""
import math

import numpy as np
import pytest
import torch

from oracle import dense_gp as orc
from tests.test_host_stack import _build_model, _oracle_gp, _problem

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("N,J,T", [(300, 20, 1), (777, 7, 5), (1000, 3, 11)])
def test_f64_kernels_match_oracle(gpu_device, N, J, T):
    from rpgp_amd import ops
    rng = np.random.default_rng(N)
    Z = rng.standard_normal((N, J))
    Z2 = rng.standard_normal((N // 2, J))
    V = rng.standard_normal((N, T))
    L, R = rng.standard_normal((N, T)), rng.standard_normal((N, T))
    Zt = torch.from_numpy(Z).to(gpu_device)
    out = ops.mvm_sym(Zt, torch.from_numpy(V).to(gpu_device), 0.3, 0.2)
    assert out.dtype == torch.float64
    assert _rel(out.cpu().numpy(), orc.mvm(Z, Z, V, 0.3, 0.2)) < 1e-12
    outr = ops.mvm_rect(torch.from_numpy(Z2).to(gpu_device), Zt, torch.from_numpy(V).to(gpu_device), 0.3)
    assert _rel(outr.cpu().numpy(), orc.mvm(Z2, Z, V, 0.3)) < 1e-12
    Kd = ops.dense(torch.from_numpy(Z2).to(gpu_device), Zt, 0.5)
    np.testing.assert_allclose(Kd.cpu().numpy(), 0.5 * orc.additive_rbf(Z2, Z), rtol=1e-12, atol=1e-13)
    gZ, gs = ops.bilinear_grad(Zt, torch.from_numpy(L).to(gpu_device), torch.from_numpy(R).to(gpu_device), 0.2)
    gZ_ref, gs_ref = orc.bilinear_grad(Z, L, R, 0.2)
    assert _rel(gZ.cpu().numpy(), gZ_ref) < 1e-11 and abs(gs.item() - gs_ref) < 1e-10 * abs(gs_ref) + 1e-9
    X = rng.standard_normal((N, 5))
    P = rng.standard_normal((5, J))
    Zp = ops.project(torch.from_numpy(X).to(gpu_device), torch.from_numpy(P).to(gpu_device))
    assert _rel(Zp.cpu().numpy(), X @ P) < 1e-14
    dP = ops.project_grad(torch.from_numpy(X).to(gpu_device), torch.from_numpy(Z).to(gpu_device))
    assert _rel(dP.cpu().numpy(), X.T @ Z) < 1e-13


@pytest.mark.parametrize("N,chol", [(277, True), (1500, False)])
def test_double_model_parity(gpu_device, N, chol):
    from rpgp_amd import settings
    X, y, P, ls, noise, s = _problem(N=N, d=6, J=20, seed=3, noise=0.25)
    model, lik, mll = _build_model(X.double().to(gpu_device), y.double().to(gpu_device), P.double(), ls.double(), noise, s)
    model = model.to(gpu_device, torch.double)
    # the oracle takes the hyper-parameters exactly as the (float32-initialised) model holds them
    ls = model.covar_module.base_kernel.lengthscale.detach().cpu().reshape(-1)
    s = model.covar_module.outputscale.item()
    noise = lik.noise.item()
    cmean = float(model.mean_module.constant.item())
    ref = orc.DenseExactGP(X.numpy(), y.numpy(), P.numpy(), ls.numpy(), s, noise, mean=cmean)
    model.train()
    with settings.cg_tolerance(1e-10), settings.deterministic_probes(True), settings.skip_logdet_forward(not chol):
        val = mll(model(model.train_inputs), model.train_targets)
        val.backward()
    if chol:
        assert abs(val.item() - ref.mll()) < 1e-10 * abs(ref.mll())
        eps = 1e-6
        raw_n = float(lik.raw_noise.detach())
        f = lambda rn: orc.DenseExactGP(X.numpy(), y.numpy(), P.numpy(), ls.numpy(), s, float(orc.softplus(rn)) + 1e-4,
                                        cmean).mll()
        fd = (f(raw_n + eps) - f(raw_n - eps)) / (2 * eps)
        assert abs(lik.raw_noise.grad.item() - fd) < 1e-6 * abs(fd) + 1e-9
    else:
        n = X.shape[0]
        expect = (-0.5 * ref.inv_quad() - 0.5 * n * math.log(2 * math.pi) + orc.smoothed_box_log_prob(noise)) / n
        assert abs(val.item() - expect) < 1e-8 * abs(expect)
    Xs = torch.randn(40, 6, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    mean_ref, var_ref = ref.predict(Xs.numpy())
    model.eval()
    with torch.no_grad(), settings.eval_cg_tolerance(1e-10):
        out = model(Xs.to(gpu_device))
    assert _rel(out.mean.cpu().numpy(), mean_ref) < 1e-8
    assert _rel(out.variance.cpu().numpy(), var_ref) < 1e-7


def test_train_exact_gp_double_flag(gpu_device):
    from rpgp_amd.training import train_exact_gp
    torch.manual_seed(0)
    X, y, P, ls, noise, s = _problem(N=400, d=4, J=5, seed=8)
    Xs = torch.randn(30, 4)
    ys = torch.sin(Xs).sum(1)
    mk = {"J": 8, "noise_prior": True, "kernel_type": "RBF", "learn_proj": False, "prescale": True}
    tk = {"verbose": False, "optimizer": "adam", "max_iter": 4, "lr": 0.1, "patience": 20, "smooth": True, "init_iters": 1}
    metrics, pred, model = train_exact_gp(X, y, Xs, ys, "additive_rp", mk, tk, devices=("cuda:0",), double=True)
    assert model.covar_module.base_kernel.raw_lengthscale.dtype == torch.float64
    assert pred.dtype == torch.float32 and np.isfinite(metrics["test_nll"])
