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


@pytest.mark.parametrize("rule,weighted", [("shared", False), ("shared", True), ("reference", False), ("reference", True)])
def test_ski_float64_kernels_match_oracle(gpu_device, rule, weighted):
    """rpgp_ski_f64_* (the `--double` path of the `ski: true` specifications) against the float64 dense SKI oracle: product
    (square with noise, rectangular), diagonal, dense block, bilinear derivative incl. the per-projection parts — at float64
    accuracy, for both grid rules and with per-projection output scales."""
    from oracle import ski as sko
    from rpgp_amd import ops
    from tests.oracle_backend import OracleBackend
    rng = np.random.default_rng(7)
    N, M, J, T, G = 900, 300, 4, 5, 128
    Z = rng.standard_normal((N, J)) * np.array([1.0, 0.5, 2.0, 1.3])
    Zs = rng.standard_normal((M, J)) * 0.8
    V, w = rng.standard_normal((N, T)), (rng.uniform(0.3, 1.4, size=J) if weighted else None)
    Zt, Zst, Vt = (torch.from_numpy(a).to(gpu_device) for a in (Z, Zs, V))
    wt = None if w is None else torch.from_numpy(w).to(gpu_device)
    gp = ops.ski_grid(Zt, Zst, G, weights=wt, rule=rule)
    assert gp.dtype == torch.float64
    grid = sko.grid_params(Z, Zs, G) if rule == "shared" else sko.grid_params_reference(Z, Zs, G)
    K = sko.dense_kernel(Z, Z, 0.7, G, grid, w)
    Kx = sko.dense_kernel(Zs, Z, 0.7, G, grid, w)
    out = ops.ski_mvm(Zt, Zt, gp, Vt, 0.7, 0.2, G)
    assert out.dtype == torch.float64 and _rel(out.cpu().numpy(), K @ V + 0.2 * V) < 1e-12
    assert _rel(ops.ski_mvm(Zst, Zt, gp, Vt, 0.7, 0.0, G).cpu().numpy(), Kx @ V) < 1e-12
    assert _rel(ops.ski_diag(Zt, gp, 0.7, G).cpu().numpy(), np.diag(K)) < 1e-12
    assert _rel(ops.ski_dense(Zst, Zt, gp, 0.7, G).cpu().numpy(), Kx) < 1e-12
    L, R = rng.standard_normal((N, T)), rng.standard_normal((N, T))
    gZ, gs, gc = ops.ski_bilinear_grad_comp(Zt, gp, torch.from_numpy(L).to(gpu_device), torch.from_numpy(R).to(gpu_device), 0.7, G)
    obj = sko.bilinear_objective(Z, L, R, 0.7, G, grid, w)
    assert abs(float(gs) - obj / 0.7) < 1e-10 * abs(obj / 0.7) + 1e-9
    assert abs(float(gc.sum()) - float(gs)) < 1e-10 * abs(float(gs)) + 1e-9
    eps = 1e-6
    for (i, j) in [(0, 0), (N // 2, J - 1), (N - 1, 1)]:
        Zp, Zm = Z.copy(), Z.copy()
        Zp[i, j] += eps
        Zm[i, j] -= eps
        fd = (sko.bilinear_objective(Zp, L, R, 0.7, G, grid, w) - sko.bilinear_objective(Zm, L, R, 0.7, G, grid, w)) / (2 * eps)
        assert abs(float(gZ[i, j]) - fd) < 1e-5 * abs(fd) + 1e-6


def test_ski_model_in_double_matches_the_float32_model(gpu_device):
    """`--double` with `ski: true` end to end (train_exact_gp, CG regime): the float64 model's objective, gradients and
    predictions against the float32 model's at the same parameters (interpolation and solver tolerances apart, the same GP)."""
    from rpgp_amd import settings
    from rpgp_amd.training import create_exact_gp
    from rpgp_amd.models import ExactMarginalLogLikelihood
    gen = torch.Generator().manual_seed(2)
    N, d, J = 2500, 5, 3
    X = torch.randn(N, d, generator=gen)
    y = torch.sin(X).sum(1) + 0.1 * torch.randn(N, generator=gen)
    y = (y - y.mean()) / y.std()
    Xs = torch.randn(40, d, generator=gen)
    res = {}
    for dt in (torch.float32, torch.float64):
        torch.manual_seed(4)
        np.random.seed(4)
        model, lik = create_exact_gp(X.to(gpu_device, dt), y.to(gpu_device, dt), "additive_rp", J=J, noise_prior=True,
                                     kernel_type="RBF", learn_proj=False, prescale=True, space_proj=True, ski=True,
                                     ski_options={"grid_size": 256, "num_dims": 1})
        model = model.to(gpu_device, dt)
        mll = ExactMarginalLogLikelihood(lik, model)
        model.train()
        with settings.deterministic_probes(True), settings.cg_tolerance(1e-5), settings.max_cg_iterations(2000):
            loss = -mll(model(model.train_inputs), model.train_targets)
            loss.backward()
        grads = {k: p.grad.detach().double().cpu().reshape(-1) for k, p in model.named_parameters() if p.grad is not None}
        model.eval()
        with torch.no_grad(), settings.eval_cg_tolerance(1e-6), settings.max_cg_iterations(2000):
            out = model(Xs.to(gpu_device, dt))
            res[dt] = (loss.item(), grads, out.mean.double().cpu(), out.variance.double().cpu())
        assert all(p.dtype == dt for p in model.parameters())
    a, b = res[torch.float32], res[torch.float64]
    assert abs(a[0] - b[0]) < 2e-2 * abs(b[0])          # (SLQ with ten float32 / float64 probe draws: different random numbers)
    assert (a[2] - b[2]).norm() < 2e-3 * b[2].norm() and (a[3] - b[3]).abs().max() < 2e-3 * b[3].abs().max() + 1e-4
    for k in b[1]:
        if "noise" in k or "outputscale" in k or "constant" in k:
            continue                                    # (log-det gradients carry the probe noise)
        assert (a[1][k] - b[1][k]).norm() < 0.15 * b[1][k].norm() + 1e-3
