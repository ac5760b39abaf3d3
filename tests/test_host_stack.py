"""Host-side solve stack on CPU with the oracle test double as compute backend: mBCG, SLQ, pivoted Cholesky,
inv_quad_logdet autograd, MLL, predictions, training loop.  (The GPU versions of the same checks run in
test_gp_gpu.py against the real HIP backend.)"""
import math
import warnings

import numpy as np
import pytest
import torch

from oracle import dense_gp as orc


def _problem(N=120, d=5, J=7, seed=0, noise=0.3, s=0.9):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(N, d, generator=g)
    P = torch.randn(d, J, generator=g)
    ls = torch.rand(d, generator=g) * 1.5 + 1.0
    y = torch.sin(X).sum(1) + 0.05 * torch.randn(N, generator=g)
    return X, y, P, ls, noise, s


def test_linear_cg_solves_spd_system_and_tridiag_gives_logdet():
    from rpgp_amd.linear_cg import linear_cg
    from rpgp_amd.inv_quad_logdet import slq_logdet
    torch.manual_seed(0)
    n = 200
    A = torch.randn(n, n, dtype=torch.float64)
    A = A @ A.t() / n + 0.5 * torch.eye(n, dtype=torch.float64)
    B = torch.randn(n, 6, dtype=torch.float64)
    X = linear_cg(lambda v: A @ v, B, tolerance=1e-10, max_iter=1000)
    assert torch.allclose(A @ X, B, atol=1e-7)
    # SLQ: many unit-norm Gaussian probes, full Lanczos depth -> log-det within a few percent
    probes = torch.randn(n, 64, dtype=torch.float64)
    probes = probes / probes.norm(dim=0, keepdim=True)
    _, T = linear_cg(lambda v: A @ v, probes, n_tridiag=64, tolerance=1e-12, max_iter=60, max_tridiag_iter=60)
    est = float(slq_logdet(T, n))
    exact = float(torch.logdet(A))
    assert abs(est - exact) < 0.05 * abs(exact) + 2.0


def test_linear_cg_warns_when_not_converged_and_handles_zero_rhs():
    from rpgp_amd.linear_cg import linear_cg, NumericalWarning
    torch.manual_seed(1)
    n = 300
    A = torch.randn(n, n, dtype=torch.float64)
    A = A @ A.t() + 1e-3 * torch.eye(n, dtype=torch.float64)
    b = torch.randn(n, 2, dtype=torch.float64)
    b[:, 1] = 0
    with pytest.warns(NumericalWarning):
        x = linear_cg(lambda v: A @ v, b, tolerance=1e-12, max_iter=12)
    assert torch.isfinite(x).all() and float(x[:, 1].abs().max()) == 0.0
    v = linear_cg(lambda z: A @ z, b[:, 0], tolerance=1e-8, max_iter=2000)   # vector rhs
    assert v.shape == (n,)


def test_pivoted_cholesky_and_woodbury(oracle_backend):
    from rpgp_amd.operators import AdditiveRPOperator
    from rpgp_amd.precond import pivoted_cholesky, WoodburyPreconditioner
    torch.manual_seed(0)
    Z = torch.randn(150, 3) * 0.5
    op = AdditiveRPOperator(Z, None, torch.tensor(1.3), weight=1.0 / 3)
    K = op.to_dense().double()
    L = pivoted_cholesky(op._diagonal(), op._get_rows, 15).double()
    assert L.shape == (150, 15)
    resid = K - L @ L.t()
    assert float(resid.diagonal().min()) > -1e-5                   # partial Cholesky never overshoots the diagonal
    assert float(resid.abs().max()) < float(K.abs().max())          # and reduces the error
    L40 = pivoted_cholesky(op._diagonal(), op._get_rows, 40).double()
    assert float((K - L40 @ L40.t()).abs().max()) < float(resid.abs().max())
    pre = WoodburyPreconditioner(L.float(), 0.2)
    M = (L @ L.t() + 0.2 * torch.eye(150, dtype=torch.float64))
    r = torch.randn(150, 4)
    assert torch.allclose(pre.solve(r).double(), torch.linalg.solve(M, r.double()), atol=2e-4)
    assert abs(pre.logdet() - float(torch.logdet(M))) < 1e-3 * abs(float(torch.logdet(M))) + 1e-3
    z = pre.sample(20000, generator=torch.Generator().manual_seed(0)).double()
    emp = z @ z.t() / 20000
    assert float((emp - M).abs().max()) < 0.15


def _build_model(X, y, P, ls, noise, s, prescale=True):
    from rpgp_amd.kernels import AdditiveStructureRBFKernel, ScaledProjectionKernel, ScaleKernel
    from rpgp_amd.likelihoods import GaussianLikelihood, SmoothedBoxPrior
    from rpgp_amd.models import ExactGPModel, ExactMarginalLogLikelihood
    d, J = P.shape
    lin = torch.nn.Linear(d, J, bias=False)
    lin.weight.data = P.t().contiguous()
    k = ScaledProjectionKernel(lin, AdditiveStructureRBFKernel(J), prescale=prescale, ard_num_dims=d if prescale else J)
    k.initialize(lengthscale=ls)
    sk = ScaleKernel(k)
    sk.outputscale = s
    lik = GaussianLikelihood(noise_prior=SmoothedBoxPrior(1e-4, 10, sigma=0.01))
    lik.noise = noise
    model = ExactGPModel(X, y, lik, sk)
    model.mean_module.constant.data.fill_(0.2)
    return model, lik, ExactMarginalLogLikelihood(lik, model)


def _oracle_gp(X, y, P, ls, noise, s, prescale=True):
    return orc.DenseExactGP(X.numpy(), y.numpy(), P.numpy(), ls.numpy(), s, noise, mean=0.2, prescale=prescale)


@pytest.mark.parametrize("prescale", [True, False])
def test_mll_value_and_gradients_cholesky_regime(oracle_backend, prescale):
    X, y, P, ls, noise, s = _problem()
    if not prescale:
        ls = torch.rand(P.shape[1], generator=torch.Generator().manual_seed(5)) + 1.0
    model, lik, mll = _build_model(X, y, P, ls, noise, s, prescale)
    model.train()
    val = mll(model(X), y)
    ref = _oracle_gp(X, y, P, ls, noise, s, prescale)
    assert abs(val.item() - ref.mll()) < 1e-4 * abs(ref.mll())
    val.backward()
    # float64 autograd reference of the same objective
    Xd, yd, Pd = X.double(), y.double(), P.double()
    raw_ls = model.covar_module.base_kernel.raw_lengthscale.detach().double().clone().requires_grad_(True)
    raw_s = model.covar_module.raw_outputscale.detach().double().clone().requires_grad_(True)
    raw_n = lik.raw_noise.detach().double().clone().requires_grad_(True)
    c = model.mean_module.constant.detach().double().clone().requires_grad_(True)
    l = torch.nn.functional.softplus(raw_ls).reshape(-1)
    Z = (Xd / l) @ Pd if prescale else (Xd @ Pd) / l
    K = torch.zeros(X.shape[0], X.shape[0], dtype=torch.float64)
    for j in range(P.shape[1]):
        dd = Z[:, j:j + 1] - Z[:, j:j + 1].t()
        K = K + torch.exp(-0.5 * dd * dd)
    n = X.shape[0]
    Kh = torch.nn.functional.softplus(raw_s) / P.shape[1] * K + (torch.nn.functional.softplus(raw_n) + 1e-4) * torch.eye(n, dtype=torch.float64)
    r = yd - c
    obj = (-0.5 * r @ torch.linalg.solve(Kh, r) - 0.5 * torch.logdet(Kh) - 0.5 * n * math.log(2 * math.pi)) / n
    obj.backward()
    got = model.covar_module.base_kernel.raw_lengthscale.grad.double()
    assert torch.allclose(got, raw_ls.grad, rtol=2e-3, atol=1e-6)
    assert torch.allclose(model.covar_module.raw_outputscale.grad.double(), raw_s.grad, rtol=2e-3, atol=1e-6)
    assert torch.allclose(lik.raw_noise.grad.double(), raw_n.grad, rtol=2e-3, atol=1e-6)
    assert torch.allclose(model.mean_module.constant.grad.double(), c.grad, rtol=2e-3, atol=1e-6)
    # frozen parameters stay frozen (test.py:597-598)
    assert model.covar_module.base_kernel.projection_module.weight.grad is None


def test_mll_cg_regime_matches_exact_with_tight_tolerance(oracle_backend):
    from rpgp_amd import settings
    X, y, P, ls, noise, s = _problem(N=260, seed=3, noise=0.2)
    model, lik, mll = _build_model(X, y, P, ls, noise, s)
    ref = _oracle_gp(X, y, P, ls, noise, s)
    model.train()
    with settings.max_cholesky_size(0), settings.cg_tolerance(1e-7), settings.num_trace_samples(60), \
            settings.max_lanczos_quadrature_iterations(60), settings.deterministic_probes(True), \
            settings.min_preconditioning_size(100):
        val = mll(model(X), y)
        val.backward()
    # inverse-quadratic part is deterministic: exact to CG tolerance; the SLQ log-det carries probe noise
    n = X.shape[0]
    exact = ref.mll()
    assert abs(val.item() - exact) < 0.03 * abs(exact)
    g_noise_cg = lik.raw_noise.grad.item()
    lik.raw_noise.grad = None
    model.zero_grad()
    val2 = mll(model(X), y)      # Cholesky regime (N <= 800)
    val2.backward()
    assert abs(g_noise_cg - lik.raw_noise.grad.item()) < 0.15 * abs(lik.raw_noise.grad.item()) + 1e-4


def test_skip_logdet_forward_and_preconditioner_off(oracle_backend):
    from rpgp_amd import settings
    X, y, P, ls, noise, s = _problem(N=150, seed=4)
    model, lik, mll = _build_model(X, y, P, ls, noise, s)
    ref = _oracle_gp(X, y, P, ls, noise, s)
    model.train()
    with settings.max_cholesky_size(0), settings.cg_tolerance(1e-8), settings.skip_logdet_forward(True), \
            settings.deterministic_probes(True):
        val = mll(model(X), y).item()
    n = X.shape[0]
    expect = (-0.5 * ref.inv_quad() - 0.5 * n * math.log(2 * math.pi) + orc.smoothed_box_log_prob(noise)) / n
    assert abs(val - expect) < 1e-4 * abs(expect)


def test_predictions_match_oracle(oracle_backend):
    from rpgp_amd import settings
    X, y, P, ls, noise, s = _problem(N=140, seed=7)
    Xs = torch.randn(37, X.shape[1], generator=torch.Generator().manual_seed(9))
    ys = torch.sin(Xs).sum(1)
    model, lik, mll = _build_model(X, y, P, ls, noise, s)
    ref = _oracle_gp(X, y, P, ls, noise, s)
    mean_ref, cov_ref = ref.predict(Xs.numpy(), full_cov=True)
    for chol in (True, False):
        model.train()
        model.eval()
        ctx = settings.max_cholesky_size(800 if chol else 0)
        with ctx, settings.eval_cg_tolerance(1e-8), torch.no_grad():
            out = model(Xs)
            np.testing.assert_allclose(out.mean.numpy(), mean_ref, rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(out.covariance_matrix.numpy(), cov_ref, rtol=1e-3, atol=2e-5)
            nll = -mll(out, ys).item()
            assert abs(nll - ref.test_nll(Xs.numpy(), ys.numpy())) < 1e-3 * abs(nll) + 1e-4
            lower, upper = lik(out).confidence_region()
            sd = np.sqrt(np.diag(cov_ref) + noise)
            np.testing.assert_allclose(lower.numpy(), mean_ref - 2 * sd, rtol=1e-3, atol=1e-4)
    model.train()
    model.eval()
    with settings.skip_posterior_variances(True), torch.no_grad():
        out = model(Xs)
        np.testing.assert_allclose(out.mean.numpy(), mean_ref, rtol=1e-4, atol=1e-5)
        assert float(out.variance.abs().max()) == 0.0


def test_train_mode_requires_train_inputs(oracle_backend):
    X, y, P, ls, noise, s = _problem(N=30)
    model, _, _ = _build_model(X, y, P, ls, noise, s)
    model.train()
    with pytest.raises(RuntimeError):
        model(torch.randn(5, X.shape[1]))


def test_gradient_step_moves_only_trainable_parameters(oracle_backend):
    """test.py:575-621: after one Adam step the projection stays fixed unless learn_proj, the outer lengthscale moves."""
    from rpgp_amd.kernels import AdditiveStructureRBFKernel, ScaledProjectionKernel
    from rpgp_amd.likelihoods import GaussianLikelihood
    from rpgp_amd.models import ExactGPModel, ExactMarginalLogLikelihood
    x = torch.tensor([[1., 2., 3.], [1.1, 2.2, 3.3]])
    y = torch.sin(x).sum(dim=1)
    for learn in (False, True):
        lin = torch.nn.Linear(3, 3, bias=False)
        lin.weight.data = torch.eye(3)
        k = ScaledProjectionKernel(lin, AdditiveStructureRBFKernel(3, weight=1.0), prescale=True, ard_num_dims=3,
                                   learn_proj=learn)
        k.initialize(lengthscale=torch.tensor([1., 2., 3.]))
        np.testing.assert_allclose(k.lengthscale.detach().numpy().ravel(), [1., 2., 3.], rtol=1e-6)
        model = ExactGPModel(x, y, GaussianLikelihood(), k)
        mll = ExactMarginalLogLikelihood(model.likelihood, model)
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=0.1)
        opt.zero_grad()
        model.train()
        loss = -mll(model(x), y)
        loss.backward()
        opt.step()
        moved_proj = not np.allclose(lin.weight.detach().numpy(), np.eye(3))
        assert moved_proj == learn
        assert not np.allclose(k.lengthscale.detach().numpy().ravel(), [1., 2., 3.])
        assert float(k.base_kernel.inner_lengthscale) == 1.0


def test_train_to_convergence_semantics(oracle_backend):
    from rpgp_amd.training import train_to_convergence
    X, y, P, ls, noise, s = _problem(N=40, seed=2)
    model, lik, mll = _build_model(X, y, P, ls, noise, s)

    class FlatObjective(torch.nn.Module):
        def forward(self, output, target):
            return 0.0 * sum(p.sum() for p in model.parameters() if p.requires_grad) + 1.0

    # a flat loss converges at the earliest possible epoch 2*patience-1 (moving average undefined before that)
    it = train_to_convergence(model, X, y, optimizer=torch.optim.Adam, objective=FlatObjective(), max_iter=100,
                              patience=20, isloss=True)
    assert it == 39
    it = train_to_convergence(model, X, y, optimizer=torch.optim.Adam, objective=FlatObjective(), max_iter=30,
                              patience=20, isloss=True, check_conv=False)
    assert it == 30
    it = train_to_convergence(model, X, y, optimizer=torch.optim.Adam, objective=FlatObjective(), max_iter=60,
                              patience=5, isloss=True, smooth=False)
    assert it == 5
    # a real objective decreases the loss and `checkpoint` restores the best state
    model.train()
    before = -mll(model(X), y).item()
    train_to_convergence(model, X, y, optimizer=torch.optim.Adam, objective=mll, max_iter=15, lr=0.1, checkpoint=True)
    model.train()
    after = -mll(model(X), y).item()
    assert after < before


def test_train_exact_gp_contract(oracle_backend):
    from rpgp_amd.training import train_exact_gp
    torch.manual_seed(0)
    np.random.seed(0)
    X, y, P, ls, noise, s = _problem(N=90, d=4, seed=11)
    Xs = torch.randn(20, 4)
    ys = torch.sin(Xs).sum(1)
    mk = {"J": "d", "noise_prior": True, "kernel_type": "RBF", "learn_proj": False, "prescale": True,
          "space_proj": True, "batch_kernel": False, "mem_efficient": True}
    tk = {"verbose": False, "optimizer": "adam", "max_iter": 6, "lr": 0.1, "patience": 20, "smooth": True,
          "init_iters": 2}
    metrics, pred, model = train_exact_gp(X, y, Xs, ys, "additive_rp", mk, tk, record_pred_unc=True)
    for key in ["trained_epochs", "prior_train_nmll", "train_mse", "train_nll", "test_nll", "test_pred_frac_in_cr",
                "test_pred_z_score", "training_warnings", "testing_warning", "state_dict_file"]:
        assert key in metrics
    assert metrics["trained_epochs"] == 6
    assert pred.shape == (20,) and pred.dtype == torch.float32 and pred.device.type == "cpu"
    assert model.covar_module.base_kernel.projection_module.weight.shape == (4, 4)       # "J": "d" placeholder
    gam = model.covar_module.base_kernel.base_kernel
    assert abs(float(gam.lengthscale) - math.log(2)) < 1e-6 and not gam.raw_lengthscale.requires_grad
    m2, _, _ = train_exact_gp(X, y, Xs, ys, "additive_rp", mk, tk, skip_posterior_variances=True,
                              skip_random_restart=True, evaluate_on_train=False)
    assert "test_nll" not in m2 and "train_mse" not in m2
    with pytest.raises(ValueError):
        train_exact_gp(X, y, Xs, ys, "nonsense", mk, tk)
    with pytest.raises(ValueError):
        train_exact_gp(X, y, Xs, ys, "additive_rp", dict(mk, k=2), tk)


def test_kernel_factory_validation_errors():
    from rpgp_amd.training import create_additive_rp_kernel, _map_to_optim
    with pytest.raises(ValueError):
        create_additive_rp_kernel(5, 3, k=2)                                  # batch kernel with k > 1
    with pytest.raises(ValueError):
        create_additive_rp_kernel(5, 3, mem_efficient=True)                   # batch_kernel default True
    with pytest.raises(ValueError):
        create_additive_rp_kernel(5, 3, mem_efficient=True, batch_kernel=False, kernel_type="Matern")
    with pytest.raises(ValueError):
        create_additive_rp_kernel(5, 3, mem_efficient=True, batch_kernel=False, ski=True)
    with pytest.raises(ValueError):
        create_additive_rp_kernel(5, 3, kernel_type="bogus")
    with pytest.raises(ValueError):
        _map_to_optim("rmsprop")
    k = create_additive_rp_kernel(6, 4, prescale=True)
    assert k.raw_lengthscale.shape == (1, 6)
    k = create_additive_rp_kernel(6, 4, prescale=False)
    assert k.raw_lengthscale.shape == (1, 4)
    k = create_additive_rp_kernel(6, 4, ard=False)
    assert k.raw_lengthscale.shape == (1, 1)
    assert not k.projection_module.weight.requires_grad


def test_ski_model_matches_dense_ski_oracle(oracle_backend):
    """`ski: true` spec path end to end on the host stack: MLL (Cholesky and CG regimes) against a dense float64
    evaluation of the same SKI kernel, and closeness to the exact-kernel MLL."""
    from rpgp_amd import settings
    from rpgp_amd.kernels import AdditiveStructureRBFKernel, ScaledProjectionKernel, ScaleKernel
    from rpgp_amd.likelihoods import GaussianLikelihood, SmoothedBoxPrior
    from rpgp_amd.models import ExactGPModel, ExactMarginalLogLikelihood
    from oracle import ski as sko
    X, y, P, ls, noise, s = _problem(N=150, d=3, J=3, seed=21)
    lin = torch.nn.Linear(3, 3, bias=False)
    lin.weight.data = P.t().contiguous()
    k = ScaledProjectionKernel(lin, AdditiveStructureRBFKernel(3, ski=True, ski_options={"grid_size": 256, "num_dims": 1}),
                               prescale=True, ard_num_dims=3)
    k.initialize(lengthscale=ls)
    sk = ScaleKernel(k)
    sk.outputscale = s
    lik = GaussianLikelihood(noise_prior=SmoothedBoxPrior(1e-4, 10, sigma=0.01))
    lik.noise = noise
    model = ExactGPModel(X, y, lik, sk)
    mll = ExactMarginalLogLikelihood(lik, model)
    model.train()
    val = mll(model(X), y)
    val.backward()
    Z = orc.project(X.numpy(), P.numpy(), ls.numpy())
    Kd = sko.dense_kernel(Z, Z, s / 3, 256) + noise * np.eye(150)
    r = y.numpy().astype(np.float64)
    ref = (-0.5 * r @ np.linalg.solve(Kd, r) - 0.5 * np.linalg.slogdet(Kd)[1] - 75 * math.log(2 * math.pi)
           + orc.smoothed_box_log_prob(noise)) / 150
    assert abs(val.item() - ref) < 1e-4 * abs(ref)
    exact = orc.DenseExactGP(X.numpy(), y.numpy(), P.numpy(), ls.numpy(), s, noise).mll()
    assert abs(val.item() - exact) < 1e-3 * abs(exact)
    assert model.covar_module.base_kernel.raw_lengthscale.grad is not None
    g_chol = model.covar_module.base_kernel.raw_lengthscale.grad.clone()
    model.zero_grad()
    with settings.max_cholesky_size(0), settings.cg_tolerance(1e-7), settings.num_trace_samples(80), \
            settings.max_lanczos_quadrature_iterations(60), settings.deterministic_probes(True):
        v2 = mll(model(X), y)
        v2.backward()
    assert abs(v2.item() - ref) < 0.03 * abs(ref)
    g_cg = model.covar_module.base_kernel.raw_lengthscale.grad
    assert torch.allclose(g_cg, g_chol, rtol=0.3, atol=5e-3)
    # predictions through the SKI cross operator
    Xs = torch.randn(20, 3, generator=torch.Generator().manual_seed(3))
    model.eval()
    with torch.no_grad():
        out = model(Xs)
    ex_mean, ex_var = orc.DenseExactGP(X.numpy(), y.numpy(), P.numpy(), ls.numpy(), s, noise).predict(Xs.numpy())
    np.testing.assert_allclose(out.mean.numpy(), ex_mean, rtol=5e-3, atol=5e-3)
    np.testing.assert_allclose(out.variance.numpy(), ex_var, rtol=2e-2, atol=1e-3)


def test_fast_pred_var_love(oracle_backend):
    """`--fast_pred` (LOVE): with a full-rank Lanczos root the predictive covariance equals the exact one; with a low
    rank it is a conservative (larger-variance) approximation that improves with the rank."""
    from rpgp_amd import settings
    X, y, P, ls, noise, s = _problem(N=120, seed=13)
    Xs = torch.randn(25, X.shape[1], generator=torch.Generator().manual_seed(4))
    model, lik, mll = _build_model(X, y, P, ls, noise, s)
    ref = _oracle_gp(X, y, P, ls, noise, s)
    mean_ref, cov_ref = ref.predict(Xs.numpy(), full_cov=True)
    errs = []
    for rank in (120, 40, 10):
        model.train()
        model.eval()
        with settings.max_cholesky_size(0), settings.eval_cg_tolerance(1e-8), settings.fast_pred_var(True), \
                settings.max_root_decomposition_size(rank), torch.no_grad():
            out = model(Xs)
        np.testing.assert_allclose(out.mean.numpy(), mean_ref, rtol=1e-4, atol=1e-5)
        cov = out.covariance_matrix.numpy()
        errs.append(np.abs(cov - cov_ref).max())
        assert (np.diag(cov) - np.diag(cov_ref)).min() > -1e-4          # never under-estimates the variance
    assert errs[0] < 1e-4 and errs[0] <= errs[1] + 1e-6 <= errs[2] + 2e-6


def test_linear_cg_gives_up_on_stagnation_instead_of_running_to_max_iter():
    """A system whose fp32 residual floor lies above the tolerance: the torch-op loop stops after
    `cg_stagnation_window` tests without progress and still warns about non-convergence."""
    from rpgp_amd import linear_cg as lcg, settings
    g = torch.Generator().manual_seed(0)
    Q, _ = torch.linalg.qr(torch.randn(200, 200, generator=g, dtype=torch.float64))
    A = ((Q * torch.logspace(0, 9, 200, dtype=torch.float64)) @ Q.t()).float()      # condition number 1e9 in fp32
    b = torch.randn(200, 3, generator=g)
    with settings.cg_stagnation_window(25), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        lcg.stats["stagnated"] = 0
        lcg.linear_cg(lambda v: A @ v, b, tolerance=1e-7, max_iter=100000)
    assert lcg.stats["stagnated"] == 1 and lcg.stats["last_iterations"] < 20000
    assert any("CG terminated" in str(x.message) for x in w)


@pytest.mark.parametrize("kernel_type", ["RBF", "Matern", "InverseMQ"])
def test_full_kernel_types_match_dense_formulas(kernel_type):
    """`kind: full` (ARD_model_spec.json, Inverse_MQ_ARD_model_spec.json; training_routines.py:275-293): the dense torch path
    with the reference's other full kernels against the float64 formulas of oracle/family.py."""
    from oracle import family as fmo
    from rpgp_amd.training import create_exact_gp
    from rpgp_amd.models import ExactMarginalLogLikelihood
    g = torch.Generator().manual_seed(0)
    X = torch.randn(60, 3, generator=g)
    y = torch.sin(X).sum(1)
    torch.manual_seed(0)
    model, lik = create_exact_gp(X, y, "full", noise_prior=False, kernel_type=kernel_type, ard=True,
                                 init_lengthscale_range=(0.7, 1.5))
    mll = ExactMarginalLogLikelihood(lik, model)
    model.train()
    val = float(mll(model(X), y).detach())
    ls = model.covar_module.base_kernel.lengthscale.detach().double().reshape(-1).numpy()
    s, noise, c = float(model.covar_module.outputscale.detach()), float(lik.noise.detach()), float(model.mean_module.constant.detach())
    Z = X.double().numpy() / ls
    K = fmo.kernel_matrix(Z, Z, "RBF", 3, [1.0], s) if kernel_type == "RBF" else None
    if K is None:
        d2 = ((Z[:, None, :] - Z[None, :, :]) ** 2).sum(-1)
        K = s * fmo._phi(kernel_type, d2)
    K = K + noise * np.eye(60)
    r = y.double().numpy() - c
    ref = (-0.5 * r @ np.linalg.solve(K, r) - 0.5 * np.linalg.slogdet(K)[1] - 30 * math.log(2 * math.pi)) / 60
    assert abs(val - ref) < 1e-4 * max(1.0, abs(ref))
    with pytest.raises(ValueError):
        create_exact_gp(X, y, "full", noise_prior=False, kernel_type="bogus")


def test_cached_modes_match_fused_solve(oracle_backend):
    """settings.cache_kernel(True): training goes through SymCachedOperator (packed symmetric cache), prediction through
    the thin cache + the dense matrix for the wide covariance solve; same MLL, gradients and predictions as the fused
    operator."""
    from rpgp_amd import settings
    X, y, P, ls, noise, sc = _problem(N=80, d=4, J=5, seed=3)
    Xs = torch.randn(9, 4, generator=torch.Generator().manual_seed(9))
    res = []
    for cached in (False, True):
        model, lik, mll = _build_model(X, y, P, ls, noise, sc)
        model.train()
        n0 = oracle_backend.calls.get("symcache_mvm", 0)
        with settings.max_cholesky_size(0), settings.cg_tolerance(1e-7), settings.eval_cg_tolerance(1e-7), \
                settings.deterministic_probes(True), settings.cache_kernel(cached):
            v = mll(model(X), y)
            v.backward()
            g = model.covar_module.base_kernel.raw_lengthscale.grad.clone()
            model.eval()
            with torch.no_grad():
                out = model(Xs)
                mean, var = out.mean.clone(), out.variance.clone()
        assert (oracle_backend.calls.get("symcache_mvm", 0) > n0) == cached
        res.append((v.item(), g, mean, var))
    assert abs(res[0][0] - res[1][0]) < 1e-6 * abs(res[0][0])
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-4, atol=1e-7)
    assert torch.allclose(res[0][2], res[1][2], rtol=1e-4, atol=1e-6)
    assert torch.allclose(res[0][3], res[1][3], rtol=1e-3, atol=1e-6)


def test_woodbury_capacitance_survives_large_kernel_norm():
    """|L^T L| ~ 1e6 with noise 0.3 in float32: the capacitance matrix must be accumulated in float64, or the Woodbury
    formula stops inverting M = L L^T + noise I (round-2 regression at the C5 shape)."""
    from rpgp_amd.precond import WoodburyPreconditioner
    g = torch.Generator().manual_seed(0)
    N, k, noise = 60000, 15, 0.3
    L = (torch.randn(N, k, generator=g) * 4.0 + 3.0).float()          # columns far from orthogonal, |L^T L| ~ 1.5e6
    pre = WoodburyPreconditioner(L, noise)
    v = torch.randn(N, 3, generator=g).float()
    Ld = L.double()
    Mv = Ld @ (Ld.t() @ v.double()) + noise * v.double()
    back = pre.solve(Mv.float()).double()
    # M^-1 (M v) = v up to the float32 representation of M v (relative 6e-8 of |M v| ~ 1e6 |v| -> ~0.1 / noise); what must
    # hold tightly is the float64 identity on a float32-representable right-hand side:
    r32 = Mv.float()
    exact = torch.linalg.solve(Ld.t() @ Ld + noise * torch.eye(k, dtype=torch.float64), Ld.t() @ r32.double())
    ref = (r32.double() - Ld @ exact) / noise
    assert float((back - ref).norm() / ref.norm()) < 1e-5
    assert abs(pre.logdet() - float(torch.logdet(Ld.t() @ Ld + noise * torch.eye(k, dtype=torch.float64)) + (N - k) * np.log(noise))) < 1e-6 * N


def test_row_sharded_woodbury_and_wide_blocks_keep_the_cancellation_in_float64():
    """RowShardedWoodbury.solve and the wide (T > 64) branch of WoodburyPreconditioner.solve subtracted
    r - L t in float32; along range(L) that difference is ~ sigma^2 / |K| of r (1e-6 here).  Same system as above
    (|L^T L| / sigma^2 ~ 5e6); also 1-D right-hand sides at N >= 32768 (gram64 indexed B.shape[1])."""
    from rpgp_amd.distributed import RowShard
    from rpgp_amd.operators import RowShardedWoodbury
    from rpgp_amd.precond import WoodburyPreconditioner
    g = torch.Generator().manual_seed(1)
    N, k, noise = 40000, 15, 0.3
    L = (torch.randn(N, k, generator=g) * 4.0 + 3.0).float()
    Ld = L.double()
    v = torch.randn(N, 70, generator=g).float()                      # 70 columns: the wide branch (two panels)
    r32 = (Ld @ (Ld.t() @ v.double()) + noise * v.double()).float()
    exact = torch.linalg.solve(Ld.t() @ Ld + noise * torch.eye(k, dtype=torch.float64), Ld.t() @ r32.double())
    ref = (r32.double() - Ld @ exact) / noise
    for pre in (WoodburyPreconditioner(L, noise), RowShardedWoodbury(L, noise, RowShard(N))):
        back = pre.solve(r32).double()
        assert float((back - ref).norm() / ref.norm()) < 1e-5, type(pre).__name__
        one = pre.solve(r32[:, 3].contiguous())                      # 1-D right-hand side
        assert one.shape == (N,) and float((one.double() - ref[:, 3]).norm() / ref[:, 3].norm()) < 1e-5
        # M^-1 must stay symmetric: u^T M^-1 w == w^T M^-1 u
        u, w = r32[:, :1], r32[:, 1:2]
        a = float((u.double() * pre.solve(w).double()).sum())
        b = float((w.double() * pre.solve(u).double()).sum())
        assert abs(a - b) < 1e-6 * max(abs(a), abs(b))
    sh = RowShardedWoodbury(L, noise, RowShard(N))
    assert abs(sh.logdet() - WoodburyPreconditioner(L, noise).logdet()) < 1e-9 * N


def test_train_posterior_closed_form_matches_explicit_covariance(oracle_backend):
    """`evaluate_on_train` (training_routines.py:551-556,567-569): the posterior at the training inputs comes back as
    models.TrainPosterior — mean y - sigma^2 alpha, log-density of the noisy version from the closed form in Khat — and
    equals what the explicit N x N posterior covariance gives (the oracle's test_nll evaluated on the train set)."""
    from rpgp_amd.models import TrainPosterior
    X, y, P, ls, noise, s = _problem(N=150, d=5, J=7, seed=3, noise=0.2)
    model, lik, mll = _build_model(X, y, P, ls, noise, s)
    ref = _oracle_gp(X, y, P, ls, noise, s)
    model.eval()
    with torch.no_grad():
        out = model(X)
        assert isinstance(out, TrainPosterior)
        nll = -mll(out, y).item()
        mean_ref, cov_ref = ref.predict(X.numpy(), full_cov=True)
        assert np.linalg.norm(out.mean.numpy() - mean_ref) / np.linalg.norm(mean_ref) < 1e-4
        assert abs(nll - ref.test_nll(X.numpy(), y.numpy())) < 1e-4 * abs(nll) + 1e-5
        # the lazily formed covariance is the general path's
        assert np.abs(out.variance.numpy() - np.diag(cov_ref)).max() < 1e-4
        # another tensor with the same values is NOT the training set object: the general path answers, same numbers
        out2 = model(X.clone())
        assert not isinstance(out2, TrainPosterior)
        assert abs(-mll(out2, y).item() - nll) < 1e-4 * abs(nll) + 1e-5


def test_fused_objective_mem_efficient_gam(oracle_backend):
    """The frozen MemoryEfficientGamKernel base (additive_spread_prescale_Jd.json: inner lengthscale ln 2, no 1/J): fused
    node against the generic path."""
    from rpgp_amd import settings, fused_mll
    from rpgp_amd.training import create_exact_gp
    from rpgp_amd.models import ExactMarginalLogLikelihood
    X, y, P, ls, noise, s = _problem(N=130, d=6, J=6, seed=21, noise=0.25)
    res = {}
    for fused in (True, False):
        torch.manual_seed(4)
        np.random.seed(4)
        model, lik = create_exact_gp(X, y, "additive_rp", J=6, noise_prior=True, kernel_type="RBF", learn_proj=False,
                                     prescale=True, space_proj=True, mem_efficient=True, batch_kernel=False)
        mll = ExactMarginalLogLikelihood(lik, model)
        model.train()
        with settings.fused_training(fused):
            assert fused_mll.applicable(model) == fused
            loss = -mll(model(X), y)
            loss.backward()
        res[fused] = (loss.item(), [p.grad.clone() for p in model.parameters() if p.requires_grad])
    assert abs(res[True][0] - res[False][0]) < 1e-6 * abs(res[False][0]) + 1e-7
    assert len(res[True][1]) == len(res[False][1]) == 4
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.allclose(a.reshape(-1), b.reshape(-1), rtol=2e-5, atol=1e-7), (a, b)


@pytest.mark.parametrize("prescale,regime", [(True, "chol"), (False, "chol"), (True, "cg")])
def test_fused_objective_equals_generic_path(oracle_backend, prescale, regime):
    """fused_mll (one autograd node for -mll of the flagship model, fitting/optimizing.py:67-72) against the generic
    operator-by-operator autograd path: same value, same four gradients (same probes in the CG regime)."""
    from rpgp_amd import settings, fused_mll
    from rpgp_amd.models import LazyPrior
    X, y, P, ls, noise, s = _problem(N=140, d=5, J=7, seed=11, noise=0.25)
    if not prescale:
        ls = torch.rand(P.shape[1], generator=torch.Generator().manual_seed(5)) + 1.0
    res = {}
    for fused in (True, False):
        model, lik, mll = _build_model(X, y, P, ls, noise, s, prescale)
        model.train()
        ctxs = [settings.fused_training(fused), settings.deterministic_probes(True)]
        if regime == "cg":
            ctxs += [settings.max_cholesky_size(10), settings.min_preconditioning_size(50), settings.cg_tolerance(1e-8),
                     settings.max_cg_iterations(500)]
        import contextlib
        with contextlib.ExitStack() as es:
            for c in ctxs:
                es.enter_context(c)
            out = model(X)
            assert isinstance(out, LazyPrior) == fused and fused_mll.applicable(model) == fused
            loss = -mll(out, y)
            loss.backward()
        res[fused] = (loss.item(), [p.grad.clone() for p in (model.covar_module.base_kernel.raw_lengthscale,
                                                              model.covar_module.raw_outputscale, lik.raw_noise,
                                                              model.mean_module.constant)])
    assert abs(res[True][0] - res[False][0]) < 1e-6 * abs(res[False][0]) + 1e-7
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.allclose(a.reshape(-1), b.reshape(-1), rtol=2e-5, atol=1e-7), (a, b)
    # a lazily returned prior still answers like the generic one when somebody reads it
    model, lik, mll = _build_model(X, y, P, ls, noise, s, prescale)
    model.train()
    out = model(X)
    assert out.mean.shape == (X.shape[0],) and out.covariance.shape[0] == X.shape[0] and out.materialized
    assert abs(-mll(out, y).item() - res[False][0]) < (5e-2 if regime == "cg" else 1e-6) * abs(res[False][0]) + 1e-7


@pytest.mark.parametrize("regime", ["chol", "cg"])
def test_negated_objective_is_bitwise_the_negation(oracle_backend, regime):
    """ExactMarginalLogLikelihood.negative (the sign folded into the fused node: what train_to_convergence calls) against
    -mll(...): the same bits for the value and for the four gradients; for a materialised prior it IS -mll(...)."""
    import contextlib
    from rpgp_amd import settings
    X, y, P, ls, noise, s = _problem(N=140, d=5, J=7, seed=13, noise=0.25)
    res = {}
    for folded in (True, False):
        model, lik, mll = _build_model(X, y, P, ls, noise, s, True)
        model.train()
        ctxs = [settings.deterministic_probes(True)]
        if regime == "cg":
            ctxs += [settings.max_cholesky_size(10), settings.min_preconditioning_size(50), settings.cg_tolerance(1e-8),
                     settings.max_cg_iterations(500)]
        with contextlib.ExitStack() as es:
            for c in ctxs:
                es.enter_context(c)
            out = model(X)
            loss = mll.negative(out, y) if folded else -mll(out, y)
            loss.backward()
        res[folded] = (loss.detach().clone(), [p.grad.clone() for p in (model.covar_module.base_kernel.raw_lengthscale,
                                                                          model.covar_module.raw_outputscale, lik.raw_noise,
                                                                          model.mean_module.constant)])
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b), (a, b)
    model, lik, mll = _build_model(X, y, P, ls, noise, s, True)
    model.train()
    out = model(X)
    _ = out.mean                                            # materialised: the generic path
    assert torch.equal(mll.negative(out, y), -mll(out, y))


def test_gpytorch_layout_state_dict_round_trip():
    """training.gpytorch_state_dict / load_gpytorch_state_dict: the reference's checkpoint key layout
    (training_routines.py:37-44; keys from memory of GPyTorch, see the table in training.py) round-trips."""
    from rpgp_amd.training import create_exact_gp, gpytorch_state_dict, load_gpytorch_state_dict
    X, y = torch.randn(40, 5), torch.randn(40)
    torch.manual_seed(0)
    a, _ = create_exact_gp(X, y, "additive_rp", J=4, noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True)
    with torch.no_grad():
        for p in a.parameters():
            p.add_(torch.randn_like(p) * 0.1)
    sd = gpytorch_state_dict(a)
    assert "likelihood.noise_covar.raw_noise" in sd and "likelihood.raw_noise" not in sd
    assert abs(float(torch.nn.functional.softplus(sd["covar_module.base_kernel.base_kernel.base_kernel.raw_outputscale"])) - 0.25) < 1e-6
    torch.manual_seed(1)
    b, _ = create_exact_gp(X, y, "additive_rp", J=4, noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True)
    load_gpytorch_state_dict(b, sd)
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.allclose(va.double(), vb.double()), ka


def test_library_slq_matches_eigendecomposition():
    """rpgp_slq_logdet (host arithmetic inside the library: implicit-shift QL on the Lanczos tridiagonals the CG coefficients
    define) against the eigendecomposition of the same matrices, incl. a column whose late alphas are masked to zero."""
    from rpgp_amd import ops, linear_cg as lcg
    from rpgp_amd.inv_quad_logdet import slq_logdet
    rng = np.random.default_rng(0)
    n, m, p = 300, 25, 10
    A = rng.standard_normal((n, n))
    A = A @ A.T / n + 0.5 * np.eye(n)
    B = rng.standard_normal((n, p))
    B /= np.linalg.norm(B, axis=0)
    X, R, P = np.zeros_like(B), B.copy(), B.copy()
    rz = (R * R).sum(0)
    ah, bh = np.zeros((m, 16), np.float32), np.zeros((m, 16), np.float32)
    for k in range(m):
        Ap = A @ P
        a = rz / (P * Ap).sum(0)
        X += a * P
        R -= a * Ap
        rzn = (R * R).sum(0)
        ah[k, :p], bh[k, :p] = a, rzn / rz
        P = R + (rzn / rz) * P
        rz = rzn
    ah[20:, 3] = 0.0
    ref = float(slq_logdet(lcg._tridiag_from_history(ah[:, :p], bh[:, :p], p, torch.float32, "cpu"), n))
    got = ops.slq_logdet_history(ah[:, :p], bh[:, :p], p, n)
    assert abs(got - ref) < 1e-12 * abs(ref)
    assert abs(got - np.linalg.slogdet(A)[1]) < 0.2 * abs(ref)        # (and it IS a log-determinant estimate)
    hist = lcg.LanczosHistory(ah[:, :p], bh[:, :p], p, torch.float32, "cpu")
    assert abs(float(slq_logdet(hist, n)) - ref) < 1e-12 * abs(ref)


def test_hostvals_drop_a_remembered_value_when_the_tensor_changes():
    """The remembered host copy of a scalar follows in-place updates (version counter) and rebinding (data pointer)."""
    from rpgp_amd.hostvals import forget, host_float, peek, prefetch, remember
    t = torch.tensor(0.3)
    assert host_float(t) == pytest.approx(0.3)
    t.add_(1.0)
    assert host_float(t) == pytest.approx(1.3)
    p = torch.nn.Parameter(torch.tensor(2.0))
    prefetch(p, t)
    assert peek(p) == pytest.approx(2.0)
    with torch.no_grad():
        p.mul_(0.5)                                            # what optimizer.step() does to a leaf
    assert peek(p) is None and host_float(p) == pytest.approx(1.0)
    p.data = torch.tensor(7.0)                                 # rebinding: new storage, same object
    assert host_float(p) == pytest.approx(7.0)
    remember(t, 5.0)                                           # a value known by other means is taken as given ...
    assert host_float(t) == 5.0
    t.zero_()                                                  # ... until the tensor changes
    assert host_float(t) == 0.0
    forget(t)
    assert peek(t) is None


def test_operator_scale_and_noise_follow_in_place_updates_of_leaf_tensors():
    """AdditiveRPOperator._scale / AddedDiagOperator._noise are read through hostvals: a leaf updated in place between two
    constructions must give the new value (advisor finding, round 4)."""
    from rpgp_amd.inv_quad_logdet import AddedDiagOperator
    from rpgp_amd.operators import AdditiveRPOperator
    Z = torch.randn(9, 3)
    s = torch.tensor(0.7, requires_grad=True)
    n = torch.tensor(0.2, requires_grad=True)
    op = AdditiveRPOperator(Z, None, outputscale=s)
    assert op._scale == pytest.approx(0.7 * op.weight)
    assert AddedDiagOperator(op, n)._noise == pytest.approx(0.2)
    with torch.no_grad():
        s.add_(0.5)
        n.mul_(3.0)
    op2 = AdditiveRPOperator(Z, None, outputscale=s)
    assert op2._scale == pytest.approx(1.2 * op2.weight)
    assert AddedDiagOperator(op2, n)._noise == pytest.approx(0.6)
