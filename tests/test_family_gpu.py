"""GPU parity tests (through the C-ABI) for the other members of the kernel family behind the same operator
(rpgp_family_*) against the float64 oracle (oracle/family.py)."""
import math

import numpy as np
import pytest
import torch

from oracle import family as fmo

pytestmark = pytest.mark.gpu

MEMBERS = [("RBF", 1, 20), ("RBF", 1, 7), ("RBF", 2, 6), ("RBF", 3, 6), ("RBF", 4, 8), ("RBF", 5, 10), ("RBF", 8, 8),
           ("RBF", 10, 20), ("RBF", 20, 20), ("Matern", 1, 20), ("Matern", 1, 3), ("InverseMQ", 1, 18),
           ("Cosine", 1, 9)]


def _rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _setup(kind, group, cols, N, T, dev, seed=0, M=None):
    rng = np.random.default_rng(seed)
    scale_z = 0.5 if group >= 8 else 1.0          # keep multi-dimensional RBFs from collapsing to the identity
    Z = (rng.normal(size=(N, cols)) * scale_z).astype(np.float32)
    V = rng.normal(size=(N, T)).astype(np.float32)
    w = rng.uniform(0.3, 1.2, size=cols // group).astype(np.float32)
    from rpgp_amd import ops
    fam = ops.Family(kind, group, torch.from_numpy(w).to(dev))
    out = [fam, Z, V, w]
    if M:
        out.append((rng.normal(size=(M, cols)) * scale_z).astype(np.float32))
    return out


@pytest.mark.parametrize("kind,group,cols", MEMBERS)
@pytest.mark.parametrize("T", [1, 11])
def test_family_mvm_matches_oracle(gpu_device, kind, group, cols, T):
    from rpgp_amd import ops
    N, M = 1237, 301
    fam, Z, V, w, Z1 = _setup(kind, group, cols, N, T, gpu_device, seed=cols + T, M=M)
    Zt, Vt, Z1t = (torch.from_numpy(a).to(gpu_device) for a in (Z, V, Z1))
    got = ops.family_mvm_sym(fam, Zt, Vt, 0.7, 0.13).cpu().numpy()
    ref = fmo.mvm(Z, Z, V, kind, group, w, 0.7, 0.13)
    assert _rel(got, ref) < 5e-6
    got_r = ops.family_mvm_rect(fam, Z1t, Zt, Vt, 0.7).cpu().numpy()
    ref_r = fmo.mvm(Z1, Z, V, kind, group, w, 0.7)
    assert _rel(got_r, ref_r) < 5e-6


@pytest.mark.parametrize("kind,group,cols", MEMBERS)
def test_family_dense_and_bilinear_match_oracle(gpu_device, kind, group, cols):
    from rpgp_amd import ops
    N, M, T = 523, 77, 3
    fam, Z, V, w, Z1 = _setup(kind, group, cols, N, T, gpu_device, seed=3 * cols, M=M)
    Zt, Z1t = torch.from_numpy(Z).to(gpu_device), torch.from_numpy(Z1).to(gpu_device)
    Kd = ops.family_dense(fam, Z1t, Zt, 1.3).cpu().numpy()
    assert np.abs(Kd - fmo.kernel_matrix(Z1, Z, kind, group, w, 1.3)).max() < 2e-5
    rng = np.random.default_rng(5)
    L, R = rng.normal(size=(N, T)).astype(np.float32), rng.normal(size=(N, T)).astype(np.float32)
    gZ, gc = ops.family_bilinear_grad(fam, Zt, torch.from_numpy(L).to(gpu_device), torch.from_numpy(R).to(gpu_device), 1.3)
    rZ, rc = fmo.bilinear_grad(Z.astype(np.float64), L, R, kind, group, w, 1.3)
    assert _rel(gZ.cpu().numpy(), rZ) < 2e-5
    assert _rel(gc.cpu().numpy(), rc) < 2e-5
    S = rng.normal(size=(N, N)).astype(np.float32)
    S = (S + S.T) / 2
    gZ2, gc2 = ops.family_bilinear_grad_dense(fam, Zt, torch.from_numpy(S).to(gpu_device), 1.3)
    rZ2, rc2 = fmo.bilinear_grad_dense(Z.astype(np.float64), S, kind, group, w, 1.3)
    assert _rel(gZ2.cpu().numpy(), rZ2) < 2e-5
    assert _rel(gc2.cpu().numpy(), rc2) < 2e-5


def test_family_large_n_symmetric_equals_rectangular_and_is_linear(gpu_device):
    """Size-independent properties at a size the dense oracle cannot reach: the symmetric sweep (each pair once, DPP
    rotation for the transposed product) equals the rectangular sweep, and the operator is linear."""
    from rpgp_amd import ops
    N = 20011
    for kind, group, cols in (("Matern", 1, 20), ("RBF", 2, 10), ("InverseMQ", 1, 5)):
        fam, Z, V, w = _setup(kind, group, cols, N, 2, gpu_device, seed=1)
        Zt, Vt = torch.from_numpy(Z).to(gpu_device), torch.from_numpy(V).to(gpu_device)
        a = ops.family_mvm_sym(fam, Zt, Vt, 0.5, 0.0)
        b = ops.family_mvm_rect(fam, Zt, Zt, Vt, 0.5)
        assert float((a - b).norm() / b.norm()) < 3e-6
        c = ops.family_mvm_sym(fam, Zt, 2.0 * Vt[:, :1] - 3.0 * Vt[:, 1:], 0.5, 0.0)
        assert float((c - (2.0 * a[:, :1] - 3.0 * a[:, 1:])).norm() / c.norm()) < 3e-6


def test_family_group1_rbf_equals_hot_path(gpu_device):
    from rpgp_amd import ops
    N, J = 3000, 20
    Z = torch.randn(N, J, generator=torch.Generator().manual_seed(0)).to(gpu_device)
    V = torch.randn(N, 4, generator=torch.Generator().manual_seed(1)).to(gpu_device)
    fam = ops.Family("RBF", 1, torch.full((J,), 1.0 / J, device=gpu_device))
    a = ops.family_mvm_sym(fam, Z, V, 1.0, 0.1)
    b = ops.mvm_sym(Z, V, 1.0 / J, 0.1)
    assert float((a - b).norm() / b.norm()) < 2e-6


def test_family_errors(gpu_device):
    from rpgp_amd import ops
    w = torch.ones(3, device=gpu_device)
    assert ops.Family("Matern", 2, w).generic and ops.Family("RBF", 7, w).generic and not ops.Family("RBF", 2, w).generic
    with pytest.raises(ValueError):
        ops.Family("RBF", 40, w)                                   # groups of at most 32 columns
    with pytest.raises(ValueError):
        ops.Family("Matern", 30, w)                                # ... and 64 columns in all
    fam = ops.Family("RBF", 2, w)
    with pytest.raises(ValueError):
        ops.family_mvm_sym(fam, torch.zeros(10, 5, device=gpu_device), torch.zeros(10, 1, device=gpu_device), 1.0)
    with pytest.raises(TypeError):
        ops.family_mvm_sym(fam, torch.zeros(10, 6, device=gpu_device, dtype=torch.float64),
                           torch.zeros(10, 1, device=gpu_device), 1.0)


# what the templated kernels do not instantiate: float64 members, radial k > 1 groups of the non-RBF types, other group sizes
GENERIC = [("Matern", 2, 6, "f32"), ("InverseMQ", 3, 9, "f32"), ("Cosine", 2, 4, "f32"), ("RBF", 6, 12, "f32"),
           ("RBF", 1, 20, "f64"), ("Matern", 1, 7, "f64"), ("InverseMQ", 4, 8, "f64"), ("Cosine", 1, 5, "f64"),
           ("RBF", 20, 20, "f64"),
           # the PRODUCT-of-1-D form of a group (RPGP_KIND_PRODUCT: the ProductKernel groups of the rp_poly kinds)
           ("Matern*", 2, 6, "f32"), ("InverseMQ*", 3, 9, "f32"), ("Cosine*", 2, 4, "f64"), ("Matern*", 5, 10, "f64"),
           ("InverseMQ*", 32, 32, "f64")]


@pytest.mark.parametrize("kind,group,cols,prec", GENERIC)
def test_generic_family_kernels_match_oracle(gpu_device, kind, group, cols, prec):
    """csrc/rpgp_family_generic.hip through the C-ABI: product (symmetric with noise, rectangular, T = 1 / 11 / 19), dense
    block, both forms of the derivative — against oracle/family.py; float64 at 1e-11, float32 at 2e-5; bitwise reproducible."""
    from rpgp_amd import ops
    dt = torch.float64 if prec == "f64" else torch.float32
    npdt = np.float64 if prec == "f64" else np.float32
    tol = 1e-11 if prec == "f64" else 2e-5
    rng = np.random.default_rng(cols * 7 + group)
    N, M = 411, 97
    sz = 0.5 if group >= 8 else 1.0
    Z, Z1 = (rng.normal(size=(N, cols)) * sz).astype(npdt), (rng.normal(size=(M, cols)) * sz).astype(npdt)
    w = rng.uniform(0.3, 1.2, size=cols // group).astype(npdt)
    product = kind.endswith("*")
    kind = kind.rstrip("*")
    fam = ops.Family(kind, group, torch.from_numpy(w).to(gpu_device), product=product)
    assert fam.generic and fam.dtype == dt and fam.product == product
    Zt, Z1t = torch.from_numpy(Z).to(gpu_device), torch.from_numpy(Z1).to(gpu_device)
    for T in (1, 11, 19):
        V = rng.normal(size=(N, T)).astype(npdt)
        Vt = torch.from_numpy(V).to(gpu_device)
        got = ops.family_mvm_sym(fam, Zt, Vt, 0.7, 0.13)
        assert got.dtype == dt and torch.equal(got, ops.family_mvm_sym(fam, Zt, Vt, 0.7, 0.13))
        assert _rel(got.cpu().numpy(), fmo.mvm(Z, Z, V, kind, group, w, 0.7, 0.13, product)) < tol
        got_r = ops.family_mvm_rect(fam, Z1t, Zt, Vt, 0.7).cpu().numpy()
        assert _rel(got_r, fmo.mvm(Z1, Z, V, kind, group, w, 0.7, 0.0, product)) < tol
    Kd = ops.family_dense(fam, Z1t, Zt, 1.3).cpu().numpy()
    assert np.abs(Kd - fmo.kernel_matrix(Z1, Z, kind, group, w, 1.3, product)).max() < tol * 10
    L, R = rng.normal(size=(N, 3)).astype(npdt), rng.normal(size=(N, 3)).astype(npdt)
    gZ, gc = ops.family_bilinear_grad(fam, Zt, torch.from_numpy(L).to(gpu_device), torch.from_numpy(R).to(gpu_device), 1.3)
    rZ, rc = fmo.bilinear_grad(Z.astype(np.float64), L, R, kind, group, w, 1.3, product)
    assert _rel(gZ.cpu().numpy(), rZ) < tol * 2 and _rel(gc.cpu().numpy(), rc) < tol * 2
    S = rng.normal(size=(N, N)).astype(npdt)
    S = (S + S.T) / 2
    gZ2, gc2 = ops.family_bilinear_grad_dense(fam, Zt, torch.from_numpy(S).to(gpu_device), 1.3)
    rZ2, rc2 = fmo.bilinear_grad_dense(Z.astype(np.float64), S, kind, group, w, 1.3, product)
    assert _rel(gZ2.cpu().numpy(), rZ2) < tol * 2 and _rel(gc2.cpu().numpy(), rc2) < tol * 2


@pytest.mark.parametrize("kind,model_kwargs", [
    ("additive_rp", dict(J=20, kernel_type="Matern", prescale=True)),
    ("additive_rp", dict(J=4, k=5, batch_kernel=False, prescale=True)),
    ("additive_rp", dict(J=3, k=6, batch_kernel=False, prescale=True)),          # any k: zero-padded to the 8-wide kernel
    ("additive_rp", dict(J=2, k=7, batch_kernel=False, prescale=False)),
    ("additive_rp", dict(J=3, k=2, batch_kernel=False, kernel_type="Matern", prescale=True)),   # radial non-RBF groups
    ("rp_poly", dict(J=8, k=1, weighted=True, kernel_type="RBF")),
    ("strictly_additive", dict(weighted=True, kernel_type="InverseMQ")),
    ("rp_poly", dict(J=3, k=2, weighted=True, kernel_type="Matern")),                   # products of non-RBF sub-kernels
])
@pytest.mark.parametrize("regime", ["chol", "cg"])
def test_family_model_mll_and_prediction(gpu_device, kind, model_kwargs, regime):
    """End to end through the kernel modules, the native mBCG executor (RPGP_OP_FAMILY) and the prediction strategy,
    against a dense float64 computation with the model's own hyper-parameters."""
    from rpgp_amd import settings
    from rpgp_amd import kernels as km
    from rpgp_amd.models import ExactMarginalLogLikelihood
    from rpgp_amd.training import create_exact_gp
    N, d = 1500, 6
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, d, generator=g)
    y = torch.sin(X).sum(1) + 0.05 * torch.randn(N, generator=g)
    y = (y - y.mean()) / y.std()
    Xs = torch.randn(40, d, generator=g)
    torch.manual_seed(1)
    if kind in ("additive_rp", "rp_poly"):
        model, lik = create_exact_gp(X, y, kind, noise_prior=True, learn_proj=False, **model_kwargs)
    else:
        model, lik = create_exact_gp(X, y, kind, noise_prior=True, **model_kwargs)
    lik.noise = 0.05
    model, lik = model.to(gpu_device), lik.to(gpu_device)
    model.train_inputs, model.train_targets = X.to(gpu_device), y.to(gpu_device)
    mll = ExactMarginalLogLikelihood(lik, model)
    base = model.covar_module.base_kernel
    if isinstance(base, km.ScaledProjectionKernel):
        P = base.projection_module.weight.detach().double().cpu().t()
        ls = base.lengthscale.detach().double().cpu().reshape(-1)
        tr = lambda A: (((A.double() / ls) @ P) if base.prescale else ((A.double() @ P) / ls)).numpy()
        ktype, group = base.base_kernel.kernel_type, base.base_kernel.group
        w = np.full(P.shape[1] // group, float(base.base_kernel.weight))
    else:
        P = base.projection_module.weight.detach().double().cpu().t()
        ls = base.lengthscales.detach().double().cpu().reshape(-1)
        tr = lambda A: ((A.double() @ P) / ls).numpy()
        ktype, group = base.kernel_type, base.k
        w = base.outputscales.detach().double().cpu().numpy()
    product = bool(getattr(base, "product", False))
    s, noise, c = float(model.covar_module.outputscale), float(lik.noise), float(model.mean_module.constant)
    Z, Zs = tr(X), tr(Xs)
    K = fmo.kernel_matrix(Z, Z, ktype, group, w, s, product) + noise * np.eye(N)
    r = y.double().numpy() - c
    alpha = np.linalg.solve(K, r)
    ref_mll = (-0.5 * r @ alpha - 0.5 * np.linalg.slogdet(K)[1] - 0.5 * N * math.log(2 * math.pi)
               + float(lik.log_prior().detach())) / N
    ctx = settings.max_cholesky_size(4000 if regime == "chol" else 100)
    with ctx, settings.cg_tolerance(1e-4), settings.eval_cg_tolerance(1e-5), settings.num_trace_samples(30), \
            settings.max_lanczos_quadrature_iterations(60), settings.max_cg_iterations(2000):
        model.train()
        val = mll(model(model.train_inputs), model.train_targets)
        val.backward()
        tol = 1e-4 if regime == "chol" else 2e-2                # SLQ log-det is stochastic in the CG regime
        assert abs(val.item() - ref_mll) < tol * max(1.0, abs(ref_mll))
        for p in model.parameters():
            if p.requires_grad:
                assert p.grad is not None and torch.isfinite(p.grad).all()
        model.eval()
        with torch.no_grad():
            out = model(Xs.to(gpu_device))
        Ks = fmo.kernel_matrix(Zs, Z, ktype, group, w, s, product)
        mean = Ks @ alpha + c
        cov = fmo.kernel_matrix(Zs, Zs, ktype, group, w, s, product) - Ks @ np.linalg.solve(K, Ks.T)
        assert _rel(out.mean.cpu().numpy(), mean) < 1e-4
        assert np.abs(out.variance.cpu().numpy() - np.diag(cov)).max() < 1e-4 * s + 1e-5


def test_mixed_group_sizes_operator_matches_oracle(gpu_device):
    """`general_rp_poly` / `create_multi_additive_kernel` (training_routines.py:192-207,247-258): multiplicative groups of
    different sizes, bucketed by size onto the family tile kernels (operators.MixedGroupOperator) — product, dense form,
    rows, diagonal and the bilinear derivative against float64."""
    from rpgp_amd.operators import MixedGroupOperator
    degrees = [1, 2, 1, 3, 2, 1, 4, 6]        # 6 is not an instantiated group size: zero-padded to 8
    N, M, T = 1237, 301, 11
    rng = np.random.default_rng(3)
    Z = (rng.normal(size=(N, sum(degrees))) * 0.8).astype(np.float32)
    Z2 = (rng.normal(size=(M, sum(degrees))) * 0.8).astype(np.float32)
    V = rng.normal(size=(N, T)).astype(np.float32)
    w = rng.uniform(0.3, 1.2, size=len(degrees)).astype(np.float32)
    s = 1.7

    def dense(A, B):
        K, col = np.zeros((A.shape[0], B.shape[0])), 0
        for c, k in enumerate(degrees):
            K += fmo.kernel_matrix(A[:, col:col + k], B[:, col:col + k], "RBF", k, [w[c]], s)
            col += k
        return K

    Zt = torch.from_numpy(Z).to(gpu_device).requires_grad_(True)
    wt = torch.from_numpy(w).to(gpu_device)
    st = torch.tensor(s, device=gpu_device)
    op = MixedGroupOperator(Zt, None, st, wt, "RBF", degrees)
    assert len(op.buckets) == 5 and op.buckets[-1][2].group == 8
    Kd = dense(Z, Z)
    out = op._matmul(torch.from_numpy(V).to(gpu_device), noise=0.3).cpu().numpy()
    assert _rel(out, Kd @ V.astype(np.float64) + 0.3 * V) < 1e-5
    assert _rel(op.to_dense().cpu().numpy(), Kd) < 1e-5
    assert np.allclose(op._diagonal().cpu().numpy(), Kd.diagonal(), rtol=1e-5)
    idx = torch.tensor([5, 1000, 5, 77], device=gpu_device)
    assert _rel(op._get_rows(idx).cpu().numpy(), Kd[idx.cpu().numpy()]) < 1e-5
    opr = MixedGroupOperator(Zt, torch.from_numpy(Z2).to(gpu_device), st, wt, "RBF", degrees)
    W = rng.normal(size=(M, 3)).astype(np.float32)
    assert _rel(opr._matmul(torch.from_numpy(W).to(gpu_device)).cpu().numpy(), dense(Z, Z2) @ W.astype(np.float64)) < 1e-5
    assert _rel(opr.t()._matmul(torch.from_numpy(V).to(gpu_device)).cpu().numpy(), dense(Z2, Z) @ V.astype(np.float64)) < 1e-5

    # the native mBCG executor on the sum of the buckets (RPGP_OP_SUM): true float64 residual of the solve
    from rpgp_amd import linear_cg as lcg
    from rpgp_amd.operators import AddedDiagOperator
    khat = AddedDiagOperator(op, torch.tensor(0.3, device=gpu_device))
    rhs = torch.from_numpy(V[:, :5].copy()).to(gpu_device)
    n0 = lcg.stats.get("native_calls", 0)
    x = lcg.linear_cg(khat._matmul, rhs, tolerance=1e-5, max_iter=2000, operator=khat)
    assert lcg.stats.get("native_calls", 0) == n0 + 1
    Kh = Kd + 0.3 * np.eye(N)
    res = np.linalg.norm(Kh @ x.cpu().double().numpy() - V[:, :5], axis=0) / np.linalg.norm(V[:, :5], axis=0)
    assert res.max() < 2e-4

    # bilinear derivative vs float64 autograd of sum((L R^T) * K)
    L = rng.normal(size=(N, 4)).astype(np.float32)
    R = rng.normal(size=(N, 4)).astype(np.float32)
    gZ, gs, gw = op._bilinear_derivative(torch.from_numpy(L).to(gpu_device), torch.from_numpy(R).to(gpu_device))
    Zd = torch.from_numpy(Z).double().requires_grad_(True)
    wd = torch.from_numpy(w).double().requires_grad_(True)
    sd = torch.tensor(s, dtype=torch.float64, requires_grad=True)
    K, col = torch.zeros(N, N, dtype=torch.float64), 0
    for c, k in enumerate(degrees):
        d2 = sum((Zd[:, col + m:col + m + 1] - Zd[:, col + m:col + m + 1].t()) ** 2 for m in range(k))
        K = K + wd[c] * torch.exp(-0.5 * d2)
        col += k
    obj = ((torch.from_numpy(L).double() @ torch.from_numpy(R).double().t()) * (sd * K)).sum()
    obj.backward()
    assert _rel(gZ.cpu().double().numpy(), Zd.grad.numpy()) < 2e-5
    assert abs(float(gs) - float(sd.grad)) < 2e-5 * max(1.0, abs(float(sd.grad)))
    assert _rel(gw.cpu().double().numpy(), wd.grad.numpy()) < 2e-5


def test_general_rp_poly_model_trains_on_gpu(gpu_device):
    """model_specs/polynomial_rp_smaller.json's shape end to end (mixed group sizes, weighted, torch mBCG loop)."""
    from rpgp_amd.training import train_exact_gp
    g = torch.Generator().manual_seed(0)
    X = torch.randn(900, 6, generator=g)
    y = torch.sin(X).sum(1) + 0.5 * X[:, 0] * X[:, 1] + 0.05 * torch.randn(900, generator=g)
    y = (y - y.mean()) / y.std()
    tk = {"verbose": False, "optimizer": "adam", "max_iter": 15, "lr": 0.1, "patience": 20, "smooth": True}
    mk = dict(degrees=[1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3], noise_prior=True, kernel_type="RBF", learn_proj=False,
              weighted=True)
    torch.manual_seed(0)
    from rpgp_amd import linear_cg as lcg, settings
    n_native = lcg.stats.get("native_calls", 0)
    with settings.max_cholesky_size(100):
        metrics, mean, model = train_exact_gp(X[:800], y[:800], X[800:], y[800:], "general_rp_poly", mk, tk,
                                              devices=(str(gpu_device),), skip_random_restart=True)
    assert lcg.stats.get("native_calls", 0) > n_native          # the training solves ran in the native executor
    metrics, mean, model = train_exact_gp(X[:800], y[:800], X[800:], y[800:], "general_rp_poly", mk, tk,
                                          devices=(str(gpu_device),), skip_random_restart=True)
    assert np.isfinite(metrics["prior_train_nmll"]) and np.isfinite(metrics["test_nll"])
    rmse = float(((mean - y[800:]) ** 2).mean().sqrt())
    assert rmse < 0.6                  # far better than the unit-variance baseline after 15 steps


@pytest.mark.parametrize("model_kwargs", [dict(J=6, kernel_type="Matern", prescale=True),
                                          dict(J=3, k=2, batch_kernel=False, kernel_type="InverseMQ", prescale=False),
                                          dict(J=4, k=3, batch_kernel=False, prescale=True)])
def test_double_family_model_parity(gpu_device, model_kwargs):
    """`--double` (training_routines.py:481) for family members: float64 model through the runtime-(kind, group) kernels —
    MLL (Cholesky and CG regime, inv-quad part), gradients finite, predictive mean / variance against the dense float64
    computation at 1e-8 (the float32 path's gates are 1e-4)."""
    from rpgp_amd import settings
    from rpgp_amd.models import ExactMarginalLogLikelihood
    from rpgp_amd.training import create_exact_gp
    N, d = 700, 5
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, d, generator=g).double()
    y = (torch.sin(X).sum(1) + 0.05 * torch.randn(N, generator=g).double())
    y = (y - y.mean()) / y.std()
    Xs = torch.randn(30, d, generator=g).double()
    torch.manual_seed(1)
    model, lik = create_exact_gp(X.float(), y.float(), "additive_rp", noise_prior=True, learn_proj=False, **model_kwargs)
    model, lik = model.to(gpu_device, torch.float64), lik.to(gpu_device, torch.float64)
    lik.noise = 0.05
    model.train_inputs, model.train_targets = X.to(gpu_device), y.to(gpu_device)
    mll = ExactMarginalLogLikelihood(lik, model)
    base = model.covar_module.base_kernel
    P = base.projection_module.weight.detach().double().cpu().t()
    ls = base.lengthscale.detach().double().cpu().reshape(-1)
    tr = lambda A: (((A / ls) @ P) if base.prescale else ((A @ P) / ls)).numpy()
    ktype, group = base.base_kernel.kernel_type, base.base_kernel.group
    w = np.full(P.shape[1] // group, float(base.base_kernel.weight))
    s, noise, c = float(model.covar_module.outputscale), float(lik.noise), float(model.mean_module.constant)
    Z, Zs = tr(X), tr(Xs)
    K = fmo.kernel_matrix(Z, Z, ktype, group, w, s) + noise * np.eye(N)
    r = y.numpy() - c
    alpha = np.linalg.solve(K, r)
    ref_mll = (-0.5 * r @ alpha - 0.5 * np.linalg.slogdet(K)[1] - 0.5 * N * math.log(2 * math.pi)
               + float(lik.log_prior().detach())) / N
    model.train()
    with settings.max_cholesky_size(4000):
        val = mll(model(model.train_inputs), model.train_targets)
        val.backward()
    assert val.dtype == torch.float64 and abs(val.item() - ref_mll) < 1e-9 * max(1.0, abs(ref_mll))
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.requires_grad)
    with settings.max_cholesky_size(100), settings.cg_tolerance(1e-10), settings.skip_logdet_forward(True), \
            settings.max_cg_iterations(3000), settings.min_preconditioning_size(100), torch.no_grad():
        v2 = mll(model(model.train_inputs), model.train_targets).item()
    expect = (-0.5 * r @ alpha - 0.5 * N * math.log(2 * math.pi) + float(lik.log_prior().detach())) / N
    assert abs(v2 - expect) < 1e-8 * abs(expect)
    model.eval()
    with torch.no_grad(), settings.max_cholesky_size(100), settings.eval_cg_tolerance(1e-11), settings.max_cg_iterations(3000):
        out = model(Xs.to(gpu_device))
    Ks = fmo.kernel_matrix(Zs, Z, ktype, group, w, s)
    mean = Ks @ alpha + c
    cov = fmo.kernel_matrix(Zs, Zs, ktype, group, w, s) - Ks @ np.linalg.solve(K, Ks.T)
    assert _rel(out.mean.cpu().numpy(), mean) < 1e-8
    assert np.abs(out.variance.cpu().numpy() - np.diag(cov)).max() < 1e-8
