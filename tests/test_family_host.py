"""Host-side tests (CPU, oracle test double) for the other members of the kernel family behind the same operator
(SURVEY.md §8(f) rank 4): `kernel_type` sub-kernels, k > 1 RBF sub-kernels, the weighted rp_poly / strictly_additive /
additive kinds.  The float64 reference of every check is a dense torch-autograd restatement written out in the test."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import family as fmo
from oracle import dense_gp as orc


def _phi_t(kind, d2):
    if kind == "RBF":
        return torch.exp(-0.5 * d2)
    r = torch.sqrt(d2 + 1e-300)
    if kind == "Matern":
        return (1 + math.sqrt(3) * r) * torch.exp(-math.sqrt(3) * r)
    if kind == "InverseMQ":
        return (d2 + 1) ** -0.5
    return torch.cos(math.pi * r)


def _dense_family(Z, kind, group, w, product=False):
    """`group`: the common group size, or the list of group sizes (in column order) for mixed sizes.  `product`: a group is
    the product of its 1-D sub-kernels (the rp_poly kinds) instead of one radial sub-kernel (additive_rp)."""
    n = Z.shape[0]
    K = torch.zeros(n, n, dtype=torch.float64)
    sizes = [group] * (Z.shape[1] // group) if isinstance(group, int) else list(group)
    col = 0
    for c, k in enumerate(sizes):
        sq = [(Z[:, col + m:col + m + 1] - Z[:, col + m:col + m + 1].t()) ** 2 for m in range(k)]
        if product:
            comp = torch.ones(n, n, dtype=torch.float64)
            for q in sq:
                comp = comp * _phi_t(kind, q)
        else:
            comp = _phi_t(kind, sum(sq))
        K = K + w[c] * comp
        col += k
    return K


def test_family_oracle_pinned_to_rbf_oracle_and_reference_formulas():
    rng = np.random.default_rng(0)
    Z1, Z2 = rng.normal(size=(30, 6)), rng.normal(size=(17, 6))
    # group 1, weights 1/J: the hot-path kernel (oracle/dense_gp.py, itself pinned to the reference's GAMFunction)
    assert np.allclose(fmo.kernel_matrix(Z1, Z2, "RBF", 1, np.full(6, 1 / 6)), orc.additive_rbf(Z1, Z2) / 6, atol=1e-14)
    # k-dimensional RBF = product of the 1-D RBFs of its group
    comps1 = fmo.component_matrices(Z1, Z2, "RBF", 1)
    comps3 = fmo.component_matrices(Z1, Z2, "RBF", 3)
    assert np.allclose(comps3[0], comps1[0] * comps1[1] * comps1[2])
    # InverseMQ: dist.add_(1).pow_(-1/2) on the squared distance (imq_kernel.py:8-9)
    d2 = (Z1[:, :1] - Z2[:, :1].T) ** 2
    assert np.allclose(fmo.component_matrices(Z1, Z2, "InverseMQ", 1)[0], (d2 + 1) ** -0.5)
    # non-RBF groups are RADIAL (training_routines.py:172-174: kernel(active_dims = the group)): a function of the
    # group's Euclidean distance, not the product of 1-D kernels
    r = np.sqrt(((Z1[:, None, :2] - Z2[None, :, :2]) ** 2).sum(-1))
    assert np.allclose(fmo.component_matrices(Z1, Z2, "Matern", 2)[0], (1 + np.sqrt(3) * r) * np.exp(-np.sqrt(3) * r))
    # ... unless `product`: the ProductKernel groups of the rp_poly kinds (polynomial_projection_kernels.py:70-86)
    for kind in ("Matern", "InverseMQ", "Cosine", "RBF"):
        c1 = fmo.component_matrices(Z1, Z2, kind, 1)
        c3 = fmo.component_matrices(Z1, Z2, kind, 3, product=True)
        assert np.allclose(c3[1], c1[3] * c1[4] * c1[5], atol=1e-15)
    # the product form's derivative against central differences of its own kernel matrix
    Zs = rng.normal(size=(9, 4))
    S = rng.normal(size=(9, 9))
    S = S + S.T
    w = np.array([0.4, 1.1])
    for kind in ("Matern", "InverseMQ", "Cosine"):
        gZ, gc = fmo.bilinear_grad_dense(Zs, S, kind, 2, w, 0.9, product=True)
        comps = fmo.component_matrices(Zs, Zs, kind, 2, product=True)
        assert np.allclose(gc, [0.5 * (S * c).sum() for c in comps])
        num = np.zeros_like(Zs)
        for i in range(9):
            for j in range(4):
                e = np.zeros_like(Zs)
                e[i, j] = 1e-6
                f = lambda Zq: 0.5 * (S * fmo.kernel_matrix(Zq, Zq, kind, 2, w, 0.9, product=True)).sum()
                num[i, j] = (f(Zs + e) - f(Zs - e)) / 2e-6
        assert np.allclose(gZ, num, rtol=1e-5, atol=1e-7), kind


def _problem(n=60, d=5, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(n, d, generator=g)
    y = torch.sin(X).sum(1) + 0.05 * torch.randn(n, generator=g)
    return X, (y - y.mean()) / y.std()


@pytest.mark.parametrize("kind,model_kwargs", [
    ("additive_rp", dict(J=6, kernel_type="Matern", prescale=True)),
    ("additive_rp", dict(J=4, kernel_type="InverseMQ", prescale=False)),
    ("additive_rp", dict(J=4, kernel_type="Cosine", prescale=True, init_lengthscale_range=(3.0, 3.0))),
    ("additive_rp", dict(J=3, k=2, batch_kernel=False, prescale=True)),
    ("additive_rp", dict(J=2, k=6, batch_kernel=False, prescale=True)),                      # any k: 6 -> the 8-wide kernel
    ("additive_rp", dict(J=1, k=7, batch_kernel=False, prescale=False)),                     # ... 7 -> 8, postscale
    ("additive_rp", dict(J=1, k=13, batch_kernel=False, prescale=True)),                     # ... 13 -> 20
    ("additive_rp", dict(J=3, k=2, batch_kernel=False, kernel_type="Matern", prescale=True)),     # radial non-RBF groups
    ("additive_rp", dict(J=2, k=3, batch_kernel=False, kernel_type="InverseMQ", prescale=False)),
    ("rp_poly", dict(J=5, k=1, weighted=True, kernel_type="RBF")),
    ("rp_poly", dict(J=3, k=2, weighted=True, kernel_type="RBF")),
    ("rp_poly", dict(J=4, k=1, weighted=False, kernel_type="Matern")),
    ("strictly_additive", dict(weighted=True, kernel_type="RBF")),
    ("strictly_additive", dict(weighted=False, kernel_type="RBF", memory_efficient=True)),
    ("additive", dict(groups=[[0, 3], [1, 4]], weighted=True)),
    ("additive", dict(groups=[[0], [1, 4], [2], [0, 2, 3]], weighted=True)),                  # unequal groups
    ("general_rp_poly", dict(degrees=[1, 2, 1, 3], weighted=True, learn_proj=False)),        # polynomial_rp.json's shape
    ("general_rp_poly", dict(degrees=[2, 1], weighted=False, learn_proj=False)),
    ("general_rp_poly", dict(degrees=[6, 1, 6], weighted=True, learn_proj=False)),           # 6 is padded to the 8-wide kernel
    ("rp_poly", dict(J=2, k=7, weighted=True, kernel_type="RBF")),                           # equal sizes, padded as well
    ("rp_poly", dict(J=3, k=2, weighted=True, kernel_type="Matern")),                        # products of non-RBF sub-kernels
    ("rp_poly", dict(J=2, k=3, weighted=False, kernel_type="InverseMQ")),
    ("general_rp_poly", dict(degrees=[1, 2, 1, 3], weighted=True, learn_proj=False, kernel_type="Matern")),
    ("general_rp_poly", dict(degrees=[2, 1], weighted=True, learn_proj=False, kernel_type="Cosine")),
])
def test_family_mll_and_gradients_match_dense_autograd(oracle_backend, kind, model_kwargs):
    from rpgp_amd.training import create_exact_gp
    from rpgp_amd.models import ExactMarginalLogLikelihood
    from rpgp_amd import kernels as km
    X, y = _problem()
    torch.manual_seed(3)
    model, lik = create_exact_gp(X, y, kind, noise_prior=True, learn_proj=False, **model_kwargs) \
        if kind in ("additive_rp", "rp_poly") else create_exact_gp(X, y, kind, noise_prior=True, **model_kwargs)
    mll = ExactMarginalLogLikelihood(lik, model)
    model.train()
    val = mll(model(X), y)
    val.backward()

    # dense float64 autograd restatement of the same objective from the model's own parameters
    base = model.covar_module.base_kernel
    Xd, yd = X.double(), y.double()
    leaves = {}

    def leaf(name, p):
        leaves[name] = p.detach().double().clone().requires_grad_(True)
        return leaves[name]

    raw_s = leaf("outputscale", model.covar_module.raw_outputscale)
    raw_n = leaf("noise", lik.raw_noise)
    c = leaf("mean", model.mean_module.constant)
    if isinstance(base, km.ScaledProjectionKernel):
        raw_ls = leaf("ls", base.raw_lengthscale)
        ls = F.softplus(raw_ls).reshape(-1)
        P = base.projection_module.weight.detach().double().t()
        Z = (Xd / ls) @ P if base.prescale else (Xd @ P) / ls
        inner = base.base_kernel
        group, ktype = inner.group, inner.kernel_type
        w = torch.full((Z.shape[1] // group,), float(inner.weight), dtype=torch.float64)
    elif isinstance(base, km.MemoryEfficientGamKernel):
        raw_ls = leaf("ls", base.raw_lengthscale)
        Z = Xd / F.softplus(raw_ls).reshape(-1)
        group, ktype = 1, "RBF"
        w = torch.ones(Z.shape[1], dtype=torch.float64)
    else:
        raw_ls = leaf("ls", base.raw_lengthscales)
        raw_w = leaf("w", base.raw_outputscales)
        P = base.projection_module.weight.detach().double().t()
        Z = (Xd @ P) / F.softplus(raw_ls).reshape(-1)
        group, ktype = (base.k if base.k is not None else base.component_degrees), base.kernel_type
        w = F.softplus(raw_w)
    n = X.shape[0]
    Kh = F.softplus(raw_s) * _dense_family(Z, ktype, group, w, product=bool(getattr(base, "product", False))) + \
        (F.softplus(raw_n) + 1e-4) * torch.eye(n, dtype=torch.float64)
    r = yd - c
    obj = (-0.5 * r @ torch.linalg.solve(Kh, r) - 0.5 * torch.logdet(Kh) - 0.5 * n * math.log(2 * math.pi)) / n
    # the noise-prior term comes from the model's own code on both sides; its gradient w.r.t. raw_noise is not compared
    prior = float(lik.log_prior().detach())
    assert abs(val.item() - (obj.item() + prior / n)) < 2e-4 * max(1.0, abs(val.item()))
    obj.backward()

    def close(a, b):
        return torch.allclose(a.double().reshape(-1), b.reshape(-1), rtol=3e-3, atol=2e-6)

    assert close(model.covar_module.raw_outputscale.grad, leaves["outputscale"].grad)
    assert close(model.mean_module.constant.grad, leaves["mean"].grad)
    ls_param = base.raw_lengthscale if hasattr(base, "raw_lengthscale") and "w" not in leaves else base.raw_lengthscales
    assert close(ls_param.grad, leaves["ls"].grad)
    if "w" in leaves:
        if base.weighted:
            assert close(base.raw_outputscales.grad, leaves["w"].grad)
        else:
            assert base.raw_outputscales.grad is None            # frozen mixing weights (polynomial_projection_kernels.py:94-98)
    proj = getattr(base, "projection_module", None)
    if proj is not None and isinstance(getattr(proj, "weight", None), torch.nn.Parameter):
        assert proj.weight.grad is None


def test_family_initialisation_follows_reference_order():
    """polynomial_projection_kernels.py:139-156: mixing weights first (normalised), then one draw per sub-kernel."""
    from rpgp_amd.training import create_rp_poly_kernel, create_strictly_additive_kernel
    torch.manual_seed(11)
    k = create_rp_poly_kernel(6, 1, 4, weighted=True, init_mixin_range=(0.5, 1.5), init_lengthscale_range=(0.5, 2.0))
    torch.manual_seed(11)
    from rpgp_amd import rp
    [rp.gen_rp(6, 1) for _ in range(4)]
    torch.nn.Linear(6, 4, bias=False)            # the projection module's default init draws from the same stream
    mix = torch.rand(4) * 1.0 + 0.5
    mix = mix / mix.sum()
    ls = torch.cat([torch.rand(1) * 1.5 + 0.5 for _ in range(4)])
    assert torch.allclose(k.outputscales.detach(), mix, atol=1e-6)
    assert torch.allclose(k.lengthscales.detach().reshape(-1), ls, atol=1e-6)
    assert abs(float(k.outputscales.sum()) - 1.0) < 1e-6
    g = create_strictly_additive_kernel(5, weighted=False)
    assert not g.raw_outputscales.requires_grad and g.raw_lengthscales.shape == (1, 5)
    assert torch.allclose(g.outputscales.detach(), torch.full((5,), 0.2), atol=1e-6)
    m = create_strictly_additive_kernel(5, memory_efficient=True, init_lengthscale_range=(2.0, 2.0))
    assert torch.allclose(m.lengthscale.detach(), torch.full((1, 5), 2.0), atol=1e-6)


def test_family_validation_errors():
    from rpgp_amd.training import create_additive_rp_kernel, create_rp_poly_kernel, create_exact_gp
    create_additive_rp_kernel(6, 3, k=2, batch_kernel=False, kernel_type="Matern")     # radial k-dimensional Matern: served
    with pytest.raises(NotImplementedError):
        create_additive_rp_kernel(6, 3, k=40, batch_kernel=False, kernel_type="Matern")
    create_additive_rp_kernel(30, 2, k=24, batch_kernel=False, kernel_type="Matern")   # non-RBF: k <= 32, J k <= 64 columns
    with pytest.raises(NotImplementedError):
        create_additive_rp_kernel(30, 3, k=24, batch_kernel=False, kernel_type="Matern")   # 72 columns
    create_additive_rp_kernel(6, 3, k=7, batch_kernel=False)            # any k <= 20 is served (padded group)
    with pytest.raises(NotImplementedError):
        create_additive_rp_kernel(6, 1, k=21, batch_kernel=False)
    # grid interpolation wraps whatever 1-D sub-kernel `_map_to_kernel` returned (training_routines.py:157-158): served for every
    # kernel type since round 6; only k > 1 (a k-dimensional grid per projection group) is not built
    create_additive_rp_kernel(6, 3, kernel_type="Matern", ski=True, ski_options={"grid_size": 64})
    with pytest.raises((NotImplementedError, ValueError)):
        create_additive_rp_kernel(6, 3, k=2, batch_kernel=False, ski=True, ski_options={"grid_size": 64})
    with pytest.raises(ValueError):
        create_rp_poly_kernel(6, 1, 3, kernel_type="bogus")
    with pytest.raises(ValueError):
        create_rp_poly_kernel(6, 1, 3, activation="relu")
    X, y = _problem(20, 4)
    # products of non-RBF sub-kernels (the ProductKernel groups of the rp_poly kinds): served since round 6
    m, _ = create_exact_gp(X, y, "general_rp_poly", noise_prior=False, degrees=[1, 2], kernel_type="Matern")
    assert m.covar_module.base_kernel.product
    with pytest.raises(NotImplementedError):
        create_exact_gp(X, y, "deep_rp_poly", noise_prior=False)
    with pytest.raises(ValueError):
        create_exact_gp(X, y, "nonsense", noise_prior=False)


def test_family_predictions_match_dense(oracle_backend):
    from rpgp_amd.training import create_exact_gp
    X, y = _problem(50, 4, seed=4)
    Xs = torch.randn(9, 4, generator=torch.Generator().manual_seed(8))
    torch.manual_seed(0)
    model, lik = create_exact_gp(X, y, "rp_poly", noise_prior=False, J=4, k=1, weighted=True, kernel_type="InverseMQ")
    model.eval(); lik.eval()
    out = model(Xs)
    base = model.covar_module.base_kernel
    P = base.projection_module.weight.detach().double().t()
    ls = base.lengthscales.detach().double().reshape(-1)
    w = base.outputscales.detach().double().numpy()
    s = float(model.covar_module.outputscale)
    noise = float(lik.noise)
    Z, Zs = ((X.double() @ P) / ls).numpy(), ((Xs.double() @ P) / ls).numpy()
    K = fmo.kernel_matrix(Z, Z, "InverseMQ", 1, w, s) + noise * np.eye(50)
    Ks = fmo.kernel_matrix(Zs, Z, "InverseMQ", 1, w, s)
    Kss = fmo.kernel_matrix(Zs, Zs, "InverseMQ", 1, w, s)
    c = float(model.mean_module.constant)
    mean = Ks @ np.linalg.solve(K, y.double().numpy() - c) + c
    cov = Kss - Ks @ np.linalg.solve(K, Ks.T)
    assert np.allclose(out.mean.numpy(), mean, rtol=1e-4, atol=1e-5)
    assert np.allclose(out.covariance.numpy(), cov, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("kind,model_kwargs", [
    ("rp_poly", dict(J=5, k=1, weighted=True, kernel_type="RBF")),
    ("strictly_additive", dict(weighted=False, kernel_type="RBF")),
    # round 6: the wrapped sub-kernel may be any of the reference's types (training_routines.py:47-88 with :157-158)
    ("rp_poly", dict(J=5, k=1, weighted=True, kernel_type="InverseMQ")),
    ("rp_poly", dict(J=4, k=1, weighted=False, kernel_type="Matern")),
])
def test_weighted_ski_kinds_track_the_exact_kernel(oracle_backend, kind, model_kwargs):
    """`ski: true` on the weighted kinds (additive_rp_J20_K1_ski.json, additive_deterministic_spec_unweighted_ski.json):
    with a 1024-point grid the interpolated operator reproduces the exact family operator's MLL and gradients, including
    the gradient of the per-component output scales (which ride in the SKI grid-parameter block)."""
    from rpgp_amd.training import create_exact_gp
    from rpgp_amd.models import ExactMarginalLogLikelihood
    from rpgp_amd.operators import SKIAdditiveOperator
    X, y = _problem(70, 4, seed=5)
    grads = {}
    for ski in (False, True):
        torch.manual_seed(7)
        kw = dict(model_kwargs, ski=ski, ski_options={"grid_size": 1024, "num_dims": 1} if ski else None,
                  init_mixin_range=(0.5, 1.5), init_lengthscale_range=(1.0, 2.0))
        if kind == "rp_poly":
            kw["learn_proj"] = False
        model, lik = create_exact_gp(X, y, kind, noise_prior=True, **kw)
        mll = ExactMarginalLogLikelihood(lik, model)
        model.train()
        out = model(X)
        assert isinstance(out.covariance, SKIAdditiveOperator) == ski
        val = mll(out, y)
        val.backward()
        base = model.covar_module.base_kernel
        grads[ski] = (val.item(), base.raw_lengthscales.grad.clone(),
                      None if base.raw_outputscales.grad is None else base.raw_outputscales.grad.clone(),
                      model.covar_module.raw_outputscale.grad.clone())
    (v0, gl0, gw0, gs0), (v1, gl1, gw1, gs1) = grads[False], grads[True]
    # (the Matern-1.5 kernel has a kink at zero distance: cubic interpolation of it converges with h^2, not h^4)
    loose = 10.0 if model_kwargs["kernel_type"] == "Matern" else 1.0
    assert abs(v0 - v1) < 2e-4 * loose * max(1.0, abs(v0))
    assert torch.allclose(gl0, gl1, rtol=3e-2 * loose, atol=2e-4 * loose)
    assert torch.allclose(gs0, gs1, rtol=3e-2 * loose, atol=2e-4 * loose)
    if model_kwargs["weighted"]:
        assert torch.allclose(gw0, gw1, rtol=3e-2 * loose, atol=2e-4 * loose)
    else:
        assert gw0 is None and gw1 is None


def test_double_family_kinds_and_ski_train(oracle_backend):
    """`--double` (training_routines.py:481): every family member trains in float64 (runtime-(kind, group) kernels on the
    device; here the CPU test double)."""
    from rpgp_amd.training import train_exact_gp
    X, y = _problem(30, 4)
    tk = {"verbose": False, "optimizer": "adam", "max_iter": 2, "lr": 0.1, "patience": 20, "smooth": True}
    for kind, mk in (("rp_poly", dict(J=3, k=1, noise_prior=True, weighted=True)),
                     ("additive_rp", dict(J=3, noise_prior=True, kernel_type="Matern", learn_proj=False, prescale=True)),
                     ("additive_rp", dict(J=2, k=2, batch_kernel=False, noise_prior=True, kernel_type="InverseMQ",
                                          learn_proj=False, prescale=True))):
        metrics, pred, model = train_exact_gp(X, y, X, y, kind, mk, tk, double=True, skip_random_restart=True)
        assert all(p.dtype == torch.float64 for p in model.parameters()) and np.isfinite(metrics["test_nll"])
    # ... and (round 4) the grid-interpolation operator: float64 parity kernels (rpgp_ski_f64.hip)
    metrics, pred, model = train_exact_gp(X, y, X, y, "additive_rp",
                                          dict(J=3, noise_prior=True, learn_proj=False, prescale=True, ski=True,
                                               ski_options={"grid_size": 64, "num_dims": 1}), tk, double=True,
                                          skip_random_restart=True)
    assert all(p.dtype == torch.float64 for p in model.parameters()) and np.isfinite(metrics["test_nll"])


def test_multi_additive_kernel_groups_and_operator(oracle_backend):
    """create_multi_additive_kernel (training_routines.py:247-258): every feature subset up to max_degree, through the
    mixed-size operator; product, diagonal, rows and dense form agree with the dense restatement."""
    from rpgp_amd.training import create_multi_additive_kernel
    from rpgp_amd.operators import MixedGroupOperator
    torch.manual_seed(5)
    k = create_multi_additive_kernel(4, 3, weighted=True, init_lengthscale_range=(0.8, 1.6), init_mixin_range=(0.5, 1.5))
    assert sorted(len(g) for g in k.groups) == [1] * 4 + [2] * 6 + [3] * 4
    assert sorted(k.groups[:4]) == [(0,), (1,), (2,), (3,)] and k.k is None
    assert abs(float(k.outputscales.sum()) - 1.0) < 1e-6
    X, _ = _problem(40, 4, seed=2)
    s = torch.tensor(1.3)
    op = k(X, None, outputscale=s) if not hasattr(k, "operator") else k.forward(X, None, outputscale=s)
    assert isinstance(op, MixedGroupOperator) and len(op.buckets) == 3
    Z = (X.double() @ k.projection_module.weight.double().t()) / k.lengthscales.detach().double().reshape(-1)
    Kd = 1.3 * _dense_family(Z, "RBF", k.component_degrees, k.outputscales.detach().double())
    V = torch.randn(40, 3, generator=torch.Generator().manual_seed(1))
    assert torch.allclose(op._matmul(V, noise=0.2).double(), Kd @ V.double() + 0.2 * V.double(), atol=2e-5)
    assert torch.allclose(op.to_dense().double(), Kd, atol=2e-6)
    assert torch.allclose(op._diagonal().double(), Kd.diagonal(), atol=2e-6)
    idx = torch.tensor([3, 17, 3])
    assert torch.allclose(op._get_rows(idx).double(), Kd[idx], atol=2e-6)
    X2, _ = _problem(9, 4, seed=3)
    opr = k.forward(X, X2, outputscale=s)
    Z2 = (X2.double() @ k.projection_module.weight.double().t()) / k.lengthscales.detach().double().reshape(-1)
    n, m = 40, 9
    Kr = torch.zeros(n, m, dtype=torch.float64)
    col = 0
    for c, kk in enumerate(k.component_degrees):
        d2 = sum((Z[:, col + q:col + q + 1] - Z2[:, col + q:col + q + 1].t()) ** 2 for q in range(kk))
        Kr += k.outputscales.detach().double()[c] * torch.exp(-0.5 * d2)
        col += kk
    W = torch.randn(9, 2, generator=torch.Generator().manual_seed(4))
    assert torch.allclose(opr._matmul(W).double(), 1.3 * Kr @ W.double(), atol=2e-5)
    assert torch.allclose(opr.t()._matmul(V).double(), 1.3 * Kr.t() @ V.double(), atol=2e-5)


def test_model_average_weights_and_routine(oracle_backend, tmp_path):
    """train_exact_gp_model_average (training_routines.py:631-676) + the runner's `kind: model_average` disambiguation
    (gp_experiment_runner.py:299-304).  ModelAverage itself is unpinned (fitting/sampling.py is not in the reference checkout):
    the mixture identities are checked instead."""
    import json
    from rpgp_amd.likelihoods import MultivariateNormal
    from rpgp_amd.training import ModelAverage, train_exact_gp_model_average
    from rpgp_amd import runner, specs
    m1, m2 = torch.tensor([0.0, 1.0, 2.0]), torch.tensor([1.0, 1.0, 1.0])
    p1 = MultivariateNormal(m1, torch.tensor([1.0, 2.0, 0.5]), diagonal_only=True)
    p2 = MultivariateNormal(m2, torch.eye(3) * 0.7)
    ma = ModelAverage([p1, p2], [-1.0, -1.0 + math.log(3.0)])
    assert torch.allclose(ma.weights, torch.tensor([0.25, 0.75], dtype=torch.float64))
    assert torch.allclose(ma.mean(), 0.25 * m1 + 0.75 * m2) and torch.allclose(ma.sample_mean(), 0.5 * (m1 + m2))
    y = torch.tensor([0.3, 0.9, 1.4])
    want = math.log(0.25 * math.exp(float(p1.log_prob(y))) + 0.75 * math.exp(float(p2.log_prob(y))))
    assert abs(float(ma.log_prob(y)) - want) < 1e-9
    one = ModelAverage([p2], [-3.0])
    assert abs(float(one.log_prob(y)) - float(p2.log_prob(y))) < 1e-9 and torch.allclose(one.mean(), m2)
    with pytest.raises(ValueError):
        ModelAverage([p1], [0.0, 1.0])

    X, y = _problem(50, 4, seed=6)
    Xt, yt = _problem(12, 4, seed=7)
    tk = {"verbose": False, "optimizer": "adam", "max_iter": 3, "lr": 0.1, "patience": 20, "smooth": True}
    mk = dict(J=3, noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True, varying_params={"J": [1, 2, 4]})
    torch.manual_seed(0)
    metrics, mean, model = train_exact_gp_model_average(X, y, Xt, yt, "additive_rp", mk, tk)
    assert model is None and mean.shape == (12,) and mk["varying_params"] == {"J": [1, 2, 4]} and mk["J"] == 3
    assert set(metrics) == {"test_nll", "sampled_mean_mse", "normal_mean_mse"}
    assert all(np.isfinite(v) for v in metrics.values())
    m2_, _, _ = train_exact_gp_model_average(X, y, Xt, yt, "additive_rp", mk, tk, skip_posterior_variances=True)
    assert "test_nll" not in m2_

    spec = specs.get("ma_dpa_gp_ard")
    spec["varying_params"] = {"J": [1, 3]}
    spec["base_model_kwargs"]["train_kwargs"].update(max_iter=2)
    f = tmp_path / "ma.json"
    json.dump(spec, open(f, "w"))
    df = runner.main(["-m", str(f), "-d", "synthetic:tiny", "-o", str(tmp_path / "ma.csv"), "--no_cv"])
    assert np.isfinite(df.iloc[0]["normal_mean_mse"]) and np.isfinite(df.iloc[0]["rmse"])


def test_utils_getters_on_this_builds_kernels():
    """The four introspection getters of the reference's utils.py (:7-52), kept as an extra."""
    from rpgp_amd import utils
    from rpgp_amd.kernels import ScaleKernel
    from rpgp_amd.training import create_additive_rp_kernel, create_rp_poly_kernel
    k = ScaleKernel(create_additive_rp_kernel(5, 4, prescale=True, init_lengthscale_range=(2.0, 2.0)))
    ls = utils.get_lengthscales(k)
    assert isinstance(ls, torch.Tensor) and torch.allclose(ls.detach().reshape(-1), torch.full((5,), 2.0), atol=1e-6)
    assert utils.get_mixins(k) is None and float(utils.get_outputscale(k)) == float(k.outputscale)
    g = create_rp_poly_kernel(5, 2, 3, weighted=True)
    assert [len(c) for c in utils.get_lengthscales(g)] == [2, 2, 2] and len(utils.get_mixins(g)) == 3
    assert utils.format_for_str([0.12345, torch.tensor([1.23456])]) == [0.123, [1.235]]
    assert utils.format_for_str("x") == ""
