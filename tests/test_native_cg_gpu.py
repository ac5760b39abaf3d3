"""Native mBCG executor (rpgp_mbcg_solve) and the torch-op loop of linear_cg.py, each against float64 oracle solves
(np.linalg.solve on the oracle's dense matrix); the two loops run the same algorithm, so their stopping iterations agree
up to a few percent wherever the tolerance is clear of the fp32 floor."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops_pair(gpu_device, N, J, noise, ski=False, spread=1.0, seed=0):
    from rpgp_amd.operators import AdditiveRPOperator, SKIAdditiveOperator, AddedDiagOperator
    g = torch.Generator().manual_seed(seed)
    Z = (torch.randn(N, J, generator=g) * spread).to(gpu_device)
    cls = SKIAdditiveOperator if ski else AdditiveRPOperator
    base = cls(Z, None, torch.tensor(0.9, device=gpu_device), 1.0 / J)
    return base, AddedDiagOperator(base, torch.tensor(noise, device=gpu_device))


def _oracle_khat(base, N, J, noise, ski):
    """float64 dense K + noise I of the operator under test, from the oracle (never from the HIP path)."""
    Zd = base.Z1.double().cpu().numpy()
    if ski:
        from oracle import ski as orc_ski
        gp = base.gp.double().cpu().numpy()
        K = orc_ski.dense_kernel(Zd, Zd, 0.9 / J, G=base.grid_size, grid=(gp[0], gp[1]))
    else:
        from oracle import dense_gp as orc
        K = orc.additive_rbf(Zd, Zd) * (0.9 / J)
    return K + noise * np.eye(N)


# (N, J, T, ski, spread, precond, tol): `tol` sits >= 10x above cond(Khat) * eps_fp32 so that the stopping iteration is
# a property of the algorithm and not of the rounding order (the N = 700 un-preconditioned case: cond ~ 2e3 -> floor ~6e-5,
# hence 1e-3)
@pytest.mark.parametrize("N,J,T,ski,spread,precond,tol", [(3000, 20, 11, False, 1.0, True, 1e-4),
                                                          (2500, 20, 1, False, 1.0, False, 1e-3),
                                                          (2600, 8, 5, False, 30.0, True, 1e-4),
                                                          (4000, 3, 11, True, 1.0, True, 1e-4),
                                                          (700, 20, 16, False, 1.0, False, 1e-3)])
def test_native_and_torch_loops_match_fp64_oracle_solve(gpu_device, N, J, T, ski, spread, precond, tol):
    """Both CG loops (native executor, torch-op loop) against np.linalg.solve on the float64 oracle matrix; iteration
    counts only with a relative slack; Lanczos tridiagonals through the SLQ log-det vs the oracle's slogdet."""
    from rpgp_amd import settings, linear_cg as lcg
    from rpgp_amd.precond import pivoted_cholesky, WoodburyPreconditioner
    noise = 0.3
    base, khat = _ops_pair(gpu_device, N, J, noise, ski, spread, seed=N)
    rhs = torch.randn(N, T, generator=torch.Generator().manual_seed(1)).to(gpu_device)
    if T > 2:
        rhs[:, 2] = 0.0                                       # a zero right-hand side stays zero
    pre = None
    if precond:
        pre = WoodburyPreconditioner(pivoted_cholesky(base._diagonal(), base._get_rows, 15), noise)
    nt = min(T, 10) if T > 1 else 0
    kw = dict(n_tridiag=nt, tolerance=tol, max_iter=500, max_tridiag_iter=20, preconditioner=pre)
    before = lcg.stats.get("native_calls", 0)
    out_n = lcg.linear_cg(khat._matmul, rhs, operator=khat, **kw)
    assert lcg.stats.get("native_calls", 0) == before + 1, "native executor was not used"
    it_native = lcg.stats["last_iterations"]
    out_t = lcg.linear_cg(khat._matmul, rhs, **kw)
    it_torch = lcg.stats["last_iterations"]
    xn, xt = (out_n[0], out_t[0]) if nt else (out_n, out_t)

    Kh = _oracle_khat(base, N, J, noise, ski)
    b = rhs.double().cpu().numpy()
    x_ref = np.linalg.solve(Kh, b)
    cond = np.linalg.cond(Kh) if N <= 1000 else None
    for name, x in (("native", xn), ("torch", xt)):
        xd = x.double().cpu().numpy()
        # true float64 residual of the returned iterate: unit-norm columns are solved to a mean residual < tol
        res = np.linalg.norm(Kh @ xd - b, axis=0) / np.maximum(np.linalg.norm(b, axis=0), 1e-300)
        assert res.mean() < 2.0 * tol, (name, res)
        # ||x - x*|| / ||x*|| <= cond * residual; the A-norm bound is loose, so gate at 1/noise-scaled tolerance
        err = np.linalg.norm(xd - x_ref) / np.linalg.norm(x_ref)
        assert err < 100.0 * tol, (name, err)
        if T > 2:
            assert float(np.abs(xd[:, 2]).max()) == 0.0
    if cond is not None:
        assert cond * 6e-8 * 10 <= tol * 1.5, "test case sits on the fp32 floor: cond %g" % cond
    # same algorithm, two rounding orders: the stopping iteration may differ by a few percent, not more
    assert abs(it_native - it_torch) <= max(2, int(0.1 * it_torch)), (it_native, it_torch)
    if nt:
        from rpgp_amd.inv_quad_logdet import slq_logdet
        tn, tt = out_n[1].cpu().double(), out_t[1].cpu().double()
        cols = [c for c in range(nt) if c != 2]
        if pre is None:
            # un-preconditioned: E[z^T log(Khat) z] over Gaussian probes = log det Khat; 9 probes -> a few percent
            ld_ref = np.linalg.slogdet(Kh)[1]
            for name, tri in (("native", tn), ("torch", tt)):
                ld = float(slq_logdet(tri[cols], N))
                assert abs(ld - ld_ref) < 0.1 * abs(ld_ref) + 0.02 * N, (name, ld, ld_ref)
        ln, lt = float(slq_logdet(tn[cols], N)), float(slq_logdet(tt[cols], N))
        assert abs(ln - lt) < 2e-2 * abs(lt) + 1.0


def test_native_cg_warns_at_max_iter(gpu_device):
    from rpgp_amd import linear_cg as lcg
    base, khat = _ops_pair(gpu_device, 2000, 20, 1e-3, seed=5)
    rhs = torch.randn(2000, 3, generator=torch.Generator().manual_seed(2)).to(gpu_device)
    with pytest.warns(lcg.NumericalWarning):
        lcg.linear_cg(khat._matmul, rhs, tolerance=1e-9, max_iter=12, operator=khat)


def test_training_step_uses_native_executor(gpu_device):
    from rpgp_amd import settings, linear_cg as lcg
    from tests.test_gp_gpu import _gpu_model
    prob, model, lik, mll = _gpu_model(gpu_device, 2600, 8, 20, 9, 0.3)
    model.train()
    before = lcg.stats.get("native_calls", 0)
    with settings.cg_tolerance(0.01), settings.deterministic_probes(True):
        v = mll(model(model.train_inputs), model.train_targets)
        v.backward()
    assert lcg.stats.get("native_calls", 0) == before + 1
    assert np.isfinite(v.item()) and torch.isfinite(model.covar_module.base_kernel.raw_lengthscale.grad).all()


def test_native_cached_k_operator(gpu_device):
    """RPGP_OP_DENSE (cached K + rocBLAS GEMM inside the native executor) solves the same system as the fused operator."""
    from rpgp_amd import linear_cg as lcg
    from rpgp_amd.operators import DenseOperator
    from rpgp_amd.precond import WoodburyPreconditioner
    from rpgp_amd import ops
    base, khat = _ops_pair(gpu_device, 3000, 20, 0.3, seed=11)
    dense = DenseOperator(base.to_dense(), 0.3)
    rhs = torch.randn(3000, 11, generator=torch.Generator().manual_seed(1)).to(gpu_device)
    pre = WoodburyPreconditioner(ops.pivoted_cholesky(base.Z1, base._scale, 15), 0.3)
    before = lcg.stats.get("native_calls", 0)
    xd, td = lcg.linear_cg(dense._matmul, rhs, n_tridiag=10, tolerance=1e-5, max_iter=500, preconditioner=pre, operator=dense)
    xf, tf = lcg.linear_cg(khat._matmul, rhs, n_tridiag=10, tolerance=1e-5, max_iter=500, preconditioner=pre, operator=khat)
    assert lcg.stats.get("native_calls", 0) == before + 2
    assert float((xd - xf).norm() / xf.norm()) < 1e-3
    resid = (dense._matmul(xd) - rhs).norm(dim=0) / rhs.norm(dim=0)
    assert float(resid.max()) < 1e-3


@pytest.mark.parametrize("wide", [False, True])
def test_native_symcache_operator(gpu_device, wide):
    """RPGP_OP_SYMCACHE (packed symmetric cache inside the native executor) solves the same system as the fused
    operator, and the operator's dense form / diagonal agree with the fused operator's."""
    from rpgp_amd import linear_cg as lcg
    from rpgp_amd.operators import SymCachedOperator
    from rpgp_amd.precond import WoodburyPreconditioner
    from rpgp_amd import ops
    base, khat = _ops_pair(gpu_device, 4200, 20, 0.3, seed=11)
    sym = SymCachedOperator(base.to_symcache(wide=wide), base._scale, 0.3, diag_value=base._scale * base.num_projections)
    rhs = torch.randn(4200, 11, generator=torch.Generator().manual_seed(1)).to(gpu_device)
    pre = WoodburyPreconditioner(ops.pivoted_cholesky(base.Z1, base._scale, 15), 0.3)
    before = lcg.stats.get("native_calls", 0)
    xs, ts = lcg.linear_cg(sym._matmul, rhs, n_tridiag=10, tolerance=1e-5, max_iter=500, preconditioner=pre, operator=sym)
    xf, tf = lcg.linear_cg(khat._matmul, rhs, n_tridiag=10, tolerance=1e-5, max_iter=500, preconditioner=pre, operator=khat)
    assert lcg.stats.get("native_calls", 0) == before + 2
    assert float((xs - xf).norm() / xf.norm()) < 1e-3
    resid = (khat._matmul(xs) - rhs).norm(dim=0) / rhs.norm(dim=0)
    assert float(resid.max()) < 1e-3
    assert torch.allclose(ts, tf, rtol=1e-2, atol=1e-3)
    small, _ = _ops_pair(gpu_device, 700, 6, 0.3, seed=3)
    sym_small = SymCachedOperator(small.to_symcache(wide=wide), small._scale, 0.3,
                                  diag_value=small._scale * small.num_projections)
    ref = small.to_dense()
    ref.diagonal().add_(0.3)
    assert float((sym_small.to_dense() - ref).abs().max()) < 1e-5
    assert torch.allclose(sym_small._diagonal(), ref.diagonal(), atol=1e-5)


def test_native_cg_stops_on_stagnation(gpu_device):
    """Badly conditioned system (noise 1e-7 on a smooth kernel): fp32 CG cannot reach the tolerance; the device-side
    stagnation rule stops the executor long before max_iter and the non-convergence warning is raised."""
    import warnings
    from rpgp_amd import linear_cg as lcg, settings
    from rpgp_amd.operators import AdditiveRPOperator, AddedDiagOperator
    N, J = 3000, 4
    Z = (torch.randn(N, J, generator=torch.Generator().manual_seed(0)) * 0.3).to(gpu_device)
    khat = AddedDiagOperator(AdditiveRPOperator(Z, None, torch.tensor(1.0, device=gpu_device), 1.0 / J),
                             torch.tensor(1e-7, device=gpu_device))
    rhs = torch.randn(N, 2, generator=torch.Generator().manual_seed(1)).to(gpu_device)
    with settings.cg_stagnation_window(50), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        lcg.linear_cg(khat._matmul, rhs, tolerance=1e-6, max_iter=100000, operator=khat)
    assert lcg.stats["last_iterations"] < 2500          # n_iter = min(max_iter, N) = 3000 without the rule
    assert any("CG terminated" in str(x.message) for x in w)


@pytest.mark.parametrize("tol,max_iter,check_every,nt", [(1e-2, 500, 1, 0), (1e-3, 500, 3, 0), (1e-30, 7, 1, 0),
                                                         (1e-30, 1, 1, 0), (0.5, 500, 1, 0), (1e-2, 500, 1, 4),
                                                         (1e-30, 25, 1, 4), (1e-3, 12, 5, 0)])
def test_native_stopping_rule_edge_cases(gpu_device, tol, max_iter, check_every, nt):
    """The device-side convergence flag + lagged host polling must reproduce the torch loop's stopping rule exactly:
    same iteration count (>= 10 iterations, after the Lanczos window, every `check_every`-th test, max_iter caps) and
    the same iterate, including when convergence is detected on the last allowed iteration."""
    from rpgp_amd import linear_cg as lcg
    base, khat = _ops_pair(gpu_device, 2048, 20, 0.5, seed=3)
    rhs = torch.randn(2048, 5, generator=torch.Generator().manual_seed(2)).to(gpu_device)
    kw = dict(n_tridiag=nt, tolerance=tol, max_iter=max_iter, max_tridiag_iter=20, check_every=check_every)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out_n = lcg.linear_cg(khat._matmul, rhs, operator=khat, **kw)
        it_n = lcg.stats["last_iterations"]
        out_t = lcg.linear_cg(khat._matmul, rhs, **kw)
        it_t = lcg.stats["last_iterations"]
    xn, xt = (out_n[0], out_t[0]) if nt else (out_n, out_t)
    # the two loops round differently, so a residual sitting on the tolerance may be accepted one test apart
    assert abs(it_n - it_t) <= check_every and (it_n - it_t) % check_every == 0 or it_n == it_t, (it_n, it_t)
    assert float((xn - xt).norm() / xt.norm()) < 2e-3
    if nt:
        assert out_n[1].shape == out_t[1].shape


def test_best_iterate_is_returned_when_fp32_cg_breaks_down(gpu_device):
    """cond(Khat) * fp32 eps >~ 1: the recurrence residual stalls or grows.  Both loops must hand back the iterate with the
    smallest tested residual (not the last one) and report that residual."""
    import warnings
    from rpgp_amd import linear_cg as lcg, settings
    from rpgp_amd.operators import AdditiveRPOperator, AddedDiagOperator
    N, J = 20000, 3
    Z = (torch.randn(N, J, generator=torch.Generator().manual_seed(0)) * 0.5).to(gpu_device)
    khat = AddedDiagOperator(AdditiveRPOperator(Z, None, torch.tensor(3.0, device=gpu_device), 1.0 / J),
                             torch.tensor(2e-2, device=gpu_device))
    rhs = torch.randn(N, 2, generator=torch.Generator().manual_seed(1)).to(gpu_device)
    for operator in (khat, None):                         # native executor, torch-op loop
        with settings.cg_stagnation_window(60), warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            x = lcg.linear_cg(khat._matmul, rhs, tolerance=1e-9, max_iter=3000, operator=operator)   # below the fp32 floor
        assert torch.isfinite(x).all()
        true_res = float(((khat._matmul(x) - rhs).norm(dim=0) / rhs.norm(dim=0)).mean())
        # (the recurrence residual may keep shrinking below the tolerance while the TRUE residual floors near
        # eps * cond: whether a non-convergence warning fires is not asserted, the quality of the returned iterate is)
        assert true_res < 0.05


def test_history_quadrature_matches_tridiagonal_eigendecomposition(gpu_device):
    """`linear_cg(..., lanczos="history")`: the library's quadrature on the executor's coefficient history against torch's
    eigendecomposition of the tridiagonal matrices of the same history; the preconditioner's log-determinant through pinned
    host memory against the device scalar."""
    from rpgp_amd import linear_cg as lcg, settings
    from rpgp_amd.inv_quad_logdet import slq_logdet
    from rpgp_amd.operators import AddedDiagOperator, AdditiveRPOperator
    from rpgp_amd.precond import build_preconditioner
    torch.manual_seed(3)
    N, J = 3000, 20
    Z = (torch.randn(N, J) * 0.7).to(gpu_device)
    op = AdditiveRPOperator(Z, None, outputscale=torch.tensor(0.9, device=gpu_device))
    khat = AddedDiagOperator(op, torch.tensor(0.2, device=gpu_device))
    pre = build_preconditioner(op, 0.2, settings)
    probes = torch.randn(N, 10, device=gpu_device)
    rhs = torch.cat([probes / probes.norm(dim=0, keepdim=True), torch.randn(N, 1, device=gpu_device)], dim=1)
    x, hist = lcg.linear_cg(khat._matmul, rhs, n_tridiag=10, tolerance=1e-4, max_iter=500, preconditioner=pre,
                            operator=khat, lanczos="history")
    host = float(slq_logdet(hist, N))
    eig = float(slq_logdet(hist.tridiagonals(), N))
    assert abs(host - eig) < 1e-10 * abs(eig)
    import math
    ld = pre.logdet()
    assert abs(ld - (float(pre._logdet_cap) + (N - pre.k) * math.log(0.2))) < 1e-9 * abs(ld)


@pytest.mark.parametrize("N,J,ski", [(3000, 20, False), (9000, 8, False), (4000, 3, True)])
def test_consumer_side_reductions_are_bitwise_the_separate_launches(gpu_device, monkeypatch, N, J, ski):
    """The executor's two reduction forms (passes adding up the slabs themselves / `k_reduce` launches, RPGP_CG_DIRECT=0) give
    bit-identical solutions, coefficient histories and iteration counts — for the exact and the grid-interpolation operator;
    so do two runs of the same form."""
    from rpgp_amd import linear_cg as lcg
    from rpgp_amd.precond import pivoted_cholesky, WoodburyPreconditioner
    noise = 0.2
    base, khat = _ops_pair(gpu_device, N, J, noise, ski, 1.0, seed=N + 1)
    rhs = torch.randn(N, 11, generator=torch.Generator().manual_seed(3)).to(gpu_device)
    pre = WoodburyPreconditioner(pivoted_cholesky(base._diagonal(), base._get_rows, 15), noise)
    kw = dict(n_tridiag=10, tolerance=1e-4, max_iter=500, max_tridiag_iter=20, preconditioner=pre, operator=khat, lanczos="history")
    outs = []
    for env in ({}, {"RPGP_CG_DIRECT": "0"}, {"RPGP_CG_DIRECT": "0"}, {}):
        for k in ("RPGP_CG_DIRECT",):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        x, hist = lcg.linear_cg(khat._matmul, rhs, **kw)
        outs.append((x.clone(), hist.alpha.copy(), hist.beta.copy(), lcg.stats["last_iterations"]))
    for x, a, b, it in outs[1:]:
        assert it == outs[0][3]
        assert torch.equal(x, outs[0][0])
        assert np.array_equal(a, outs[0][1]) and np.array_equal(b, outs[0][2])


