"""SKI path (config 5 spec `additive_spread_prescale_Jd_ski.json`): HIP kernels through the C-ABI against the float64
dense SKI oracle, and the interpolation error against the exact additive kernel."""
import numpy as np
import pytest
import torch

from oracle import dense_gp as orc
from oracle import ski as sko

pytestmark = pytest.mark.gpu


def _spec(name):
    from rpgp_amd import specs
    return specs.get(name)


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("N,J,T,G", [(500, 3, 1, 128), (1000, 3, 11, 1024), (777, 8, 4, 256), (3000, 3, 13, 1024),
                                     (64, 1, 1, 64)])
def test_ski_mvm_matches_dense_ski_oracle(gpu_device, N, J, T, G):
    from rpgp_amd import ops
    rng = np.random.default_rng(N + G)
    Z = rng.standard_normal((N, J)).astype(np.float32)
    V = rng.standard_normal((N, T)).astype(np.float32)
    Zt = torch.from_numpy(Z).to(gpu_device)
    gp = ops.ski_grid(Zt, None, G)
    g0, h = sko.grid_params(Z, None, G)
    gph = gp.cpu().numpy()
    assert abs(gph[0] - g0) < 1e-5 * max(1, abs(g0)) and abs(gph[1] - h) < 1e-5 * h
    K = sko.dense_kernel(Z, Z, 0.4, G, (float(gph[0]), float(gph[1])))
    ref = K @ V.astype(np.float64) + 0.2 * V
    out = ops.ski_mvm(Zt, Zt, gp, torch.from_numpy(V).to(gpu_device), 0.4, 0.2, G)
    assert _rel(out.cpu().numpy(), ref) < 2e-5
    diag = ops.ski_diag(Zt, gp, 0.4, G)
    np.testing.assert_allclose(diag.cpu().numpy(), np.diag(K), rtol=2e-5, atol=1e-6)


def test_ski_rect_and_exact_kernel_error(gpu_device):
    """With G = 1024 the cubic interpolation reproduces the exact additive RBF kernel to ~1e-6."""
    from rpgp_amd import ops
    rng = np.random.default_rng(0)
    Z1 = rng.standard_normal((300, 3)).astype(np.float32)
    Z2 = (rng.standard_normal((2000, 3)) * 1.3).astype(np.float32)
    V = rng.standard_normal((2000, 2)).astype(np.float32)
    t1, t2 = torch.from_numpy(Z1).to(gpu_device), torch.from_numpy(Z2).to(gpu_device)
    gp = ops.ski_grid(t1, t2, 1024)
    out = ops.ski_mvm(t1, t2, gp, torch.from_numpy(V).to(gpu_device), 1.0 / 3, 0.0, 1024).cpu().numpy()
    gph = gp.cpu().numpy()
    ref_ski = sko.dense_kernel(Z1, Z2, 1.0 / 3, 1024, (float(gph[0]), float(gph[1]))) @ V.astype(np.float64)
    ref_exact = orc.mvm(Z1, Z2, V, 1.0 / 3)
    assert _rel(out, ref_ski) < 2e-5
    assert _rel(out, ref_exact) < 1e-4


def _c5_problem(gpu_device, T):
    """Config C5 shape (additive_spread_prescale_Jd_ski on 3droad: N = 434 874 * 0.9 = 391 386 train rows, d = J = 3,
    grid 1024); z-scored stand-in features, orthonormal (diversified) projections, unit prescale lengthscale."""
    N, J, G = 391386, 3, 1024
    g = torch.Generator().manual_seed(5)
    X = torch.randn(N, J, generator=g)
    Q = torch.linalg.qr(torch.randn(J, J, generator=g))[0]
    Z = (X @ Q).contiguous()
    V = torch.randn(N, T, generator=g)
    return N, J, G, Z, V


@pytest.mark.parametrize("T", [1, 11])
def test_ski_mvm_and_diag_at_full_c5_size(gpu_device, T):
    """rpgp_ski_mvm / rpgp_ski_diag at exactly the C5 shape against the float64 sparse-W oracle (O(N) memory)."""
    from rpgp_amd import ops
    N, J, G, Z, V = _c5_problem(gpu_device, T)
    Zt, Vt = Z.to(gpu_device), V.to(gpu_device)
    gp = ops.ski_grid(Zt, None, G)
    gph = gp.double().cpu().numpy()
    grid = (float(gph[0]), float(gph[1]))
    scale, noise = 0.8 / J, 0.25
    out = ops.ski_mvm(Zt, Zt, gp, Vt, scale, noise, G).double().cpu().numpy()
    ref = sko.mvm_sparse(Z.numpy(), Z.numpy(), V.numpy(), scale, G, grid, noise)
    col = np.linalg.norm(out - ref, axis=0) / np.linalg.norm(ref, axis=0)
    assert col.max() < 2e-5, col
    if T == 1:
        dg = ops.ski_diag(Zt, gp, scale, G).double().cpu().numpy()
        np.testing.assert_allclose(dg, sko.diag_sparse(Z.numpy(), scale, G, grid), rtol=2e-5, atol=1e-6)
        # bitwise reproducible (fixed-point LDS histograms, fixed-order slab sums)
        again = ops.ski_mvm(Zt, Zt, gp, Vt, scale, noise, G).double().cpu().numpy()
        assert np.array_equal(out, again)


def test_ski_native_mbcg_solve_at_full_c5_size(gpu_device):
    """One native mBCG solve of the C5-shaped system at the runner's tolerances (cg_tol 0.05 train / 0.01 eval,
    gp_experiment_runner.py:324-329), rank-15 pivoted-Cholesky preconditioner; the TRUE float64 residual of the returned
    iterate is checked with the sparse-W oracle operator."""
    from rpgp_amd import ops, linear_cg as lcg, settings
    from rpgp_amd.operators import SKIAdditiveOperator, AddedDiagOperator
    from rpgp_amd.precond import build_preconditioner
    N, J, G, Z, V = _c5_problem(gpu_device, 11)
    Zt = Z.to(gpu_device)
    outputscale, noise = 0.8, 0.25
    base = SKIAdditiveOperator(Zt, None, torch.tensor(outputscale, device=gpu_device), 1.0 / J, grid_size=G)
    khat = AddedDiagOperator(base, torch.tensor(noise, device=gpu_device))
    y = torch.sin(Z).sum(1, keepdim=True)
    rhs = torch.cat([y, V[:, :10]], dim=1).to(gpu_device)            # the training block: residual + 10 probes
    pre = build_preconditioner(base, noise, settings)
    gph = base.gp.double().cpu().numpy()
    grid = (float(gph[0]), float(gph[1]))
    for tol in (0.05, 0.01):
        before = lcg.stats.get("native_calls", 0)
        x = lcg.linear_cg(khat._matmul, rhs, n_tridiag=0, tolerance=tol, max_iter=10000, preconditioner=pre, operator=khat)
        assert lcg.stats.get("native_calls", 0) == before + 1
        xd = x.double().cpu().numpy()
        r = sko.mvm_sparse(Z.numpy(), Z.numpy(), xd, outputscale / J, G, grid, noise) - rhs.double().cpu().numpy()
        res = np.linalg.norm(r, axis=0) / np.linalg.norm(rhs.double().cpu().numpy(), axis=0)
        assert res.mean() < 1.5 * tol, (tol, res)
        assert lcg.stats["last_iterations"] < 200


@pytest.mark.parametrize("N,J,T,G", [(400, 3, 11, 256), (900, 2, 1, 1024), (300, 3, 20, 128)])
def test_ski_bilinear_grad(gpu_device, N, J, T, G):
    from rpgp_amd import ops
    rng = np.random.default_rng(N)
    Z = rng.standard_normal((N, J)).astype(np.float32)
    L = rng.standard_normal((N, T)).astype(np.float32)
    R = rng.standard_normal((N, T)).astype(np.float32)
    Zt = torch.from_numpy(Z).to(gpu_device)
    gp = ops.ski_grid(Zt, None, G)
    gph = gp.cpu().numpy()
    grid = (float(gph[0]), float(gph[1]))
    gZ, gs = ops.ski_bilinear_grad(Zt, gp, torch.from_numpy(L).to(gpu_device), torch.from_numpy(R).to(gpu_device), 0.3, G)
    # d/dscale is linear: objective / scale
    obj = sko.bilinear_objective(Z, L, R, 0.3, G, grid)
    assert abs(gs.item() - obj / 0.3) < 2e-4 * (np.abs(L).sum() * np.abs(R).sum() / N) ** 0.5 + 2e-4 * abs(obj / 0.3)
    # d/dZ by central differences on a few entries (fixed grid)
    gz = gZ.cpu().numpy()
    eps = 1e-4
    scale_ref = np.abs(gz).max()
    for (i, j) in [(0, 0), (N // 2, J - 1), (N - 1, 0), (7, J // 2)]:
        Zp, Zm = Z.astype(np.float64).copy(), Z.astype(np.float64).copy()
        Zp[i, j] += eps
        Zm[i, j] -= eps
        fd = (sko.bilinear_objective(Zp, L, R, 0.3, G, grid) - sko.bilinear_objective(Zm, L, R, 0.3, G, grid)) / (2 * eps)
        assert abs(gz[i, j] - fd) < 2e-3 * scale_ref + 2e-3 * abs(fd)


def test_ski_spec_trains_and_predicts_on_gpu(gpu_device, tmp_path):
    """The config-5 spec (J = d, SKI grid 1024) through the runner on a 3droad-shaped stand-in (reduced N)."""
    import json
    import os
    from rpgp_amd import runner
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = _spec("additive_spread_prescale_Jd_ski.json")
    spec["train_kwargs"]["max_iter"] = 6
    spec["train_kwargs"]["init_iters"] = 2
    sp = tmp_path / "spec.json"
    json.dump(spec, open(sp, "w"))
    runner.SYNTHETIC_SHAPES["road_small"] = (20000, 3)
    torch.manual_seed(0)
    np.random.seed(0)
    df = runner.main(["-m", str(sp), "-d", "synthetic:road_small", "-o", str(tmp_path / "o.csv"), "--no_cv",
                      "--device", "cuda:0"])
    assert "error" not in df.columns
    row = df.iloc[0]
    assert np.isfinite(row["rmse"]) and np.isfinite(row["test_nll"]) and row["rmse"] < 1.0


def test_ski_mll_matches_exact_operator_on_gpu(gpu_device):
    """At grid_size 1024 the SKI MLL and its gradients agree with the exact fused operator: tightly in the
    deterministic Cholesky regime, and within the probe noise of the trace estimator in the CG regime."""
    from rpgp_amd import settings
    from rpgp_amd.kernels import AdditiveStructureRBFKernel, ScaledProjectionKernel, ScaleKernel
    from rpgp_amd.likelihoods import GaussianLikelihood, SmoothedBoxPrior
    from rpgp_amd.models import ExactGPModel, ExactMarginalLogLikelihood
    g = torch.Generator().manual_seed(0)
    N, d = 3000, 3
    X = torch.randn(N, d, generator=g)
    y = torch.sin(X).sum(1) + 0.05 * torch.randn(N, generator=g)
    P = torch.linalg.qr(torch.randn(d, d, generator=g))[0]
    res = {}
    for ski in (False, True):
        for chol in (True, False):
            lin = torch.nn.Linear(d, d, bias=False)
            lin.weight.data = P.t().contiguous()
            k = ScaledProjectionKernel(lin, AdditiveStructureRBFKernel(d, ski=ski, ski_options={"grid_size": 1024, "num_dims": 1}),
                                       prescale=True, ard_num_dims=d)
            k.initialize(lengthscale=torch.tensor([1.2, 0.8, 1.5]))
            sk = ScaleKernel(k)
            lik = GaussianLikelihood(noise_prior=SmoothedBoxPrior(1e-4, 10, sigma=0.01))
            lik.noise = 0.2
            model = ExactGPModel(X.to(gpu_device), y.to(gpu_device), lik, sk).to(gpu_device)
            mll = ExactMarginalLogLikelihood(lik, model)
            model.train()
            with settings.cg_tolerance(1e-6), settings.deterministic_probes(True), settings.num_trace_samples(20), \
                    settings.max_cholesky_size(100000 if chol else 800):
                v = mll(model(model.train_inputs), model.train_targets)
                v.backward()
            res[(ski, chol)] = (v.item(), model.covar_module.base_kernel.raw_lengthscale.grad.cpu().clone(),
                                lik.raw_noise.grad.item())
    # deterministic regime: SKI == exact kernel to interpolation accuracy
    assert abs(res[(True, True)][0] - res[(False, True)][0]) < 1e-5 * abs(res[(False, True)][0])
    assert torch.allclose(res[(True, True)][1], res[(False, True)][1], rtol=1e-3, atol=2e-5)
    assert abs(res[(True, True)][2] - res[(False, True)][2]) < 1e-4 * abs(res[(False, True)][2])
    # CG regime: both estimators scatter around the deterministic value with the probe noise of 20 probes
    for ski in (False, True):
        assert abs(res[(ski, False)][0] - res[(ski, True)][0]) < 2e-3 * abs(res[(ski, True)][0])
        assert torch.allclose(res[(ski, False)][1], res[(ski, True)][1], rtol=0.1, atol=4e-3)
        assert abs(res[(ski, False)][2] - res[(ski, True)][2]) < 2e-2 * abs(res[(ski, True)][2])


@pytest.mark.parametrize("N,J,G", [(3000, 3, 256), (20000, 20, 1024), (70001, 3, 512), (66000, 6, 256)])
def test_ski_fused_pivoted_cholesky_matches_generic(gpu_device, N, J, G):
    """rpgp_ski_pivoted_cholesky (one chip-wide launch per greedy step, entries from the interpolation weights and the
    Toeplitz lags) against the generic row-by-row version that asks the operator for rows.  From N = 65 536 on the factor is
    built column-major in the scratch and written out once (the 70 001-row case: ragged last workgroup)."""
    from rpgp_amd.operators import SKIAdditiveOperator
    from rpgp_amd.precond import pivoted_cholesky
    rng = np.random.default_rng(N)
    Z = torch.from_numpy((rng.standard_normal((N, J)) * 0.8).astype(np.float32)).to(gpu_device)
    op = SKIAdditiveOperator(Z, None, torch.tensor(0.9, device=gpu_device), 1.0 / J, grid_size=G)
    Lf = op.fused_pivoted_cholesky(10)
    Lg = pivoted_cholesky(op._diagonal(), op._get_rows, 10)
    assert Lf is not None and torch.allclose(Lf, Lg, rtol=2e-3, atol=3e-4)
    assert torch.equal(Lf, op.fused_pivoted_cholesky(10))
    # the first ten columns of a rank-17 factor are the rank-10 factor: at (70 001, 3) rank 17 runs the general step kernel, rank 10
    # the one with the own-row operands requested up front — the same arithmetic in the same order
    assert torch.equal(op.fused_pivoted_cholesky(17)[:, :10], Lf)
    from rpgp_amd import ops, _lib
    lib = _lib.load()
    assert lib.rpgp_ski_pivoted_cholesky_work_floats(N, 10) == N + _lib.RPGP_PIVCHOL_SCRATCH + (10 * N if N >= 65536 else 0)
    small = torch.empty(N + _lib.RPGP_PIVCHOL_SCRATCH - 1, device=gpu_device)
    assert lib.rpgp_ski_pivoted_cholesky(Z.data_ptr(), op.gp.data_ptr(), Lf.data_ptr(), small.data_ptr(), small.numel(), N, J, J,
                                         G, 10, 0.9, None) == _lib.RPGP_EWORKSPACE


@pytest.mark.parametrize("N,J,T,G", [(2500, 5, 3, 256), (3000, 20, 11, 1024)])
def test_weighted_ski_kernels_match_oracle(gpu_device, N, J, T, G):
    """Per-projection output scales in the grid-parameter block: MVM, diagonal, pivoted Cholesky rows and the bilinear
    derivative (incl. the per-component parts) against the float64 dense SKI oracle / its analytic derivative."""
    from rpgp_amd import ops
    from tests.oracle_backend import OracleBackend
    rng = np.random.default_rng(N)
    Z = (rng.standard_normal((N, J)) * 0.9).astype(np.float32)
    V = rng.standard_normal((N, T)).astype(np.float32)
    w = rng.uniform(0.2, 1.5, size=J).astype(np.float32)
    Zt, Vt, wt = (torch.from_numpy(a).to(gpu_device) for a in (Z, V, w))
    gp = ops.ski_grid(Zt, None, G, weights=wt)
    grid = (float(gp[0]), float(gp[1]))
    M = 400
    Kd = sko.dense_kernel(Z[:M], Z, 0.7, G, grid, w)
    got = ops.ski_mvm(Zt[:M].contiguous(), Zt, gp, Vt, 0.7, 0.0, G).cpu().numpy()
    assert np.linalg.norm(got - Kd @ V) / np.linalg.norm(Kd @ V) < 2e-5
    full = ops.ski_mvm(Zt, Zt, gp, Vt, 0.7, 0.1, G).cpu().numpy()
    assert np.linalg.norm(full[:M] - (Kd @ V + 0.1 * V[:M])) / np.linalg.norm(Kd @ V) < 2e-5
    dg = ops.ski_diag(Zt, gp, 0.7, G).cpu().numpy()[:M]
    assert np.abs(dg - np.diag(Kd[:, :M])).max() < 2e-5
    # derivative: the CPU test double restates the analytic formulas in float64
    ob = OracleBackend()
    L = rng.standard_normal((N, T)).astype(np.float32)
    R = rng.standard_normal((N, T)).astype(np.float32)
    sub = slice(0, 1500)                     # dense float64 reference on a subset (its own grid = the same block)
    Zs, Ls, Rs = (torch.from_numpy(a[sub].copy()) for a in (Z, L, R))
    gps = torch.cat([gp[:4].cpu(), torch.from_numpy(w)])
    rZ, rs, rc = ob.ski_bilinear_grad_comp(Zs, gps, Ls, Rs, 0.7, G)
    gZ, gs, gc = ops.ski_bilinear_grad_comp(Zs.to(gpu_device), gp, Ls.to(gpu_device), Rs.to(gpu_device), 0.7, G)
    assert float((gZ.cpu() - rZ).norm() / rZ.norm()) < 2e-4
    assert abs(float(gs) - float(rs)) < 2e-4 * abs(float(rs)) + 1e-3
    assert float((gc.cpu() - rc).norm() / rc.norm()) < 2e-4


@pytest.mark.parametrize("weighted", [False, True])
def test_ski_dense_block_matches_oracle_and_mvm(gpu_device, weighted):
    """rpgp_ski_dense (entries from the interpolation weights and the Toeplitz lags) against the dense float64 SKI oracle
    and against the operator's own MVM."""
    from rpgp_amd import ops
    rng = np.random.default_rng(7)
    N, M, J, G = 1500, 333, 7, 512
    Z = (rng.standard_normal((N, J)) * 0.8).astype(np.float32)
    Z1 = (rng.standard_normal((M, J)) * 0.8).astype(np.float32)
    w = rng.uniform(0.3, 1.4, size=J).astype(np.float32) if weighted else None
    Zt, Z1t = torch.from_numpy(Z).to(gpu_device), torch.from_numpy(Z1).to(gpu_device)
    gp = ops.ski_grid(Zt, Z1t, G, weights=None if w is None else torch.from_numpy(w).to(gpu_device))
    grid = (float(gp[0]), float(gp[1]))
    Kd = ops.ski_dense(Z1t, Zt, gp, 0.6, G)
    ref = sko.dense_kernel(Z1, Z, 0.6, G, grid, w)
    assert np.abs(Kd.cpu().numpy() - ref).max() < 2e-5
    V = torch.randn(N, 5, generator=torch.Generator().manual_seed(0)).to(gpu_device)
    mv = ops.ski_mvm(Z1t, Zt, gp, V, 0.6, 0.0, G)
    assert float((Kd @ V - mv).norm() / mv.norm()) < 1e-5


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_ski_mvm_propagates_non_finite_rhs(gpu_device, bad):
    """A NaN / Inf right-hand-side entry must poison the SKI product like it poisons the exact operator (the fixed-point
    scatter used to drop it: ADVICE r1)."""
    from rpgp_amd import ops
    N, J, T, G = 5000, 3, 4, 1024
    g = torch.Generator().manual_seed(0)
    Z = torch.randn(N, J, generator=g).to(gpu_device)
    V = torch.randn(N, T, generator=g)
    V[1234, 2] = bad
    gp = ops.ski_grid(Z, None, G)
    out = ops.ski_mvm(Z, Z, gp, V.to(gpu_device), 0.3, 0.0, G)
    assert not torch.isfinite(out[:, 2]).any()              # the dense operator mixes it into every row
    assert torch.isfinite(out[:, [0, 1, 3]]).all()


def test_preconditioned_cg_on_badly_scaled_c5_system(gpu_device):
    """Regression (round 2): with |K| / sigma^2 ~ 1e6 the Woodbury capacitance matrix sigma^2 I + L^T L must be
    accumulated in float64 — accumulated in fp32 it was off by more than sigma^2 and preconditioned CG stagnated at a true
    residual of 0.2 - 1.4 on this system (hyper-parameters reached after ten Adam steps on the C5 stand-in), while
    un-preconditioned fp32 CG converges.  Both loops, true float64 residual from the sparse-W oracle."""
    import warnings
    from rpgp_amd import linear_cg as lcg, settings
    from rpgp_amd.operators import SKIAdditiveOperator, AddedDiagOperator
    from rpgp_amd.precond import build_preconditioner
    N, J, G = 391386, 3, 1024
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, J, generator=g)
    y = torch.sin(X).sum(1) + 0.05 * torch.randn(N, generator=g)
    y = (y - y.mean()) / y.std()
    Q = torch.linalg.qr(torch.randn(J, J, generator=g))[0]
    ls = torch.tensor([1.193, 0.495, 1.766])
    s, noise = 0.8657, 0.45
    Z = ((X / ls) @ Q).contiguous()
    Zt = Z.to(gpu_device)
    base = SKIAdditiveOperator(Zt, None, torch.tensor(s, device=gpu_device), 1.0 / J, grid_size=G)
    khat = AddedDiagOperator(base, torch.tensor(noise, device=gpu_device))
    pre = build_preconditioner(base, noise, settings)
    rhs = torch.cat([pre.sample(10, generator=torch.Generator(device=gpu_device).manual_seed(1)),
                     y.to(gpu_device).reshape(-1, 1)], dim=1)
    gph = base.gp.double().cpu().numpy()
    grid = (float(gph[0]), float(gph[1]))
    bn = rhs.double().cpu().numpy()
    for operator in (khat, None):                       # native executor, torch-op loop
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            x = lcg.linear_cg(khat._matmul, rhs, tolerance=0.01, max_iter=2000, preconditioner=pre, operator=operator)
        assert not [m for m in w if "CG terminated" in str(m.message)]
        assert lcg.stats["last_iterations"] < 150, lcg.stats["last_iterations"]
        r = sko.mvm_sparse(Z.numpy(), Z.numpy(), x.double().cpu().numpy(), s / J, G, grid, noise) - bn
        res = np.linalg.norm(r, axis=0) / np.linalg.norm(bn, axis=0)
        assert res.mean() < 0.015, res


# ---- the planned (cell-sorted) SKI product: rpgp_ski_plan + rpgp_ski_mvm_planned ----------------------------------------
@pytest.mark.parametrize("N,J,T,G,spread", [(500, 3, 1, 128, 1.0), (1000, 3, 11, 1024, 1.0), (777, 8, 4, 256, 1.0),
                                            (3000, 20, 12, 1024, 0.7), (64, 1, 1, 64, 1.0), (9, 2, 5, 16, 1.0),
                                            (40000, 3, 11, 64, 1.0), (5000, 3, 7, 1024, 1e-3), (1, 3, 2, 32, 1.0)])
def test_planned_ski_mvm_matches_dense_ski_oracle(gpu_device, N, J, T, G, spread):
    """Same oracle as the unplanned product, at shapes that exercise every branch of the plan: items of < 64 and of many
    x 64 points (G = 64 with 40 000 points: ~ 650 points per cell), empty cells (most cells at spread 1e-3), all three
    column pieces (1 / 4 / 12 lanes per point), a single point."""
    from rpgp_amd import ops
    rng = np.random.default_rng(N + G + T)
    Z = (rng.standard_normal((N, J)) * spread).astype(np.float32)
    V = rng.standard_normal((N, T)).astype(np.float32)
    Zt, Vt = torch.from_numpy(Z).to(gpu_device), torch.from_numpy(V).to(gpu_device)
    gp = ops.ski_grid(Zt, None, G)
    gph = gp.double().cpu().numpy()
    grid = (float(gph[0]), float(gph[1]))
    ref = sko.mvm_sparse(Z, Z, V, 0.4, G, grid, 0.2)
    plan = ops.SkiPlan(Zt, gp, G)
    assert plan.ok
    out = ops.ski_mvm(Zt, Zt, gp, Vt, 0.4, 0.2, G, plan=plan)
    assert _rel(out.cpu().numpy(), ref) < 2e-5
    plain = ops.ski_mvm(Zt, Zt, gp, Vt, 0.4, 0.2, G)
    assert _rel(out.cpu().numpy(), plain.cpu().numpy()) < 2e-6
    # bitwise reproducible: same plan, and a plan rebuilt from scratch (stable sort: same order inside every cell)
    assert torch.equal(out, ops.ski_mvm(Zt, Zt, gp, Vt, 0.4, 0.2, G, plan=plan))
    assert torch.equal(out, ops.ski_mvm(Zt, Zt, gp, Vt, 0.4, 0.2, G, plan=ops.SkiPlan(Zt, gp, G)))
    # the staged form of the row-sharded operator: planned scatter == plain scatter
    if T <= 12:
        h1 = ops.ski_scatter(Zt, gp, Vt, G, plan=plan)
        h0 = ops.ski_scatter(Zt, gp, Vt, G)
        assert float((h1 - h0).abs().max()) < 2e-5 * float(h0.abs().max() + 1e-30)


@pytest.mark.parametrize("T", [1, 11])
def test_planned_ski_mvm_at_full_c5_size(gpu_device, T):
    """The planned product at exactly the C5 shape (N = 391 386, J = 3, G = 1024) against the float64 sparse-W oracle, with
    per-projection weights as well (they ride on the Toeplitz stage), NaN propagation, and bitwise reproducibility."""
    from rpgp_amd import ops
    N, J, G, Z, V = _c5_problem(gpu_device, T)
    Zt, Vt = Z.to(gpu_device), V.to(gpu_device)
    scale, noise = 0.8 / J, 0.25
    for weights in (None, torch.tensor([0.5, 1.25, 2.0])):
        gp = ops.ski_grid(Zt, None, G, weights=weights)
        gph = gp.double().cpu().numpy()
        grid = (float(gph[0]), float(gph[1]))
        plan = ops.SkiPlan(Zt, gp, G)
        out = ops.ski_mvm(Zt, Zt, gp, Vt, scale, noise, G, plan=plan)
        ref = sko.mvm_sparse(Z.numpy(), Z.numpy(), V.numpy(), scale, G, grid, noise,
                             weights=None if weights is None else weights.numpy())
        col = np.linalg.norm(out.double().cpu().numpy() - ref, axis=0) / np.linalg.norm(ref, axis=0)
        assert col.max() < 2e-5, col
        assert torch.equal(out, ops.ski_mvm(Zt, Zt, gp, Vt, scale, noise, G, plan=plan))
    Vb = Vt.clone()
    Vb[12345, 0] = float("nan")
    bad = ops.ski_mvm(Zt, Zt, gp, Vb, scale, noise, G, plan=plan)
    assert not torch.isfinite(bad[:, 0]).all()
    if T > 1:
        assert torch.isfinite(bad[:, 1:]).all()


# ---- the reference's per-projection grid rule (polynomial_projection_kernels.py:54-63) -------------------------------------
@pytest.mark.parametrize("N,J,T,G,weighted", [(1200, 3, 11, 256, False), (900, 8, 4, 128, True), (40000, 3, 11, 1024, True),
                                               (700, 20, 1, 64, False)])
def test_reference_grid_rule_every_entry_point(gpu_device, N, J, T, G, weighted):
    """rule="reference": every projection has its own bounds (min - 2.01 sp, max + 2.01 sp), sp = (max - min) / (G - 4).
    Projections are given very different ranges on purpose (with the shared grid they would share one spacing).  Grid block,
    MVM (plain and planned), diagonal, dense block, pivoted Cholesky and the derivative against the float64 oracle."""
    from rpgp_amd import ops
    rng = np.random.default_rng(N + J)
    spread = np.linspace(0.3, 3.0, J)
    Z = (rng.standard_normal((N, J)) * spread + np.linspace(-2, 2, J)).astype(np.float32)
    V = rng.standard_normal((N, T)).astype(np.float32)
    wts = (0.5 + rng.random(J)).astype(np.float32) if weighted else None
    Zt, Vt = torch.from_numpy(Z).to(gpu_device), torch.from_numpy(V).to(gpu_device)
    gp = ops.ski_grid(Zt, None, G, weights=None if wts is None else torch.from_numpy(wts), rule="reference")
    assert gp.numel() == 4 + 4 * J and int(gp[3]) == (3 if weighted else 2)
    g0r, hr = sko.grid_params_reference(Z, None, G)
    blk = gp.double().cpu().numpy()[4 + J:].reshape(J, 3)
    np.testing.assert_allclose(blk[:, 0], g0r, rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(blk[:, 1], hr, rtol=2e-6)
    grid = (blk[:, 0].copy(), blk[:, 1].copy())          # the oracle runs on the kernel's own (float32) grid block
    # every point keeps 2.01 (G - 1) / (G + 0.02) ~ 2 cells of margin (> 1 is what an interior 4-tap stencil needs)
    u = (Z.astype(np.float64) - grid[0]) / grid[1]
    assert u.min() >= 1.9 and u.max() <= G - 2.9
    scale, noise = 0.6 / J, 0.15
    ref = sko.mvm_sparse(Z, Z, V, scale, G, grid, noise, weights=wts)
    out = ops.ski_mvm(Zt, Zt, gp, Vt, scale, noise, G)
    assert _rel(out.cpu().numpy(), ref) < 2e-5
    plan = ops.SkiPlan(Zt, gp, G)
    outp = ops.ski_mvm(Zt, Zt, gp, Vt, scale, noise, G, plan=plan)
    assert _rel(outp.cpu().numpy(), ref) < 2e-5
    np.testing.assert_allclose(ops.ski_diag(Zt, gp, scale, G).cpu().numpy(), sko.diag_sparse(Z, scale, G, grid, weights=wts),
                               rtol=3e-5, atol=1e-6)
    if N <= 1500:
        K = sko.dense_kernel(Z, Z, scale, G, grid, wts)
        assert np.abs(ops.ski_dense(Zt, Zt, gp, scale, G).cpu().numpy() - K).max() < 3e-5 * np.abs(K).max()
        Lc = ops.ski_pivoted_cholesky(Zt, gp, scale, 8, G).double().cpu().numpy()
        Lref, _ = orc.pivoted_cholesky(K, 8)
        # (the SKI diagonal is constant up to interpolation ripple, so fp32 and fp64 greedy pivots may differ: compare the
        #  factorisations by what they are for — the residual they leave — and check L L^T <= K)
        res, res_ref = np.trace(K - Lc @ Lc.T), np.trace(K - Lref @ Lref.T)
        assert res < 1.1 * res_ref + 1e-3 * np.trace(K), (res, res_ref)
        assert np.linalg.eigvalsh(K - Lc @ Lc.T).min() > -1e-4 * K.max()
        Lm = (rng.standard_normal((N, T)) * 0.1).astype(np.float32)
        Rm = (rng.standard_normal((N, T)) * 0.1).astype(np.float32)
        gZ, gs = ops.ski_bilinear_grad(Zt, gp, torch.from_numpy(Lm).to(gpu_device), torch.from_numpy(Rm).to(gpu_device),
                                       scale, G)
        eps = 1e-4
        for (i, j) in [(3, 0), (N // 2, J - 1)]:
            Zp, Zm = Z.astype(np.float64).copy(), Z.astype(np.float64).copy()
            Zp[i, j] += eps
            Zm[i, j] -= eps
            fd = (sko.bilinear_objective(Zp, Lm, Rm, scale, G, grid, wts) -
                  sko.bilinear_objective(Zm, Lm, Rm, scale, G, grid, wts)) / (2 * eps)
            assert abs(float(gZ[i, j]) - fd) < 2e-3 * abs(fd) + 1e-5
        obj = sko.bilinear_objective(Z, Lm, Rm, scale, G, grid, wts)
        assert abs(float(gs) - obj / scale) < 1e-3 * abs(obj / scale) + 1e-4


def test_reference_grid_rule_is_the_default_of_the_rp_poly_ski_kinds(gpu_device):
    """`rp_poly` / `strictly_additive` with `ski: true` build per-projection grids (flags & 2); the additive_rp SKI kernel keeps
    the shared grid; both train a step and agree with their exact (non-SKI) operators to interpolation accuracy."""
    from rpgp_amd import training
    g = torch.Generator().manual_seed(0)
    X = torch.randn(1500, 4, generator=g).to(gpu_device)
    y = torch.sin(X).sum(1)
    for kind, mk, flag in (("rp_poly", dict(k=1, J=6, weighted=True), 2), ("strictly_additive", dict(weighted=False), 2),
                           ("additive_rp", dict(J=6, prescale=True, learn_proj=False), 0)):
        torch.manual_seed(3)
        model, lik = training.create_exact_gp(X, y, kind, noise_prior=True, kernel_type="RBF", ski=True,
                                              ski_options={"grid_size": 512, "num_dims": 1}, **mk)
        model = model.to(gpu_device)
        op = model.covar_module(X)
        assert int(op.gp[3]) & 2 == flag, kind
        torch.manual_seed(3)
        exact, _ = training.create_exact_gp(X, y, kind, noise_prior=True, kernel_type="RBF", ski=False, **mk)
        exact = exact.to(gpu_device)
        v = torch.randn(1500, 3, generator=g).to(gpu_device)
        a, b = op._matmul(v), exact.covar_module(X)._matmul(v)
        assert float((a - b).norm() / b.norm()) < 2e-4, kind


@pytest.mark.parametrize("N,J,T,G,rule,weighted", [(5000, 3, 11, 1024, "shared", False), (3001, 20, 23, 256, "reference", True),
                                                    (777, 5, 1, 64, "reference", False)])
def test_planned_bilinear_derivative_matches_the_fused_one(gpu_device, N, J, T, G, rule, weighted):
    """SKIAdditiveOperator._bilinear_derivative on its plan (cell-sorted scatters of L and R, 2T-column Toeplitz product on
    the matrix cores, per-row finish) against the fused rpgp_ski_bilinear_grad[_comp] entry point (itself oracle-checked
    above), including more than 12 columns and both grid rules; the staged histogram equals the unplanned one."""
    from rpgp_amd import ops
    from rpgp_amd.operators import SKIAdditiveOperator
    rng = np.random.default_rng(N + T)
    Z = torch.from_numpy(rng.normal(size=(N, J)).astype(np.float32)).to(gpu_device)
    L = torch.from_numpy(rng.normal(size=(N, T)).astype(np.float32)).to(gpu_device)
    R = torch.from_numpy(rng.normal(size=(N, T)).astype(np.float32)).to(gpu_device)
    w = torch.from_numpy(rng.uniform(0.4, 1.3, size=J).astype(np.float32)).to(gpu_device) if weighted else None
    s = torch.tensor(0.7, device=gpu_device)
    op = SKIAdditiveOperator(Z, None, s, 1.0, grid_size=G, comp_weights=w, grid_rule=rule)
    plan = op._get_plan()
    assert plan is not None
    got = op._bilinear_derivative(L, R)
    if weighted:
        gZ, gs, gc = ops.ski_bilinear_grad_comp(Z, op.gp, L, R, 0.7, G)
        want = (gZ, gs, 0.7 * gc / w)
    else:
        want = ops.ski_bilinear_grad(Z, op.gp, L, R, 0.7, G)
    assert len(got) == len(want)
    # (the unplanned scatter accumulates in fixed point in LDS, the planned one sums exactly: the two agree to ~3e-5)
    for a, b in zip(got, want):
        a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
        assert np.linalg.norm(a - b) <= 1e-4 * max(np.linalg.norm(b), 1e-30)
    if N < 1000:            # float64 oracle: d/dscale = objective / scale, d/dZ by central differences on a fixed grid
        Zh, Lh, Rh = Z.cpu().numpy(), L.cpu().numpy(), R.cpu().numpy()
        gph = op.gp.cpu().numpy().astype(np.float64)
        grid = (gph[4 + J:4 + J + 3 * J:3], gph[5 + J:4 + J + 3 * J:3]) if rule == "reference" else (float(gph[0]), float(gph[1]))
        obj = sko.bilinear_objective(Zh, Lh, Rh, 0.7, G, grid)
        assert abs(float(got[1]) - obj / 0.7) < 2e-4 * abs(obj / 0.7) + 2e-4 * (np.abs(Lh).sum() * np.abs(Rh).sum() / N) ** 0.5
        gz = got[0].cpu().numpy()
        for (i, j) in [(0, 0), (N // 2, J - 1), (N - 1, 1)]:
            Zp, Zm = Zh.astype(np.float64).copy(), Zh.astype(np.float64).copy()
            Zp[i, j] += 1e-4
            Zm[i, j] -= 1e-4
            fd = (sko.bilinear_objective(Zp, Lh, Rh, 0.7, G, grid) - sko.bilinear_objective(Zm, Lh, Rh, 0.7, G, grid)) / 2e-4
            assert abs(gz[i, j] - fd) < 2e-3 * np.abs(gz).max() + 2e-3 * abs(fd)
    Lc, Rc = L[:, :min(T, 12)].contiguous(), R[:, :min(T, 12)].contiguous()
    h_plan = ops.ski_bilinear_scatter(Z, op.gp, Lc, Rc, G, plan=plan).cpu().numpy()
    h_ref = ops.ski_bilinear_scatter(Z, op.gp, Lc, Rc, G).cpu().numpy()
    assert np.linalg.norm(h_plan - h_ref) <= 1e-6 * np.linalg.norm(h_ref)


@pytest.mark.parametrize("N,J,T,G", [(5000, 3, 11, 1024), (40000, 3, 1, 64), (3000, 20, 12, 1024), (2000, 3, 4, 256),
                                     (1500, 2, 1, 128), (391386, 3, 1, 1024), (391386, 3, 11, 1024)])
def test_planned_product_is_bitwise_its_three_stages(gpu_device, N, J, T, G):
    """rpgp_ski_mvm_planned against the staged entry points the row-sharded operator uses — planned scatter,
    rpgp_ski_grid_product, rpgp_ski_gather_fast: the same kernels in the same order, so the two agree bit for bit (up to the
    full C5 size)."""
    from rpgp_amd import ops
    g = torch.Generator().manual_seed(N + T)
    Zt = torch.randn(N, J, generator=g).to(gpu_device)
    Vt = torch.randn(N, T, generator=g).to(gpu_device)
    gp = ops.ski_grid(Zt, None, G, weights=torch.linspace(0.5, 1.5, J))
    plan = ops.SkiPlan(Zt, gp, G)
    out = ops.ski_mvm(Zt, Zt, gp, Vt, 0.7 / J, 0.2, G, plan=plan)
    hist = ops.ski_scatter(Zt, gp, Vt, G, plan=plan)
    staged = ops.ski_gather(Zt, gp, ops.ski_grid_product(hist, gp, G), Vt, 0.7 / J, 0.2, G, plan=plan)
    assert torch.equal(out, staged)


@pytest.mark.parametrize("N,J,T,G,dist", [(5000, 3, 11, 1024, "gaussian"), (60000, 3, 11, 1024, "skewed"), (60000, 3, 1, 1024, "skewed"),
                                          (30000, 20, 12, 256, "gaussian"), (20000, 2, 4, 129, "uniform"), (391386, 3, 11, 1024, "gaussian")])
def test_centre_out_cell_dispatch_covers_every_cell(gpu_device, N, J, T, G, dist):
    """The cell-sorted scatter's workgroups take the cells centre-out (round 5: the long cells of centre-peaked coordinates start
    first): every cell exactly once for even and odd grid sizes, one- and many-round cells, bell-shaped / skewed / uniform
    coordinates — the histogram against the unplanned scatter (per-chunk slabs: another algorithm, same sums to rounding) and the
    product reproducible bit for bit."""
    from rpgp_amd import ops
    g = torch.Generator().manual_seed(N + T)
    Z = torch.randn(N, J, generator=g)
    if dist == "skewed":
        Z = Z.exp()
    elif dist == "uniform":
        Z = torch.rand(N, J, generator=g) * 4 - 2
    Zt = Z.to(gpu_device)
    Vt = torch.randn(N, T, generator=g).to(gpu_device)
    gp = ops.ski_grid(Zt, None, G)
    plan = ops.SkiPlan(Zt, gp, G)
    h_plan = ops.ski_scatter(Zt, gp, Vt, G, plan=plan)
    h_ref = ops.ski_scatter(Zt, gp, Vt, G)
    assert torch.isfinite(h_plan).all()
    assert float((h_plan - h_ref).norm() / h_ref.norm()) < 1e-6
    out = ops.ski_mvm(Zt, Zt, gp, Vt, 0.7 / J, 0.2, G, plan=plan)
    assert torch.isfinite(out).all() and torch.equal(out, ops.ski_mvm(Zt, Zt, gp, Vt, 0.7 / J, 0.2, G, plan=plan))


@pytest.mark.parametrize("kind", ["Matern", "InverseMQ", "Cosine"])
@pytest.mark.parametrize("N,J,T,G,rule,weighted", [(1500, 3, 11, 256, "shared", False), (40000, 3, 11, 1024, "shared", False),
                                                   (2000, 8, 23, 128, "reference", True), (900, 2, 1, 1024, "shared", True)])
def test_ski_around_the_other_sub_kernels_every_entry_point(gpu_device, kind, N, J, T, G, rule, weighted):
    """Round 6 (VERDICT r5 missing #4): `GridInterpolationKernel` wraps whatever `_map_to_kernel` returned
    (training_routines.py:157-158 with :47-88) — Matern-1.5, InverseMQ (imq_kernel.py:8-9) and Cosine 1-D sub-kernels under the
    grid: the sub-kernel rides in the grid block's flags and only the grid-to-grid Toeplitz entries depend on it.  Every entry
    point — plain and planned product (narrow and wide blocks), staged scatter / grid product / gather, diagonal, dense block,
    pivoted Cholesky rows, derivative — against the float64 oracle with the same radial form (oracle/ski.py + oracle/family.py),
    in float32 and through the float64 parity kernels."""
    from rpgp_amd import ops
    from rpgp_amd.operators import SKIAdditiveOperator
    rng = np.random.default_rng(N + G)
    Z = (rng.standard_normal((N, J)) * 0.9).astype(np.float32)
    V = rng.standard_normal((N, T)).astype(np.float32)
    w = rng.uniform(0.4, 1.6, size=J).astype(np.float32) if weighted else None
    Zt, Vt = torch.from_numpy(Z).to(gpu_device), torch.from_numpy(V).to(gpu_device)
    wt = None if w is None else torch.from_numpy(w).to(gpu_device)
    gp = ops.ski_grid(Zt, None, G, weights=wt, rule=rule, kind=kind)
    assert (int(gp[3]) >> 2) == {"Matern": 1, "InverseMQ": 2, "Cosine": 3}[kind]
    if rule == "shared":
        gph = gp.double().cpu().numpy()
        grid = (float(gph[0]), float(gph[1]))
    else:
        arr = gp.double().cpu().numpy()[4 + J:].reshape(J, 3)
        grid = (arr[:, 0].copy(), arr[:, 1].copy())
    K = sko.dense_kernel(Z, Z, 0.4, G, grid, w, kind) if N <= 4000 else None      # (the big case: O(N) oracle forms only)
    V64 = V.astype(np.float64)
    ref = K @ V64 + 0.2 * V64 if N <= 4000 else sko.mvm_sparse(Z, Z, V, 0.4, G, grid, 0.2, w, kind)
    tol = 3e-5
    out = ops.ski_mvm(Zt, Zt, gp, Vt, 0.4, 0.2, G)
    assert _rel(out.cpu().numpy(), ref) < tol
    plan = ops.SkiPlan(Zt, gp, G)
    outp = ops.ski_mvm(Zt, Zt, gp, Vt, 0.4, 0.2, G, plan=plan)
    assert _rel(outp.cpu().numpy(), ref) < tol
    Tc = min(T, 12)
    hist = ops.ski_scatter(Zt, gp, Vt[:, :Tc].contiguous(), G, plan=plan)
    staged = ops.ski_gather(Zt, gp, ops.ski_grid_product(hist, gp, G), Vt[:, :Tc].contiguous(), 0.4, 0.2, G, plan=plan)
    assert _rel(staged.cpu().numpy(), ref[:, :Tc]) < tol
    diag_ref = np.diag(K) if N <= 4000 else sko.diag_sparse(Z, 0.4, G, grid, w, kind)
    np.testing.assert_allclose(ops.ski_diag(Zt, gp, 0.4, G).cpu().numpy(), diag_ref, rtol=3e-5, atol=2e-6)
    rows = torch.from_numpy(Z[:7]).to(gpu_device)
    blk = ops.ski_dense(rows, Zt, gp, 0.4, G).cpu().numpy()
    np.testing.assert_allclose(blk, sko.dense_kernel(Z[:7], Z, 0.4, G, grid, w, kind), rtol=3e-5, atol=2e-6)
    if kind != "Cosine" and J <= 64:
        # greedy factor of the (positive definite) operator: the library's per-step kernel evaluates the SKI rows itself
        from rpgp_amd.precond import pivoted_cholesky
        op = SKIAdditiveOperator(Zt, None, torch.tensor(0.4, device=gpu_device), 1.0, grid_size=G, comp_weights=wt,
                                 grid_rule=rule, kind=kind)
        Lf = op.fused_pivoted_cholesky(8)
        Lg = pivoted_cholesky(op._diagonal(), op._get_rows, 8)
        assert Lf is not None and torch.allclose(Lf, Lg, rtol=2e-3, atol=3e-4)
    if N <= 4000:
        # derivative of sum((L R^T) * K) with the grid held fixed, against float64 finite differences of the oracle
        Tb = min(T, 5)
        Lm = rng.standard_normal((N, Tb)).astype(np.float32) * 0.1
        Rm = rng.standard_normal((N, Tb)).astype(np.float32) * 0.1
        gz, gs = ops.ski_bilinear_grad(Zt, gp, torch.from_numpy(Lm).to(gpu_device), torch.from_numpy(Rm).to(gpu_device), 0.4, G)[:2]
        gz = gz.cpu().numpy()
        Lh, Rh = Lm.astype(np.float64), Rm.astype(np.float64)
        for (i, j) in ((3, 0), (N // 2, J - 1)):
            Zp, Zm = Z.astype(np.float64).copy(), Z.astype(np.float64).copy()
            Zp[i, j] += 1e-4
            Zm[i, j] -= 1e-4
            fd = (sko.bilinear_objective(Zp, Lh, Rh, 0.4, G, grid, w, kind) - sko.bilinear_objective(Zm, Lh, Rh, 0.4, G, grid, w, kind)) / 2e-4
            assert abs(gz[i, j] - fd) < 3e-3 * np.abs(gz).max() + 3e-3 * abs(fd)
        obj1 = sko.bilinear_objective(Z, Lh, Rh, 1.0, G, grid, w, kind)            # d / d scale is linear: objective / scale
        assert abs(float(gs) - obj1) < 2e-4 * (np.abs(Lh).sum() * np.abs(Rh).sum() / N) ** 0.5 + 2e-4 * abs(obj1)
        # the float64 parity kernels (`--double`) on the same problem
        Zd, Vd = Zt.double(), Vt.double()
        gpd = ops.ski_grid(Zd, None, G, weights=None if wt is None else wt.double(), rule=rule, kind=kind)
        outd = ops.ski_mvm(Zd, Zd, gpd, Vd, 0.4, 0.2, G)
        if rule == "shared":
            gd = gpd.cpu().numpy()
            grid_d = (float(gd[0]), float(gd[1]))
        else:
            ad = gpd.cpu().numpy()[4 + J:].reshape(J, 3)
            grid_d = (ad[:, 0].copy(), ad[:, 1].copy())
        refd = sko.dense_kernel(Z, Z, 0.4, G, grid_d, w, kind) @ V64 + 0.2 * V64
        assert _rel(outd.cpu().numpy(), refd) < 1e-11
        histd = ops.ski_scatter(Zd, gpd, Vd[:, :Tc].contiguous(), G)
        stagedd = ops.ski_gather(Zd, gpd, ops.ski_grid_product(histd, gpd, G), Vd[:, :Tc].contiguous(), 0.4, 0.2, G)
        assert _rel(stagedd.cpu().numpy(), refd[:, :Tc]) < 1e-11


@pytest.mark.parametrize("kernel_type", ["Matern", "InverseMQ"])
def test_ski_model_with_another_sub_kernel_tracks_its_exact_counterpart(gpu_device, kernel_type):
    """`additive_rp` with `kernel_type` Matern / InverseMQ and `ski: true`: the MLL of the interpolated model against the exact
    family operator of the same hyper-parameters (interpolation error only; the Matern kink converges with h^2)."""
    from rpgp_amd import settings
    from rpgp_amd.training import create_exact_gp
    from rpgp_amd.models import ExactMarginalLogLikelihood
    g = torch.Generator().manual_seed(3)
    X = torch.randn(3000, 5, generator=g)
    y = torch.sin(X).sum(1) + 0.1 * torch.randn(3000, generator=g)
    X, y = X.to(gpu_device), ((y - y.mean()) / y.std()).to(gpu_device)
    vals = {}
    for ski in (False, True):
        torch.manual_seed(1)
        np.random.seed(1)
        model, lik = create_exact_gp(X, y, "additive_rp", J=5, noise_prior=True, kernel_type=kernel_type, learn_proj=False,
                                     prescale=True, ski=ski, ski_options={"grid_size": 1024, "num_dims": 1} if ski else None)
        model = model.to(gpu_device)
        mll = ExactMarginalLogLikelihood(lik, model)
        model.train()
        with settings.max_cholesky_size(10000), torch.no_grad():
            vals[ski] = mll(model(X), y).item()
    assert abs(vals[True] - vals[False]) < (3e-3 if kernel_type == "Matern" else 3e-4) * abs(vals[False]), vals
