"""GPU parity of the raw HIP kernels (through the C-ABI) against the float64 oracle."""
import math

import numpy as np
import pytest
import torch

from oracle import dense_gp as orc
from tests.oracle_fast import omvm

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _data(N, J, T, seed=0, spread=1.0):
    rng = np.random.default_rng(seed)
    Z = (rng.standard_normal((N, J)) * spread).astype(np.float32)
    V = rng.standard_normal((N, T)).astype(np.float32)
    return Z, V


@pytest.mark.parametrize("N,J,T", [(1, 20, 1), (63, 20, 1), (257, 20, 1), (300, 3, 1), (777, 18, 4), (1000, 20, 11),
                                   (2049, 8, 1), (4096, 20, 1), (1500, 7, 13)])
def test_mvm_sym_matches_oracle(gpu_device, N, J, T):
    from rpgp_amd import ops
    Z, V = _data(N, J, T, seed=N + J)
    scale, noise = 0.7 / J, 0.1
    ref = omvm(Z, Z, V, scale, noise)
    out = ops.mvm_sym(torch.from_numpy(Z).to(gpu_device), torch.from_numpy(V).to(gpu_device), scale, noise)
    # tolerance: fp32 accumulation of N terms + v_exp_f32 (1 ulp) -> 1e-5 relative in the 2-norm
    assert _rel(out.cpu().numpy(), ref) < 1e-5


def test_mvm_sym_large_two_rows_per_lane(gpu_device):
    from rpgp_amd import ops
    N, J, T = 16500, 20, 1
    Z, V = _data(N, J, T, seed=5)
    ref = omvm(Z, Z, V, 1.0 / J, 0.05)
    out = ops.mvm_sym(torch.from_numpy(Z).to(gpu_device), torch.from_numpy(V).to(gpu_device), 1.0 / J, 0.05)
    assert _rel(out.cpu().numpy(), ref) < 1e-5


def test_mvm_sym_vector_rhs_and_jrange(gpu_device):
    from rpgp_amd import ops
    N, J = 900, 20
    Z, V = _data(N, J, 1, seed=3)
    Zt = torch.from_numpy(Z).to(gpu_device)
    v = torch.from_numpy(V[:, 0].copy()).to(gpu_device)
    full = ops.mvm_sym(Zt, v, 0.05, 0.0)
    parts = sum(ops.mvm_sym(Zt, v, 0.05, 0.0, j0=a, j1=b) for a, b in [(0, 3), (3, 6), (6, 13), (13, 20)])
    assert full.shape == (N,)
    assert _rel(parts.cpu().numpy(), full.cpu().numpy()) < 1e-6
    ref = omvm(Z[:, 3:6], Z[:, 3:6], V, 0.05)
    got = ops.mvm_sym(Zt, v, 0.05, 0.0, j0=3, j1=6)
    assert _rel(got.cpu().numpy(), ref[:, 0]) < 1e-5


@pytest.mark.parametrize("M,N,J,T", [(31, 277, 20, 1), (820, 3000, 20, 1), (100, 1000, 3, 5), (513, 700, 20, 12),
                                     (1, 50, 2, 1)])
def test_mvm_rect_matches_oracle(gpu_device, M, N, J, T):
    from rpgp_amd import ops
    rng = np.random.default_rng(M)
    Z1 = rng.standard_normal((M, J)).astype(np.float32)
    Z2 = rng.standard_normal((N, J)).astype(np.float32)
    V = rng.standard_normal((N, T)).astype(np.float32)
    ref = omvm(Z1, Z2, V, 0.3)
    out = ops.mvm_rect(torch.from_numpy(Z1).to(gpu_device), torch.from_numpy(Z2).to(gpu_device),
                       torch.from_numpy(V).to(gpu_device), 0.3)
    assert _rel(out.cpu().numpy(), ref) < 1e-5


@pytest.mark.parametrize("M,N,J", [(15, 1000, 20), (277, 277, 6), (3, 2, 3), (700, 300, 18)])
def test_dense_matches_oracle(gpu_device, M, N, J):
    from rpgp_amd import ops
    rng = np.random.default_rng(J)
    Z1 = rng.standard_normal((M, J)).astype(np.float32)
    Z2 = rng.standard_normal((N, J)).astype(np.float32)
    ref = 0.5 * orc.additive_rbf(Z1, Z2)
    out = ops.dense(torch.from_numpy(Z1).to(gpu_device), torch.from_numpy(Z2).to(gpu_device), 0.5)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-6, atol=2e-6)


def test_known_answer_prescale(gpu_device):
    """test.py:533-573: x=[[1,2,3],[1.1,2.2,3.3]], P=I, l=[1,2,3] -> K = 3*RBF(x[:,0])."""
    from rpgp_amd import ops
    x = torch.tensor([[1., 2., 3.], [1.1, 2.2, 3.3]], device=gpu_device)
    ls = torch.tensor([1., 2., 3.], device=gpu_device)
    P = torch.eye(3, device=gpu_device)
    Z = ops.project(x, P / ls[:, None])
    K = ops.dense(Z, Z, 1.0).cpu().numpy()
    e = 3.0 * np.exp(-0.5 * 0.1 ** 2)
    np.testing.assert_allclose(K, np.array([[3.0, e], [e, 3.0]]), rtol=1e-6)


@pytest.mark.parametrize("N,d,J", [(1000, 8, 20), (37, 6, 6), (5000, 20, 20), (300, 3, 3)])
def test_project_and_grad(gpu_device, N, d, J):
    from rpgp_amd import ops
    rng = np.random.default_rng(N)
    X = rng.standard_normal((N, d)).astype(np.float32)
    P = rng.standard_normal((d, J)).astype(np.float32)
    G = rng.standard_normal((N, J)).astype(np.float32)
    Z = ops.project(torch.from_numpy(X).to(gpu_device), torch.from_numpy(P).to(gpu_device))
    assert _rel(Z.cpu().numpy(), X.astype(np.float64) @ P.astype(np.float64)) < 1e-6
    dP = ops.project_grad(torch.from_numpy(X).to(gpu_device), torch.from_numpy(G).to(gpu_device))
    assert _rel(dP.cpu().numpy(), X.astype(np.float64).T @ G.astype(np.float64)) < 1e-5


@pytest.mark.parametrize("N,J,T", [(300, 20, 11), (1000, 3, 1), (2500, 18, 11), (64, 20, 4), (500, 20, 20)])
def test_bilinear_grad_matches_oracle(gpu_device, N, J, T):
    from rpgp_amd import ops
    rng = np.random.default_rng(N + T)
    Z = rng.standard_normal((N, J)).astype(np.float32)
    L = rng.standard_normal((N, T)).astype(np.float32)
    R = rng.standard_normal((N, T)).astype(np.float32)
    gZ_ref, gs_ref = orc.bilinear_grad(Z, L, R, 0.2)
    gZ, gs = ops.bilinear_grad(torch.from_numpy(Z).to(gpu_device), torch.from_numpy(L).to(gpu_device),
                               torch.from_numpy(R).to(gpu_device), 0.2)
    assert _rel(gZ.cpu().numpy(), gZ_ref) < 2e-5
    assert abs(gs.item() - gs_ref) <= 2e-5 * max(abs(gs_ref), np.abs(L).sum() * 1e-3)


@pytest.mark.parametrize("N,T", [(1000, 1), (2049, 11), (300, 3)])
def test_dense_mvm(gpu_device, N, T):
    from rpgp_amd import ops
    rng = np.random.default_rng(N)
    Z = rng.standard_normal((N, 5)).astype(np.float32)
    V = rng.standard_normal((N, T)).astype(np.float32)
    Zt = torch.from_numpy(Z).to(gpu_device)
    Kd = ops.dense(Zt, Zt, 0.2)
    out = ops.dense_mvm(Kd, torch.from_numpy(V).to(gpu_device), 0.3)
    ref = omvm(Z, Z, V, 0.2, 0.3)
    assert _rel(out.cpu().numpy(), ref) < 1e-5


@pytest.mark.parametrize("N,T", [(1027, 1), (2050, 11), (3001, 4), (1285, 12), (1285, 16), (2050, 37)])
def test_dense_mvm_padded_rows_ragged_edge(gpu_device, N, T):
    """Cached-K layout (rows padded to a multiple of 64 floats) with N % 4 != 0: the 16-byte loads of the last column
    group read into the row padding; whatever the padding holds (NaN here) must never reach the result."""
    from rpgp_amd import ops
    rng = np.random.default_rng(N + T)
    Z = rng.standard_normal((N, 5)).astype(np.float32)
    V = rng.standard_normal((N, T)).astype(np.float32)
    Zt = torch.from_numpy(Z).to(gpu_device)
    ld = (N + 63) // 64 * 64
    base = torch.full((N, ld), float("nan"), device=gpu_device)
    base[:, :N] = ops.dense(Zt, Zt, 0.2)
    Kd = base[:, :N]
    out = ops.dense_mvm(Kd, torch.from_numpy(V).to(gpu_device), 0.3)
    ref = omvm(Z, Z, V, 0.2, 0.3)
    assert torch.isfinite(out).all()
    assert _rel(out.cpu().numpy(), ref) < 1e-5
    Kp = ops.dense(Zt, Zt, 0.2, pad=True)
    assert Kp.stride(0) % 64 == 0 and Kp.shape == (N, N)
    out2 = ops.dense_mvm(Kp, torch.from_numpy(V).to(gpu_device), 0.3)
    assert _rel(out2.cpu().numpy(), ref) < 1e-5


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("N,J,T", [(300, 5, 1), (1000, 20, 11), (4097, 20, 12), (5000, 23, 3), (2049, 3, 25),
                                   (6211, 8, 5), (4500, 20, 16), (777, 2, 37), (1, 3, 2), (17, 4, 11), (64, 1, 1),
                                   (65, 20, 4), (257, 5, 11), (513, 5, 1)])
def test_symcache_product_matches_oracle(gpu_device, N, J, T, wide):
    """Packed symmetric cache (rpgp_symcache_build / _mvm): every unordered pair stored once, the product equals the
    float64 oracle's K V (one- and two-row-per-lane plans, ragged edges, J split into compiled pieces, T into passes),
    in both layouts: rotation order (VALU sweep) and 16 x 16 matrix-core tiles (exact-fp32 MFMA, in-register transpose)."""
    from rpgp_amd import ops
    rng = np.random.default_rng(N + J + T)
    Z = rng.standard_normal((N, J)).astype(np.float32)
    V = rng.standard_normal((N, T)).astype(np.float32)
    Zt, Vt = torch.from_numpy(Z).to(gpu_device), torch.from_numpy(V).to(gpu_device)
    cache = ops.SymCache(Zt, wide=wide)
    assert cache.nbytes < 0.75 * 4 * N * N + (1 << 22)            # about half of the dense matrix (plus block padding)
    out = ops.symcache_mvm(cache, Vt, 0.2, 0.3)
    ref = omvm(Z, Z, V, 0.2, 0.3)
    assert _rel(out.cpu().numpy(), ref) < 2e-6
    # identical arithmetic to the direct fused sweep: same sums in the same order
    fused = ops.mvm_sym(Zt, Vt, 0.2, 0.3)
    assert _rel(out.cpu().numpy(), fused.cpu().numpy()) < 2e-6
    # a column range of the projections
    if J >= 5:
        part = ops.symcache_mvm(ops.SymCache(Zt, j0=1, j1=J - 1, wide=wide), Vt, 0.2, 0.0)
        assert _rel(part.cpu().numpy(), omvm(Z[:, 1:J - 1], Z[:, 1:J - 1], V, 0.2, 0.0)) < 2e-6


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("N,world", [(3000, 2), (9000, 3)])
def test_symcache_pair_shards_sum_to_whole(gpu_device, N, world, wide):
    from rpgp_amd import ops
    rng = np.random.default_rng(N)
    Zt = torch.from_numpy(rng.standard_normal((N, 20)).astype(np.float32)).to(gpu_device)
    Vt = torch.from_numpy(rng.standard_normal((N, 11)).astype(np.float32)).to(gpu_device)
    whole = ops.symcache_mvm(ops.SymCache(Zt, wide=wide), Vt, 0.05, 0.1)
    total, nbytes = None, 0
    for r in range(world):
        c = ops.SymCache(Zt, shard=(world, r), wide=wide)
        nbytes += c.nbytes
        o = ops.symcache_mvm(c, Vt, 0.05, 0.1 if r == 0 else 0.0)
        total = o if total is None else total + o
    assert float((total - whole).norm() / whole.norm()) < 1e-6
    assert nbytes >= ops.SymCache(Zt).nbytes                      # the shards cover the cache (each carries its own pad)


@pytest.mark.parametrize("N,T,shard", [(257, 11, None), (4097, 11, None), (4500, 16, None), (6211, 5, None), (5000, 20, None),
                                       (9000, 11, (3, 1)), (12345, 11, None), (2049, 37, (2, 0))])
def test_symcache_wide_product_ragged_sizes_and_shards(gpu_device, N, T, shard):
    """The wide product (one workgroup barrier per subtile, double-buffered LDS images, contiguous slab stores) for ragged
    sizes, one and two rows per lane, pair shards, blocks of more than 16 right-hand sides (second pass at t0 = 16: the strided
    slab stores): against the oracle (whole cache), reproducible bit for bit, and — whatever the load policy the cache's size
    selects — equal to the thin layout's product of the same columns to rounding."""
    from rpgp_amd import ops
    rng = np.random.default_rng(N + T)
    Z = rng.standard_normal((N, 20)).astype(np.float32)
    V = rng.standard_normal((N, T)).astype(np.float32)
    Zt, Vt = torch.from_numpy(Z).to(gpu_device), torch.from_numpy(V).to(gpu_device)
    cache = ops.SymCache(Zt, wide=True, shard=shard)
    out = ops.symcache_mvm(cache, Vt, 0.2, 0.3)
    assert torch.isfinite(out).all()
    assert torch.equal(ops.symcache_mvm(cache, Vt, 0.2, 0.3), out)
    thin = ops.SymCache(Zt, wide=False, shard=shard)
    ref_thin = ops.symcache_mvm(thin, Vt[:, :3].contiguous(), 0.2, 0.3)
    assert float((out[:, :3] - ref_thin).norm() / ref_thin.norm()) < 2e-6
    if shard is None:
        assert _rel(out.cpu().numpy(), omvm(Z, Z, V, 0.2, 0.3)) < 2e-6


@pytest.mark.parametrize("N", [4095, 4096, 16384, 16385, 36000, 36001])
def test_symcache_layout_rule_boundaries(gpu_device, N):
    """symk_plan's thresholds (one row-tile set per wave up to N = 16384; one exactly filled round of workgroups up to 36000,
    three above): both layouts against the fused sweep on either side of each."""
    from rpgp_amd import ops
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, 20, generator=g).to(gpu_device)
    V = torch.randn(N, 11, generator=g).to(gpu_device)
    ref = ops.mvm_sym(Z, V, 0.05, 0.1)
    wide = ops.symcache_mvm(ops.SymCache(Z, wide=True), V, 0.05, 0.1)
    assert float((wide - ref).norm() / ref.norm()) < 2e-6
    if N <= 20000:
        thin = ops.symcache_mvm(ops.SymCache(Z, wide=False), V[:, :3].contiguous(), 0.05, 0.1)
        assert float((thin - ref[:, :3]).norm() / ref[:, :3].norm()) < 2e-6


def test_symcache_rejects_mismatched_arguments(gpu_device):
    from rpgp_amd import ops, _lib
    Zt = torch.randn(2000, 4, device=gpu_device)
    cache = ops.SymCache(Zt)
    with pytest.raises(ValueError):
        ops.symcache_mvm(cache, torch.zeros(1999, 1, device=gpu_device), 1.0)
    lib = _lib.load()
    out = torch.empty(2000, 1, device=gpu_device)
    V = torch.zeros(2000, 1, device=gpu_device)
    # a cache that is too small for the problem is refused, as is a missing workspace
    assert lib.rpgp_symcache_mvm(cache.buf.data_ptr(), cache.nbytes // 2, 0, V.data_ptr(), out.data_ptr(), 2000, 1, 1.0, 0.0,
                                 1, 0, None, 0, None) != 0
    assert lib.rpgp_symcache_mvm(cache.buf.data_ptr(), cache.nbytes, 7, V.data_ptr(), out.data_ptr(), 2000, 1, 1.0, 0.0,
                                 1, 0, None, 0, None) != 0          # unknown layout
    assert lib.rpgp_symcache_bytes(2000, 2, 2) == 0


def test_errors_are_python_exceptions(gpu_device):
    from rpgp_amd import ops
    Z = torch.zeros((10, 4), device=gpu_device)
    with pytest.raises(ValueError):
        ops.mvm_sym(Z, torch.zeros((9, 1), device=gpu_device), 1.0)
    with pytest.raises(ValueError):
        ops.mvm_sym(Z, torch.zeros((10, 1), device=gpu_device), 1.0, j0=3, j1=2)
    with pytest.raises(RuntimeError):
        ops.mvm_sym(Z.cpu(), torch.zeros((10, 1)), 1.0)
    with pytest.raises(TypeError):
        ops.mvm_sym(Z.half(), torch.zeros((10, 1), device=gpu_device).half(), 1.0)
    with pytest.raises(TypeError):
        ops.Prepared(Z.double())                          # the factorised fast path is fp32 only


@pytest.mark.parametrize("N,J", [(277, 20), (700, 6), (64, 3)])
def test_bilinear_grad_dense_matches_oracle(gpu_device, N, J):
    from rpgp_amd import ops
    rng = np.random.default_rng(N)
    Z = rng.standard_normal((N, J)).astype(np.float32)
    A = rng.standard_normal((N, N)).astype(np.float32)
    S = (A + A.T).astype(np.float32)
    gZ, gs = ops.bilinear_grad_dense(torch.from_numpy(Z).to(gpu_device), torch.from_numpy(S).to(gpu_device), 0.3)
    z = Z.astype(np.float64)
    s = S.astype(np.float64)
    g_ref = np.zeros_like(z)
    ks = np.zeros((N, N))
    for j in range(J):
        d = z[:, j:j + 1] - z[:, j:j + 1].T
        e = np.exp(-0.5 * d * d)
        ks += e
        g_ref[:, j] = -0.3 * (s * e * d).sum(axis=1)
    assert _rel(gZ.cpu().numpy(), g_ref) < 2e-5
    gs_ref = 0.5 * (s * ks).sum()
    assert abs(gs.item() - gs_ref) < 2e-4 * np.abs(s * ks).sum() * 0.5 / N + 2e-5 * abs(gs_ref)


@pytest.mark.parametrize("N,J,T,shift", [(63, 20, 1, 0.0), (1000, 20, 1, 5.0), (777, 18, 4, -3.0), (2049, 8, 1, 100.0),
                                         (1500, 7, 13, 0.0), (4096, 20, 11, 1.0), (300, 3, 1, 0.0), (16500, 20, 1, 0.5)])
def test_mvm_sym_prepared_matches_oracle(gpu_device, N, J, T, shift):
    """Factorised fast path (rpgp_prepare + rpgp_mvm_sym_prepared): same contract as rpgp_mvm_sym; the centring makes
    it invariant to a common shift of the projected coordinates."""
    from rpgp_amd import ops
    Z, V = _data(N, J, T, seed=N + J + 1)
    Z = (Z + np.float32(shift)).astype(np.float32)
    scale, noise = 0.7 / J, 0.1
    ref = omvm(Z, Z, V, scale, noise)
    Zt = torch.from_numpy(Z).to(gpu_device)
    prep = ops.Prepared(Zt)
    assert prep.fast_ok and prep.max_abs < 10.0
    out = ops.mvm_sym_prepared(prep, torch.from_numpy(V).to(gpu_device), scale, noise)
    assert _rel(out.cpu().numpy(), ref) < 1e-5
    part = ops.mvm_sym_prepared(prep, torch.from_numpy(V).to(gpu_device), scale, 0.0, j0=1, j1=J)
    ref_p = omvm(Z[:, 1:], Z[:, 1:], V, scale) if J > 1 else None
    if ref_p is not None:
        assert _rel(part.cpu().numpy(), ref_p) < 1e-5


def test_prepared_range_guard(gpu_device):
    """Coordinates spread beyond the safe exponent range are flagged; the exact direct kernel still handles them."""
    from rpgp_amd import ops
    rng = np.random.default_rng(0)
    Z = (rng.standard_normal((500, 6)) * 40.0).astype(np.float32)
    V = rng.standard_normal((500, 1)).astype(np.float32)
    Zt = torch.from_numpy(Z).to(gpu_device)
    prep = ops.Prepared(Zt)
    assert not prep.fast_ok and prep.max_abs > 10.0
    with pytest.raises(RuntimeError):
        ops.mvm_sym_prepared(prep, torch.from_numpy(V).to(gpu_device), 1.0)
    out = ops.mvm_sym(Zt, torch.from_numpy(V).to(gpu_device), 1.0, 0.0)
    assert _rel(out.cpu().numpy(), omvm(Z, Z, V, 1.0)) < 1e-5
    bad = torch.from_numpy(np.full((10, 2), np.nan, dtype=np.float32)).to(gpu_device)
    assert not ops.Prepared(bad).fast_ok


@pytest.mark.parametrize("N,J,rank", [(500, 20, 15), (3000, 3, 15), (64, 8, 10), (2049, 20, 15), (20000, 20, 15),
                                      (131073, 3, 8)])
def test_fused_pivoted_cholesky_matches_generic(gpu_device, N, J, rank):
    """rpgp_pivoted_cholesky (one workgroup for N <= 2048, one chip-wide launch per greedy step above) against the
    generic row-by-row implementation on the same operator."""
    from rpgp_amd import ops
    from rpgp_amd.operators import AdditiveRPOperator
    from rpgp_amd.precond import pivoted_cholesky
    rng = np.random.default_rng(N)
    Z = torch.from_numpy((rng.standard_normal((N, J)) * 0.7).astype(np.float32)).to(gpu_device)
    op = AdditiveRPOperator(Z, None, torch.tensor(0.8, device=gpu_device), 1.0 / J)
    Lf = ops.pivoted_cholesky(Z, 0.8 / J, rank)
    Lg = pivoted_cholesky(op._diagonal(), op._get_rows, rank)
    if N <= 20000:
        K = op.to_dense().double()
        ef = (K - Lf.double() @ Lf.double().t()).abs().max().item()
        eg = (K - Lg.double() @ Lg.double().t()).abs().max().item()
        assert ef <= eg * 1.05 + 1e-5
    assert torch.allclose(Lf, Lg, rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("kind,group,cols,rank", [("Matern", 1, 6, 12), ("RBF", 2, 8, 12), ("InverseMQ", 1, 20, 12),
                                                  ("Cosine", 1, 3, 4)])      # cos kernel: exact rank 2 per column
def test_family_pivoted_cholesky_matches_generic(gpu_device, kind, group, cols, rank):
    from rpgp_amd.operators import FamilyAdditiveOperator
    from rpgp_amd.precond import pivoted_cholesky
    rng = np.random.default_rng(cols)
    N = 5000
    Z = torch.from_numpy((rng.standard_normal((N, cols)) * 0.7).astype(np.float32)).to(gpu_device)
    w = torch.from_numpy(rng.uniform(0.3, 1.0, size=cols // group).astype(np.float32)).to(gpu_device)
    op = FamilyAdditiveOperator(Z, None, torch.tensor(0.8, device=gpu_device), w, kind, group)
    Lf = op.fused_pivoted_cholesky(rank)
    Lg = pivoted_cholesky(op._diagonal(), op._get_rows, rank)
    assert Lf is not None and torch.allclose(Lf, Lg, rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("N,J,rank", [(2049, 20, 15), (7372, 20, 15), (14939, 8, 16), (50000, 20, 15), (131072, 3, 15),
                                      (131073, 3, 15), (9000, 32, 15), (9000, 33, 15), (4097, 20, 1), (9999, 64, 7)])
def test_pivoted_cholesky_step_kernels_on_both_sides_of_their_limits(gpu_device, N, J, rank):
    """pivchol_step_fast_kernel (the flagship operator's per-step kernel: own-row operands requested before the pivot is
    known; <= 32 columns, one row per thread up to N = 131072) and the general per-step kernel beyond its limits: the factor
    against the generic row-by-row greedy factorisation of the same operator, reproducible on one scratch buffer."""
    from rpgp_amd import ops
    from rpgp_amd.operators import AdditiveRPOperator
    from rpgp_amd.precond import pivoted_cholesky
    rng = np.random.default_rng(N + J)
    Z = torch.from_numpy((rng.standard_normal((N, J)) * 0.8).astype(np.float32)).to(gpu_device)
    L = ops.pivoted_cholesky(Z, 0.7 / J, rank)
    assert torch.isfinite(L).all()
    assert torch.equal(ops.pivoted_cholesky(Z, 0.7 / J, rank), L)
    op = AdditiveRPOperator(Z, None, torch.tensor(0.7, device=gpu_device), 1.0 / J)
    Lg = pivoted_cholesky(op._diagonal(), op._get_rows, rank)
    assert torch.allclose(L, Lg, rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("N,T,world", [(3000, 1, 3), (20000, 1, 8), (5000, 11, 4), (700, 4, 2), (300, 1, 8)])
def test_pair_sharded_mvm_sums_to_full(gpu_device, N, T, world):
    """Pair-sharding (rpgp_mvm_sym[_prepared]_range): the per-rank partial products sum to the full MVM, for the
    direct and the prepared kernels, including j-ranges."""
    from rpgp_amd import ops
    Z, V = _data(N, 20, T, seed=N)
    Zt, Vt = torch.from_numpy(Z).to(gpu_device), torch.from_numpy(V).to(gpu_device)
    full = ops.mvm_sym(Zt, Vt, 0.05, 0.0)
    prep = ops.Prepared(Zt)
    acc_d = sum(ops.mvm_sym(Zt, Vt, 0.05, 0.0, shard=(world, r)) for r in range(world))
    acc_p = sum(ops.mvm_sym_prepared(prep, Vt, 0.05, 0.0, shard=(world, r)) for r in range(world))
    assert _rel(acc_d.cpu().numpy(), full.cpu().numpy()) < 2e-6
    assert _rel(acc_p.cpu().numpy(), full.cpu().numpy()) < 2e-6
    part = sum(ops.mvm_sym_prepared(prep, Vt, 0.05, 0.0, j0=3, j1=11, shard=(world, r)) for r in range(world))
    assert _rel(part.cpu().numpy(), ops.mvm_sym(Zt, Vt, 0.05, 0.0, j0=3, j1=11).cpu().numpy()) < 2e-6


def test_edge_shapes(gpu_device):
    """Ragged / extreme shapes: N = 2, J = 70 (> the prepared path's 64-projection limit, pieces 20+20+20+10), a zero
    right-hand side, T = 33, duplicate points."""
    from rpgp_amd import ops
    from rpgp_amd.operators import AdditiveRPOperator
    rng = np.random.default_rng(0)
    Z = rng.standard_normal((2, 3)).astype(np.float32)
    V = rng.standard_normal((2, 1)).astype(np.float32)
    out = ops.mvm_sym(torch.from_numpy(Z).to(gpu_device), torch.from_numpy(V).to(gpu_device), 1.0, 0.5)
    assert _rel(out.cpu().numpy(), omvm(Z, Z, V, 1.0, 0.5)) < 1e-6
    Z = rng.standard_normal((600, 70)).astype(np.float32)
    V = rng.standard_normal((600, 33)).astype(np.float32)
    Zt, Vt = torch.from_numpy(Z).to(gpu_device), torch.from_numpy(V).to(gpu_device)
    ref = omvm(Z, Z, V, 0.01, 0.1)
    assert _rel(ops.mvm_sym(Zt, Vt, 0.01, 0.1).cpu().numpy(), ref) < 1e-5
    op = AdditiveRPOperator(Zt, None, torch.tensor(0.7, device=gpu_device), 1.0 / 70)
    assert _rel(op._matmul(Vt, noise=0.1).cpu().numpy(), omvm(Z, Z, V, 0.01, 0.1)) < 1e-5      # J > 64: direct kernel
    assert op._prep is not None and not op._prep.fast_ok
    zero = torch.zeros(600, 2, device=gpu_device)
    assert float(ops.mvm_sym(Zt, zero, 0.01, 0.1).abs().max()) == 0.0
    Zd = np.repeat(rng.standard_normal((1, 5)).astype(np.float32), 300, axis=0)       # all points identical: K = s*J*11^T
    Vd = rng.standard_normal((300, 1)).astype(np.float32)
    Zdt = torch.from_numpy(Zd).to(gpu_device)
    for fn in (lambda: ops.mvm_sym(Zdt, torch.from_numpy(Vd).to(gpu_device), 0.2, 0.0),
               lambda: ops.mvm_sym_prepared(ops.Prepared(Zdt), torch.from_numpy(Vd).to(gpu_device), 0.2, 0.0)):
        np.testing.assert_allclose(fn().cpu().numpy().ravel(), np.full(300, 0.2 * 5 * Vd.sum()), rtol=2e-5, atol=2e-4)


def test_space_equally_on_device_matches_reference_golden(gpu_device):
    """rpgp_space_equally (one launch) against the golden outputs captured from the reference's rp.space_equally (5000
    autograd steps on the host), same tolerance as the CPU restatement's golden test, plus the reference's own
    properties (test.py:460-491): unit rows, non-trivial final energy."""
    import os
    from rpgp_amd import rp, ops
    s = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "space_equally.npz"))
    for (J, d, seed) in [(20, 18, 0), (20, 8, 1), (4, 2, 3)]:
        torch.manual_seed(seed)
        np.random.seed(seed)
        P0 = torch.cat([rp.gen_rp(d, 1, "gaussian") for _ in range(J)], dim=1).t().contiguous()
        out, loss = ops.space_equally(P0.to(gpu_device), 0.1, 5000)
        ref = s["J%d_d%d_s%d_out" % (J, d, seed)]
        np.testing.assert_allclose(out.cpu().numpy(), ref, atol=5e-5)
        np.testing.assert_allclose(np.linalg.norm(out.cpu().numpy(), axis=1), 1.0, atol=1e-5)
        assert float(loss) > 1e-3
        out2, _ = rp.space_equally(P0.clone(), 0.1, 5000)          # the module-level entry takes the device path here
        assert out2.device.type == "cpu" and np.allclose(out2.numpy(), ref, atol=5e-5)


@pytest.mark.parametrize("N", [14939, 7372, 5000])
def test_workspace_contents_never_leak_into_results(gpu_device, N):
    """Regression: with a NaN-poisoned (re-used) workspace every op must return what it returns on a fresh one.  The
    bilinear derivative used to read the slabs of column splits that own no columns (rounding of the split width)."""
    from rpgp_amd import ops
    rng = np.random.default_rng(N)
    Z = torch.from_numpy(rng.standard_normal((N, 20)).astype(np.float32)).to(gpu_device)
    L = torch.from_numpy(rng.standard_normal((N, 3)).astype(np.float32)).to(gpu_device)
    R = torch.from_numpy(rng.standard_normal((N, 3)).astype(np.float32)).to(gpu_device)
    fam = ops.Family("Matern", 1, torch.full((20,), 0.05, device=gpu_device))

    def run():
        gp = ops.ski_grid(Z, None, 256)
        return [ops.mvm_sym(Z, L, 0.05, 0.1), ops.mvm_sym_prepared(ops.Prepared(Z), L, 0.05, 0.1),
                *ops.bilinear_grad(Z, L, R, 0.05), *ops.family_bilinear_grad(fam, Z, L, R, 0.9),
                ops.family_mvm_sym(fam, Z, L, 0.9, 0.1), ops.ski_mvm(Z, Z, gp, L, 0.05, 0.1, 256),
                *ops.ski_bilinear_grad(Z, gp, L, R, 0.05, 256), ops.dense_mvm(ops.dense(Z[:4096], Z[:4096], 0.05), L[:4096])]

    ref = run()
    for buf in ops._workspaces.values():
        buf.view(torch.float32).fill_(float("nan"))
    got = run()
    for a, b in zip(ref, got):
        assert torch.isfinite(b).all()
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_no_op_reads_uninitialised_scratch(gpu_device, monkeypatch):
    """Every scratch / output tensor the wrappers allocate with torch.empty is NaN-filled here: results must not change
    (an op that reads memory it has not written shows up as NaN or as a difference)."""
    from rpgp_amd import ops
    rng = np.random.default_rng(3)
    N = 14939
    Z = torch.from_numpy(rng.standard_normal((N, 20)).astype(np.float32)).to(gpu_device)
    Zs = torch.from_numpy(rng.standard_normal((700, 20)).astype(np.float32)).to(gpu_device)
    X = torch.from_numpy(rng.standard_normal((N, 18)).astype(np.float32)).to(gpu_device)
    Pm = torch.from_numpy(rng.standard_normal((18, 20)).astype(np.float32)).to(gpu_device)
    L = torch.from_numpy(rng.standard_normal((N, 11)).astype(np.float32)).to(gpu_device)
    R = torch.from_numpy(rng.standard_normal((N, 11)).astype(np.float32)).to(gpu_device)
    w = torch.from_numpy(rng.uniform(0.3, 1.0, 20).astype(np.float32)).to(gpu_device)

    def run():
        fam = ops.Family("InverseMQ", 1, w)
        famg = ops.Family("RBF", 4, w[:5])
        gp = ops.ski_grid(Z, Zs, 512, weights=w)
        prep = ops.Prepared(Z)
        desc, keep = ops.make_operator_desc(1, N, 20, 0.05, 0.3, prep=prep)
        x, ah, bh, it, mres = ops.mbcg_solve(desc, L, 1e-3, 200, hist_len=5)
        return [ops.project(X, Pm), ops.project_grad(X, Z), ops.mvm_sym(Z, L, 0.05, 0.1), ops.mvm_sym_prepared(prep, L, 0.05, 0.1),
                ops.mvm_sym_prepared(prep, L, 0.05, 0.0, shard=(3, 1)), ops.mvm_rect(Zs, Z, L, 0.05), ops.dense(Zs, Z, 0.05),
                *ops.bilinear_grad(Z, L, R, 0.05), ops.pivoted_cholesky(Z, 0.05, 15), ops.family_mvm_sym(fam, Z, L, 0.9, 0.1),
                ops.family_mvm_sym(famg, Z, L, 0.9, 0.1), ops.family_mvm_rect(fam, Zs, Z, L, 0.9), ops.family_dense(fam, Zs, Z, 0.9),
                *ops.family_bilinear_grad(famg, Z, L, R, 0.9), ops.family_pivoted_cholesky(fam, Z, 0.9, 8, float(w.sum())),
                ops.ski_mvm(Z, Z, gp, L, 0.05, 0.1, 512), ops.ski_mvm(Zs, Z, gp, L, 0.05, 0.0, 512), ops.ski_diag(Z, gp, 0.05, 512),
                ops.ski_dense(Zs, Z, gp, 0.05, 512), *ops.ski_bilinear_grad_comp(Z, gp, L, R, 0.05, 512),
                ops.ski_pivoted_cholesky(Z, gp, 0.05, 8, 512), x, torch.from_numpy(ah), torch.from_numpy(bh),
                ops.space_equally(Pm.t().contiguous(), 0.1, 50)[0]]

    ref = [t.clone() for t in run()]
    real_empty, real_empty_like = torch.empty, torch.empty_like

    def nan_empty(*a, **k):
        t = real_empty(*a, **k)
        return t.fill_(float("nan")) if t.is_floating_point() else t

    def nan_empty_like(*a, **k):
        t = real_empty_like(*a, **k)
        return t.fill_(float("nan")) if t.is_floating_point() else t

    monkeypatch.setattr(ops.torch, "empty", nan_empty)
    monkeypatch.setattr(ops.torch, "empty_like", nan_empty_like)
    for buf in ops._workspaces.values():
        buf.view(real_empty(0).dtype if False else torch.float32).fill_(float("nan"))
    got = run()
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.isfinite(b).all(), "output %d has non-finite entries" % i
        assert torch.allclose(a.cpu(), b.cpu(), rtol=1e-5, atol=1e-6), "output %d changed" % i


@pytest.mark.parametrize("N,K,T", [(1, 1, 1), (7, 15, 10), (1237, 15, 11), (7372, 15, 10), (50001, 16, 64), (4099, 33, 17),
                                   (2048, 64, 64), (300, 20, 48)])
def test_gram_f64_and_woodbury_apply_match_float64(gpu_device, N, K, T):
    """rpgp_gram_f64 / rpgp_woodbury_apply (the preconditioner's Gram products and cancelling update outside the executor)
    against float64 numpy; the Gram product of fp32 inputs is exact up to float64 summation order; strided inputs."""
    from rpgp_amd import ops
    rng = np.random.default_rng(N + K + T)
    A = rng.normal(size=(N, K)).astype(np.float32) * 3.0
    Bw = rng.normal(size=(N, T + 5)).astype(np.float32)              # B is a column slice of a wider matrix
    At, Bt = torch.from_numpy(A).to(gpu_device), torch.from_numpy(Bw).to(gpu_device)[:, 2:2 + T]
    B = Bw[:, 2:2 + T]
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    got = ops.gram_f64(At, Bt).cpu().numpy()
    assert got.shape == (K, T) and np.allclose(got, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    got2 = ops.gram_f64(At, Bt).cpu().numpy()
    assert np.array_equal(got, got2)                                  # fixed summation order
    sym = ops.gram_f64(At, At).cpu().numpy()
    assert np.allclose(sym, A.astype(np.float64).T @ A.astype(np.float64), rtol=1e-12, atol=1e-9)
    Tm = rng.normal(size=(K, T))
    noise = 0.37
    out = ops.woodbury_apply(At, Bt, torch.from_numpy(Tm).to(gpu_device), noise).cpu().numpy()
    want = (B.astype(np.float64) - A.astype(np.float64) @ Tm) / noise
    assert np.allclose(out, want.astype(np.float32), rtol=2e-7, atol=1e-6 * np.abs(want).max())
    with pytest.raises(ValueError):
        ops.gram_f64(At, torch.zeros(N + 1, T, device=gpu_device))
    with pytest.raises(TypeError):
        ops.gram_f64(At.double(), Bt)


def test_woodbury_preconditioner_uses_the_in_tree_kernels(gpu_device):
    """precond.WoodburyPreconditioner on the HIP backend: M^-1 (M r) = r in the ill-conditioned regime of the C5 shape
    (|L^T L| / sigma^2 ~ 1e6) and agreement with the float64 dense inverse; wide blocks go through in panels."""
    from rpgp_amd.precond import WoodburyPreconditioner
    rng = np.random.default_rng(0)
    N, K = 20000, 15
    L = (rng.normal(size=(N, K)) * np.linspace(3.0, 0.05, K)).astype(np.float32)
    noise = 0.05
    pre = WoodburyPreconditioner(torch.from_numpy(L).to(gpu_device), noise)
    assert pre._be is not None and pre._L64c is None
    L64 = L.astype(np.float64)
    cap = noise * np.eye(K) + L64.T @ L64
    for T in (1, 11, 130):
        r = rng.normal(size=(N, T)).astype(np.float32)
        want = (r - L64 @ np.linalg.solve(cap, L64.T @ r)) / noise
        got = pre.solve(torch.from_numpy(r).to(gpu_device)).cpu().numpy()
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 5e-7
    v = pre.solve(torch.from_numpy(r[:, 0].copy()).to(gpu_device))
    assert v.shape == (N,)
    ld = pre.logdet()
    assert abs(ld - (np.linalg.slogdet(cap)[1] + (N - K) * math.log(noise))) < 1e-6 * abs(ld)
    assert pre._logdet_cap is not None                              # rpgp_woodbury_setup served the capacitance matrix
    assert np.allclose(pre.cinv().cpu().numpy(), np.linalg.inv(cap), rtol=1e-9, atol=1e-12 * np.abs(np.linalg.inv(cap)).max())
    assert np.allclose(pre._cap_chol.cpu().numpy(), np.linalg.cholesky(cap), rtol=1e-10, atol=1e-12)
    assert pre._L64c is None                                        # the float64 copy of L is never materialised


@pytest.mark.parametrize("K", [1, 2, 15, 16, 33, 64])
def test_woodbury_setup_matches_numpy(gpu_device, K):
    """rpgp_woodbury_setup: Cholesky factor, inverse and log-determinant of gram + noise I in one launch (float64)."""
    from rpgp_amd import ops
    rng = np.random.default_rng(K)
    A = rng.normal(size=(3 * K + 5, K))
    G = A.T @ A * np.linspace(0.1, 30.0, K)[:, None] * np.linspace(0.1, 30.0, K)[None, :]     # ill-scaled SPD Gram matrix
    noise = 0.05
    chol, cinv, ld = ops.woodbury_setup(torch.from_numpy(G).to(gpu_device), noise)
    C = G + noise * np.eye(K)
    assert np.allclose(chol.cpu().numpy(), np.linalg.cholesky(C), rtol=1e-9, atol=1e-11 * np.abs(C).max())
    assert np.allclose(cinv.cpu().numpy() @ C, np.eye(K), atol=1e-7)
    assert abs(float(ld) - np.linalg.slogdet(C)[1]) < 1e-9 * max(1.0, abs(np.linalg.slogdet(C)[1]))
    bad = -np.eye(K)
    _, cinv_b, ld_b = ops.woodbury_setup(torch.from_numpy(bad).to(gpu_device), 0.5)          # not positive definite -> NaN
    assert np.isnan(float(ld_b)) and np.isnan(cinv_b.cpu().numpy()).all()
    with pytest.raises(TypeError):
        ops.woodbury_setup(torch.zeros(65, 65, dtype=torch.float64, device=gpu_device), 1.0)


def test_step_kernels_against_torch_formulas(gpu_device):
    """csrc/rpgp_step.hip, each entry point against the torch operations it replaces in the optimiser step (fused_mll.py)."""
    import math
    from torch.nn import functional as F
    from rpgp_amd import ops
    g = torch.Generator().manual_seed(11)
    for d, J, prescale, n_ls in ((8, 20, True, 8), (6, 5, False, 5), (7, 3, True, 1)):
        raw_ls = torch.randn(n_ls, generator=g).to(gpu_device)
        raw_os, raw_nz, mean = (torch.randn(1, generator=g).to(gpu_device) for _ in range(3))
        W = torch.randn(J, d, generator=g).to(gpu_device)
        Peff, hyp, os_f, nz_f, mu_f = ops.step_hyper(raw_ls, raw_os, raw_nz, mean, W, prescale, 1e-4)
        ls = F.softplus(raw_ls)
        col = ls if n_ls == 1 else (ls.reshape(-1, 1) if prescale else ls.reshape(1, -1))
        assert torch.allclose(Peff, W.t() / col, rtol=2e-6, atol=0)
        assert abs(os_f - float(F.softplus(raw_os))) < 2e-6 * abs(os_f) and abs(nz_f - float(F.softplus(raw_nz) + 1e-4)) < 2e-6 * nz_f
        assert mu_f == float(mean)
        assert torch.allclose(hyp[:3].cpu(), torch.tensor([os_f, nz_f, mu_f]))
        assert torch.allclose(hyp[3:5], torch.sigmoid(torch.cat([raw_os, raw_nz])), rtol=2e-6)
        assert torch.allclose(hyp[8:8 + n_ls], ls, rtol=2e-6) and torch.allclose(hyp[8 + n_ls:], torch.sigmoid(raw_ls), rtol=2e-6)
        # chain rule back
        dPeff = torch.randn(d, J, generator=g).to(gpu_device)
        gs, gten = torch.randn(1, generator=g).to(gpu_device), torch.tensor([-1.0], device=gpu_device)
        part = torch.randn(2 * 37, generator=g).to(gpu_device)
        zfac, gscale, dlp_n = 0.83, -0.5 / 1234, 0.017
        g_ls, g_os, g_nz, g_mu = ops.step_hyper_backward(dPeff, W, n_ls, prescale, zfac, hyp, gs, part, 37, gten, gscale, dlp_n,
                                                         gs_scale=0.7)
        t = dPeff * zfac * W.t()
        ref_ls = -(t.sum() if n_ls == 1 else (t.sum(1) if prescale else t.sum(0))) / (ls * ls) * torch.sigmoid(raw_ls)
        assert torch.allclose(g_ls, ref_ls.reshape(-1), rtol=1e-5, atol=1e-6)
        assert torch.allclose(g_os, gs * 0.7 * torch.sigmoid(raw_os), rtol=1e-6)
        gq = float(gten) * gscale
        assert abs(float(g_nz) - (float(part[0::2].double().sum()) + float(gten) * dlp_n) * float(torch.sigmoid(raw_nz))) < 1e-5
        assert abs(float(g_mu) + 2 * gq * float(part[1::2].double().sum())) < 1e-6
    # (400 001 rows: the C5 size — several batches per thread, the workgroup caps; rank 64 with 16 probes: the 88 KB tile of the
    #  probe kernel)
    for N, k, p in ((3001, 15, 10), (70000, 9, 7), (400001, 15, 10), (5000, 64, 16)):
        L = torch.randn(N, k, generator=g).to(gpu_device)
        e1, e2 = torch.randn(k, p, generator=g).to(gpu_device), torch.randn(N, p, generator=g).to(gpu_device)
        y, mean = torch.randn(N, generator=g).to(gpu_device), torch.tensor([0.3], device=gpu_device)
        full_rhs = ops.step_probes(L, e1, e2, math.sqrt(0.2), y, mean)
        ref = L.double() @ e1.double() + math.sqrt(0.2) * e2.double()
        assert (full_rhs[:, :p].double() - ref).abs().max() < 1e-5 * ref.abs().max()
        assert torch.equal(full_rhs[:, p], y - 0.3)
        assert torch.equal(ops.step_probes(L, e1, e2, math.sqrt(0.2), y, mean), full_rhs)
        sol = torch.randn(N, p + 1, generator=g).to(gpu_device)
        out = ops.step_value(full_rhs, sol, p, 12.5, -0.5 / N, 0.75)
        iq = float((full_rhs[:, p].double() * sol[:, p].double()).sum())
        assert abs(float(out[1]) - iq) < 1e-6 * max(1.0, abs(iq)) * 10
        assert abs(float(out[0]) - ((iq + 12.5) * (-0.5 / N) + 0.75)) < 1e-6
        pre_probes, gten = torch.randn(N, p, generator=g).to(gpu_device), torch.tensor([-1.0], device=gpu_device)
        wide = torch.zeros(N, p + 3, device=gpu_device)                 # a strided view as the probes' side
        wide[:, :p] = pre_probes
        left, right, part, nparts = ops.step_lr(sol, wide[:, :p], gten, -0.5 / N)
        gq = -1.0 * (-0.5 / N)
        assert torch.allclose(left[:, :p], sol[:, :p] * (gq / p), rtol=2e-6, atol=1e-12)
        assert torch.allclose(left[:, p], -gq * sol[:, p], rtol=2e-6) and torch.equal(right[:, :p], pre_probes)
        assert torch.equal(right[:, p], sol[:, p])
        lr = float((left.double() * right.double()).sum())
        assert abs(float(part[:2 * nparts][0::2].double().sum()) - lr) < 1e-5 * max(abs(lr), 1e-6) + 1e-9
        assert abs(float(part[:2 * nparts][1::2].double().sum()) - float(sol[:, p].double().sum())) < 1e-3
