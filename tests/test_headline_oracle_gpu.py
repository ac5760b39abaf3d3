"""What bench.py times, oracle-checked at the size it times it (BASELINE config C4: N = 50 000, d = 20, J = 20, bench
seeds).  Three 256-row blocks of  K[rows, :] @ V + sigma^2 V[rows]  — first, middle (straddling a 512-row tile
boundary) and last (the ragged tail: 50 000 = 97 * 512 + 336) — plus a seeded random sample of 256 rows spread over
every row-block class (so that no interior tile can be wrong unseen) are evaluated in FLOAT64 numpy from the reference's
formula (gp_models/kernels/memory_efficient_gam_kernel.py:20-30: sum_j exp(-0.5 (z_ij - z_i'j)^2)) and every product on the
path is compared with them through the C-ABI: the prepared kernel `mvm_fact_kernel` (T = 1: the benchmark's launch, and
the T = 11 training block), the direct kernel, the packed symmetric cache in both layouts, the dense cached-K stream, and
a pair-sharded sum.  Tolerance 1e-5 relative (2-norm per column) — the bound north_star states is 1e-4.
The bilinear derivative (memory_efficient_gam_kernel.py:33-59) is checked on 64 rows of gZ."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, D, J = 50000, 20, 20
SCALE, NOISE = 1.0 / J, 0.1
BLOCKS = [(0, 256), (24960, 25216), (N - 256, N)]
RANDOM_ROWS = np.sort(np.random.default_rng(20250104).choice(N, size=256, replace=False))
ROW_SETS = [np.arange(a, b) for a, b in BLOCKS] + [RANDOM_ROWS]


def _rows_ref(Zh, rows, V):
    """float64: scale * K[rows, :] @ V + noise * V[rows] with K = sum_j exp(-0.5 (z_ij - z_i'j)^2)."""
    K = np.zeros((len(rows), Zh.shape[0]))
    for j in range(Zh.shape[1]):
        dj = Zh[rows, j][:, None] - Zh[None, :, j]
        dj *= dj
        dj *= -0.5
        np.exp(dj, out=dj)
        K += dj
    return SCALE * (K @ V) + NOISE * V[rows], K


@pytest.fixture(scope="module")
def c4(gpu_device):
    import bench
    from oracle import dense_gp as orc
    from rpgp_amd import ops
    T = 11
    X, P, ls, V = bench.make_inputs(N, D, J, T, gpu_device)
    Z = ops.project(X, (P / ls[:, None]).contiguous())
    # the projection itself against the oracle's float64 (X / l) P
    Zo = orc.project(X.cpu().numpy(), P.cpu().numpy(), ls.cpu().numpy())
    assert np.abs(Z.double().cpu().numpy() - Zo).max() < 5e-6
    Zh = Z.double().cpu().numpy()                       # the oracle runs on the SAME projected inputs as the kernels
    Vh = V.double().cpu().numpy()
    refs = []
    for rows in ROW_SETS:
        ref, _ = _rows_ref(Zh, rows, Vh)
        refs.append((rows, ref))
    return {"Z": Z, "V": V, "Zh": Zh, "Vh": Vh, "refs": refs}


def _check(out, refs, cols, what):
    o = out.double().cpu().numpy()
    if o.ndim == 1:
        o = o[:, None]
    for rows, ref in refs:
        r = ref[:, cols]
        err = np.linalg.norm(o[rows] - r, axis=0) / np.linalg.norm(r, axis=0)
        assert err.max() < 1e-5, (what, int(rows[0]), err.max())


def test_prepared_kernel_bench_launch_t1_and_t11(c4):
    """`ops.mvm_sym_prepared` = mvm_fact_kernel<20, 1, 2> at T = 1 (exactly what bench.py launches) and the T = 11 block."""
    from rpgp_amd import ops
    prep = ops.Prepared(c4["Z"])
    assert prep.fast_ok
    V = c4["V"]
    _check(ops.mvm_sym_prepared(prep, V[:, :1].contiguous(), SCALE, NOISE), c4["refs"], slice(0, 1), "prepared T=1")
    _check(ops.mvm_sym_prepared(prep, V, SCALE, NOISE), c4["refs"], slice(0, 11), "prepared T=11")


def test_bench_launch_every_row_against_the_c_oracle(c4):
    """ALL 50 000 output rows of the benchmark's launch against the float64 C/OpenMP restatement (oracle/cmvm.c: the full
    2.5e9-entry kernel matrix, 5e10 exponentials — seconds on the GPU box's host cores): no tile of the decomposition is
    left unchecked."""
    from oracle import cmvm
    from rpgp_amd import ops
    ref = cmvm.mvm(c4["Zh"], c4["Zh"], c4["Vh"][:, :1], SCALE, NOISE)
    prep = ops.Prepared(c4["Z"])
    out = ops.mvm_sym_prepared(prep, c4["V"][:, :1].contiguous(), SCALE, NOISE).double().cpu().numpy()
    assert np.linalg.norm(out - ref) / np.linalg.norm(ref) < 1e-5
    assert np.abs(out - ref).max() < 1e-5 * np.abs(ref).max()
    # ... and the row subsets used by the other tests are the same numbers the numpy formula gives
    for rows, r in c4["refs"]:
        assert np.abs(ref[rows] - r[:, :1]).max() < 1e-11 * np.abs(r).max()


def test_direct_kernel_t1_and_t11(c4):
    from rpgp_amd import ops
    V = c4["V"]
    _check(ops.mvm_sym(c4["Z"], V[:, :1].contiguous(), SCALE, NOISE), c4["refs"], slice(0, 1), "direct T=1")
    _check(ops.mvm_sym(c4["Z"], V, SCALE, NOISE), c4["refs"], slice(0, 11), "direct T=11")


def test_pair_sharded_partials_sum_to_the_oracle(c4):
    """Three emulated ranks (prepared kernel, noise on rank 0 only): the sum of the partial products is the oracle's."""
    from rpgp_amd import ops
    prep = ops.Prepared(c4["Z"])
    V1 = c4["V"][:, :1].contiguous()
    tot = None
    for r in range(3):
        part = ops.mvm_sym_prepared(prep, V1, SCALE, NOISE if r == 0 else 0.0, shard=(3, r))
        tot = part if tot is None else tot + part
    _check(tot, c4["refs"], slice(0, 1), "pair-sharded x3")
    tot = None
    for r in range(8):                                  # the direct kernel, 8 ranks, the T = 11 block
        part = ops.mvm_sym(c4["Z"], c4["V"], SCALE, NOISE if r == 0 else 0.0, shard=(8, r))
        tot = part if tot is None else tot + part
    _check(tot, c4["refs"], slice(0, 11), "pair-sharded x8 direct T=11")


@pytest.mark.parametrize("wide", [False, True])
def test_symcache_both_layouts(c4, wide):
    from rpgp_amd import ops
    c = ops.SymCache(c4["Z"], wide=wide)
    V = c4["V"]
    _check(ops.symcache_mvm(c, V[:, :1].contiguous(), SCALE, NOISE), c4["refs"], slice(0, 1), "symcache T=1 wide=%s" % wide)
    _check(ops.symcache_mvm(c, V, SCALE, NOISE), c4["refs"], slice(0, 11), "symcache T=11 wide=%s" % wide)
    _check(ops.symcache_mvm(c, V[:, :4].contiguous(), SCALE, NOISE), c4["refs"], slice(0, 4), "symcache T=4 wide=%s" % wide)
    del c
    torch.cuda.empty_cache()


def test_dense_cached_k_stream(c4):
    from rpgp_amd import ops
    Kd = ops.dense(c4["Z"], c4["Z"], SCALE, pad=True)
    # the stored entries themselves on the three row blocks
    for rows, _ in c4["refs"]:
        _, K = _rows_ref(c4["Zh"], rows[:32], c4["Vh"])
        got = Kd.index_select(0, torch.from_numpy(rows[:32]).to(Kd.device))[:, :N]
        assert np.abs(got.double().cpu().numpy() - SCALE * K).max() < 2e-6
    V = c4["V"]
    _check(ops.dense_mvm(Kd, V[:, :1].contiguous(), NOISE), c4["refs"], slice(0, 1), "dense T=1")
    _check(ops.dense_mvm(Kd, V, NOISE), c4["refs"], slice(0, 11), "dense T=11")
    del Kd
    torch.cuda.empty_cache()


def test_bilinear_derivative_rows(c4, gpu_device):
    """rpgp_bilinear_grad at C4 with the training shapes (T = 11): 64 rows of gZ (first / middle / ragged tail) against
    float64; gscale against the checksum  sum(L * (K R)) / scale  with K R from the oracle-checked MVM above."""
    from rpgp_amd import ops
    T = 11
    g = torch.Generator().manual_seed(5)
    L = torch.randn(N, T, generator=g) * 0.1
    R = torch.randn(N, T, generator=torch.Generator().manual_seed(6)) * 0.1
    Lt, Rt = L.to(gpu_device), R.to(gpu_device)
    gZ, gs = ops.bilinear_grad(c4["Z"], Lt, Rt, SCALE)
    Zh, Lh, Rh = c4["Zh"], L.double().numpy(), R.double().numpy()
    rows = np.concatenate([np.arange(0, 22), np.arange(25077, 25098), np.arange(N - 21, N)])
    S = Lh[rows] @ Rh.T + Rh[rows] @ Lh.T                 # 64 x N
    ref = np.zeros((len(rows), J))
    for j in range(J):
        dj = Zh[rows, j][:, None] - Zh[None, :, j]
        e = np.exp(-0.5 * dj * dj)
        ref[:, j] = -SCALE * (S * e * dj).sum(axis=1)
    got = gZ.double().cpu().numpy()[rows]
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 2e-5
    assert np.abs(got - ref).max() < 5e-5 * np.abs(ref).max()
    KR = ops.mvm_sym(c4["Z"], Rt, SCALE, 0.0)
    gs_chk = float((Lt.double() * KR.double()).sum()) / SCALE
    assert abs(float(gs) - gs_chk) < 2e-5 * abs(gs_chk) + 1e-4
