"""Pins the float64 oracle (oracle/dense_gp.py) to the reference: golden vectors captured from the reference's own
GAMFunction / gen_rp / space_equally (tests/golden/make_golden.py) and the closed-form known answers of test.py."""
import os

import numpy as np
import pytest
import torch

from oracle import dense_gp as orc
from oracle import cpu_path

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_gam_forward_backward_match_reference_golden():
    g = np.load(os.path.join(GOLD, "gam.npz"))
    for name in g["names"]:
        x1, x2, ls = g[name + "_x1"], g[name + "_x2"], g[name + "_ls"]
        K = orc.gam_forward(x1, x2, ls)
        np.testing.assert_allclose(K, g[name + "_K"], rtol=1e-12, atol=1e-14)
        g1, g2, gl = orc.gam_backward(x1, x2, ls, g[name + "_go"])
        np.testing.assert_allclose(g1, g[name + "_gx1"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(g2, g[name + "_gx2"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(gl, g[name + "_gls"], rtol=1e-10, atol=1e-12)


def test_gam_testpy_literal_values():
    """K of test.py:641-657 inputs, quoted to 4 decimals."""
    g = np.load(os.path.join(GOLD, "gam.npz"))
    K = orc.gam_forward(g["testpy_x1"], g["testpy_x2"], g["testpy_ls"])
    np.testing.assert_allclose(K, [[1.4434, 1.3455], [2.6477, 1.0429], [1.8704, 1.8185]], atol=5e-5)


def test_bilinear_grad_consistent_with_gam_backward():
    """oracle.bilinear_grad (what the HIP derivative kernel is checked against) agrees with the reference-pinned
    GAM backward when x1 == x2 and grad_output = L R^T."""
    rng = np.random.default_rng(0)
    n, J, T = 23, 5, 3
    Z = rng.standard_normal((n, J))
    L = rng.standard_normal((n, T))
    R = rng.standard_normal((n, T))
    W = L @ R.T
    g1, g2, _ = orc.gam_backward(Z, Z, np.ones(J), W)
    gZ, gs = orc.bilinear_grad(Z, L, R, 1.0)
    np.testing.assert_allclose(gZ, g1 + g2, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(gs, (W * orc.gam_forward(Z, Z, np.ones(J))).sum(), rtol=1e-12)


@pytest.mark.parametrize("prescale", [True, False])
def test_known_answer_manual_rescale(prescale):
    """test.py:533-573: x=[[1,2,3],[1.1,2.2,3.3]], P=I3, l=[1,2,3], AdditiveStructureKernel(RBF(l=1),3) (no 1/J):
    K == 3*RBF(x[:,0]) for both prescale and postscale."""
    x = np.array([[1., 2., 3.], [1.1, 2.2, 3.3]])
    K = orc.kernel_matrix(x, x, np.eye(3), [1., 2., 3.], 1.0, prescale=prescale, weight=1.0)
    e = 3.0 * np.exp(-0.5 * 0.1 ** 2)
    np.testing.assert_allclose(K, [[3.0, e], [e, 3.0]], rtol=1e-7)


def test_known_answer_mem_efficient_gam():
    """test.py:625-635: MemoryEfficientGamKernel() (default l = ln 2) == sum of unit-outputscale 1-D RBFs with l = ln 2."""
    x = np.array([[1., 2., 3.], [1.1, 2.2, 3.3]])
    ln2 = np.log(2.0)
    K = orc.additive_rbf(x, x, weight=1.0, inner_lengthscale=ln2)
    expect = sum(np.exp(-0.5 * ((x[:, i:i + 1] - x[:, i:i + 1].T) / ln2) ** 2) for i in range(3))
    np.testing.assert_allclose(K, expect, atol=1e-12)
    np.testing.assert_allclose(K, orc.gam_forward(x, x, np.full(3, ln2)), atol=1e-12)


def test_gen_rp_and_space_equally_match_reference_golden():
    from rpgp_amd import rp
    g = np.load(os.path.join(GOLD, "gen_rp.npz"))
    for key in g.files:
        dist, d, k, s = key.rsplit("_", 3)
        torch.manual_seed(int(s[1:]))
        W = rp.gen_rp(int(d[1:]), int(k[1:]), dist)
        assert np.array_equal(W.numpy(), g[key]), key
    s = np.load(os.path.join(GOLD, "space_equally.npz"))
    for (J, d, seed) in [(20, 18, 0), (20, 8, 1), (3, 3, 2), (4, 2, 3)]:
        torch.manual_seed(seed)
        np.random.seed(seed)
        P0 = torch.cat([rp.gen_rp(d, 1, "gaussian") for _ in range(J)], dim=1).t().contiguous()
        out, loss = rp.space_equally(P0.clone(), 0.1, 5000 if d < J else 10)
        ref = s["J%d_d%d_s%d_out" % (J, d, seed)]
        np.testing.assert_allclose(out.numpy(), ref, atol=5e-5)
        np.testing.assert_allclose(np.linalg.norm(out.numpy(), axis=1), 1.0, atol=1e-5)   # test.py:465-468
        if d >= J:
            np.testing.assert_allclose(out.numpy() @ out.numpy().T, np.eye(J), atol=1e-6)
        else:
            assert float(loss) > 1e-3                                                       # test.py:484-491


def test_rp_generator_statistics():
    """test.py:52-109: sphere columns have equal norm; gaussian projections roughly preserve distances."""
    from rpgp_amd import rp
    torch.manual_seed(0)
    W = rp.gen_rp(50, 30, "sphere")
    norms = W.norm(dim=0)
    assert torch.allclose(norms, norms[0].expand_as(norms), atol=1e-5)
    x = torch.randn(20, 100)
    for dist in ["gaussian", "sphere", "bernoulli", "uniform"]:
        P = rp.gen_rp(100, 1000, dist)
        d0 = torch.cdist(x, x)
        d1 = torch.cdist(x @ P, x @ P)
        mask = ~torch.eye(20, dtype=torch.bool)
        assert ((d1 - d0).abs() / d0)[mask].mean() < 0.1


def test_cpu_path_port_matches_oracle():
    """The fp32 torch-CPU port timed as cpu_baseline computes the same MVM as the fp64 oracle."""
    torch.manual_seed(0)
    N, d, J, T = 700, 8, 20, 3
    X = torch.randn(N, d)
    P = torch.randn(d, J)
    ls = torch.full((d,), d ** 0.5)
    V = torch.randn(N, T)
    Z = cpu_path.project(X, P, ls)
    out = cpu_path.mvm(Z, V, 0.7, 0.1, row_chunk=256)
    Zo = orc.project(X.numpy(), P.numpy(), ls.numpy())
    ref = orc.mvm(Zo, Zo, V.numpy(), 0.7 / J, 0.1)
    assert np.linalg.norm(out.numpy() - ref) / np.linalg.norm(ref) < 1e-5


def test_smoothed_box_prior_constant_inside_box():
    """inside [1e-4, 10] the log-density is -log(sqrt(2 pi) 0.01 + 9.9999) ~ -2.3051."""
    v = orc.smoothed_box_log_prob(1.0)
    assert abs(v - (-np.log(np.sqrt(2 * np.pi) * 0.01 + 9.9999))) < 1e-12
    assert abs(v + 2.3051) < 1e-3
    assert orc.smoothed_box_log_prob(10.5) < v - 100


def test_ski_sparse_oracle_equals_dense_oracle():
    """The O(N)-memory SKI forms used at config-C5 size are the dense SKI oracle rewritten with sparse W."""
    from oracle import ski as sko
    rng = np.random.default_rng(3)
    N, J, G, T = 700, 3, 256, 4
    Z = rng.standard_normal((N, J))
    V = rng.standard_normal((N, T))
    w = rng.uniform(0.3, 1.2, size=J)
    grid = sko.grid_params(Z, None, G)
    for weights in (None, w):
        K = sko.dense_kernel(Z, Z, 0.7, G, grid, weights)
        np.testing.assert_allclose(sko.mvm_sparse(Z, Z, V, 0.7, G, grid, 0.2, weights), K @ V + 0.2 * V, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(sko.diag_sparse(Z, 0.7, G, grid, weights), np.diag(K), rtol=1e-11, atol=1e-12)
    Z1 = rng.standard_normal((90, J)) * 0.5
    Kr = sko.dense_kernel(Z1, Z, 0.7, G, grid)
    np.testing.assert_allclose(sko.mvm_sparse(Z1, Z, V, 0.7, G, grid), Kr @ V, rtol=1e-11, atol=1e-11)


def test_c_oracle_matches_reference_golden_and_numpy_oracle():
    """oracle/cmvm.c (the C/OpenMP restatement the BASELINE-size GPU parity tests call) reproduces the reference-generated
    GAMFunction golden matrices and the numpy oracle's products; the dense-GP oracle's switch to it does not change K."""
    from oracle import cmvm
    g = np.load(os.path.join(GOLD, "gam.npz"))
    for name in g["names"]:
        x1, x2, ls = g[name + "_x1"], g[name + "_x2"], np.asarray(g[name + "_ls"], dtype=np.float64).reshape(1, -1)
        np.testing.assert_allclose(cmvm.kernel(x1 / ls, x2 / ls), g[name + "_K"], rtol=1e-12, atol=1e-14)
    rng = np.random.default_rng(11)
    Z, Z2 = rng.standard_normal((1300, 20)), rng.standard_normal((411, 20))       # spans the 1024-column tile edge
    V = rng.standard_normal((1300, 5))
    np.testing.assert_allclose(cmvm.kernel(Z2, Z, 0.3), 0.3 * orc.additive_rbf(Z2, Z), rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(cmvm.mvm(Z, Z, V, 0.05, 0.1), orc.mvm(Z, Z, V, 0.05, 0.1), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(cmvm.mvm(Z2, Z, V[:, 0], 0.05), orc.mvm(Z2, Z, V[:, :1], 0.05)[:, 0], rtol=1e-12, atol=1e-12)
    X, P, ls = rng.standard_normal((600, 6)), rng.standard_normal((6, 7)), rng.uniform(1, 2, 6)
    big = orc.BIG_PAIR_TERMS
    try:
        K_np = orc.kernel_matrix(X, X, P, ls, 0.9, weight=1.0, inner_lengthscale=np.log(2.0))
        orc.BIG_PAIR_TERMS = 0.0
        K_c = orc.kernel_matrix(X, X, P, ls, 0.9, weight=1.0, inner_lengthscale=np.log(2.0))
    finally:
        orc.BIG_PAIR_TERMS = big
    np.testing.assert_allclose(K_c, K_np, rtol=1e-13, atol=1e-14)
    # the C form of the derivative's d/dZ (tests/test_parity_gpu.py at the C2 / C3 sizes) against the reference-pinned GAM
    # backward: x1 == x2, grad_output = L R^T (memory_efficient_gam_kernel.py:53-58), and against oracle.bilinear_grad
    n, J, T = 137, 7, 4
    Zs, L, R = rng.standard_normal((n, J)), rng.standard_normal((n, T)), rng.standard_normal((n, T))
    W = L @ R.T
    g1, g2, _ = orc.gam_backward(Zs, Zs, np.ones(J), W)
    gz_c = cmvm.bilinear_gz(Zs, W + W.T, 1.0)
    np.testing.assert_allclose(gz_c, g1 + g2, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(cmvm.bilinear_gz(Zs, W + W.T, 0.37), orc.bilinear_grad(Zs, L, R, 0.37)[0], rtol=1e-10, atol=1e-12)
