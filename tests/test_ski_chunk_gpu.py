"""The chunked form of the planned SKI product (round 5, csrc/rpgp_ski.hip "chunked product"; opt-in): against the float64
sparse-W oracle, against the cell-sorted form it replaces (rpgp_ski_chunk_mode switches between them on ONE plan), bitwise
reproducibility, and the edge shapes of its tables (last chunk short, all rows in one cell, windows that cover the grid)."""
import numpy as np
import pytest
import torch

from oracle import ski as sko

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture
def chunk_mode():
    from rpgp_amd import ops
    prev = ops.ski_chunk_mode()
    yield ops.ski_chunk_mode
    ops.ski_chunk_mode(prev)


def _problem(N, J, T, seed, spread=1.0, ordered=False):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(N, J, generator=g) * spread
    if ordered:
        from rpgp_amd.training import locality_order
        X = X[locality_order(X)]
    Q = torch.linalg.qr(torch.randn(J, J, generator=g))[0] if J > 1 else torch.ones(1, 1)
    return (X @ Q).contiguous(), torch.randn(N, T, generator=g)


@pytest.mark.parametrize("N,J,T,G,spread,ordered", [
    (20000, 3, 11, 1024, 1.0, True),      # the flagship layout: three column groups per row, compact windows
    (20000, 3, 11, 1024, 1.0, False),     # file order: every window covers most of the grid
    (50001, 1, 1, 256, 1.0, False),       # one projection, one column (one lane per cell), a short last chunk
    (33000, 4, 7, 512, 1.0, True),        # two column groups, four projections
    (17000, 2, 12, 2048, 1.0, True),      # the largest grid, a full last column group
    (16400, 3, 5, 64, 1e-3, False),       # nearly all rows in a handful of cells (hundreds of rows per cell and chunk)
    (70000, 3, 11, 1024, 0.0, False),     # constant columns: every row in ONE cell
    (391386, 3, 11, 1024, 1.0, True),     # config C5 in the row order train_exact_gp stores
])
def test_chunked_product_matches_oracle_and_cell_sorted_form(gpu_device, chunk_mode, N, J, T, G, spread, ordered):
    from rpgp_amd import ops
    Z, V = _problem(N, J, T, N + G + T, spread, ordered)
    Zt, Vt = Z.to(gpu_device), V.to(gpu_device)
    gp = ops.ski_grid(Zt, None, G)
    gph = gp.double().cpu().numpy()
    grid = (float(gph[0]), float(gph[1]))
    scale, noise = 0.8 / J, 0.25
    chunk_mode(True)
    plan = ops.SkiPlan(Zt, gp, G)
    assert plan.ok and plan.chunked
    out = ops.ski_mvm(Zt, Zt, gp, Vt, scale, noise, G, plan=plan)
    hist = ops.ski_scatter(Zt, gp, Vt, G, plan=plan)
    H = ops.ski_grid_product(hist, gp, G)
    gat = ops.ski_gather(Zt, gp, H, Vt, scale, noise, G, plan=plan)
    assert torch.equal(out, gat)                                  # the staged entry points are the same kernels
    assert torch.equal(out, ops.ski_mvm(Zt, Zt, gp, Vt, scale, noise, G, plan=plan))          # same plan: same bits
    assert torch.equal(out, ops.ski_mvm(Zt, Zt, gp, Vt, scale, noise, G, plan=ops.SkiPlan(Zt, gp, G)))   # rebuilt plan
    chunk_mode(False)
    old = ops.ski_mvm(Zt, Zt, gp, Vt, scale, noise, G, plan=plan)
    hist_old = ops.ski_scatter(Zt, gp, Vt, G, plan=plan)
    gat_old = ops.ski_gather(Zt, gp, H, Vt, scale, noise, G, plan=plan)
    chunk_mode(True)
    # the gather forms every product the same way: identical bits on the same H; the scatter adds in another order
    assert torch.equal(gat, gat_old)
    assert float((hist - hist_old).abs().max()) < 2e-5 * float(hist_old.abs().max() + 1e-30)
    assert _rel(out.cpu().numpy(), old.cpu().numpy()) < 2e-6
    ref = sko.mvm_sparse(Z.numpy(), Z.numpy(), V.numpy(), scale, G, grid, noise)
    col = np.linalg.norm(out.double().cpu().numpy() - ref, axis=0) / np.linalg.norm(ref, axis=0)
    assert col.max() < 2e-5, col


def test_chunked_product_propagates_non_finite_rhs_and_serves_the_derivative_scatter(gpu_device, chunk_mode):
    from rpgp_amd import ops
    N, J, T, G = 40000, 3, 11, 1024
    Z, V = _problem(N, J, T, 3, ordered=True)
    Zt, Vt = Z.to(gpu_device), V.to(gpu_device)
    gp = ops.ski_grid(Zt, None, G)
    chunk_mode(True)
    plan = ops.SkiPlan(Zt, gp, G)
    assert plan.chunked
    chunk_mode(False)
    assert not ops.SkiPlan(Zt, gp, G).chunked                    # the default: no chunk tables, the cell-sorted product
    chunk_mode(True)
    Vb = Vt.clone()
    Vb[12345, 2] = float("nan")
    bad = ops.ski_mvm(Zt, Zt, gp, Vb, 0.3, 0.1, G, plan=plan)
    assert not torch.isfinite(bad[:, 2]).all()
    assert torch.isfinite(bad[:, :2]).all() and torch.isfinite(bad[:, 3:]).all()
    # the operator's derivative scatters [L | R] through the same plan (two column blocks of one histogram)
    L, R = Vt, torch.flip(Vt, dims=[0]).contiguous()
    h1 = ops.ski_bilinear_scatter(Zt, gp, L, R, G, plan=plan)
    chunk_mode(False)
    h0 = ops.ski_bilinear_scatter(Zt, gp, L, R, G, plan=plan)
    chunk_mode(True)
    assert h1.shape == (J, G, 2 * T)
    assert float((h1 - h0).abs().max()) < 2e-5 * float(h0.abs().max())
    assert torch.equal(h1, ops.ski_bilinear_scatter(Zt, gp, L, R, G, plan=plan))
