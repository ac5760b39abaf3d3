"""ORACLE — test infrastructure only.  Float64 dense restatement of the SKI path the reference enables with
`ski=True` (training_routines.py:157-158: `GridInterpolationKernel(kernel, **ski_options)` around each 1-D sub-kernel;
model_specs/additive_spread_prescale_Jd_ski.json: grid_size 1024, num_dims 1) — SURVEY.md Appendix E.

K ~= scale * sum_j W_j Tm W_j^T with cubic-convolution interpolation (Keys 1981, the 4-tap kernel GPyTorch's
`Interpolation` uses: u<=1: ((1.5u-2.5)u)u+1 ; 1<u<=2: ((-0.5u+2.5)u-4)u+2) onto one regular grid shared by all
projections, and Tm[m,m'] = k1((m-m')h) for the wrapped 1-D sub-kernel k1 — the RBF exp(-0.5 d^2) by default, or whatever
`_map_to_kernel` returned (training_routines.py:47-88 with :157-158): `kind` in oracle.family.KINDS, the radial forms of
oracle/family.py (`_phi`).  Two grid rules: this build's shared grid (`grid_params`: h = (max-min)/(G-5),
g0 = min - 2h over ALL projections, every stencil interior) and the reference's per-projection rule
(`grid_params_reference`, polynomial_projection_kernels.py:54-63); every function takes `grid` as (g0, h) floats or as
per-projection arrays.  **Parity unpinned**: GPyTorch is not installable and the reference has no SKI tests
(SURVEY.md §4); this oracle pins the HIP kernels to the stated math and to the exact kernel (interpolation error)."""
import numpy as np


def grid_params(Z1, Z2=None, G=1024):
    z = np.asarray(Z1, dtype=np.float64).ravel()
    if Z2 is not None:
        z = np.concatenate([z, np.asarray(Z2, dtype=np.float64).ravel()])
    mn, mx = z.min(), z.max()
    rng = max(mx - mn, 1e-12)
    h = rng / (G - 5)
    return mn - 2.0 * h, h


def grid_params_reference(Z1, Z2=None, G=1024):
    """The REFERENCE's grid rule (gp_models/kernels/polynomial_projection_kernels.py:54-63): per projection
    spacing = (max - min) / (G - 4), bounds = [min - 2.01 spacing, max + 2.01 spacing]; the G grid points span the bounds.
    Returns (g0 [J], h [J]).  With fixed projections the reference's static bounds (computed once from X at construction)
    scale with 1 / lengthscale_j exactly like the current coordinates, so evaluating the rule on the current Z gives the
    same grid.  What GPyTorch's GridInterpolationKernel does with explicit bounds beyond this is not restated
    (**unpinned**: GPyTorch is not installable here)."""
    z = np.asarray(Z1, dtype=np.float64)
    if Z2 is not None:
        z = np.concatenate([z, np.asarray(Z2, dtype=np.float64)], axis=0)
    mn, mx = z.min(axis=0), z.max(axis=0)
    rng = np.maximum(mx - mn, 1e-12)
    spacing = rng / (G - 4)
    b0, b1 = mn - 2.01 * spacing, mx + 2.01 * spacing
    return b0, (b1 - b0) / (G - 1)


def _grid_j(grid, j):
    """(g0, h) of projection j from a shared grid (two floats) or per-projection arrays."""
    g0, h = grid
    if np.ndim(g0) == 0:
        return float(g0), float(h)
    return float(np.asarray(g0)[j]), float(np.asarray(h)[j])


def _cubic(U):
    U = np.abs(U)
    return np.where(U < 1.0, ((1.5 * U - 2.5) * U) * U + 1.0, ((-0.5 * U + 2.5) * U - 4.0) * U + 2.0)


def interp_matrix(z, g0, h, G):
    """Dense N x G interpolation matrix of one projection (4 non-zeros per row)."""
    z = np.asarray(z, dtype=np.float64)
    u = np.clip((z - g0) / h, 1.0, G - 2.0)
    fl = np.floor(u)
    fr = u - fl
    idx0 = np.clip(fl.astype(np.int64) - 1, 0, G - 4)
    W = np.zeros((z.shape[0], G))
    rows = np.arange(z.shape[0])
    for k, s in enumerate([fr + 1.0, fr, 1.0 - fr, 2.0 - fr]):
        W[rows, idx0 + k] += _cubic(s)
    return W


def toeplitz(h, G, kind="RBF"):
    from .family import _phi
    m = np.arange(G)
    d = (m[:, None] - m[None, :]) * h
    return _phi(kind, d * d)


def dense_kernel(Z1, Z2, scale, G=1024, grid=None, weights=None, kind="RBF"):
    """scale * sum_j w_j W_j(Z1) Tm W_j(Z2)^T; `weights` = per-projection output scales (the `weighted` components of
    polynomial_projection_kernels.py:88-98), default all one."""
    Z1 = np.asarray(Z1, dtype=np.float64)
    Z2 = np.asarray(Z2, dtype=np.float64)
    grid = grid if grid is not None else grid_params(Z1, None if Z2 is Z1 else Z2, G)
    K = np.zeros((Z1.shape[0], Z2.shape[0]))
    w = np.ones(Z1.shape[1]) if weights is None else np.asarray(weights, dtype=np.float64).reshape(-1)
    for j in range(Z1.shape[1]):
        g0, h = _grid_j(grid, j)
        K += w[j] * (interp_matrix(Z1[:, j], g0, h, G) @ toeplitz(h, G, kind) @ interp_matrix(Z2[:, j], g0, h, G).T)
    return scale * K


def bilinear_objective(Z, L, R, scale, G, grid, weights=None, kind="RBF"):
    """sum((L R^T) * K_ski(Z, Z)) with a FIXED grid (the grid is a buffer, not differentiated)."""
    K = dense_kernel(Z, Z, scale, G, grid, weights, kind)
    return float((np.asarray(L, dtype=np.float64) @ np.asarray(R, dtype=np.float64).T * K).sum())


# ---- O(N)-memory forms for full-size problems (config C5: N ~ 391k, J = d = 3, G = 1024) ------------------------
def interp_sparse(z, g0, h, G):
    """N x G interpolation matrix of one projection as scipy CSR (4 non-zeros per row); same stencil rule as
    `interp_matrix`."""
    import scipy.sparse as sp
    z = np.asarray(z, dtype=np.float64)
    n = z.shape[0]
    u = np.clip((z - g0) / h, 1.0, G - 2.0)
    fl = np.floor(u)
    fr = u - fl
    idx0 = np.clip(fl.astype(np.int64) - 1, 0, G - 4)
    vals = np.stack([_cubic(fr + 1.0), _cubic(fr), _cubic(1.0 - fr), _cubic(2.0 - fr)], axis=1)
    cols = idx0[:, None] + np.arange(4)[None, :]
    indptr = np.arange(0, 4 * n + 1, 4)
    return sp.csr_matrix((vals.ravel(), cols.ravel(), indptr), shape=(n, G))


def mvm_sparse(Z1, Z2, V, scale, G, grid, noise=0.0, weights=None, kind="RBF"):
    """scale * sum_j w_j W_j(Z1) (Tm (W_j(Z2)^T V)) + noise * V in float64 with sparse W and a dense G x G Toeplitz:
    O(N (J + T)) memory, the full-size counterpart of `dense_kernel(...) @ V`."""
    Z1 = np.asarray(Z1, dtype=np.float64)
    Z2 = np.asarray(Z2, dtype=np.float64)
    V = np.asarray(V, dtype=np.float64).reshape(Z2.shape[0], -1)
    w = np.ones(Z1.shape[1]) if weights is None else np.asarray(weights, dtype=np.float64).reshape(-1)
    out = np.zeros((Z1.shape[0], V.shape[1]))
    Tm, h_prev = None, None
    for j in range(Z1.shape[1]):
        g0, h = _grid_j(grid, j)
        if h != h_prev:
            Tm, h_prev = toeplitz(h, G, kind), h
        W2 = interp_sparse(Z2[:, j], g0, h, G)
        W1 = W2 if Z1 is Z2 else interp_sparse(Z1[:, j], g0, h, G)
        out += w[j] * (W1 @ (Tm @ (W2.T @ V)))
    out *= scale
    if noise:
        out += noise * V
    return out


def diag_sparse(Z, scale, G, grid, weights=None, kind="RBF"):
    """diag(scale * sum_j w_j W_j Tm W_j^T) in float64 with O(N) memory: per row the 4 x 4 quadratic form of its
    stencil weights with the Toeplitz lags 0..3."""
    Z = np.asarray(Z, dtype=np.float64)
    w = np.ones(Z.shape[1]) if weights is None else np.asarray(weights, dtype=np.float64).reshape(-1)
    d = np.zeros(Z.shape[0])
    for j in range(Z.shape[1]):
        g0, h = _grid_j(grid, j)
        from .family import _phi
        lag = _phi(kind, (np.arange(4) * h) ** 2)
        u = np.clip((Z[:, j] - g0) / h, 1.0, G - 2.0)
        fr = u - np.floor(u)
        vals = [_cubic(fr + 1.0), _cubic(fr), _cubic(1.0 - fr), _cubic(2.0 - fr)]
        q = np.zeros(Z.shape[0])
        for k in range(4):
            for l in range(4):
                q += vals[k] * vals[l] * lag[abs(k - l)]
        d += w[j] * q
    return scale * d
