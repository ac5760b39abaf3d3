"""ORACLE — test infrastructure only (never imported by the product package).

Float64 dense restatement of the generalised additive family behind the same operator:

    K[i,i'] = scale * sum_c w_c * phi_kind(group c of the columns of Z)

  RBF       exp(-r^2 / 2), r^2 summed over the group's columns: gpytorch RBFKernel on `active_dims` of size k
            (training_routines.py:148-156,172-174) and the ProductKernel groups of 1-D RBFs of
            polynomial_projection_kernels.py:70-86 (the two coincide for the RBF)
  Matern    nu = 1.5 (training_routines.py:57-63):  (1 + sqrt3 r) exp(-sqrt3 r)        [GPyTorch MaternKernel formula]
  InverseMQ (r^2 + 1)^(-1/2)                                     (gp_models/kernels/imq_kernel.py:8-9,18-22)
  Cosine    cos(pi r / p) with period p = 1 (training_routines.py:151-152)              [GPyTorch CosineKernel formula]
with Z already divided by the lengthscales.  PARITY UNPINNED for Matern / Cosine (their arithmetic lives in GPyTorch,
which is absent here; the formulas are GPyTorch's documented ones); the InverseMQ formula is the reference's own
source; the RBF members are pinned through oracle/dense_gp.py (same function for group = 1, weights 1/J).
"""
import numpy as np

KINDS = ("RBF", "Matern", "InverseMQ", "Cosine")


def _phi(kind, d2):
    """Stationary function of the squared distance d2 over the group's columns (1-D, or radial for a group of k columns)."""
    if kind == "RBF":
        return np.exp(-0.5 * d2)
    r = np.sqrt(d2)
    if kind == "Matern":
        return (1.0 + np.sqrt(3.0) * r) * np.exp(-np.sqrt(3.0) * r)
    if kind == "InverseMQ":
        return 1.0 / np.sqrt(d2 + 1.0)
    if kind == "Cosine":
        return np.cos(np.pi * r)
    raise ValueError("Unknown kernel type")


def component_matrices(Z1, Z2, kind, group, product=False):
    """[ncomp] list of M x N float64 matrices phi_c(Z1, Z2)."""
    Z1 = np.asarray(Z1, dtype=np.float64)
    Z2 = np.asarray(Z2, dtype=np.float64)
    # group > 1 with a non-RBF kind: the RADIAL k-dimensional kernel of `additive_rp` (training_routines.py:172-174:
    # kernel(active_dims = the group's columns)), or — `product` — the ProductKernel of the group's 1-D sub-kernels that the
    # rp_poly kinds build (polynomial_projection_kernels.py:70-86)
    ncomp = Z1.shape[1] // group
    out = []
    for c in range(ncomp):
        d2 = np.zeros((Z1.shape[0], Z2.shape[0]))
        pr = np.ones((Z1.shape[0], Z2.shape[0]))
        for m in range(group):
            j = c * group + m
            d = Z1[:, j:j + 1] - Z2[:, j:j + 1].T
            d2 += d * d
            if product:
                pr *= _phi(kind, d * d)
        out.append(pr if product else _phi(kind, d2))
    return out


def _dphi(kind, d2, phi):
    """d phi / d(d2)."""
    if kind == "RBF":
        return -0.5 * phi
    if kind == "Matern":
        return -1.5 * np.exp(-np.sqrt(3.0) * np.sqrt(d2))
    if kind == "InverseMQ":
        return -0.5 * (d2 + 1.0) ** -1.5
    # Cosine: d cos(pi r)/d(d2) = -pi sin(pi r) / (2 r)  -> pi^2/2 * sinc
    return -0.5 * np.pi ** 2 * np.sinc(np.sqrt(d2))          # np.sinc(r) = sin(pi r) / (pi r)


def kernel_matrix(Z1, Z2, kind, group, weights, scale=1.0, product=False):
    w = np.asarray(weights, dtype=np.float64).reshape(-1)
    comps = component_matrices(Z1, Z2, kind, group, product)
    K = np.zeros_like(comps[0])
    for c, Kc in enumerate(comps):
        K += w[c] * Kc
    return scale * K


def mvm(Z1, Z2, V, kind, group, weights, scale=1.0, noise=0.0, product=False):
    V = np.asarray(V, dtype=np.float64)
    out = kernel_matrix(Z1, Z2, kind, group, weights, scale, product) @ V
    if noise:
        out = out + noise * V
    return out


def bilinear_grad_dense(Z, S, kind, group, weights, scale=1.0, product=False):
    """d/dZ and the unweighted per-component sums of 0.5 * sum(S * K(Z,Z)) for a symmetric S (analytic, float64):
    returns (gZ [N x cols], gcomp [ncomp]) with gcomp[c] = 0.5 * sum(S * phi_c)."""
    Z = np.asarray(Z, dtype=np.float64)
    S = np.asarray(S, dtype=np.float64)
    w = np.asarray(weights, dtype=np.float64).reshape(-1)
    ncomp = Z.shape[1] // group
    gZ = np.zeros_like(Z)
    gc = np.zeros(ncomp)
    for c in range(ncomp):
        diffs = [Z[:, c * group + m:c * group + m + 1] - Z[:, c * group + m:c * group + m + 1].T for m in range(group)]
        if product:
            phis = [_phi(kind, d * d) for d in diffs]
            phi = np.ones_like(phis[0])
            for f in phis:
                phi = phi * f
            gc[c] = 0.5 * (S * phi).sum()
            for m in range(group):
                others = np.ones_like(phi)                   # product of the OTHER factors (no division: the cosine has roots)
                for q in range(group):
                    if q != m:
                        others = others * phis[q]
                dphi = _dphi(kind, diffs[m] * diffs[m], phis[m])
                gZ[:, c * group + m] = scale * w[c] * (S * others * dphi * 2.0 * diffs[m]).sum(axis=1)
            continue
        d2 = sum(d * d for d in diffs)
        phi = _phi(kind, d2)
        gc[c] = 0.5 * (S * phi).sum()
        dphi = _dphi(kind, d2, phi)
        for m in range(group):
            # sum_i' S_ii' dK_ii'/dz_i = sum_i' S_ii' w_c dphi * 2 (z_i - z_i')
            gZ[:, c * group + m] = scale * w[c] * (S * dphi * 2.0 * diffs[m]).sum(axis=1)
    return gZ, gc


def bilinear_grad(Z, L, R, kind, group, weights, scale=1.0, product=False):
    """Same for S = L R^T + R L^T given as N x T factors (the derivative of sum((L R^T) * K))."""
    L = np.asarray(L, dtype=np.float64).reshape(Z.shape[0], -1)
    R = np.asarray(R, dtype=np.float64).reshape(Z.shape[0], -1)
    S = L @ R.T + R @ L.T
    return bilinear_grad_dense(Z, S, kind, group, weights, scale, product)
