"""TEST DOUBLE for rpgp_amd.backend: routes the backend protocol through the float64 oracle on CPU so that host logic
(CG, SLQ, preconditioner, autograd plumbing, sharding, training loop, runner) can be tested without a GPU.
Lives under tests/ on purpose: the product package never imports it."""
import numpy as np
import torch

from oracle import dense_gp as orc


def _np(t):
    return t.detach().cpu().double().numpy()


def _t(a, like):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype=like.dtype, device=like.device)


class OracleBackend:
    name = "oracle-cpu (tests only)"

    def __init__(self):
        self.calls = {"mvm_sym": 0, "mvm_rect": 0, "bilinear_grad": 0, "dense": 0}

    def project(self, X, Peff):
        return _t(_np(X) @ _np(Peff), X)

    def project_grad(self, X, G):
        return _t(_np(X).T @ _np(G), X)

    def mvm_sym(self, Z, V, scale, noise=0.0, j0=0, j1=None, out=None):
        self.calls["mvm_sym"] += 1
        z = _np(Z)[:, j0:j1]
        squeeze = V.dim() == 1
        v = _np(V).reshape(z.shape[0], -1)
        r = _t(orc.mvm(z, z, v, scale, noise), V)
        return r.squeeze(1) if squeeze else r

    def mvm_rect(self, Z1, Z2, V, scale, j0=0, j1=None):
        self.calls["mvm_rect"] += 1
        squeeze = V.dim() == 1
        v = _np(V).reshape(Z2.shape[0], -1)
        r = _t(orc.mvm(_np(Z1)[:, j0:j1], _np(Z2)[:, j0:j1], v, scale), V)
        return r.squeeze(1) if squeeze else r

    def dense(self, Z1, Z2, scale, j0=0, j1=None):
        self.calls["dense"] += 1
        return _t(scale * orc.additive_rbf(_np(Z1)[:, j0:j1], _np(Z2)[:, j0:j1]), Z1)

    def bilinear_grad(self, Z, L, R, scale, j0=0, j1=None):
        self.calls["bilinear_grad"] += 1
        z = _np(Z)
        j1 = z.shape[1] if j1 is None else j1
        g, gs = orc.bilinear_grad(z[:, j0:j1], _np(L), _np(R), scale)
        full = np.zeros_like(z)
        full[:, j0:j1] = g
        return _t(full, Z), _t(np.array(gs), Z)

    def bilinear_grad_dense(self, Z, S, scale, j0=0, j1=None):
        z = _np(Z)
        j1 = z.shape[1] if j1 is None else j1
        s = _np(S)
        g = np.zeros_like(z)
        ks = np.zeros((z.shape[0], z.shape[0]))
        for j in range(j0, j1):
            d = z[:, j:j + 1] - z[:, j:j + 1].T
            e = np.exp(-0.5 * d * d)
            ks += e
            g[:, j] = -scale * (s * e * d).sum(axis=1)
        return _t(g, Z), _t(np.array(0.5 * (s * ks).sum()), Z)

    def dense_mvm(self, Kd, V, noise=0.0):
        squeeze = V.dim() == 1
        v = _np(V).reshape(Kd.shape[0], -1)
        r = _t(_np(Kd) @ v + noise * v, V)
        return r.squeeze(1) if squeeze else r
