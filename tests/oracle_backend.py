"""TEST DOUBLE for rpgp_amd.backend: routes the backend protocol through the float64 oracle on CPU so that host logic
(CG, SLQ, preconditioner, autograd plumbing, sharding, training loop, runner) can be tested without a GPU.
Lives under tests/ on purpose: the product package never imports it."""
import numpy as np
import torch

from oracle import dense_gp as orc
from oracle import ski as sko
from oracle import family as fmo


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

    # pair-sharding emulation (distributed.JShard mode "pairs"): rank r of `world` owns the unordered pairs {i, i'} with
    # tile index ((min // 8) + (max // 8)) % world == r -- a symmetric 0/1 mask per rank, masks sum to all-ones -- and
    # evaluates ALL projections on them.  (The HIP backend splits the workgroup range of its tile decomposition instead;
    # what the host logic relies on is only "partials over ranks sum to K v, noise handed to one rank".)
    supports_pair_shard = True

    def mvm_sym(self, Z, V, scale, noise=0.0, j0=0, j1=None, out=None, shard=None):
        self.calls["mvm_sym"] += 1
        z = _np(Z)[:, j0:j1]
        squeeze = V.dim() == 1
        v = _np(V).reshape(z.shape[0], -1)
        if shard is not None and shard[0] > 1:
            world, rank = shard
            self.calls["pair_shard"] = self.calls.get("pair_shard", 0) + 1
            t = np.arange(z.shape[0]) // 8
            mask = ((t[:, None] + t[None, :]) % world) == rank
            r = _t((scale * orc.additive_rbf(z, z) * mask) @ v + noise * v, V)
        else:
            r = _t(orc.mvm(z, z, v, scale, noise), V)
        return r.squeeze(1) if squeeze else r

    # packed symmetric cache emulation: the (masked, for a pair shard) unscaled kernel matrix kept dense
    supports_symcache = True

    class _SymCache:
        pass

    def symcache(self, Z, j0=0, j1=None, shard=None, wide=False):
        self.calls["symcache"] = self.calls.get("symcache", 0) + 1
        z = _np(Z)[:, j0:j1]
        c = OracleBackend._SymCache()
        c.N = z.shape[0]
        c.world, c.rank = (1, 0) if shard is None else (int(shard[0]), int(shard[1]))
        K = orc.additive_rbf(z, z)
        if c.world > 1:
            t = np.arange(c.N) // 8
            K = K * (((t[:, None] + t[None, :]) % c.world) == c.rank)
        c.K, c.wide, c.nbytes, c.device = K, bool(wide), K.size * 2, Z.device
        return c

    def symcache_mvm(self, cache, V, scale, noise=0.0):
        self.calls["symcache_mvm"] = self.calls.get("symcache_mvm", 0) + 1
        squeeze = V.dim() == 1
        v = _np(V).reshape(cache.N, -1)
        r = _t(scale * (cache.K @ v) + noise * v, V)
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

    # ---- generalised family ----------------------------------------------------------------------------------
    class _Fam:
        def __init__(self, kind, group, weights, product=False):
            self.kind, self.group, self.w = kind, int(group), _np(weights).reshape(-1)
            self.ncomp = self.w.size
            self.product = bool(product) and self.group > 1 and kind != "RBF"

    def make_family(self, kind, group, weights, product=False):
        return OracleBackend._Fam(kind, group, weights, product)

    def family_mvm_sym(self, fam, Z, V, scale, noise=0.0):
        squeeze = V.dim() == 1
        v = _np(V).reshape(Z.shape[0], -1)
        r = _t(fmo.mvm(_np(Z), _np(Z), v, fam.kind, fam.group, fam.w, scale, noise, fam.product), V)
        return r.squeeze(1) if squeeze else r

    def family_mvm_rect(self, fam, Z1, Z2, V, scale):
        squeeze = V.dim() == 1
        v = _np(V).reshape(Z2.shape[0], -1)
        r = _t(fmo.mvm(_np(Z1), _np(Z2), v, fam.kind, fam.group, fam.w, scale, 0.0, fam.product), V)
        return r.squeeze(1) if squeeze else r

    def family_dense(self, fam, Z1, Z2, scale):
        return _t(fmo.kernel_matrix(_np(Z1), _np(Z2), fam.kind, fam.group, fam.w, scale, fam.product), Z1)

    def family_bilinear_grad(self, fam, Z, L, R, scale):
        g, gc = fmo.bilinear_grad(_np(Z), _np(L), _np(R), fam.kind, fam.group, fam.w, scale, fam.product)
        return _t(g, Z), _t(gc, Z)

    def family_bilinear_grad_dense(self, fam, Z, S, scale):
        g, gc = fmo.bilinear_grad_dense(_np(Z), _np(S), fam.kind, fam.group, fam.w, scale, fam.product)
        return _t(g, Z), _t(gc, Z)

    # ---- Woodbury preconditioner pieces (rpgp_gram_f64 / rpgp_woodbury_apply) ---------------------------------
    def gram_f64(self, A, B):
        return A.double().t() @ B.double()

    def woodbury_setup(self, gram, noise):
        C = gram.double().clone()
        C.diagonal().add_(float(noise))
        chol = torch.linalg.cholesky(C)
        return chol, torch.cholesky_inverse(chol), (2.0 * torch.log(chol.diagonal()).sum()).reshape(1)

    def woodbury_apply(self, L, R, Tm, noise):
        return ((R.double() - L.double() @ Tm) / float(noise)).to(torch.float32)

    # ---- SKI path -------------------------------------------------------------------------------------------
    _KINDS = ("RBF", "Matern", "InverseMQ", "Cosine")

    def _kind(self, gp):
        return self._KINDS[(int(gp[3]) >> 2) & 3]

    def ski_grid(self, Z1, Z2=None, grid_size=1024, weights=None, rule="shared", kind="RBF"):
        kflag = 4.0 * self._KINDS.index(kind)
        J = Z1.shape[1]
        if rule == "reference":          # per-projection grids: [., ., ., flags, w_0..w_{J-1}, (g0_j, h_j, 1/h_j) x J]
            g0, h = sko.grid_params_reference(_np(Z1), None if Z2 is None else _np(Z2), grid_size)
            head = [0.0, 1.0, 1.0, (2.0 if weights is None else 3.0) + kflag]
            w = [1.0] * J if weights is None else [float(x) for x in weights.detach().reshape(-1)]
            tail = [v for j in range(J) for v in (g0[j], h[j], 1.0 / h[j])]
            return torch.tensor(head + w + tail, dtype=torch.float64)
        g0, h = sko.grid_params(_np(Z1), None if Z2 is None else _np(Z2), grid_size)
        head = [g0, h, 1.0 / h, (0.0 if weights is None else 1.0) + kflag]
        tail = [] if weights is None else [float(x) for x in weights.detach().reshape(-1)]
        return torch.tensor(head + tail, dtype=Z1.dtype)

    def _grid(self, gp):
        g = gp.double()
        if int(g[3]) & 2:
            J = (g.numel() - 4) // 4
            arr = g[4 + J:].reshape(J, 3).numpy()
            return arr[:, 0].copy(), arr[:, 1].copy()
        return float(g[0]), float(g[1])

    def _w(self, gp):
        flags = int(gp[3])
        if not flags & 1:
            return None
        if flags & 2:
            J = (gp.numel() - 4) // 4
            return gp[4:4 + J].double().numpy()
        return gp[4:].double().numpy()

    def ski_mvm(self, Z1, Z2, gp, V, scale, noise=0.0, grid_size=1024):
        squeeze = V.dim() == 1
        v = _np(V).reshape(Z2.shape[0], -1)
        K = sko.dense_kernel(_np(Z1), _np(Z2), scale, grid_size, self._grid(gp), self._w(gp), self._kind(gp))
        out = K @ v
        if noise:
            out = out + noise * v
        r = _t(out, V)
        return r.squeeze(1) if squeeze else r

    def ski_dense(self, Z1, Z2, gp, scale, grid_size=1024):
        return _t(sko.dense_kernel(_np(Z1), _np(Z2), scale, grid_size, self._grid(gp), self._w(gp), self._kind(gp)), Z1)

    # ---- staged SKI (row-sharded operator): scatter -> all-reduce -> grid product -> gather ----------------------
    def ski_grid_from_range(self, zmin, zmax, grid_size, device, weights=None, kind="RBF", dtype=None):
        rng = max(float(zmax) - float(zmin), 1e-12)
        h = rng / (grid_size - 5)
        head = [float(zmin) - 2.0 * h, h, 1.0 / h, (0.0 if weights is None else 1.0) + 4.0 * self._KINDS.index(kind)]
        tail = [] if weights is None else [float(x) for x in weights.detach().reshape(-1)]
        return torch.tensor(head + tail, dtype=torch.float64)

    def ski_scatter(self, Z, gp, V, grid_size=1024):
        z, v = _np(Z), _np(V).reshape(Z.shape[0], -1)
        grid = self._grid(gp)
        hist = np.stack([sko.interp_sparse(z[:, j], *sko._grid_j(grid, j), grid_size).T @ v for j in range(z.shape[1])])
        return torch.from_numpy(hist)

    def ski_grid_product(self, hist, gp, grid_size=1024):
        grid = self._grid(gp)
        w = self._w(gp)
        H = np.stack([(1.0 if w is None else w[j]) * (sko.toeplitz(sko._grid_j(grid, j)[1], grid_size, self._kind(gp)) @ hist[j].double().numpy())
                      for j in range(hist.shape[0])])
        return torch.from_numpy(H)

    def ski_gather(self, Z, gp, H, V, scale, noise=0.0, grid_size=1024):
        z = _np(Z)
        grid = self._grid(gp)
        out = np.zeros((z.shape[0], H.shape[2]))
        for j in range(z.shape[1]):
            out += sko.interp_sparse(z[:, j], *sko._grid_j(grid, j), grid_size) @ H[j].double().numpy()
        out *= scale
        if noise:
            out += noise * _np(V).reshape(z.shape[0], -1)
        return _t(out, Z)

    def ski_diag(self, Z, gp, scale, grid_size=1024):
        return _t(np.diag(sko.dense_kernel(_np(Z), _np(Z), scale, grid_size, self._grid(gp), self._w(gp), self._kind(gp))).copy(), Z)

    def ski_bilinear_grad(self, Z, gp, L, R, scale, grid_size=1024):
        """Analytic derivative of sum((L R^T) * K_ski) in float64 (same formulas as the HIP kernel, dense)."""
        z = _np(Z)
        Ld, Rd = _np(L).reshape(z.shape[0], -1), _np(R).reshape(z.shape[0], -1)
        grid = self._grid(gp)
        G = grid_size
        gZ = np.zeros_like(z)
        gs = 0.0
        wts = self._w(gp)
        self._last_comp = np.zeros(z.shape[1])
        for j in range(z.shape[1]):
            g0, h = sko._grid_j(grid, j)
            Tm = sko.toeplitz(h, G, self._kind(gp))
            wj = 1.0 if wts is None else wts[j]
            u = np.clip((z[:, j] - g0) / h, 1.0, G - 2.0)
            fl = np.floor(u)
            fr = u - fl
            idx0 = np.clip(fl.astype(np.int64) - 1, 0, G - 4)
            W = sko.interp_matrix(z[:, j], g0, h, G)
            HR = Tm @ (W.T @ Rd)
            HL = Tm @ (W.T @ Ld)
            self._last_comp[j] = wj * (Ld * (W @ HR)).sum()
            gs += self._last_comp[j]
            s = [fr + 1.0, fr, 1.0 - fr, 2.0 - fr]
            sign = [1.0, 1.0, -1.0, -1.0]
            for k in range(4):
                U = s[k]
                d = np.where(U < 1.0, (4.5 * U - 5.0) * U, (-1.5 * U + 5.0) * U - 4.0) * sign[k] / h
                rows = idx0 + k
                gZ[:, j] += wj * scale * d * ((Ld * HR[rows]).sum(1) + (Rd * HL[rows]).sum(1))
        return _t(gZ, Z), _t(np.array(gs), Z)

    def ski_bilinear_grad_comp(self, Z, gp, L, R, scale, grid_size=1024):
        gZ, gs = self.ski_bilinear_grad(Z, gp, L, R, scale, grid_size)
        return gZ, gs, _t(self._last_comp.copy(), Z)

    # ---- staged derivative of the row-sharded SKI operator: scatter -> all-reduce -> finish ---------------------------
    def ski_bilinear_scatter(self, Z, gp, L, R, grid_size=1024, plan=None):
        z = _np(Z)
        Ld, Rd = _np(L).reshape(z.shape[0], -1), _np(R).reshape(z.shape[0], -1)
        grid = self._grid(gp)
        hist = np.zeros((z.shape[1], grid_size, 2 * Ld.shape[1]))
        for j in range(z.shape[1]):
            if z.shape[0]:
                W = sko.interp_sparse(z[:, j], *sko._grid_j(grid, j), grid_size)
                hist[j] = np.concatenate([W.T @ Ld, W.T @ Rd], axis=1)
        return torch.from_numpy(hist)

    def ski_bilinear_finish(self, Z, gp, hist2, L, R, scale, grid_size=1024, comp=False):
        z = _np(Z)
        Ld, Rd = _np(L).reshape(z.shape[0], -1), _np(R).reshape(z.shape[0], -1)
        T = Ld.shape[1]
        grid = self._grid(gp)
        G = grid_size
        wts = self._w(gp)
        gZ = np.zeros_like(z)
        gs, gc = 0.0, np.zeros(z.shape[1])
        hh = hist2.double().numpy()
        for j in range(z.shape[1]):
            g0, h = sko._grid_j(grid, j)
            Tm = sko.toeplitz(h, G, self._kind(gp))
            wj = 1.0 if wts is None else wts[j]
            HL, HR = Tm @ hh[j][:, :T], Tm @ hh[j][:, T:]
            if not z.shape[0]:
                continue
            u = np.clip((z[:, j] - g0) / h, 1.0, G - 2.0)
            fl = np.floor(u)
            fr = u - fl
            idx0 = np.clip(fl.astype(np.int64) - 1, 0, G - 4)
            W = sko.interp_sparse(z[:, j], g0, h, G)
            gc[j] = wj * (Ld * (W @ HR)).sum()
            gs += gc[j]
            s = [fr + 1.0, fr, 1.0 - fr, 2.0 - fr]
            sign = [1.0, 1.0, -1.0, -1.0]
            for k in range(4):
                U = s[k]
                d = np.where(U < 1.0, (4.5 * U - 5.0) * U, (-1.5 * U + 5.0) * U - 4.0) * sign[k] / h
                rows = idx0 + k
                gZ[:, j] += wj * scale * d * ((Ld * HR[rows]).sum(1) + (Rd * HL[rows]).sum(1))
        return _t(gZ, Z), _t(np.array(gs), Z), (_t(gc, Z) if comp else None)
