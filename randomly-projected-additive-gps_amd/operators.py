"""LinearOperator-shaped classes for the additive randomly-projected kernel (drop-in boundary).

Method names follow GPyTorch's LazyTensor / `linear_operator.LinearOperator` protocol (`_matmul`, `_size`,
`_transpose_nonbatch`, `_diagonal`, `_bilinear_derivative`/`_quad_form_derivative`, `to_dense`/`evaluate`,
`representation`) so the solver stack (linear_cg, pivoted Cholesky, prediction strategy) reads like the
reference's dependency.  All arithmetic goes through the compute backend (HIP library); K is never stored unless
`to_dense()` / the cached mode is requested.
"""
import torch

from . import backend as _backend
from .hostvals import host_float
from . import settings


class LinearOperator:
    """Minimal protocol base."""

    def _size(self):
        raise NotImplementedError

    @property
    def shape(self):
        return self._size()

    def size(self, dim=None):
        s = self._size()
        return s if dim is None else s[dim]

    @property
    def dtype(self):
        raise NotImplementedError

    @property
    def device(self):
        raise NotImplementedError

    def _matmul(self, rhs):
        raise NotImplementedError

    def matmul(self, rhs):
        if rhs.dim() == 1:
            return self._matmul(rhs.unsqueeze(-1)).squeeze(-1)
        return self._matmul(rhs)

    __matmul__ = matmul

    def _transpose_nonbatch(self):
        raise NotImplementedError

    def t(self):
        return self._transpose_nonbatch()

    def _diagonal(self):
        raise NotImplementedError

    def diagonal(self):
        return self._diagonal()

    diag = diagonal

    def _get_rows(self, idx):
        """Dense rows K[idx, :] (the `_getitem` use of pivoted Cholesky)."""
        raise NotImplementedError

    def to_dense(self):
        raise NotImplementedError

    evaluate = to_dense

    def representation(self):
        raise NotImplementedError

    def _bilinear_derivative(self, left_vecs, right_vecs):
        """Gradients of sum((left right^T) * self) w.r.t. each tensor of `representation()`."""
        raise NotImplementedError

    _quad_form_derivative = _bilinear_derivative

    def add_diag(self, diag_value):
        return AddedDiagOperator(self, diag_value)

    add_diagonal = add_diag


class AdditiveRPOperator(LinearOperator):
    """K(Z1, Z2) = outputscale * weight * sum_j exp(-0.5 (Z1[:,j]-Z2[:,j]^T)^2)  (SURVEY.md A.1).

    Z1, Z2: projected inputs (N x J), produced by `ScaledProjectionKernel` (scaled_projection_kernel.py:21-37).
    `Z2 is None` means the symmetric train-train kernel: each unordered pair is evaluated once.
    `shard` (rpgp_amd.distributed.JShard) splits the J terms across ranks with one all-reduce per MVM.
    """

    def __init__(self, Z1, Z2=None, outputscale=None, weight=1.0, shard=None):
        self.Z1 = Z1
        self.Z2 = Z2
        self.symmetric = Z2 is None
        if outputscale is None:
            outputscale = torch.ones((), dtype=Z1.dtype, device=Z1.device)
        self.outputscale = outputscale
        self.weight = float(weight)
        self.shard = shard
        # kernels take the scale by value: fetched on first use (hostvals: the marginal log-likelihood prefetches it together
        # with the noise in ONE device-to-host copy per optimiser step)
        self._scale_value = None
        self._prep = None          # rpgp_prepare tables for the factorised fast path (built on first use)

    @property
    def _scale(self):
        if self._scale_value is None:
            self._scale_value = host_float(self.outputscale) * self.weight
        return self._scale_value

    @_scale.setter
    def _scale(self, value):
        self._scale_value = float(value)

    # ---- protocol -------------------------------------------------------------------------------------------
    def _size(self):
        n2 = self.Z1.shape[0] if self.symmetric else self.Z2.shape[0]
        return torch.Size((self.Z1.shape[0], n2))

    @property
    def dtype(self):
        return self.Z1.dtype

    @property
    def device(self):
        return self.Z1.device

    @property
    def num_projections(self):
        return self.Z1.shape[1]

    def _jrange(self):
        if self.shard is None:
            return 0, self.num_projections
        return self.shard.j0, self.shard.j1

    def _local_matmul(self, rhs, noise=0.0):
        be = _backend.get_backend()
        j0, j1 = self._jrange()
        z1 = self.Z1.detach()
        if self.symmetric:
            kw = {}
            if self.shard is not None and self.shard.world_size > 1:
                ps = self.shard.pair_shard(be)
                if ps is not None and z1.dtype == torch.float32:    # pair-sharding: all projections, 1/world of the pairs
                    j0, j1 = 0, self.num_projections
                    kw = {"shard": ps}
                elif j1 <= j0:                      # J-sharding with more ranks than projections: nothing to do here
                    return torch.zeros_like(rhs) if not noise else rhs * noise
            prepare = getattr(be, "prepare", None)
            if prepare is not None and z1.dtype == torch.float32:
                if self._prep is None:
                    self._prep = prepare(z1)        # once per operator (= per hyper-parameter step), one host sync
                if self._prep.fast_ok:
                    return be.mvm_sym_prepared(self._prep, rhs, self._scale, noise, j0=j0, j1=j1, **kw)
            return be.mvm_sym(z1, rhs, self._scale, noise, j0=j0, j1=j1, **kw)
        return be.mvm_rect(z1, self.Z2.detach(), rhs, self._scale, j0=j0, j1=j1)

    def _matmul(self, rhs, noise=0.0):
        rhs = rhs.detach()
        if self.shard is None or self.shard.world_size == 1:
            out = self._local_matmul(rhs, noise if self.symmetric else 0.0)
            if noise and not self.symmetric:
                raise ValueError("a diagonal can only be added to the square symmetric operator")
            return out
        if noise and not self.symmetric:
            raise ValueError("a diagonal can only be added to the square symmetric operator")
        return self.shard.sharded_mvm(lambda j0, j1, nz: self._local_matmul(rhs, nz), rhs, float(noise))

    def fused_pivoted_cholesky(self, rank):
        """Single-launch pivoted Cholesky (rpgp_pivoted_cholesky) when the backend has it and the kernel is not sharded;
        None otherwise (callers fall back to the generic row-by-row version)."""
        be = _backend.get_backend()
        if type(self) is not AdditiveRPOperator or not self.symmetric or not hasattr(be, "pivoted_cholesky") or \
                self.Z1.dtype != torch.float32 or \
                (self.shard is not None and self.shard.world_size > 1) or self.num_projections > 64 or rank > 64:
            return None
        return be.pivoted_cholesky(self.Z1.detach().contiguous(), self._scale, min(rank, self.Z1.shape[0]))

    def native_descriptor(self, noise=0.0):
        """`struct rpgp_operator` for the native mBCG executor, or None when the operator must stay on the Python path
        (rectangular, float64, or a backend without the executor).  A sharded operator describes THIS RANK's share —
        its (world, rank) pair-shard with all projections, or its [j0, j1) slice of the projections — and
        `native_sharding()` tells the executor to sum the partial products over the ranks."""
        be = _backend.get_backend()
        if not self.symmetric or not hasattr(be, "mbcg_solve") or self.Z1.dtype != torch.float32:
            return None
        sharded = self.shard is not None and self.shard.world_size > 1
        if sharded and not settings.native_sharded_cg.on():
            return None
        from . import _lib
        z1 = self.Z1.detach()
        j0, j1 = self._jrange()
        kw = {}
        if sharded:
            ps = self.shard.pair_shard(be)
            if ps is not None:
                j0, j1 = 0, self.num_projections
                kw = {"world": ps[0], "rank": ps[1]}
        prepare = getattr(be, "prepare", None)
        if prepare is not None:
            if self._prep is None:
                self._prep = prepare(z1)
            if self._prep.fast_ok:
                return be.make_operator_desc(_lib.RPGP_OP_FUSED_PREPARED, z1.shape[0], z1.shape[1], self._scale, noise,
                                             prep=self._prep, j0=j0, j1=j1, **kw)
        return be.make_operator_desc(_lib.RPGP_OP_FUSED, z1.shape[0], z1.shape[1], self._scale, noise, Z=z1.contiguous(),
                                     j0=j0, j1=j1, **kw)

    def native_sharding(self):
        """(mode, reducer, global_N) for the native executor, or None for an unsharded operator."""
        if self.shard is None or self.shard.world_size <= 1:
            return None
        return ("partial", self.shard.reducer, self.Z1.shape[0])

    def _transpose_nonbatch(self):
        if self.symmetric:
            return self
        return AdditiveRPOperator(self.Z2, self.Z1, self.outputscale, self.weight, self.shard)

    def _diagonal(self):
        if not self.symmetric:
            raise RuntimeError("diagonal of a rectangular cross-covariance requested")
        n = self.Z1.shape[0]
        # k(x,x) = outputscale * weight * J  (diag(K) = s*w*J)
        return torch.full((n,), self._scale * self.num_projections, dtype=self.dtype, device=self.device)

    def _get_rows(self, idx):
        be = _backend.get_backend()
        z2 = self.Z1 if self.symmetric else self.Z2
        return be.dense(self.Z1.detach().index_select(0, idx).contiguous(), z2.detach(), self._scale)

    def to_dense(self):
        be = _backend.get_backend()
        z2 = self.Z1 if self.symmetric else self.Z2
        return be.dense(self.Z1.detach(), z2.detach(), self._scale)

    evaluate = to_dense

    def to_dense_cached(self):
        """The matrix of the cached-K mode: same entries as to_dense(), rows padded to 256 bytes when the backend can
        (a view with stride(0) >= N), so that the HBM stream of DenseOperator runs on aligned 16-byte loads."""
        be = _backend.get_backend()
        if type(self) is AdditiveRPOperator and self.symmetric and self.Z1.dtype == torch.float32 and \
                getattr(be, "supports_padded_dense", False):
            return be.dense(self.Z1.detach(), self.Z1.detach(), self._scale, pad=True)
        return self.to_dense()

    def to_symcache(self, wide=False):
        """Packed symmetric cache of this operator (every unordered pair once: half the bytes of to_dense_cached()), or
        None when the backend / operator cannot provide one (then the dense matrix is the cached-K form).  `wide`: the
        matrix-core tile layout for blocks of 5..16 right-hand sides (training), else the rotation order (thin solves)."""
        be = _backend.get_backend()
        if type(self) is not AdditiveRPOperator or not self.symmetric or self.Z1.dtype != torch.float32 or \
                not getattr(be, "supports_symcache", False):
            return None
        if self.shard is None or self.shard.world_size == 1:
            return be.symcache(self.Z1.detach(), wide=wide)
        ps = self.shard.pair_shard(be)          # multi-GPU: this rank's share of the pairs (pair-sharding only)
        if ps is None:
            return None
        return be.symcache(self.Z1.detach(), shard=ps, wide=wide)

    def representation(self):
        if self.symmetric:
            return (self.Z1, self.outputscale)
        return (self.Z1, self.Z2, self.outputscale)

    def _bilinear_derivative(self, left_vecs, right_vecs):
        if not self.symmetric:
            raise NotImplementedError("derivatives are only needed for the train-train kernel")
        be = _backend.get_backend()
        j0, j1 = self._jrange()
        if j1 > j0:
            gZ, gs = be.bilinear_grad(self.Z1.detach(), left_vecs.detach(), right_vecs.detach(), self._scale,
                                      j0=j0, j1=j1)
        else:
            gZ = torch.zeros_like(self.Z1)
            gs = torch.zeros((), dtype=self.dtype, device=self.device)
        if self.shard is not None and self.shard.world_size > 1:
            from .distributed import all_reduce_sum_
            all_reduce_sum_(gZ, self.shard.group)
            gs = gs.reshape(1)
            all_reduce_sum_(gs, self.shard.group)
            gs = gs.reshape(())
        return gZ, (gs if self.weight == 1.0 else gs * self.weight)

    _quad_form_derivative = _bilinear_derivative

    def dense_weight_derivative(self, S):
        """Gradients of 0.5*sum(S * self) for an explicit symmetric S (Cholesky regime)."""
        be = _backend.get_backend()
        gZ, gs = be.bilinear_grad_dense(self.Z1.detach(), S, self._scale)
        return gZ, gs * self.weight


class SKIAdditiveOperator(AdditiveRPOperator):
    """K ~= outputscale * weight * sum_j W_j Tm W_j^T  — the `ski=True` variant (training_routines.py:157-158;
    SURVEY.md Appendix E): cubic interpolation of every projection onto one shared regular grid of `grid_size` points
    and a Toeplitz matrix of the wrapped 1-D sub-kernel (`kind`: RBF by default) on the grid.  O(N) per MVM (scatter / Toeplitz / gather kernels); same protocol as the exact
    operator so the whole solve stack is unchanged.  The grid is recomputed from the data range at construction
    (once per hyper-parameter step) and is not differentiated (it is a buffer in GPyTorch as well)."""

    def __init__(self, Z1, Z2=None, outputscale=None, weight=1.0, shard=None, grid_size=1024, comp_weights=None,
                 row_shard=None, grid_rule="shared", kind="RBF"):
        super().__init__(Z1, Z2, outputscale, weight, shard=None)     # (no J-sharding: the SKI product is O(N))
        self.grid_size = int(grid_size)
        # the wrapped 1-D sub-kernel (training_routines.py:157-158 wraps whatever `_map_to_kernel` returned): only the
        # grid-to-grid Toeplitz entries depend on it; scatter, gather and the derivative's staging are kernel-agnostic
        self.kind = kind
        self._plan = None
        self._local_plan = None       # plan of this rank's rows (row-sharded solve + derivative)
        # Multi-GPU: the N training rows are split over the ranks (distributed.RowShard).  The operator itself keeps the
        # replicated interface (full Z, full vectors: prediction and the generic call sites are unchanged); the solves of
        # the marginal likelihood and of the mean cache run on `row_sharded(noise)` — every rank scatters / gathers ITS rows.
        self.row_shard = row_shard if (row_shard is not None and Z2 is None and row_shard.world_size > 1 and
                                       row_shard.N == Z1.shape[0]) else None
        # per-projection output scales (weighted rp_poly / strictly_additive kinds with `ski: true`): they ride in the
        # grid parameter block and every SKI kernel applies them on the Toeplitz stage
        self.comp_weights = comp_weights
        # grid rule: "shared" = one grid over all projections (this build's rule for the additive_rp kinds, whose reference
        # `GridInterpolationKernel(kernel, **ski_options)` at training_routines.py:157-158 passes no bounds); "reference" =
        # the per-projection bounds of polynomial_projection_kernels.py:54-63 (rp_poly / strictly_additive / additive kinds)
        self.grid_rule = grid_rule
        be = _backend.get_backend()
        kw = {} if grid_rule == "shared" else {"rule": grid_rule}
        if kind != "RBF":
            kw["kind"] = kind
        if comp_weights is None:
            self.gp = be.ski_grid(Z1.detach(), None if Z2 is None else Z2.detach(), self.grid_size, **kw)
        else:
            self.gp = be.ski_grid(Z1.detach(), None if Z2 is None else Z2.detach(), self.grid_size,
                                  weights=comp_weights.detach(), **kw)

    def fused_pivoted_cholesky(self, rank):
        be = _backend.get_backend()
        if not self.symmetric or not hasattr(be, "ski_pivoted_cholesky") or self.Z1.shape[1] > 64 or rank > 64 or \
                self.Z1.dtype != torch.float32:            # (float64 parity path: the generic greedy factor on _get_rows)
            return None
        return be.ski_pivoted_cholesky(self.Z1.detach().contiguous(), self.gp, self._scale,
                                       min(rank, self.Z1.shape[0]), self.grid_size)

    def _get_plan(self):
        """rpgp_ski_plan of the square operator (points sorted by interpolation cell), built on first use — once per
        operator = once per hyper-parameter step; None when the backend has no planned product."""
        be = _backend.get_backend()
        if not self.symmetric or not hasattr(be, "ski_plan") or self.Z1.dtype != torch.float32:
            return None
        if self._plan is None:
            self._plan = be.ski_plan(self.Z1.detach().contiguous(), self.gp, self.grid_size)
        return self._plan if self._plan.ok else None

    def _local_matmul(self, rhs, noise=0.0):
        be = _backend.get_backend()
        z1 = self.Z1.detach()
        z2 = z1 if self.symmetric else self.Z2.detach()
        plan = self._get_plan() if rhs.shape[-1] <= 12 or rhs.dim() == 1 else None
        if plan is not None:
            return be.ski_mvm(z1, z2, self.gp, rhs, self._scale, noise, self.grid_size, plan=plan)
        return be.ski_mvm(z1, z2, self.gp, rhs, self._scale, noise if self.symmetric else 0.0, self.grid_size)

    def native_descriptor(self, noise=0.0):
        be = _backend.get_backend()
        if not self.symmetric or not hasattr(be, "mbcg_solve") or self.Z1.dtype != torch.float32:
            return None
        from . import _lib
        z1 = self.Z1.detach().contiguous()
        return be.make_operator_desc(_lib.RPGP_OP_SKI, z1.shape[0], z1.shape[1], self._scale, noise, Z=z1, gp=self.gp,
                                     G=self.grid_size, prep=self._get_plan())

    def native_sharding(self):
        return None

    def _matmul(self, rhs, noise=0.0):
        if noise and not self.symmetric:
            raise ValueError("a diagonal can only be added to the square symmetric operator")
        return self._local_matmul(rhs.detach(), noise)

    def _transpose_nonbatch(self):
        if self.symmetric:
            return self
        t = SKIAdditiveOperator.__new__(SKIAdditiveOperator)
        AdditiveRPOperator.__init__(t, self.Z2, self.Z1, self.outputscale, self.weight, None)
        t.grid_size, t.gp, t.comp_weights, t.row_shard, t._plan = self.grid_size, self.gp, self.comp_weights, None, None
        t._local_plan = None
        t.grid_rule = self.grid_rule
        t.kind = self.kind
        return t

    def row_sharded(self, noise):
        """This rank's rows of (self + noise I) as a RowShardedSKIOperator on the SAME grid block (and weights)."""
        rs = self.row_shard
        op = RowShardedSKIOperator(self.Z1.detach()[rs.r0:rs.r1], self._scale, 1.0, rs, grid_size=self.grid_size,
                                   noise=noise, gp=self.gp)
        # one plan of the local rows per hyper-parameter step, shared by the forward solve and the derivative
        if self._local_plan is None:
            self._local_plan = op._get_plan()
        else:
            op._plan = self._local_plan
        return op

    def _get_local_plan(self, Zl):
        be = _backend.get_backend()
        if not hasattr(be, "ski_plan") or Zl.dtype != torch.float32 or Zl.shape[0] == 0:
            return None
        if self._local_plan is None:
            self._local_plan = be.ski_plan(Zl, self.gp, self.grid_size)
        return self._local_plan if self._local_plan.ok else None

    def row_sharded_bilinear_derivative(self, left_local, right_local):
        """`_bilinear_derivative` with the vectors given as this rank's rows: per 12-column piece the 2T-column grid
        histogram is all-reduced (J x G x 2T float64), the Toeplitz stage is replicated and every rank differentiates ITS
        rows; the d+1 (+J) parameter sums are all-reduced once.  Returns the same tuple as `_bilinear_derivative`, with
        gZ assembled for all N rows (zero-padded all-reduce) so that the projection's backward runs replicated."""
        be = _backend.get_backend()
        rs = self.row_shard
        Zl = self.Z1.detach()[rs.r0:rs.r1].contiguous()
        comp = self.comp_weights is not None
        J = self.Z1.shape[1]
        gZ = torch.zeros_like(self.Z1.detach())
        tail = torch.zeros(1 + (J if comp else 0), dtype=self.dtype, device=self.device)
        for t0 in range(0, left_local.shape[1], 12):
            Lc = left_local[:, t0:t0 + 12].detach().contiguous()
            Rc = right_local[:, t0:t0 + 12].detach().contiguous()
            lplan = self._get_local_plan(Zl)            # cell-sorted scatters of the local rows (same sums as one process)
            hist = be.ski_bilinear_scatter(Zl, self.gp, Lc, Rc, self.grid_size, plan=lplan) if lplan is not None else \
                be.ski_bilinear_scatter(Zl, self.gp, Lc, Rc, self.grid_size)
            rs.all_reduce_(hist, "sum")
            gzl, gs, gc = be.ski_bilinear_finish(Zl, self.gp, hist, Lc, Rc, self._scale, self.grid_size, comp=comp)
            gZ[rs.r0:rs.r1] += gzl
            tail[0] += gs.reshape(())
            if comp:
                tail[1:] += gc.reshape(-1)
        rs.all_reduce_(gZ, "sum")
        rs.all_reduce_(tail, "sum")
        if comp:
            w = self.comp_weights.detach().to(tail)
            return gZ, tail[0] * self.weight, self._scale * tail[1:] / w
        return gZ, tail[0] * self.weight

    def _diagonal(self):
        if not self.symmetric:
            raise RuntimeError("diagonal of a rectangular cross-covariance requested")
        return _backend.get_backend().ski_diag(self.Z1.detach(), self.gp, self._scale, self.grid_size)

    def representation(self):
        base = super().representation()
        return base if self.comp_weights is None else base + (self.comp_weights,)

    def _get_rows(self, idx):
        be = _backend.get_backend()
        z1 = self.Z1.detach()
        z2 = z1 if self.symmetric else self.Z2.detach()
        if hasattr(be, "ski_dense") and z1.shape[1] <= 64:
            return be.ski_dense(z1.index_select(0, idx).contiguous(), z2, self.gp, self._scale, self.grid_size)
        k = idx.numel()
        eye = torch.eye(k, dtype=self.dtype, device=self.device)
        # K[idx, :] = (K(Z2, Z1[idx]))^T : interpolate the k selected points, gather at all columns
        cols = be.ski_mvm(z2, z1.index_select(0, idx).contiguous(), self.gp, eye, self._scale, 0.0, self.grid_size)
        return cols.t().contiguous()

    def to_dense(self):
        be = _backend.get_backend()
        z1 = self.Z1.detach()
        z2 = z1 if self.symmetric else self.Z2.detach()
        if hasattr(be, "ski_dense") and z1.shape[1] <= 64:
            # straight from the interpolation weights and the Toeplitz lags (no N x N identity through the MVM)
            return be.ski_dense(z1, z2, self.gp, self._scale, self.grid_size)
        eye = torch.eye(z2.shape[0], dtype=self.dtype, device=self.device)
        return be.ski_mvm(z1, z2, self.gp, eye, self._scale, 0.0, self.grid_size)

    evaluate = to_dense

    def _bilinear_derivative(self, left_vecs, right_vecs):
        if not self.symmetric:
            raise NotImplementedError("derivatives are only needed for the train-train kernel")
        be = _backend.get_backend()
        plan = self._get_plan()
        if plan is not None and hasattr(be, "ski_bilinear_scatter"):
            # staged form on the operator's plan: cell-sorted scatters of L and R (no atomics), the 2T-column Toeplitz product
            # on the matrix cores, then the per-row derivative — 12 columns per piece
            z = self.Z1.detach()
            comp = self.comp_weights is not None
            gZ = gs = gc = None
            for t0 in range(0, left_vecs.shape[1], 12):
                Lc = left_vecs[:, t0:t0 + 12].detach().contiguous()
                Rc = right_vecs[:, t0:t0 + 12].detach().contiguous()
                hist = be.ski_bilinear_scatter(z, self.gp, Lc, Rc, self.grid_size, plan=plan)
                gz_p, gs_p, gc_p = be.ski_bilinear_finish(z, self.gp, hist, Lc, Rc, self._scale, self.grid_size, comp=comp)
                gZ = gz_p if gZ is None else gZ.add_(gz_p)
                gs = gs_p if gs is None else gs.add_(gs_p)
                if comp:
                    gc = gc_p if gc is None else gc.add_(gc_p)
            if comp:
                w = self.comp_weights.detach().to(gc)
                return gZ, gs * self.weight, self._scale * gc / w
            return gZ, gs * self.weight
        if self.comp_weights is not None:
            gZ, gs, gc = be.ski_bilinear_grad_comp(self.Z1.detach(), self.gp, left_vecs.detach(), right_vecs.detach(),
                                                   self._scale, self.grid_size)
            w = self.comp_weights.detach().to(gc)
            return gZ, gs * self.weight, self._scale * gc / w       # d/dZ, d/d outputscale, d/d comp_weights
        gZ, gs = be.ski_bilinear_grad(self.Z1.detach(), self.gp, left_vecs.detach(), right_vecs.detach(), self._scale,
                                      self.grid_size)
        return gZ, gs * self.weight

    _quad_form_derivative = _bilinear_derivative

    def dense_weight_derivative(self, S):
        """Gradients of 0.5*sum(S * self) for an explicit symmetric S: bilinear form with L = S/2, R = I."""
        eye = torch.eye(S.shape[0], dtype=self.dtype, device=self.device)
        return self._bilinear_derivative((0.5 * S).contiguous(), eye)


class RowShardedSKIOperator(LinearOperator):
    """The SKI operator (+ noise I) with the N data rows split across ranks — the MI355X counterpart of wrapping the
    SKI kernel in `MultiDeviceKernel` (training_routines.py:407-408 with :157-158).  Every vector it touches is a
    rank's LOCAL block of rows:
        hist_r = W_r^T v_r  (J x G x T float64, rpgp_ski_scatter)  ->  all-reduce SUM (RCCL; 270 KB at J = 3, T = 11)
        H = Tm hist         (replicated float64 Toeplitz product, rpgp_ski_grid_product)
        out_r = scale * W_r H + noise * v_r                        (rpgp_ski_gather)
    The interpolation grid is built from the GLOBAL coordinate range (MIN / MAX all-reduce), so all ranks interpolate onto
    the same grid and the sharded product equals the single-process SKI product up to the float64 summation order of the
    histogram.  CG on it runs with all-reduced inner products (`linear_cg(..., reduce=shard.all_reduce_)`); the rank-k
    pivoted-Cholesky preconditioner is built from distributed pivots (`row_sharded_preconditioner`)."""

    def __init__(self, Z_local, outputscale, weight, row_shard, grid_size=1024, noise=0.0, gp=None, kind="RBF"):
        self.Z1 = Z_local.detach().contiguous()
        self.row_shard = row_shard
        self.grid_size = int(grid_size)
        self._scale = float(outputscale) * float(weight)
        self._noise = float(noise)
        be = _backend.get_backend()
        self._plan = None
        if gp is not None:                 # the grid block of the replicated operator this one was split from
            self.gp = gp
            return
        if self.Z1.shape[0] > 0:
            rng = torch.stack([self.Z1.min(), -self.Z1.max()])
        else:
            rng = torch.full((2,), float("inf"), device=self.Z1.device, dtype=self.Z1.dtype)
        row_shard.all_reduce_(rng, "min")                     # one collective for (min, -max)
        kw = {} if kind == "RBF" else {"kind": kind}
        if self.Z1.dtype == torch.float64:
            kw["dtype"] = torch.float64
        self.gp = be.ski_grid_from_range(float(rng[0]), float(-rng[1]), self.grid_size, self.Z1.device, **kw)

    def _size(self):
        n = self.Z1.shape[0]
        return torch.Size((n, n))

    @property
    def dtype(self):
        return self.Z1.dtype

    @property
    def device(self):
        return self.Z1.device

    def _matmul(self, rhs):
        be = _backend.get_backend()
        rhs = rhs.detach()
        squeeze = rhs.dim() == 1
        V = rhs.reshape(self.Z1.shape[0], rhs.shape[1] if rhs.dim() == 2 else 1).contiguous()
        outs = []
        for c0 in range(0, V.shape[1], 12):                   # the staged entry points take T <= 12 columns
            Vp = V[:, c0:c0 + 12].contiguous()
            J = self.Z1.shape[1]
            if self.Z1.shape[0] > 0:
                plan = self._get_plan()
                hist = be.ski_scatter(self.Z1, self.gp, Vp, self.grid_size, plan=plan) if plan is not None else \
                    be.ski_scatter(self.Z1, self.gp, Vp, self.grid_size)
            else:
                hist = torch.zeros(J, self.grid_size, Vp.shape[1], dtype=torch.float64, device=self.Z1.device)
            self.row_shard.all_reduce_(hist, "sum")
            if self.Z1.shape[0] > 0:
                H = be.ski_grid_product(hist, self.gp, self.grid_size)
                if plan is not None:
                    outs.append(be.ski_gather(self.Z1, self.gp, H, Vp, self._scale, self._noise, self.grid_size, plan=plan))
                else:
                    outs.append(be.ski_gather(self.Z1, self.gp, H, Vp, self._scale, self._noise, self.grid_size))
            else:
                outs.append(Vp.clone())
        out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)
        return out.reshape(-1) if squeeze else out

    def _diagonal(self):
        if self.Z1.shape[0] == 0:
            return torch.zeros(0, dtype=self.dtype, device=self.device)
        return _backend.get_backend().ski_diag(self.Z1, self.gp, self._scale, self.grid_size) + self._noise

    def _get_plan(self):
        be = _backend.get_backend()
        if not hasattr(be, "ski_plan") or self.Z1.dtype != torch.float32 or self.Z1.shape[0] == 0:
            return None
        if self._plan is None:
            self._plan = be.ski_plan(self.Z1, self.gp, self.grid_size)
        return self._plan if self._plan.ok else None

    def native_descriptor(self):
        """The local rows as an RPGP_OP_SKI descriptor (noise included); with `native_sharding()` the executor all-reduces
        the grid histogram and the inner products, so the solve is the row-sharded one."""
        be = _backend.get_backend()
        if not hasattr(be, "mbcg_solve") or self.Z1.dtype != torch.float32 or not settings.native_sharded_cg.on():
            return None
        from . import _lib
        z = self.Z1 if self.Z1.shape[0] > 0 else torch.zeros(1, self.Z1.shape[1], dtype=self.dtype, device=self.device)
        return be.make_operator_desc(_lib.RPGP_OP_SKI, self.Z1.shape[0], self.Z1.shape[1], self._scale, self._noise, Z=z,
                                     gp=self.gp, G=self.grid_size, prep=self._get_plan())

    def native_sharding(self):
        return ("rows", self.row_shard.reducer, self.row_shard.N)

    def kernel_rows(self, z_rows):
        """scale * K_ski(z_rows, local rows): k x N_local block for rows given by their (global) coordinates."""
        return _backend.get_backend().ski_dense(z_rows.contiguous(), self.Z1, self.gp, self._scale, self.grid_size)


class RowShardedWoodbury:
    """M = L L^T + noise I with L split by rows: every product with L^T is a k x T all-reduce."""

    def __init__(self, L_local, noise, row_shard):
        self.L = L_local
        self.noise = float(noise)
        self.row_shard = row_shard
        from .precond import gram64
        be = _backend.get_backend()
        on_device = L_local.is_cuda or getattr(be, "name", "") != "hip-gfx950"
        # in-tree float64 Gram / update kernels on the fp32 factor (rpgp_gram_f64, rpgp_woodbury_apply), as in
        # precond.WoodburyPreconditioner; a rank without rows contributes zeros
        self._be = be if (L_local.dtype == torch.float32 and 0 < L_local.shape[1] <= 64 and hasattr(be, "gram_f64")
                          and on_device and L_local.shape[0] > 0) else None
        self._L64 = L_local.double() if self._be is None else None
        cap = self._be.gram_f64(L_local, L_local) if self._be is not None else gram64(self._L64, self._L64)
        row_shard.all_reduce_(cap, "sum")
        self._cinv = self._logdet_cap = None
        if hasattr(be, "woodbury_setup") and on_device and cap.shape[0] <= 64:
            self._cap_chol, self._cinv, self._logdet_cap = be.woodbury_setup(cap, self.noise)   # one launch, no sync
        else:
            cap.diagonal().add_(self.noise)
            self._cap_chol = torch.linalg.cholesky(cap)

    def solve(self, r):
        """(r - L C^-1 L^T r) / noise on the local rows.  As in precond.WoodburyPreconditioner.solve the subtraction
        cancels to ~sigma^2 / |K| of r along the range of L (1e-6 at N = 391k), so t and r - L t stay float64; wide
        blocks go through in 64-column panels (one k x T all-reduce per panel)."""
        from .precond import gram64, _PANEL
        squeeze = r.dim() == 1
        if squeeze:
            r = r.unsqueeze(-1)
        out = torch.empty_like(r)
        for c0 in range(0, max(r.shape[1], 1), _PANEL):
            if self._be is not None and r.dtype == torch.float32:
                rp = r[:, c0:c0 + _PANEL]
                t = self._be.gram_f64(self.L, rp)
                self.row_shard.all_reduce_(t, "sum")
                t = self._cinv @ t if self._cinv is not None else torch.cholesky_solve(t, self._cap_chol)
                out[:, c0:c0 + _PANEL] = self._be.woodbury_apply(self.L, rp, t, self.noise)
                continue
            if self._L64 is None:
                self._L64 = self.L.double()
            rd = r[:, c0:c0 + _PANEL].double()
            t = gram64(self._L64, rd)
            self.row_shard.all_reduce_(t, "sum")
            t = torch.cholesky_solve(t, self._cap_chol)
            out[:, c0:c0 + _PANEL] = torch.addmm(rd, self._L64, t, alpha=-1.0).div_(self.noise).to(r.dtype)
        return out.squeeze(-1) if squeeze else out

    def logdet(self):
        """log|M| = log|noise I_k + L^T L| + (N - k) log noise with the GLOBAL N."""
        import math
        if self._logdet_cap is not None:
            ld_cap = float(self._logdet_cap)
            if ld_cap != ld_cap:
                raise RuntimeError("the preconditioner's capacitance matrix is not positive definite")
        else:
            ld_cap = float(2.0 * torch.log(self._cap_chol.diagonal()).sum())
        return ld_cap + (self.row_shard.N - self.L.shape[1]) * math.log(self.noise)

    def cinv(self):
        """(noise I + L^T L)^-1 in float64 (k x k), identical on every rank, for the native mBCG executor."""
        if self._cinv is not None:
            return self._cinv
        return torch.cholesky_inverse(self._cap_chol).contiguous()

    __call__ = solve


def row_sharded_preconditioner(op, rank):
    """Rank-`rank` pivoted Cholesky of the noise-free row-sharded SKI operator with GLOBAL greedy pivots: per step one
    MAX all-reduce of (largest local residual diagonal), one broadcast of the pivot's J coordinates from its owner, and
    the usual downdate of the local rows.  Same pivots as the single-process factorisation (ties go to the lowest rank)."""
    sh = op.row_shard
    dev, dt = op.device, op.dtype
    nloc, J = op.Z1.shape
    d = (op._diagonal() - op._noise).clone() if nloc else torch.zeros(0, device=dev, dtype=dt)
    dmax0 = torch.tensor([float(d.max()) if nloc else 0.0], device=dev, dtype=torch.float64)
    sh.all_reduce_(dmax0, "max")
    L = torch.zeros(rank, nloc, dtype=dt, device=dev)
    for m in range(rank):
        # (value, -rank) packed so that ONE max all-reduce picks the largest value and, among ties, the lowest rank
        loc = float(d.max()) if nloc else -1.0
        key = torch.tensor([loc], device=dev, dtype=torch.float64)
        sh.all_reduce_(key, "max")
        best = float(key[0])
        claim = torch.tensor([sh.rank if (nloc and loc == best) else sh.world_size], device=dev, dtype=torch.float64)
        sh.all_reduce_(claim, "min")
        owner = int(claim[0])
        zp = torch.zeros(1, J, device=dev, dtype=dt)
        lp = torch.zeros(m + 1, device=dev, dtype=dt)             # the pivot's own entries in the previous columns
        if sh.rank == owner:
            p = int(torch.argmax(d))
            zp[0] = op.Z1[p]
            lp[:m] = L[:m, p]
            lp[m] = d[p]
            d[p] = 0.0
            own_p = p
        sh.broadcast_(zp, owner)
        sh.broadcast_(lp, owner)
        dp = float(lp[m])
        if not dp > 1e-10 * float(dmax0[0]):
            continue                                              # exhausted: zero column (as the single-process loop)
        if nloc:
            row = op.kernel_rows(zp)[0]
            if m > 0:
                row = row - L[:m].t() @ lp[:m]
            l = row / (dp ** 0.5)
            if sh.rank == owner:
                l[own_p] = dp ** 0.5
            L[m] = l
            d = (d - l * l).clamp_min(0.0)
            if sh.rank == owner:
                d[own_p] = 0.0
    return RowShardedWoodbury(L.t().contiguous(), op._noise, sh)


class FamilyAdditiveOperator(AdditiveRPOperator):
    """K = outputscale * sum_c w_c phi_kind(group c of Z's columns) — the other members of the family behind the same
    operator: `kernel_type` Matern / InverseMQ / Cosine sub-kernels (training_routines.py:47-88),
    k > 1 RBF sub-kernels (:172-174) and per-component output scales (polynomial_projection_kernels.py:88-98).
    Z is already divided by the lengthscales.  Same fused tile kernels with a different kernel-function policy
    (rpgp_family_*); runs replicated (no J-sharding), fp32 only."""

    def __init__(self, Z1, Z2=None, outputscale=None, comp_weights=None, kind="RBF", group=1, product=False):
        super().__init__(Z1, Z2, outputscale, 1.0, shard=None)
        self.kind, self.group = kind, int(group)
        # a k > 1 group as the PRODUCT of its 1-D sub-kernels (polynomial_projection_kernels.py:70-86) instead of the radial
        # k-dimensional sub-kernel of `additive_rp` (training_routines.py:172-174); one and the same for the RBF
        self.product = bool(product) and self.group > 1 and kind != "RBF"
        if Z1.shape[1] % self.group:
            raise ValueError("the number of columns must be a multiple of the sub-kernel dimension")
        ncomp = Z1.shape[1] // self.group
        if comp_weights is None:
            comp_weights = torch.full((ncomp,), 1.0 / ncomp, dtype=Z1.dtype, device=Z1.device)
        self.comp_weights = comp_weights
        # float64 operators (`--double`) keep float64 weights: the backend then serves them with the runtime-(kind, group)
        # kernels (csrc/rpgp_family_generic.hip), as it does k > 1 sub-kernels of the non-RBF types
        wdt = torch.float64 if Z1.dtype == torch.float64 else torch.float32
        w = comp_weights.detach().to(device=Z1.device, dtype=wdt).reshape(-1).contiguous()
        self.fam = _backend.get_backend().make_family(kind, self.group, w, self.product) if self.product else \
            _backend.get_backend().make_family(kind, self.group, w)
        self._generic = bool(getattr(self.fam, "generic", False))
        self._wsum = float(w.sum())                 # one host sync per construction (= per optimiser step)

    def _local_matmul(self, rhs, noise=0.0):
        be = _backend.get_backend()
        z1 = self.Z1.detach()
        if self.symmetric:
            return be.family_mvm_sym(self.fam, z1, rhs, self._scale, noise)
        return be.family_mvm_rect(self.fam, z1, self.Z2.detach(), rhs, self._scale)

    def _matmul(self, rhs, noise=0.0):
        if noise and not self.symmetric:
            raise ValueError("a diagonal can only be added to the square symmetric operator")
        return self._local_matmul(rhs.detach(), noise if self.symmetric else 0.0)

    def fused_pivoted_cholesky(self, rank):
        be = _backend.get_backend()
        if not self.symmetric or not hasattr(be, "family_pivoted_cholesky") or self.Z1.shape[1] > 64 or rank > 64 or \
                self._generic:
            return None
        return be.family_pivoted_cholesky(self.fam, self.Z1.detach().contiguous(), self._scale,
                                          min(rank, self.Z1.shape[0]), self._wsum)

    def native_descriptor(self, noise=0.0):
        be = _backend.get_backend()
        if not self.symmetric or not hasattr(be, "mbcg_solve") or self._generic:
            return None
        from . import _lib
        z1 = self.Z1.detach().contiguous()
        return be.make_operator_desc(_lib.RPGP_OP_FAMILY, z1.shape[0], z1.shape[1], self._scale, noise, Z=z1,
                                     family=self.fam)

    def native_sharding(self):
        return None

    def _transpose_nonbatch(self):
        if self.symmetric:
            return self
        return FamilyAdditiveOperator(self.Z2, self.Z1, self.outputscale, self.comp_weights, self.kind, self.group, self.product)

    def _diagonal(self):
        if not self.symmetric:
            raise RuntimeError("diagonal of a rectangular cross-covariance requested")
        # phi(0) = 1 for every member: k(x,x) = outputscale * sum_c w_c
        return torch.full((self.Z1.shape[0],), self._scale * self._wsum, dtype=self.dtype, device=self.device)

    def _get_rows(self, idx):
        be = _backend.get_backend()
        z2 = self.Z1 if self.symmetric else self.Z2
        return be.family_dense(self.fam, self.Z1.detach().index_select(0, idx).contiguous(), z2.detach(), self._scale)

    def to_dense(self):
        be = _backend.get_backend()
        z2 = self.Z1 if self.symmetric else self.Z2
        return be.family_dense(self.fam, self.Z1.detach(), z2.detach(), self._scale)

    evaluate = to_dense

    def representation(self):
        if self.symmetric:
            return (self.Z1, self.outputscale, self.comp_weights)
        return (self.Z1, self.Z2, self.outputscale, self.comp_weights)

    def _finish_grads(self, gZ, gcomp):
        w = self.comp_weights.detach().to(gcomp)
        return gZ, (w * gcomp).sum(), self._scale * gcomp          # d/dZ, d/d outputscale, d/d comp_weights

    def _bilinear_derivative(self, left_vecs, right_vecs):
        if not self.symmetric:
            raise NotImplementedError("derivatives are only needed for the train-train kernel")
        gZ, gc = _backend.get_backend().family_bilinear_grad(self.fam, self.Z1.detach(), left_vecs.detach(),
                                                            right_vecs.detach(), self._scale)
        return self._finish_grads(gZ, gc)

    _quad_form_derivative = _bilinear_derivative

    def dense_weight_derivative(self, S):
        gZ, gc = _backend.get_backend().family_bilinear_grad_dense(self.fam, self.Z1.detach(), S, self._scale)
        return self._finish_grads(gZ, gc)


# group sizes the family tile kernels are instantiated for (ops.FAMILY_GROUPS); other sizes are padded with zero columns
_FAMILY_GROUPS = (1, 2, 3, 4, 5, 8, 10, 20)


def padded_group_size(k):
    """Smallest instantiated group size >= k (an RBF group padded with all-zero coordinates is the same kernel:
    exp(-0.5 sum d^2) gains factors exp(0))."""
    for g in _FAMILY_GROUPS:
        if g >= k:
            return g
    raise NotImplementedError("multiplicative groups of more than %d dimensions are not built" % _FAMILY_GROUPS[-1])


class MixedGroupOperator(AdditiveRPOperator):
    """K = outputscale * sum_c w_c prod_{m in group c} k1(z_m - z_m') with multiplicative groups of DIFFERENT sizes
    (`general_rp_poly` with e.g. degrees [1, 1, 2, 3], training_routines.py:192-207; `create_multi_additive_kernel`,
    :247-258: every feature subset up to a degree).  The components are bucketed by group size and every bucket is one
    FamilyAdditiveOperator on its own column slice of Z — same fused tile kernels, one launch per distinct size per
    product; sums, diagonals, rows and derivatives are assembled here.  Runs replicated, fp32; solves in the native mBCG
    executor as an RPGP_OP_SUM of the buckets."""

    def __init__(self, Z1, Z2=None, outputscale=None, comp_weights=None, kind="RBF", degrees=(1,), product=False):
        super().__init__(Z1, Z2, outputscale, 1.0, shard=None)
        self.kind = kind
        self.product = bool(product)
        self.degrees = [int(dg) for dg in degrees]
        if sum(self.degrees) != Z1.shape[1]:
            raise ValueError("the group sizes must add up to the number of columns")
        ncomp = len(self.degrees)
        if comp_weights is None:
            comp_weights = torch.full((ncomp,), 1.0 / ncomp, dtype=Z1.dtype, device=Z1.device)
        self.comp_weights = comp_weights
        starts = [0]
        for dg in self.degrees:
            starts.append(starts[-1] + dg)
        self.buckets = []                           # (component indices, column indices, operator) per distinct size
        def gather(Zm, co, k, kp, ncomp):
            z = Zm.detach().index_select(1, co)
            if kp == k:
                return z.contiguous()
            zp = torch.zeros(z.shape[0], ncomp, kp, dtype=z.dtype, device=z.device)     # zero-padded groups (see padded_group_size)
            zp[:, :, :k] = z.reshape(z.shape[0], ncomp, k)
            return zp.reshape(z.shape[0], ncomp * kp)

        for k in sorted(set(self.degrees)):
            comps = [c for c, dg in enumerate(self.degrees) if dg == k]
            cols = [starts[c] + m for c in comps for m in range(k)]
            kp = padded_group_size(k) if kind == "RBF" else k
            ci = torch.as_tensor(comps, dtype=torch.long, device=Z1.device)
            co = torch.as_tensor(cols, dtype=torch.long, device=Z1.device)
            z1 = gather(Z1, co, k, kp, len(comps))
            z2 = None if Z2 is None else gather(Z2, co, k, kp, len(comps))
            part = FamilyAdditiveOperator(z1, z2, outputscale, comp_weights.detach().index_select(0, ci), kind, kp, product)
            part._true_group = k
            self.buckets.append((ci, co, part))
        self._wsum = sum(part._wsum for _, _, part in self.buckets)

    def _local_matmul(self, rhs, noise=0.0):
        out = None
        for i, (_, _, part) in enumerate(self.buckets):
            o = part._local_matmul(rhs, noise if i == 0 else 0.0)
            out = o if out is None else out.add_(o)
        return out

    def _matmul(self, rhs, noise=0.0):
        if noise and not self.symmetric:
            raise ValueError("a diagonal can only be added to the square symmetric operator")
        return self._local_matmul(rhs.detach(), noise if self.symmetric else 0.0)

    def fused_pivoted_cholesky(self, rank):
        return None

    def native_descriptor(self, noise=0.0):
        """RPGP_OP_SUM of the buckets' RPGP_OP_FAMILY descriptors (the noise rides on the first one)."""
        be = _backend.get_backend()
        if not self.symmetric or not hasattr(be, "make_sum_operator_desc"):
            return None
        parts = []
        for i, (_, _, part) in enumerate(self.buckets):
            made = part.native_descriptor(noise if i == 0 else 0.0)
            if made is None:
                return None
            parts.append(made)
        return be.make_sum_operator_desc(parts)

    def native_sharding(self):
        return None

    def _transpose_nonbatch(self):
        if self.symmetric:
            return self
        return MixedGroupOperator(self.Z2, self.Z1, self.outputscale, self.comp_weights, self.kind, self.degrees, self.product)

    def _diagonal(self):
        if not self.symmetric:
            raise RuntimeError("diagonal of a rectangular cross-covariance requested")
        return torch.full((self.Z1.shape[0],), self._scale * self._wsum, dtype=self.dtype, device=self.device)

    def _get_rows(self, idx):
        out = None
        for _, _, part in self.buckets:
            o = part._get_rows(idx)
            out = o if out is None else out.add_(o)
        return out

    def to_dense(self):
        out = None
        for _, _, part in self.buckets:
            o = part.to_dense()
            out = o if out is None else out.add_(o)
        return out

    evaluate = to_dense

    def to_dense_cached(self):
        return self.to_dense()

    def to_symcache(self, wide=False):
        return None

    def representation(self):
        if self.symmetric:
            return (self.Z1, self.outputscale, self.comp_weights)
        return (self.Z1, self.Z2, self.outputscale, self.comp_weights)

    def _assemble(self, grads):
        gZ = torch.zeros_like(self.Z1)
        gc = torch.zeros(len(self.degrees), dtype=self.Z1.dtype, device=self.Z1.device)
        gs = torch.zeros((), dtype=self.Z1.dtype, device=self.Z1.device)
        for (ci, co, part), (gz_p, gs_p, gc_p) in zip(self.buckets, grads):
            k, kp = part._true_group, part.group
            if kp != k:                              # drop the derivatives of the padding columns
                gz_p = gz_p.reshape(gz_p.shape[0], -1, kp)[:, :, :k].reshape(gz_p.shape[0], -1)
            gZ.index_copy_(1, co, gz_p.to(gZ))
            gc.index_copy_(0, ci, gc_p.to(gc).reshape(-1))
            gs = gs + gs_p.to(gs)
        return gZ, gs, gc

    def _bilinear_derivative(self, left_vecs, right_vecs):
        if not self.symmetric:
            raise NotImplementedError("derivatives are only needed for the train-train kernel")
        return self._assemble([part._bilinear_derivative(left_vecs, right_vecs) for _, _, part in self.buckets])

    _quad_form_derivative = _bilinear_derivative

    def dense_weight_derivative(self, S):
        return self._assemble([part.dense_weight_derivative(S) for _, _, part in self.buckets])


class AddedDiagOperator(LinearOperator):
    """base + noise * I  (the likelihood's AddedDiagLazyTensor); the noise term is fused into the MVM kernel."""

    def __init__(self, base, noise, noise_value=None):
        if not isinstance(noise, torch.Tensor):
            noise = torch.as_tensor(float(noise), dtype=base.dtype, device=base.device)
        self.base = base
        self.noise = noise
        # host value of the noise: given by the caller, or the one `hostvals.prefetch` / the step kernels left on the tensor —
        # a plain `float(noise)` is a device-to-host copy and a synchronisation in the middle of the step's set-up
        self._noise = float(noise_value) if noise_value is not None else host_float(noise)

    def _size(self):
        return self.base._size()

    @property
    def dtype(self):
        return self.base.dtype

    @property
    def device(self):
        return self.base.device

    def _matmul(self, rhs):
        if isinstance(self.base, AdditiveRPOperator):
            return self.base._matmul(rhs, noise=self._noise)
        return self.base._matmul(rhs) + self._noise * rhs

    def native_descriptor(self):
        fn = getattr(self.base, "native_descriptor", None)
        return fn(self._noise) if fn is not None else None

    def native_sharding(self):
        fn = getattr(self.base, "native_sharding", None)
        return fn() if fn is not None else None

    def _transpose_nonbatch(self):
        return self

    def _diagonal(self):
        return self.base._diagonal() + self._noise

    def _get_rows(self, idx):
        rows = self.base._get_rows(idx)
        rows[torch.arange(idx.numel(), device=rows.device), idx] += self._noise
        return rows

    def to_dense(self):
        d = self.base.to_dense()
        d.diagonal().add_(self._noise)
        return d

    evaluate = to_dense

    def representation(self):
        return self.base.representation() + (self.noise,)

    def _bilinear_derivative(self, left_vecs, right_vecs):
        g_noise = (left_vecs * right_vecs).sum()
        return self.base._bilinear_derivative(left_vecs, right_vecs) + (g_noise,)

    _quad_form_derivative = _bilinear_derivative


class SymCachedOperator(LinearOperator):
    """Cached-K mode on the packed symmetric cache (rpgp_symcache_*): Khat = scale * K + noise I applied as ONE stream
    over N^2 / 2 stored kernel values per block of up to 12 right-hand sides.  `diag_value` = scale * (number of
    projections), the constant diagonal of the additive RBF kernel."""

    def __init__(self, cache, scale, noise=0.0, diag_value=None, shard=None):
        self.cache = cache
        self._scale = float(scale)
        self._noise = float(noise)
        self._diag_value = diag_value
        self.shard = shard if (shard is not None and shard.world_size > 1) else None   # cache = this rank's pair share

    def _size(self):
        return torch.Size((self.cache.N, self.cache.N))

    @property
    def dtype(self):
        return torch.float32

    @property
    def device(self):
        return self.cache.device

    def _matmul(self, rhs):
        be = _backend.get_backend()
        if self.shard is None:
            return be.symcache_mvm(self.cache, rhs.detach(), self._scale, self._noise)
        # one all-reduce of the N x T partial products per MVM, the noise term added by rank 0's slab reduce only
        return self.shard.sharded_mvm(lambda j0, j1, nz: be.symcache_mvm(self.cache, rhs.detach(), self._scale, nz), rhs,
                                      self._noise)

    def native_descriptor(self):
        be = _backend.get_backend()
        if not hasattr(be, "mbcg_solve") or (self.cache.world != 1) != (self.shard is not None):
            return None
        if self.shard is not None and not settings.native_sharded_cg.on():
            return None
        from . import _lib
        return be.make_operator_desc(_lib.RPGP_OP_SYMCACHE, self.cache.N, 0, self._scale, self._noise, symcache=self.cache)

    def native_sharding(self):
        return None if self.shard is None else ("partial", self.shard.reducer, self.cache.N)

    def _transpose_nonbatch(self):
        return self

    def _diagonal(self):
        if self._diag_value is None:
            raise RuntimeError("SymCachedOperator was built without its diagonal value")
        return torch.full((self.cache.N,), float(self._diag_value) + self._noise, dtype=torch.float32, device=self.device)

    def to_dense(self):
        n = self.cache.N
        out = torch.empty((n, n), dtype=torch.float32, device=self.device)
        eye = torch.zeros((n, 12), dtype=torch.float32, device=self.device)
        for c0 in range(0, n, 12):
            w = min(12, n - c0)
            eye.zero_()
            eye[c0:c0 + w, :w].fill_diagonal_(1.0)
            out[:, c0:c0 + w] = self._matmul(eye)[:, :w]
        return out

    evaluate = to_dense


class DenseOperator(LinearOperator):
    """Cached-K mode: an explicit symmetric kernel matrix in HBM; every MVM is an HBM stream
    (rpgp_dense_mvm for thin right-hand sides, a library GEMM for wide ones)."""

    def __init__(self, Kd, noise=0.0):
        self.Kd = Kd
        self._noise = float(noise)

    def _size(self):
        return self.Kd.shape

    @property
    def dtype(self):
        return self.Kd.dtype

    @property
    def device(self):
        return self.Kd.device

    def _matmul(self, rhs):
        rhs = rhs.detach()
        if rhs.shape[-1] <= 32 and rhs.is_cuda and rhs.dtype == torch.float32:
            return _backend.get_backend().dense_mvm(self.Kd, rhs, self._noise)      # HBM-bound MFMA thin GEMM
        out = self.Kd @ rhs                                                        # wide blocks: library GEMM
        if self._noise:
            out.add_(rhs, alpha=self._noise)
        return out

    def native_descriptor(self):
        be = _backend.get_backend()
        if not hasattr(be, "mbcg_solve") or not self.Kd.is_cuda or self.Kd.dtype != torch.float32 or \
                self.Kd.stride(1) != 1 or self.Kd.shape[0] >= 2 ** 31:
            return None
        from . import _lib
        return be.make_operator_desc(_lib.RPGP_OP_DENSE, self.Kd.shape[0], 0, 1.0, self._noise, Kd=self.Kd)

    def _transpose_nonbatch(self):
        return self

    def _diagonal(self):
        return self.Kd.diagonal() + self._noise

    def _get_rows(self, idx):
        rows = self.Kd.index_select(0, idx).clone()
        if self._noise:
            rows[torch.arange(idx.numel(), device=rows.device), idx] += self._noise
        return rows

    def to_dense(self):
        d = self.Kd.clone()
        if self._noise:
            d.diagonal().add_(self._noise)
        return d

    evaluate = to_dense
