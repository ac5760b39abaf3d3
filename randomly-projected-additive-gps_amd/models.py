"""Exact GP model, marginal log-likelihood and prediction strategy (counterparts of gp_models/models.py:10-20,
gpytorch.models.ExactGP, gpytorch.mlls.ExactMarginalLogLikelihood and DefaultPredictionStrategy as the reference uses
them from fitting/optimizing.py:65-73 and training_routines.py:545-579; semantics per SURVEY.md §3.3-3.4, A.3, B.7)."""
import math
import warnings

import torch
from torch import nn

from . import settings
from .dense_ops import DenseKernelOperator
from .inv_quad_logdet import inv_quad_logdet, psd_safe_cholesky, use_cholesky
from .likelihoods import ConstantMean, GaussianLikelihood, MultivariateNormal, LOG2PI
from .hostvals import host_float, prefetch
from .linear_cg import linear_cg
from .operators import AddedDiagOperator, AdditiveRPOperator, DenseOperator, SKIAdditiveOperator, SymCachedOperator
from .precond import build_preconditioner


def lanczos_inverse_root(matmul, init_vec, rank):
    """LOVE cache (GPyTorch `fast_pred_var`, `--fast_pred`, gp_experiment_runner.py:235,327): R (N x k) with
    Khat^-1 ~= R R^T from k = `max_root_decomposition_size` Lanczos steps with full re-orthogonalisation started at
    `init_vec`; k operator applications with T = 1."""
    N = init_vec.shape[0]
    k = min(rank, N)
    Q = torch.zeros(N, k, dtype=init_vec.dtype, device=init_vec.device)
    alpha = torch.zeros(k, dtype=torch.float64)
    beta = torch.zeros(k, dtype=torch.float64)
    q = init_vec / init_vec.norm()
    m = 0
    for i in range(k):
        Q[:, i] = q
        w = matmul(q.reshape(-1, 1)).reshape(-1)
        a = torch.dot(w, q)
        w = w - a * q - (beta[i - 1].to(w.dtype) * Q[:, i - 1] if i > 0 else 0.0)
        for _ in range(2):                                   # full re-orthogonalisation (twice is enough)
            w = w - Q[:, :i + 1] @ (Q[:, :i + 1].t() @ w)
        b = w.norm()
        alpha[i] = float(a)
        m = i + 1
        if float(b) < 1e-6 * abs(float(a)) or i == k - 1:
            break
        beta[i] = float(b)
        q = w / b
    Tm = torch.diag(alpha[:m]) + torch.diag(beta[:m - 1], 1) + torch.diag(beta[:m - 1], -1)
    evals, evecs = torch.linalg.eigh(Tm)
    evals = evals.clamp_min(1e-10)
    return Q[:, :m] @ (evecs / evals.sqrt()).to(Q.dtype).to(Q.device)


class PredictionStrategy:
    """Caches alpha = Khat^-1 (y - c) (the `mean_cache`) and produces predictive mean / covariance:
       mu* = K(X*,X) alpha + c ;  Sigma* = K(X*,X*) - K(X*,X) Khat^-1 K(X,X*)."""

    def __init__(self, model):
        self.model = model
        x, y = model.train_inputs, model.train_targets
        with torch.no_grad():
            self.mean_const = model.mean_module(x)
            self.op = model.covar_module(x)                        # train-train operator
            self.noise = model.likelihood.noise.reshape(()).detach()
            self.r = (y - self.mean_const).reshape(-1, 1)
            N = x.shape[0]
            self.dense_path = isinstance(self.op, DenseKernelOperator) or use_cholesky(N) or \
                not settings.fast_computations.solves()
            if self.dense_path:
                Kd = self.op.to_dense()
                Kd.diagonal().add_(self.noise)
                self.chol = psd_safe_cholesky(Kd)
                self.alpha = torch.cholesky_solve(self.r, self.chol)
                self.khat = None
            elif getattr(self.op, "row_shard", None) is not None:
                # row-sharded SKI model: the mean-cache solve runs on this rank's rows (all-reduces inside the solve), the
                # solution is assembled on every rank; everything downstream (cross-covariance products, wide solves of the
                # predictive covariance) uses the replicated operator
                from .operators import row_sharded_preconditioner
                self.chol = None
                rs = self.op.row_shard
                sop = self.op.row_sharded(host_float(self.noise))
                rank_k = settings.max_preconditioner_size.value()
                pre_sh = row_sharded_preconditioner(sop, rank_k) \
                    if (N >= settings.min_preconditioning_size.value() and rank_k > 0) else None
                a_loc = linear_cg(sop._matmul, self.r[rs.r0:rs.r1].contiguous(), tolerance=settings.eval_cg_tolerance.value(),
                                  max_iter=settings.max_cg_iterations.value(), preconditioner=pre_sh, operator=sop,
                                  reduce=rs.all_reduce_, global_size=N)
                self.alpha = torch.zeros_like(self.r)
                self.alpha[rs.r0:rs.r1] = a_loc
                rs.all_reduce_(self.alpha, "sum")
                self.khat = AddedDiagOperator(self.op, self.noise)
                self.pre = build_preconditioner(self.op, host_float(self.noise), settings)
            else:
                self.chol = None
                shard = getattr(self.op, "shard", None)
                sharded = shard is not None and shard.world_size > 1
                cacheable = isinstance(self.op, AdditiveRPOperator) and not isinstance(self.op, SKIAdditiveOperator) and \
                    x.dtype == torch.float32
                per_rank = 2.0 / (shard.world_size if sharded else 1)     # pair-sharded ranks cache their own share
                cache = self.op.to_symcache() if cacheable and settings.use_cached_kernel(N, x.device, per_rank) else None
                if cache is not None:
                    # thin solves (the mean cache, LOVE's Lanczos) stream the packed symmetric cache: half the bytes of
                    # the dense matrix; the N_test-wide covariance solve builds the dense matrix on demand (solve())
                    self.khat = SymCachedOperator(cache, self.op._scale, host_float(self.noise),
                                                  diag_value=self.op._scale * self.op.num_projections,
                                                  shard=shard if sharded else None)
                elif cacheable and not sharded and settings.use_cached_kernel(N, x.device):
                    self.khat = DenseOperator(self.op.to_dense_cached(), host_float(self.noise))
                    self._dense_khat = self.khat
                else:
                    self.khat = AddedDiagOperator(self.op, self.noise)
                self.pre = build_preconditioner(self.op, host_float(self.noise), settings)
                self.alpha = self._solve_thin(self.r)
                if not sharded:
                    self._refine_mean_cache(model, x)

    def _solve_thin(self, B):
        """The float32 solver of the thin blocks (mean cache, refinement corrections): preconditioned mBCG on `self.khat`."""
        return linear_cg(self.khat._matmul, B, tolerance=settings.eval_cg_tolerance.value(),
                         max_iter=settings.max_cg_iterations.value(), preconditioner=self.pre, operator=self.khat)

    def _refine_mean_cache(self, model, x):
        """Mixed-precision refinement of alpha = Khat^-1 (y - c) (settings.solve_refinement): residuals with the float64 twin
        of the fused operator (float64 projection, float64 exponentials: rpgp_mvm_f64; 41 ms at N = 50 000), corrections by
        the float32 solver.  float32 CG stalls at a TRUE residual of ~1e-4 at N = 50 000 (measured against the float64
        oracle); one round reaches 4e-8.  The refined solution is kept in float64 (`alpha64`) and the predictive mean is
        summed with the float64 cross-covariance operator: with an exact alpha the float32 sum of 50 000 cancelling terms
        was still 1.7e-4 off."""
        self.alpha64 = None
        rounds = settings.solve_refinement.value()
        N = x.shape[0]
        if rounds <= 0 or x.dtype != torch.float32 or not x.is_cuda or N < settings.solve_refinement_min_size.value():
            return
        f64 = getattr(model.covar_module, "float64_operator", None)
        op64 = f64(x) if f64 is not None else None
        if op64 is None:
            return
        noise64 = host_float(self.noise)
        r64 = self.r.double()
        a64 = self.alpha.double()
        rnorm = float(r64.norm())
        self.refinement_residuals = []
        for _ in range(rounds):
            res = r64 - op64._matmul(a64, noise64)
            rel = float(res.norm()) / max(rnorm, 1e-300)
            self.refinement_residuals.append(rel)
            if rel < 1e-9:
                break
            scale = float(res.abs().max())
            if scale == 0.0:
                break
            a64 = a64 + self._solve_thin((res / scale).float()).double() * scale
        self.alpha = a64.float()
        self.alpha64 = a64

    # ---- mixed-precision solve of the N_test-wide covariance block (settings.solve_refinement) ----------------------------
    def _mixed_precision_ready(self, like, n_train, n_test):
        """float32 Cholesky factor of the stored dense Khat as the solver, float64 residuals against a float64 copy of the
        same matrix (LAPACK dsposv's scheme): needs 4N^2 (matrix) + 4N^2 (factor) + 8N^2 (float64 copy) bytes + the blocks."""
        if settings.solve_refinement.value() <= 0 or self.dense_path or not like.is_cuda or like.dtype != torch.float32:
            return False
        if n_train < settings.solve_refinement_min_size.value() or n_train > settings.cholesky_precond_size.value():
            return False
        if not isinstance(self.op, AdditiveRPOperator) or isinstance(self.op, SKIAdditiveOperator):
            return False
        shard = getattr(self.op, "shard", None)
        if shard is not None and shard.world_size > 1:
            return False
        if getattr(self, "_mp", None) is None:
            # decided on the memory that is FREE now (the symmetric / dense cache, the LOVE and solve buffers of this
            # strategy and, with several ranks on one device, the other processes already hold theirs), minus a margin
            # for the library's potrf / trsm workspaces; what the allocator caches but does not use counts as free
            free, _total = torch.cuda.mem_get_info(like.device)
            free += torch.cuda.memory_reserved(like.device) - torch.cuda.memory_allocated(like.device)
            c = min(n_test, max(32, settings.predictive_block_floats.value() // max(n_train, 1)))
            have = 4.0 * n_train * n_train if getattr(self, "_dense_khat", None) is not None else 0.0
            need = 20.0 * n_train * n_train - have + 32.0 * n_train * c + 12.0 * n_test * n_test
            if need > 0.8 * free - 2.0e9:
                return False
            try:
                if getattr(self, "_dense_khat", None) is None:
                    self._dense_khat = DenseOperator(self.op.to_dense_cached() if hasattr(self.op, "to_dense_cached")
                                                     else self.op.to_dense(), host_float(self.noise))
                Kh = self._dense_khat.to_dense()                   # float32, noise on the diagonal
                K64 = Kh.double()
                from .precond import blocked_cholesky
                Lc, info = blocked_cholesky(Kh)                    # (fp16x3 trailing updates beyond N = 16k: precond.py)
                del Kh
            except torch.OutOfMemoryError:                         # lost the race for the memory: the general path serves
                Kh = K64 = Lc = None
                self._mp = False
                torch.cuda.empty_cache()
                return False
            if int(info) != 0:
                self._mp = False
            else:
                self._mp = (Lc, K64)
                if getattr(self, "_chol_pre", None) is None:       # the same factor serves solve()'s preconditioned CG
                    from .precond import CholeskyPreconditioner
                    self._chol_pre = CholeskyPreconditioner(Lc)
        return bool(self._mp)

    def _mp_solve(self, B):
        """Khat^-1 B in float64 for a float32 block B (N x c): factor solve + refinement until the float64 residual is below
        1e-9 relative (each round gains ~kappa * eps32; at most six)."""
        Lc, K64 = self._mp

        def fsolve(R):
            out = torch.empty_like(R)
            for c0 in range(0, R.shape[1], 2048):                  # (column panels: library trsm workspace)
                out[:, c0:c0 + 2048] = torch.cholesky_solve(R[:, c0:c0 + 2048].contiguous(), Lc)
            return out
        B64 = B.double()
        S = fsolve(B).double()
        bn = B64.norm(dim=0).clamp_min(1e-300)
        self.wide_refinement_residuals = []
        for _ in range(6):
            R = B64 - K64 @ S
            rel = float((R.norm(dim=0) / bn).max())
            self.wide_refinement_residuals.append(rel)
            if rel < 1e-9:
                break
            scale = R.abs().amax(dim=0, keepdim=True).clamp_min(1e-300)
            S += fsolve((R / scale).float()).double() * scale
        return S

    def _mixed_precision_covariance(self, cross, cov, n_train, n_test):
        """Sigma* = K** - K*x Khat^-1 Kx* with the solves of `_mp_solve`, the products in float64, in column blocks of test
        points."""
        budget = settings.predictive_block_floats.value()
        c = max(32, min(n_test, budget // max(n_train, 1)))
        dev = cov.device
        total = torch.cuda.get_device_properties(dev).total_memory
        Kc = cross._get_rows(torch.arange(n_test, device=dev)) if 4.0 * n_train * n_test <= 0.15 * total else None   # K(X*, X)
        cov64 = cov.double()
        for c0 in range(0, n_test, c):
            idx = torch.arange(c0, min(c0 + c, n_test), device=dev)
            Kx_blk = (Kc[c0:c0 + idx.numel()] if Kc is not None else cross._get_rows(idx)).t().contiguous()     # K(X, X*[idx])
            S = self._mp_solve(Kx_blk)
            if idx.numel() == n_test:
                cov64 -= Kx_blk.double().t() @ S
            else:
                for r0 in range(0, n_test, 8192):                  # K(X*, X) S in row panels (float64 copies of 8192 rows)
                    ridx = torch.arange(r0, min(r0 + 8192, n_test), device=dev)
                    rows = Kc[r0:r0 + ridx.numel()] if Kc is not None else cross._get_rows(ridx)
                    cov64[r0:r0 + ridx.numel(), c0:c0 + idx.numel()] -= rows.double() @ S
        return cov64.to(cov.dtype)

    # ---- posterior at the training inputs in closed form -----------------------------------------------------------------
    def _train_closed_form_ready(self, like):
        """Closed form available: the dense (Cholesky) regime, or the CG regime where the float32 factor + float64 copy of
        the mixed-precision solve exist / fit (settings.solve_refinement on)."""
        if self.dense_path:
            return True
        n = self.r.shape[0]
        return self._mixed_precision_ready(like, n, 1)

    def train_log_prob(self, target):
        """log N(target | mu_train, Sigma_train + sigma^2 I) with Sigma_train = K - K Khat^-1 K = sigma^2 (I - sigma^2 Khat^-1):
             C := Sigma_train + sigma^2 I = 2 sigma^2 I - sigma^4 Khat^-1 = sigma^2 Khat^-1 B,   B := 2 K + sigma^2 I
             C^-1 = B^-1 Khat / sigma^2 ,   log|C| = N log sigma^2 - log|Khat| + log|B|
        i.e. one more factorisation (of B, same kernel matrix with another diagonal) instead of the N x N posterior
        covariance, its N^3 products and a float64 Cholesky of it.  Replaces `mll(train_outputs, trainY)` of
        training_routines.py:567-569 for the train set."""
        N = self.r.shape[0]
        s2 = host_float(self.noise)
        mean = (self.model.train_targets.double().reshape(-1, 1) - s2 *
                (self.alpha64 if getattr(self, "alpha64", None) is not None else self.alpha.double()))
        d = target.double().reshape(-1, 1) - mean
        if self.dense_path:
            Lk = self.chol.double()
            Khat = Lk @ Lk.t()
            B = 2.0 * Khat
            B.diagonal().sub_(s2)
            Lb = psd_safe_cholesky(B)
            w = torch.cholesky_solve(d, Lb)
            quad = float((d * (Khat @ w)).sum()) / s2
            logdet = N * math.log(s2) - 2.0 * float(torch.log(Lk.diagonal()).sum()) + 2.0 * float(torch.log(Lb.diagonal()).sum())
            return -0.5 * (quad + logdet + N * LOG2PI)
        Lc, K64 = self._mp                                       # float32 factor and float64 copy of Khat
        if N <= settings.dense_solve_size.value():
            # mid sizes: everything in float64 on the stored matrix (as the predictive covariance does below dense_solve_size).
            # A confident fit (a GAM with sigma^2 ~ 1e-4 s) leaves 2K + sigma^2 I numerically indefinite in float32 — found by
            # the round-4 soak over the served specifications; psd_safe_cholesky adds the float32-scale jitter the matrix's own
            # rounding calls for.
            c64 = self._chol64_factor(target)
            if c64 is None:
                c64 = psd_safe_cholesky(K64, jitter=1e-6)
            B64 = 2.0 * K64
            B64.diagonal().sub_(s2)
            Lb = psd_safe_cholesky(B64, jitter=1e-6)
            w = torch.cholesky_solve(d, Lb)
            quad = float((d * (K64 @ w)).sum()) / s2
            logdet = N * math.log(s2) - 2.0 * float(torch.log(c64.diagonal()).sum()) + 2.0 * float(torch.log(Lb.diagonal()).sum())
            return -0.5 * (quad + logdet + N * LOG2PI)
        logdet_khat = 2.0 * float(torch.log(Lc.diagonal().double()).sum())
        # B in float32 from the stored kernel matrix (one N x N float32 block, factorised); B w in float64 is
        # 2 Khat_64 w - sigma^2 w: no float64 copy of B
        Bf = self._dense_khat.Kd[:, :N] * 2.0
        Bf.diagonal().add_(s2)
        from .precond import blocked_cholesky
        Lb, info = blocked_cholesky(Bf)                           # (fp16x3 trailing updates beyond N = 16k)
        del Bf
        if int(info) != 0:
            raise RuntimeError("2 K + sigma^2 I is not positive definite in float32")
        logdet_b = 2.0 * float(torch.log(Lb.diagonal().double()).sum())
        w = torch.cholesky_solve(d.float(), Lb).double()
        dn = float(d.norm())
        for _ in range(4):                                       # float64 residuals against B, float32 factor corrections
            res = d - (2.0 * (K64 @ w) - s2 * w)
            if float(res.norm()) < 1e-9 * dn:
                break
            scale = float(res.abs().max())
            w = w + torch.cholesky_solve((res / scale).float(), Lb).double() * scale
        del Lb
        quad = float((d * (K64 @ w)).sum()) / s2
        logdet = N * math.log(s2) - logdet_khat + logdet_b
        return -0.5 * (quad + logdet + N * LOG2PI)

    def solve(self, B):
        if self.dense_path:
            return torch.cholesky_solve(B, self.chol)
        khat = self.khat
        if B.shape[1] > 12 and isinstance(self.op, AdditiveRPOperator):
            # wide right-hand sides (predictive covariance, T = N_test) on the exact fused operator: materialise K once
            # so every CG iteration is a library GEMM on the matrix cores instead of N_test/12 fused sweeps.  The SKI
            # operator's own wide scatter/gather is O(N J T) and wins above N ~ 32k (measured: 35 vs 88 ms per product
            # at N = 50k, T = 2000; 115 vs 45 ms at N = T = 14 939), so it is densified only below that.
            N = B.shape[0]
            fits = (4.0 * N * N <= 0.25 * torch.cuda.get_device_properties(B.device).total_memory) if B.is_cuda else True
            if isinstance(self.op, SKIAdditiveOperator) and N > 32768:
                fits = False
            if fits:
                if getattr(self, "_dense_khat", None) is None:
                    self._dense_khat = DenseOperator(self.op.to_dense_cached() if hasattr(self.op, "to_dense_cached") else self.op.to_dense(), host_float(self.noise))
                khat = self._dense_khat
                if N <= settings.dense_solve_size.value():
                    # with Khat in HBM anyway, a float64 Cholesky (rocSOLVER) solves the N_test-wide block exactly and
                    # faster than ~50 CG iterations of N x N x N_test GEMMs (N = 15k: 0.6 s against 3 s)
                    if getattr(self, "_chol64", None) is None:
                        self._chol64 = psd_safe_cholesky(khat.to_dense().double())
                    # column panels of 1024: hipSOLVER potrs / rocBLAS trsm run out of workspace on N x N right-hand sides
                    out = torch.empty_like(B)
                    for c0 in range(0, B.shape[1], 1024):
                        blk = B[:, c0:c0 + 1024].double().contiguous()
                        out[:, c0:c0 + 1024] = torch.cholesky_solve(blk, self._chol64).to(B.dtype)
                    return out
                # (live at the factorisation: the dense matrix, its noise-added clone, the library's working copy and the
                #  factor = 4 x 4N^2 bytes; only taken when that fits 60 % of the device)
                room = 16.0 * N * N <= 0.6 * torch.cuda.get_device_properties(B.device).total_memory if B.is_cuda else False
                if N <= settings.cholesky_precond_size.value() and room and B.dtype == torch.float32:
                    # beyond the float64 direct solve: one fp32 factorisation of Khat (N = 50k: 0.8 s), used as the
                    # preconditioner — the wide CG then converges in a handful of 91 ms GEMMs instead of ~40
                    if getattr(self, "_chol_pre", None) is None:
                        from .precond import CholeskyPreconditioner
                        from .precond import blocked_cholesky
                        Kh = khat.to_dense()
                        Lc, info = blocked_cholesky(Kh)
                        del Kh
                        self._chol_pre = CholeskyPreconditioner(Lc) if int(info) == 0 else False
                    if self._chol_pre:
                        return linear_cg(khat._matmul, B, tolerance=settings.eval_cg_tolerance.value(),
                                         max_iter=settings.max_cg_iterations.value(), preconditioner=self._chol_pre,
                                         operator=khat, min_iter=1)
        return linear_cg(khat._matmul, B, tolerance=settings.eval_cg_tolerance.value(),
                         max_iter=settings.max_cg_iterations.value(), preconditioner=getattr(self, "pre", None),
                         operator=khat)

    def _chol64_factor(self, like):
        """Float64 Cholesky factor of the dense Khat when the exact wide solve applies (see solve()), else None."""
        N = self.r.shape[0]
        if self.dense_path or not isinstance(self.op, AdditiveRPOperator) or N > settings.dense_solve_size.value():
            return None
        if isinstance(self.op, SKIAdditiveOperator) and N > 32768:
            return None
        if like.is_cuda and 4.0 * N * N > 0.25 * torch.cuda.get_device_properties(like.device).total_memory:
            return None
        if getattr(self, "_dense_khat", None) is None:
            self._dense_khat = DenseOperator(self.op.to_dense_cached() if hasattr(self.op, "to_dense_cached")
                                             else self.op.to_dense(), host_float(self.noise))
        if getattr(self, "_chol64", None) is None:
            self._chol64 = psd_safe_cholesky(self._dense_khat.to_dense().double())
        return self._chol64

    def predict(self, xs):
        model = self.model
        at_train = xs is model.train_inputs or (xs.shape == model.train_inputs.shape and
                                                xs.data_ptr() == model.train_inputs.data_ptr())
        with torch.no_grad():
            if at_train and self._train_closed_form_ready(xs):
                # the posterior AT the training inputs (`evaluate_on_train`, training_routines.py:551-556): the mean is
                # y - sigma^2 alpha, and log N(y | mu, Sigma + sigma^2 I) has a closed form in Khat — no N x N posterior
                # covariance is formed unless somebody asks for it (C4: 19 s of library GEMMs + a 50 000^2 float64 Cholesky)
                a = self.alpha64 if getattr(self, "alpha64", None) is not None else self.alpha.double()
                mean = (model.train_targets.double().reshape(-1, 1) - host_float(self.noise) * a).reshape(-1).to(xs.dtype)
                if settings.skip_posterior_variances.on():
                    return MultivariateNormal(mean, torch.zeros_like(mean), diagonal_only=True)
                return TrainPosterior(mean, self, xs)
            return self._predict_general(xs)

    def _predict_general(self, xs):
        model = self.model
        with torch.no_grad():
            cross = model.covar_module(xs, model.train_inputs)      # K(X*, X) operator
            if getattr(self, "alpha64", None) is not None:        # refined mean cache: the sum in float64 as well
                cross64 = model.covar_module.float64_operator(xs, model.train_inputs)
                mean = cross64._matmul(self.alpha64).reshape(-1).to(xs.dtype) + model.mean_module(xs)
            else:
                mean = cross._matmul(self.alpha).reshape(-1) + model.mean_module(xs)
            if settings.skip_posterior_variances.on():
                return MultivariateNormal(mean, torch.zeros_like(mean), diagonal_only=True)
            # Sigma* = K** - K*x Khat^-1 Kx*.  With an iterative solve S ~= Khat^-1 Kx* the plain product K*x S is
            # neither symmetric nor bounded by the exact quadratic form; the variational form
            #     Kx*^T Khat^-1 Kx*  >=  2 Kx*^T S - S^T Khat S          (equality at the exact solve)
            # is symmetric and keeps Sigma* positive semi-definite for ANY S (slightly conservative variances at loose
            # CG tolerances; the plain product has a FIRST-order error amplified by ||K|| / sigma^2).  The solves run in
            # column blocks of test points; S and Khat S are kept (2 x N x N* floats).
            n_train, n_test = model.train_inputs.shape[0], xs.shape[0]
            cov = model.covar_module(xs).to_dense()                 # K(X*, X*)
            budget = settings.predictive_block_floats.value()       # floats per N x c block of CG state (1 GiB)
            c = max(32, min(n_test, budget // max(n_train, 1)))
            if self.dense_path:
                Kx = cross._get_rows(torch.arange(n_test, device=xs.device)).t().contiguous()   # K(X, X*)
                cov -= Kx.t() @ torch.cholesky_solve(Kx, self.chol)
            elif settings.fast_pred_var.on():
                # LOVE: Sigma* ~= K** - (K*x R)(K*x R)^T with the cached rank-k Lanczos inverse root R
                if getattr(self, "_love_root", None) is None:
                    init = cross._transpose_nonbatch()._matmul(torch.ones(n_test, 1, dtype=cov.dtype, device=cov.device))
                    self._love_root = lanczos_inverse_root(self.khat._matmul, init.reshape(-1),
                                                           settings.max_root_decomposition_size.value())
                KR = cross._matmul(self._love_root)                  # (N* x k)
                cov -= KR @ KR.t()
            elif self._chol64_factor(cov) is not None and 8.0 * n_train * n_test <= 4.0e9:
                # Khat is in HBM with a float64 factor (N <= dense_solve_size): the whole quadratic form in float64.  For a
                # confident model Sigma* is a difference of O(s J) quantities that cancels to 1e-5: in fp32 GEMMs the result was
                # indefinite by 3e-4 (kin8nm-shaped GAM fit), which a small fitted noise does not cover.
                c64 = self._chol64_factor(cov)
                Kx = cross._get_rows(torch.arange(n_test, device=xs.device)).t().contiguous().double()   # K(X, X*)
                sol = torch.empty_like(Kx)
                for c0 in range(0, n_test, 1024):            # (column panels: the library's triangular solves run out of
                    sol[:, c0:c0 + 1024] = torch.cholesky_solve(Kx[:, c0:c0 + 1024].contiguous(), c64)   # workspace on wide blocks)
                cov = (cov.double() - Kx.t() @ sol).to(cov.dtype)
            elif self._mixed_precision_ready(cov, n_train, n_test):
                # beyond the float64 direct solve: float32 factor + float64 residuals (measured at N = 50 000 against the
                # float64 oracle: the float32 CG form below is 8.7e-4 off in the variances, this one < 1e-5)
                cov = self._mixed_precision_covariance(cross, cov, n_train, n_test)
            else:
                khat = self.khat
                total_mem = torch.cuda.get_device_properties(xs.device).total_memory if xs.is_cuda else float("inf")
                if 8.0 * n_train * n_test > 0.35 * total_mem:
                    raise RuntimeError("the full %d x %d predictive covariance needs two %d x %d work arrays (%.0f GB): "
                                       "use --skip_posterior_variances" % (n_test, n_test, n_train, n_test,
                                                                          8e-9 * n_train * n_test))
                S = torch.empty(n_train, n_test, dtype=cov.dtype, device=cov.device)
                KS = torch.empty_like(S)
                BtS = torch.empty(n_test, n_test, dtype=cov.dtype, device=cov.device)
                # several column blocks (evaluate-on-train at N = 45 000: eight): K(X*, X) is materialised once when it fits
                # beside S and KS, so that K(X*, X) sol is a library GEMM — the fused rectangular product takes the block 12
                # columns at a time (500 sweeps per block: 20 of the 30 s of that evaluation)
                Kc = None
                if c < n_test and n_test > 12 and 12.0 * n_train * n_test <= 0.35 * total_mem:
                    Kc = cross._get_rows(torch.arange(n_test, device=xs.device))       # (N* x N)
                for c0 in range(0, n_test, c):
                    idx = torch.arange(c0, min(c0 + c, n_test), device=xs.device)
                    Kx_blk = (Kc[c0:c0 + idx.numel()] if Kc is not None else cross._get_rows(idx)).t().contiguous()   # K(X, X*[idx])
                    sol = self.solve(Kx_blk)                            # ~ Khat^-1 K(X, X*[idx])
                    S[:, idx] = sol
                    # Khat sol through the matrix the solve used (a library GEMM when it was densified)
                    KS[:, idx] = (getattr(self, "_dense_khat", None) or khat)._matmul(sol)
                    if idx.numel() == n_test and sol.shape[1] > 12:
                        BtS[:, idx] = Kx_blk.t() @ sol                  # single block: K(X*, X) is Kx_blk^T, already dense
                    elif Kc is not None:
                        BtS[:, idx] = Kc @ sol
                    else:
                        BtS[:, idx] = cross._matmul(sol)                # K(X*, X) sol
                cov -= BtS + BtS.t() - S.t() @ KS
            cov = 0.5 * (cov + cov.t())
        return MultivariateNormal(mean, cov)


class TrainPosterior(MultivariateNormal):
    """Posterior at the training inputs.  `mean` is exact (y - sigma^2 alpha); the log-density of the noisy version comes from
    the strategy's closed form; the N x N covariance (or its diagonal) is only computed when somebody reads it."""

    def __init__(self, mean, strategy, xs, noise=None):
        self.mean = mean
        self.diagonal_only = False
        self._strategy, self._xs, self._noise, self._cov = strategy, xs, noise, None

    @property
    def covariance(self):
        if self._cov is None:
            cov = self._strategy._predict_general(self._xs).covariance
            if self._noise is not None:
                cov = cov.clone()
                cov.diagonal().add_(self._noise)
            self._cov = cov
        return self._cov

    def with_observation_noise(self, noise):
        """p(y | f) at the training inputs (what `likelihood(train_outputs)` returns)."""
        return TrainPosterior(self.mean, self._strategy, self._xs, noise=noise)

    def log_prob(self, value):
        strategy = self._strategy
        same_noise = self._noise is not None and abs(float(self._noise) - float(strategy.noise)) <= 1e-12 * float(strategy.noise)
        if not same_noise:
            return super().log_prob(value)
        try:
            lp = strategy.train_log_prob(value)
        except RuntimeError:
            # (a factorisation of the closed form failed: the explicit posterior covariance and its jittered float64 factor)
            return super().log_prob(value)
        return torch.as_tensor(lp, dtype=value.dtype, device=value.device)


class LazyPrior(MultivariateNormal):
    """Train-mode output of a model whose objective has a fused form (fused_mll): nothing is evaluated until somebody reads
    `mean` / `covariance` — the marginal log-likelihood recognises the object and runs the fused node instead."""

    def __init__(self, model):
        self.diagonal_only = False
        self._model, self._mvn = model, None

    def _materialize(self):
        if self._mvn is None:
            self._mvn = self._model.forward(self._model.train_inputs)
        return self._mvn

    @property
    def materialized(self):
        return self._mvn is not None

    @property
    def mean(self):
        return self._materialize().mean

    @property
    def covariance(self):
        return self._materialize().covariance


class ExactGP(nn.Module):
    """Train mode: `model(train_x)` returns the prior MultivariateNormal(mean, K operator).
    Eval mode: `model(x)` returns the posterior at x (prediction strategy cached across calls, B.7)."""

    def __init__(self, train_inputs, train_targets, likelihood):
        super().__init__()
        self.train_inputs = train_inputs
        self.train_targets = train_targets
        self.likelihood = likelihood
        self.prediction_strategy = None

    def train(self, mode=True):
        if mode:
            self.prediction_strategy = None
        return super().train(mode)

    def forward(self, x):
        raise NotImplementedError

    def __call__(self, x):
        if self.training:
            if not (x is self.train_inputs or (x.shape == self.train_inputs.shape and torch.equal(x, self.train_inputs))):
                raise RuntimeError("You must train on the training inputs!")
            from . import fused_mll
            if fused_mll.applicable(self):
                return LazyPrior(self)
            return self.forward(self.train_inputs)
        if self.prediction_strategy is None:
            self.prediction_strategy = PredictionStrategy(self)
        return self.prediction_strategy.predict(x)


class ExactGPModel(ExactGP):
    """Basic exact GP with constant mean and a provided kernel (gp_models/models.py:10-20)."""

    def __init__(self, train_x, train_y, likelihood, kernel):
        super().__init__(train_x, train_y, likelihood)
        self.mean_module = ConstantMean()
        self.covar_module = kernel

    def forward(self, x):
        mean_x = self.mean_module(x)
        covar_x = self.covar_module(x)
        return MultivariateNormal(mean_x, covar_x)


class _DenseGaussianLogProb(torch.autograd.Function):
    """log N(r; 0, K) for a dense K (the `kind: full` path) with the analytic gradient 0.5 (alpha alpha^T - K^-1): the
    backward of an autograd-tracked `torch.linalg.cholesky` runs N x N triangular solves whose library workspace fails at a
    few thousand rows on this stack (HIPBLAS_STATUS_ALLOC_FAILED at N = 7372)."""

    @staticmethod
    def forward(ctx, K, r):
        Lc = psd_safe_cholesky(K.detach())
        alpha = torch.cholesky_solve(r.detach().unsqueeze(-1), Lc)
        ctx.save_for_backward(Lc, alpha)
        n = r.shape[0]
        return -0.5 * (r.detach().unsqueeze(-1) * alpha).sum() - torch.log(Lc.diagonal()).sum() - 0.5 * n * LOG2PI

    @staticmethod
    def backward(ctx, g):
        Lc, alpha = ctx.saved_tensors
        gK = gr = None
        if ctx.needs_input_grad[0]:
            gK = torch.cholesky_inverse(Lc).neg_()
            gK.addmm_(alpha, alpha.t())
            gK.mul_(0.5 * g)
        if ctx.needs_input_grad[1]:
            gr = (-g * alpha).reshape(-1)
        return gK, gr


class ExactMarginalLogLikelihood(nn.Module):
    """mll(output, target) = (1/N) [ log N(target | mean, K + sigma^2 I) + sum log-priors ]   (SURVEY.md A.3)."""

    def __init__(self, likelihood, model):
        super().__init__()
        self.likelihood = likelihood
        self.model = model

    def negative(self, output, target):
        """-mll(output, target): the training loss of fitting/optimizing.py:70.  For the fused objective the sign rides inside
        the one autograd node (same bits as negating its value: the separate negation was a launch each way per step)."""
        if isinstance(output, LazyPrior) and not output.materialized:
            from . import fused_mll, settings
            if output._model is self.model and self.likelihood is self.model.likelihood and settings.fused_training.on():
                return fused_mll.evaluate(self.model, self.likelihood, target, negate=True)
        return -self(output, target)

    def negative_and_backward(self, output, target):
        """`loss = -mll(output, target); loss.backward()` of the training loop (fitting/optimizing.py:70-72), returning the loss.
        For the fused objective on the step kernels the gradients are written into `.grad` without the autograd engine
        (fused_mll.value_and_grad: same values, ~120 us of host time less per step); otherwise exactly the two statements."""
        if isinstance(output, LazyPrior) and not output.materialized:
            from . import fused_mll, settings
            if output._model is self.model and self.likelihood is self.model.likelihood and settings.fused_training.on():
                loss = fused_mll.value_and_grad(self.model, self.likelihood, target, negate=True)
                if loss is not None:
                    return loss
        loss = self.negative(output, target)
        loss.backward()
        return loss

    def forward(self, output, target):
        n = target.shape[0]
        if isinstance(output, LazyPrior) and not output.materialized:
            # (a LazyPrior only exists because fused_mll.applicable(model) held a moment ago, with the model's own likelihood)
            from . import fused_mll, settings
            if output._model is self.model and self.likelihood is self.model.likelihood and settings.fused_training.on():
                return fused_mll.evaluate(self.model, self.likelihood, target)
        noise = self.likelihood.noise.reshape(())
        if isinstance(output, TrainPosterior):
            res = self.likelihood(output).log_prob(target) + self.likelihood.log_prior().to(target.dtype)
            return res / n
        cov = output.covariance
        if isinstance(cov, AdditiveRPOperator):
            r = target - output.mean
            # the two scalars the kernels take by value, in ONE device-to-host copy per step (hostvals)
            prefetch(cov.outputscale, noise)
            inv_quad, logdet = inv_quad_logdet(cov, noise, r)
            log_prob = -0.5 * (inv_quad + logdet + n * LOG2PI)
        elif isinstance(cov, DenseKernelOperator):
            K = cov.to_dense_autograd() + noise * torch.eye(n, dtype=target.dtype, device=target.device)
            log_prob = _DenseGaussianLogProb.apply(K, target - output.mean)
        else:
            # posterior (eval-mode) output: dense covariance or marginal variances
            log_prob = self.likelihood(output).log_prob(target)
        res = log_prob + self.likelihood.log_prior().to(log_prob.dtype)
        return res / n
