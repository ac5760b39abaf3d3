"""Preconditioned batched conjugate gradients with Lanczos tridiagonalisation (mBCG).

Re-implementation of the algorithm GPyTorch's `gpytorch.utils.linear_cg` runs for the reference (SURVEY.md §8(a)
row a9, Appendix B.2; settings at gp_experiment_runner.py:324-329):
  * right-hand-side columns are normalised to unit 2-norm, x0 = 0;
  * per iteration  alpha = (r.z)/(p.Ap),  x += alpha p,  r -= alpha Ap,  z = M^-1 r,  beta = (r+.z+)/(r.z),  p = z + beta p;
  * for the first `n_tridiag` columns and the first `max_tridiag_iter` iterations the Lanczos tridiagonal is
    T[k,k] = 1/alpha_k + beta_{k-1}/alpha_{k-1},  T[k,k-1] = T[k-1,k] = sqrt(beta_{k-1})/alpha_{k-1};
  * stop when the mean column residual norm < tolerance after >= 10 iterations (and the tridiagonal steps are done),
    or at max_iter, in which case a NumericalWarning is issued (the reference counts these warnings,
    training_routines.py:535,581).
Every vector op is a torch op on the operator's device; the only host sync is the stopping test.
"""
import warnings

import torch

from . import settings


class NumericalWarning(RuntimeWarning):
    """Issued when CG stops at max_iter without reaching the tolerance."""


# diagnostics of the most recent solves (iterations, number of right-hand sides); read by benchmarks / tests
stats = {"calls": 0, "iterations": 0, "last_iterations": 0, "last_rhs": 0}


def _tridiag_from_history(alpha, beta, n_tridiag, dtype, device):
    """Lanczos tridiagonals from the CG coefficient history (same formulas as the loop below, incl. the
    reciprocal -> 1 rule for masked alphas)."""
    import numpy as np
    m = alpha.shape[0]
    a = alpha[:, :n_tridiag].astype(np.float64)
    b = beta[:, :n_tridiag].astype(np.float64)
    inv_a = np.where(np.abs(a) > 1e-30, 1.0 / np.where(np.abs(a) > 1e-30, a, 1.0), 1.0)
    t = np.zeros((n_tridiag, m, m))
    idx = np.arange(m)
    diag = inv_a.copy()                                   # [m, n_tridiag]
    diag[1:] += b[:-1] * inv_a[:-1]
    t[:, idx, idx] = diag.T
    if m > 1:
        off = (np.sqrt(np.clip(b[:-1], 0.0, None)) * inv_a[:-1]).T
        t[:, idx[1:], idx[:-1]] = off
        t[:, idx[:-1], idx[1:]] = off
    # the tridiagonals stay on the HOST: they are only eigendecomposed (10 matrices of 20 x 20), which costs ~1 ms
    # as GPU eigh launches and microseconds on the CPU
    return torch.from_numpy(t)


class LanczosHistory:
    """The CG coefficient history of the native executor ([iters x 16] float32 host arrays, the first `n_tridiag` columns
    are the probe columns): what `linear_cg(..., lanczos="history")` returns in place of the tridiagonal matrices, for callers
    that only want the quadrature (the optimiser step: `slq_logdet` runs in the library, no matrices are laid out)."""

    def __init__(self, alpha, beta, n_tridiag, dtype, device):
        self.alpha, self.beta, self.n_tridiag, self.dtype, self.device = alpha, beta, n_tridiag, dtype, device

    def tridiagonals(self):
        return _tridiag_from_history(self.alpha, self.beta, self.n_tridiag, self.dtype, self.device)

    def slq_logdet(self, n):
        from . import backend as _backend
        return _backend.get_backend().slq_logdet_history(self.alpha, self.beta, self.n_tridiag, n)


def _native_linear_cg(operator, rhs, n_tridiag, tolerance, max_iter, max_tridiag_iter, preconditioner, check_every,
                      min_iter=10, row_sharded=False, lanczos="tridiag"):
    """Route the solve through the native mBCG executor (rpgp_mbcg_solve) when everything it needs is available;
    returns None to fall through to the torch-op loop.  A sharded operator (`native_sharding()`) runs the same
    executor: its all-reduces are issued on the launch stream by the reducer's hook, the convergence flag stays on the
    device, so there is no host synchronisation per iteration either."""
    from . import backend as _backend
    be = _backend.get_backend()
    if operator is None or not hasattr(be, "mbcg_solve") or not rhs.is_cuda or rhs.dtype != torch.float32 or \
            rhs.shape[1] > 16 or max_tridiag_iter > 64:
        return None
    shf = getattr(operator, "native_sharding", None)
    sharding = shf() if shf is not None else None
    if row_sharded != (sharding is not None and sharding[0] == "rows"):
        return None                              # local-row vectors need the row-sharded executor mode (and vice versa)
    if sharding is not None and sharding[0] == "rows" and rhs.shape[1] > 12:
        return None
    fn = getattr(operator, "native_descriptor", None)
    made = fn() if fn is not None else None
    if made is None:
        return None
    desc, keep = made
    L = Cinv = None
    sigma2 = 1.0
    if preconditioner is not None:
        if not (hasattr(preconditioner, "L") and hasattr(preconditioner, "cinv")) or preconditioner.L.shape[1] > 16:
            return None
        L, Cinv, sigma2 = preconditioner.L.contiguous(), preconditioner.cinv(), preconditioner.noise
    N, T = rhs.shape
    n_glob = N if sharding is None else int(sharding[2])
    n_iter = min(max_iter, n_glob)
    hist = min(max_tridiag_iter, n_iter) if n_tridiag else 0
    x, ah, bh, iters, mres = be.mbcg_solve(desc, rhs.contiguous(), tolerance, n_iter, min_iter=min_iter, hist_len=hist,
                                           check_every=check_every, L=L, Cinv=Cinv, sigma2=sigma2,
                                           stagnation_window=settings.cg_stagnation_window.value(), sharding=sharding)
    del keep
    stats["calls"] += 1
    stats["iterations"] += iters
    stats["last_iterations"] = iters
    stats["last_rhs"] = T
    stats["native_calls"] = stats.get("native_calls", 0) + 1
    if sharding is not None:
        stats["native_sharded_calls"] = stats.get("native_sharded_calls", 0) + 1
        if sharding[0] == "rows":
            stats["row_sharded_calls"] = stats.get("row_sharded_calls", 0) + 1
    if mres >= tolerance:                       # ran out of iterations, or stagnated at the fp32 floor
        warnings.warn(
            "CG terminated in {} iterations with average residual norm {} which is larger than the tolerance of {} "
            "specified by rpgp_amd.settings.cg_tolerance. If performance is affected, consider raising the maximum "
            "number of CG iterations by running code in a rpgp_amd.settings.max_cg_iterations(value) context."
            .format(iters, mres, tolerance), NumericalWarning)
    if n_tridiag:
        hist_ = LanczosHistory(ah, bh, n_tridiag, rhs.dtype, rhs.device)
        return x, (hist_ if lanczos == "history" else hist_.tridiagonals())
    return x


def linear_cg(matmul_closure, rhs, n_tridiag=0, tolerance=None, eps=None, stop_updating_after=1e-10, max_iter=None,
              max_tridiag_iter=None, initial_guess=None, preconditioner=None, check_every=1, operator=None,
              reduce=None, global_size=None, min_iter=10, lanczos="tridiag"):
    """Solve A X = rhs for symmetric positive definite A given as `matmul_closure`.

    rhs: (N x T).  Returns X, or (X, tridiag [n_tridiag x k x k]) when n_tridiag > 0.
    `operator` (optional): the LinearOperator behind `matmul_closure`; when it exposes `native_descriptor()` and
    T <= 16 the whole loop runs in the native executor (fused vector kernels, device-resident scalars).
    `reduce` (optional): in-place SUM all-reduce of a small tensor.  With it the vectors are a rank's LOCAL rows of a
    row-sharded system (operators.RowShardedSKIOperator): every column inner product / norm is all-reduced, so all ranks
    take identical steps and stop at the same iteration; `global_size` is the global N (iteration cap).
    `lanczos="history"`: the native executor's coefficient history (LanczosHistory) instead of the tridiagonal matrices
    (the torch-op loop below always returns matrices).
    """
    if rhs.dim() == 1:
        squeeze = True
        rhs = rhs.unsqueeze(-1)
    else:
        squeeze = False
    if tolerance is None:
        tolerance = settings.cg_tolerance.value()
    if max_iter is None:
        max_iter = settings.max_cg_iterations.value()
    if max_tridiag_iter is None:
        max_tridiag_iter = settings.max_lanczos_quadrature_iterations.value()
    if initial_guess is None and rhs.dim() == 2:
        res = _native_linear_cg(operator, rhs, n_tridiag, tolerance, max_iter, max_tridiag_iter, preconditioner,
                                check_every, min_iter=min_iter, row_sharded=reduce is not None, lanczos=lanczos)
        if res is not None:
            if squeeze:
                return (res[0].squeeze(-1), res[1]) if n_tridiag else res.squeeze(-1)
            return res
    if preconditioner is None:
        def preconditioner(x):
            return x

    if eps is None:
        # only guards divisions by exactly-vanished inner products; GPyTorch's fixed 1e-10 puts a floor of ~1e-5 on
        # the reachable residual, which is too coarse for the 1e-4 parity gates
        eps = 1e3 * torch.finfo(rhs.dtype).tiny
    N, T = rhs.shape
    n_iter = min(max_iter, (N if global_size is None else int(global_size)) + 0)
    n_tridiag_iter = min(max_tridiag_iter, n_iter)

    def colnorm(a):
        if reduce is None:
            return a.norm(2, dim=0, keepdim=True)
        sq = a.double().pow(2).sum(0, keepdim=True)
        reduce(sq)
        return sq.sqrt().to(a.dtype)

    rhs_norm = colnorm(rhs)
    rhs_is_zero = rhs_norm.lt(1e-10)
    rhs_norm = rhs_norm.masked_fill(rhs_is_zero, 1.0)
    rhs = rhs / rhs_norm

    if initial_guess is None:
        result = torch.zeros_like(rhs)
        residual = rhs.clone()
    else:
        result = initial_guess / rhs_norm
        residual = rhs - matmul_closure(result)
    bad = (~torch.isfinite(residual)).any().to(torch.float32).reshape(1)
    if reduce is not None:
        reduce(bad)                        # row-sharded: every rank must take the same decision, or the others hang
    if float(bad) > 0:
        raise RuntimeError("NaNs encountered when trying to perform matrix-vector multiplication")

    z = preconditioner(residual)
    p = z.clone()
    # wide blocks (predictive covariance: N x N_test) make every N x T temporary a GB-sized allocation; the loop below
    # updates result / residual / p in place and routes the column-wise inner products through ONE scratch buffer
    scratch = torch.empty_like(rhs)

    def coldot(a, b):
        s_ = torch.mul(a, b, out=scratch).sum(0, keepdim=True)
        if reduce is not None:
            reduce(s_)
        return s_

    rz = coldot(residual, z)

    if n_tridiag:
        t_mat = torch.zeros(n_tridiag, n_tridiag_iter, n_tridiag_iter, dtype=rhs.dtype, device=rhs.device)
        inv_alpha_prev = torch.ones(1, n_tridiag, dtype=rhs.dtype, device=rhs.device)
        beta_prev = torch.zeros(1, n_tridiag, dtype=rhs.dtype, device=rhs.device)
    update_tridiag = bool(n_tridiag)
    last_tridiag_iter = 0

    tolerance_reached = False
    best_res, since_best = float("inf"), 0
    stagnation_window = settings.cg_stagnation_window.value()
    keep_best, snap_res, snap_x = T <= 64, float("inf"), None      # (wide blocks: a GB-sized copy per improvement is not worth it)
    residual_norm = None
    min_iters = min(int(min_iter), n_iter - 1)     # GPyTorch tests the tolerance from iteration 10 on; a caller with a
                                                   # near-exact preconditioner (fp32 Cholesky factor) lowers it
    k = 0
    for k in range(n_iter):
        Ap = matmul_closure(p)
        pAp = coldot(p, Ap)
        safe = pAp.abs().gt(eps)
        alpha = torch.where(safe, rz / torch.where(safe, pAp, torch.ones_like(pAp)), torch.zeros_like(pAp))
        # columns that already converged stop moving (keeps alpha/beta finite)
        if residual_norm is not None:
            alpha = alpha.masked_fill(residual_norm.lt(stop_updating_after), 0.0)
        result.addcmul_(p, alpha)
        residual.addcmul_(Ap, alpha, value=-1.0)
        del Ap

        residual_norm = colnorm(residual).masked_fill(rhs_is_zero, 0.0)

        z = preconditioner(residual)
        rz_new = coldot(residual, z)
        safe_rz = rz.abs().gt(eps)
        beta = torch.where(safe_rz, rz_new / torch.where(safe_rz, rz, torch.ones_like(rz)), torch.zeros_like(rz))
        rz = rz_new
        p.mul_(beta).add_(z)

        if update_tridiag and k < n_tridiag_iter:
            a_t = alpha[:, :n_tridiag]
            nz = a_t.abs().gt(eps)
            # reciprocal -> 1 where alpha was masked to zero (converged column): keeps log(T) finite and the
            # decoupled trailing block carries zero quadrature weight
            inv_a = torch.where(nz, 1.0 / torch.where(nz, a_t, torch.ones_like(a_t)), torch.ones_like(a_t))
            if k == 0:
                t_mat[:, 0, 0] = inv_a[0]
            else:
                t_mat[:, k, k] = (inv_a + beta_prev * inv_alpha_prev)[0]
                off = (beta_prev.clamp_min(0).sqrt() * inv_alpha_prev)[0]
                t_mat[:, k, k - 1] = off
                t_mat[:, k - 1, k] = off
            inv_alpha_prev = inv_a
            beta_prev = beta[:, :n_tridiag].clone()
            last_tridiag_iter = k

        if k >= min_iters and (k % check_every == 0 or k == n_iter - 1):
            tridiag_pending = bool(n_tridiag) and k < min(n_tridiag_iter, n_iter) - 1
            if not tridiag_pending:
                mean_res = float(residual_norm.mean())  # host sync (the only one per iteration)
                if mean_res != mean_res:
                    raise RuntimeError("NaNs encountered in CG residuals")
                if mean_res < tolerance:
                    tolerance_reached = True
                    break
                # fp32 floor: on a badly conditioned system the recurrence stops making progress long before
                # max_cg_iterations (10 000 in the runner); give up once the best residual has not improved by 1 %
                # over `cg_stagnation_window` consecutive tests (the non-convergence warning below still fires)
                # best-iterate safeguard (see rpgp_cg.hip k_direction): keep the iterate with the smallest tested residual
                if keep_best and mean_res < snap_res:
                    snap_res = mean_res
                    snap_x = result.clone() if snap_x is None else snap_x.copy_(result)
                if snap_res < 1.0 and mean_res > 100.0 * snap_res:
                    stats["stagnated"] = stats.get("stagnated", 0) + 1
                    break
                if mean_res < 0.99 * best_res:
                    best_res, since_best = mean_res, 0
                else:
                    since_best += 1
                    if stagnation_window and since_best >= stagnation_window:
                        stats["stagnated"] = stats.get("stagnated", 0) + 1
                        break

    if not tolerance_reached and n_iter > 0:
        mean_res = float(residual_norm.mean()) if residual_norm is not None else 0.0
        if snap_x is not None and snap_res < mean_res:
            result, mean_res = snap_x, snap_res
        if mean_res >= tolerance:
            warnings.warn(
                "CG terminated in {} iterations with average residual norm {} which is larger than the tolerance of {} "
                "specified by rpgp_amd.settings.cg_tolerance. If performance is affected, consider raising the maximum "
                "number of CG iterations by running code in a rpgp_amd.settings.max_cg_iterations(value) context."
                .format(k + 1, mean_res, tolerance), NumericalWarning)

    stats["calls"] += 1
    stats["iterations"] += k + 1
    stats["last_iterations"] = k + 1
    stats["last_rhs"] = T
    if reduce is not None:
        stats["row_sharded_calls"] = stats.get("row_sharded_calls", 0) + 1
    result.mul_(rhs_norm)
    if squeeze:
        result = result.squeeze(-1)
    if n_tridiag:
        m = last_tridiag_iter + 1
        return result, t_mat[:, :m, :m]
    return result

