"""inv_quad / log-det of Khat = K + sigma^2 I with gradients.

Two regimes, switched like GPyTorch does (Appendix B.1):
  * N <= settings.max_cholesky_size or fast_computations(log_prob=False): dense Cholesky of rpgp_dense(K);
    backward through the explicit-weight HIP derivative kernel (rpgp_bilinear_grad_dense);
  * otherwise: preconditioned mBCG on [probes | y - c] (T = num_trace_samples + 1 = 11), SLQ log-det from the
    Lanczos tridiagonals (+ log|M| of the preconditioner), backward through the fused bilinear-derivative kernel
    with L = [Khat^-1 z_p / p, -Khat^-1 r], R = [M^-1 z_p, Khat^-1 r]   (replaces `_quad_form_derivative`
    triggered by loss.backward() at fitting/optimizing.py:72).
"""
import torch

from . import settings
from .hostvals import host_float, peek
from .linear_cg import linear_cg
from .operators import AddedDiagOperator, DenseOperator, SKIAdditiveOperator, SymCachedOperator
from .precond import build_preconditioner


def _probe_generator(device):
    if settings.deterministic_probes.on():
        g = torch.Generator(device=device)
        g.manual_seed(12345)
        return g
    return None


def _check_comm(op):
    """Sharded operators: report a bounded wait of the IPC all-reduce that ran out (Reducer.check) — the kernel has already
    poisoned the result with NaN; this turns it into an exception on the rank that saw it."""
    for sh in (getattr(op, "shard", None), getattr(op, "row_shard", None)):
        if sh is not None and getattr(sh, "world_size", 1) > 1:
            sh.reducer.check()


def use_cholesky(N):
    return N <= settings.max_cholesky_size.value() or not settings.fast_computations.log_prob()


def psd_safe_cholesky(A, max_tries=4, jitter=None):
    """Cholesky with escalating jitter (GPyTorch's psd_safe_cholesky behaviour).  `jitter`: the first value tried (default
    by dtype; a float64 copy of a matrix that was COMPUTED in float32 should pass the float32 value)."""
    L, info = torch.linalg.cholesky_ex(A)
    if not bool(info.any()):
        return L
    if jitter is None:
        jitter = 1e-6 if A.dtype == torch.float32 else 1e-8
    # relative to the size of the diagonal when that exceeds one (a kernel with outputscale 50 has rounding errors 50 x those of
    # a unit-scale one; GPyTorch's absolute values are the unit-scale case)
    jitter = jitter * max(1.0, float(A.diagonal().abs().mean()))
    Aj = A.clone()
    prev = 0.0
    for i in range(max_tries):
        new = jitter * (10 ** i)
        Aj.diagonal().add_(new - prev)
        prev = new
        L, info = torch.linalg.cholesky_ex(Aj)
        if not bool(info.any()):
            import warnings
            from .linear_cg import NumericalWarning
            warnings.warn("A not p.d., added jitter of %.1e to the diagonal" % new, NumericalWarning)
            return L
    raise RuntimeError("Matrix not positive definite after repeatedly adding jitter up to %.1e" % new)


def slq_logdet(t_mat, n):
    """log|A| estimate from Lanczos tridiagonals of unit-norm probes:  (n/p) sum_p sum_m Q_p[0,m]^2 log lambda_pm.
    `t_mat`: the matrices [p x m x m], or the native executor's coefficient history (linear_cg.LanczosHistory: the
    quadrature then runs in the library, rpgp_slq_logdet)."""
    if hasattr(t_mat, "slq_logdet"):
        return t_mat.slq_logdet(n)
    evals, evecs = torch.linalg.eigh(t_mat.double())
    evals = evals.clamp_min(1e-30)
    w = evecs[:, 0, :] ** 2
    return float(n) * (w * torch.log(evals)).sum(-1).mean()


def solve_operator(op, khat, Z, noise_f, width):
    """(matmul closure, operator for the native executor, sharded?) of the CG solve on Khat: the cached forms when they fit.
    cached-K mode: evaluate the kernel once per hyper-parameter step so each CG iteration on the
    T = 11 block is one HBM-bound pass over stored values; the backward pass stays fused.  Preferred form: the packed
    symmetric cache (every unordered pair once, half the bytes and half the build); otherwise the dense matrix (rpgp_dense).
    Multi-GPU (pair-sharding): every rank caches its own 1/world of the pairs and the partial products are summed by the same
    single all-reduce per MVM as the fused sharded sweep."""
    N = Z.shape[0]
    matmul = khat._matmul
    native_op = khat
    sharded = op.shard is not None and op.shard.world_size > 1
    if not isinstance(op, SKIAdditiveOperator) and Z.dtype == torch.float32:
        per_rank = 2.0 / (op.shard.world_size if sharded else 1)
        cache = op.to_symcache(wide=width > 4) if hasattr(op, "to_symcache") and \
            settings.use_cached_kernel(N, Z.device, per_rank) else None
        if cache is not None:
            native_op = SymCachedOperator(cache, op._scale, noise_f,
                                          diag_value=op._scale * op.num_projections, shard=op.shard if sharded else None)
            matmul = native_op._matmul
        elif not sharded and settings.use_cached_kernel(N, Z.device):
            native_op = DenseOperator(op.to_dense_cached(), noise_f)
            matmul = native_op._matmul
    return matmul, native_op, sharded


class InvQuadLogDet(torch.autograd.Function):
    """(inv_quad, logdet) = (r^T Khat^-1 r, log|Khat|) for Khat = K(Z) * outputscale + noise I."""

    @staticmethod
    def forward(ctx, Z, outputscale, noise, rhs, op, comp_weights=None):
        # `op` is the AdditiveRPOperator built on (Z, outputscale); passed as a non-tensor argument
        N = Z.shape[0]
        noise_f = op._noise_host if getattr(op, "_noise_host", None) is not None else host_float(noise)
        khat = AddedDiagOperator(op, noise.detach(), noise_value=noise_f)
        r = rhs.detach().reshape(N, 1)
        ctx.op = op
        ctx.N = N
        if use_cholesky(N):
            Kd = khat.to_dense()
            Lc = psd_safe_cholesky(Kd)
            alpha = torch.cholesky_solve(r, Lc)
            inv_quad = (r * alpha).sum()
            logdet = 2.0 * torch.log(Lc.diagonal()).sum()
            ctx.mode = "chol"
            ctx.save_for_backward(Lc, alpha)
            return inv_quad, logdet

        num_probes = settings.num_trace_samples.value()
        if getattr(op, "row_shard", None) is not None:
            return _row_sharded_forward(ctx, Z, noise, r, op, num_probes)
        pre = build_preconditioner(op, noise_f, settings)
        gen = _probe_generator(Z.device)
        if pre is not None:
            probes = pre.sample(num_probes, generator=gen)
        else:
            probes = torch.randn(N, num_probes, generator=gen, device=Z.device, dtype=Z.dtype)
        if op.shard is not None and op.shard.world_size > 1:
            # every rank must run the identical CG recurrences: rank 0's probes are broadcast
            import torch.distributed as dist
            dist.broadcast(probes, src=0, group=op.shard.group)
        probe_norms = probes.norm(2, dim=0, keepdim=True)
        probes_n = probes / probe_norms
        full_rhs = torch.cat([probes_n, r], dim=1)
        matmul, native_op, sharded = solve_operator(op, khat, Z, noise_f, full_rhs.shape[1])
        solves, t_mat = linear_cg(matmul, full_rhs, n_tridiag=num_probes, operator=native_op,
                                  tolerance=settings.cg_tolerance.value(),
                                  max_iter=settings.max_cg_iterations.value(),
                                  max_tridiag_iter=settings.max_lanczos_quadrature_iterations.value(),
                                  preconditioner=pre, lanczos="history")
        alpha = solves[:, num_probes:]
        inv_quad = (r * alpha).sum()
        if settings.skip_logdet_forward.on():
            logdet = torch.zeros((), dtype=Z.dtype, device=Z.device)
        else:
            # log|M| of the preconditioner is read AFTER the solve (its value went to pinned host memory when the preconditioner
            # was built: no copy, no synchronisation here); the value goes back as a fill, not as a host-to-device copy
            logdet_correction = pre.logdet() if pre is not None else 0.0
            logdet = torch.full((), float(slq_logdet(t_mat, N)) + logdet_correction, dtype=Z.dtype, device=Z.device)
        ctx.mode = "cg"
        ctx.pre = pre
        ctx.num_probes = num_probes
        probe_solves = solves[:, :num_probes] * probe_norms          # Khat^-1 z_p
        ctx.save_for_backward(probe_solves, probes, alpha)
        if sharded:
            _check_comm(op)
        return inv_quad, logdet

    @staticmethod
    def backward(ctx, g_inv_quad, g_logdet):
        op = ctx.op
        need = ctx.needs_input_grad
        gZ = gs = gn = gr = gw = None
        if ctx.mode == "chol":
            Lc, alpha = ctx.saved_tensors
            Kinv = torch.cholesky_inverse(Lc)
            # d(g_iq * inv_quad + g_ld * logdet) = sum_{ii'} (g_ld Kinv - g_iq alpha alpha^T)[i,i'] dKhat[i,i']
            S = g_logdet * Kinv - g_inv_quad * (alpha @ alpha.t())
            if need[0] or need[1] or need[5]:
                gZ, gs, *rest = op.dense_weight_derivative((2.0 * S).contiguous())
                gw = rest[0] if rest else None
            if need[2]:
                gn = S.diagonal().sum()
            if need[3]:
                gr = (2.0 * g_inv_quad * alpha).reshape(-1)
        elif ctx.mode == "cg_rows":
            return _row_sharded_backward(ctx, g_inv_quad, g_logdet)
        else:
            probe_solves, probes, alpha = ctx.saved_tensors
            p = ctx.num_probes
            pre_probes = ctx.pre.solve(probes) if ctx.pre is not None else probes
            left = torch.cat([probe_solves * (g_logdet / p), -g_inv_quad * alpha], dim=1).contiguous()
            right = torch.cat([pre_probes, alpha], dim=1).contiguous()
            if need[0] or need[1] or need[5]:
                gZ, gs, *rest = op._bilinear_derivative(left, right)
                gw = rest[0] if rest else None
            if need[2]:
                gn = (left * right).sum()
            if need[3]:
                gr = (2.0 * g_inv_quad * alpha).reshape(-1)
        if gs is not None:
            gs = gs.reshape(ctx.op.outputscale.shape)
        if gn is not None:
            gn = gn.reshape(())
        if gw is not None:
            gw = gw.reshape(ctx.op.comp_weights.shape).to(ctx.op.comp_weights.dtype)
        _check_comm(op)
        return gZ, gs, gn, gr, None, gw


def _shared_generator(device, shard):
    """A generator every rank seeds identically: the fixed one of `deterministic_probes`, else a seed drawn by rank 0 and
    broadcast (8 bytes instead of the N x p probe block)."""
    gen = _probe_generator(device)
    if gen is not None:
        return gen
    seed = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64).to(device)
    shard.broadcast_(seed, 0)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    return gen


def _row_sharded_forward(ctx, Z, noise, r, op, num_probes):
    """mBCG + SLQ with the N rows split over the ranks (SKI operator; replaces the MultiDeviceKernel split of
    training_routines.py:407-408).  Every rank solves for ITS rows of [probes | y - c]; the grid histogram and the inner
    products are all-reduced inside the solve (native executor: on the launch stream), so alpha / beta — hence the Lanczos
    tridiagonals and the SLQ log-determinant — are identical on every rank.  The probes are the ones the single-process
    path draws (same generator, same order: e1 then e2), sliced to the local rows."""
    from .operators import row_sharded_preconditioner
    import math
    rs = op.row_shard
    N = Z.shape[0]
    noise_f = float(noise.detach())
    sop = op.row_sharded(noise_f)
    rank_k = settings.max_preconditioner_size.value()
    pre = row_sharded_preconditioner(sop, rank_k) if (N >= settings.min_preconditioning_size.value() and rank_k > 0) \
        else None
    gen = _shared_generator(Z.device, rs)
    if pre is not None:
        k = pre.L.shape[1]
        e1 = torch.randn(k, num_probes, generator=gen, device=Z.device, dtype=Z.dtype)
        e2 = torch.randn(N, num_probes, generator=gen, device=Z.device, dtype=Z.dtype)
        probes = pre.L @ e1 + math.sqrt(noise_f) * e2[rs.r0:rs.r1]
        logdet_correction = pre.logdet()
    else:
        probes = torch.randn(N, num_probes, generator=gen, device=Z.device, dtype=Z.dtype)[rs.r0:rs.r1].contiguous()
        logdet_correction = 0.0
    sq = probes.double().pow(2).sum(0, keepdim=True)
    rs.all_reduce_(sq, "sum")
    probe_norms = sq.sqrt().to(Z.dtype)
    r_loc = r[rs.r0:rs.r1]
    full_rhs = torch.cat([probes / probe_norms, r_loc], dim=1).contiguous()
    solves, t_mat = linear_cg(sop._matmul, full_rhs, n_tridiag=num_probes, operator=sop,
                              tolerance=settings.cg_tolerance.value(), max_iter=settings.max_cg_iterations.value(),
                              max_tridiag_iter=settings.max_lanczos_quadrature_iterations.value(), preconditioner=pre,
                              reduce=rs.all_reduce_, global_size=N)
    alpha = solves[:, num_probes:]
    iq = (r_loc.double() * alpha.double()).sum().reshape(1)
    rs.all_reduce_(iq, "sum")
    inv_quad = iq[0].to(Z.dtype)
    if settings.skip_logdet_forward.on():
        logdet = torch.zeros((), dtype=Z.dtype, device=Z.device)
    else:
        logdet = torch.as_tensor(float(slq_logdet(t_mat, N)) + logdet_correction, dtype=Z.dtype, device=Z.device)
    ctx.mode = "cg_rows"
    ctx.pre = pre
    ctx.num_probes = num_probes
    ctx.save_for_backward(solves[:, :num_probes] * probe_norms, probes, alpha)
    _check_comm(op)
    return inv_quad, logdet


def _row_sharded_backward(ctx, g_inv_quad, g_logdet):
    op, need = ctx.op, ctx.needs_input_grad
    rs = op.row_shard
    probe_solves, probes, alpha = ctx.saved_tensors
    p = ctx.num_probes
    pre_probes = ctx.pre.solve(probes) if ctx.pre is not None else probes
    left = torch.cat([probe_solves * (g_logdet / p), -g_inv_quad * alpha], dim=1).contiguous()
    right = torch.cat([pre_probes, alpha], dim=1).contiguous()
    gZ = gs = gn = gr = gw = None
    if need[0] or need[1] or need[5]:
        gZ, gs, *rest = op.row_sharded_bilinear_derivative(left, right)
        gw = rest[0] if rest else None
    if need[2]:
        gn = (left.double() * right.double()).sum().reshape(1)
        rs.all_reduce_(gn, "sum")
        gn = gn[0].to(left.dtype)
    if need[3]:
        gr = torch.zeros(ctx.N, dtype=alpha.dtype, device=alpha.device)
        gr[rs.r0:rs.r1] = (2.0 * g_inv_quad * alpha).reshape(-1)
        rs.all_reduce_(gr, "sum")
    if gs is not None:
        gs = gs.reshape(op.outputscale.shape)
    if gn is not None:
        gn = gn.reshape(())
    if gw is not None:
        gw = gw.reshape(op.comp_weights.shape).to(op.comp_weights.dtype)
    _check_comm(op)
    return gZ, gs, gn, gr, None, gw


def inv_quad_logdet(op, noise, rhs):
    """op: symmetric AdditiveRPOperator on (Z, outputscale) (both may require grad); noise: 0-dim tensor; rhs: (N,)."""
    noise_t = noise.reshape(())
    op._noise_host = peek(noise)                              # (set by hostvals.prefetch in the marginal log-likelihood)
    gr = InvQuadLogDet.apply(op.Z1, op.outputscale, noise_t, rhs, op, getattr(op, "comp_weights", None))
    return gr
