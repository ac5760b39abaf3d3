"""One autograd node for the training objective of the flagship model.

`-mll(model(X), y)` of fitting/optimizing.py:67-72 walks, in the generic stack, ~20 autograd nodes (softplus x3, the
lengthscale division, the projection, the operator's inv_quad / log-det function, the prior, the arithmetic of the
marginal log-likelihood) with one to three tiny launches each way: outside the solve the optimiser step is bound by the
HOST time of ~130 such launches (profiles/r4_step_C2_step_gaps.txt), not by the device.  For the model every served
`additive_rp` specification builds —

    ExactGPModel( ConstantMean,  ScaleKernel( ScaledProjectionKernel( frozen Linear, AdditiveStructureRBFKernel or the
                                                                      frozen MemoryEfficientGamKernel ) ) ),
    GaussianLikelihood with (or without) the SmoothedBoxPrior on the noise                  (training_routines.py:131-189,
                                                                                              325-410; models.py:10-20)

— the objective has FOUR trainable tensors (raw lengthscale, raw outputscale, raw noise, the constant mean) and its
gradient is a handful of closed-form chain-rule lines around the same native calls.  `evaluate`
runs it as ONE `torch.autograd.Function`: forward and backward call `InvQuadLogDet`'s own forward / backward (same
preconditioner, probes, solve, SLQ, bilinear derivative — same numbers) without the autograd bookkeeping around them, the
prior and the constants are host arithmetic, and the hyper-parameter scalars reach the host in one copy.
Anything outside that shape (learned projections, component weights, sharded kernels, dense `full` kernels, the older
family) takes the generic path unchanged; `settings.fused_training(False)` forces it."""
import math

import torch
from torch.nn import functional as F

from . import backend as _backend
from . import settings
from .hostvals import host_float, prefetch, remember
from .ops import trace_range

LOG2PI = math.log(2.0 * math.pi)


class _Ctx:
    """What `InvQuadLogDet.forward / backward` need from an autograd context."""

    needs_input_grad = (True, True, True, True, False, False)
    saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


def applicable(model, likelihood=None):
    """Is `model` the flagship structure on one device?"""
    if not settings.fused_training.on():
        return False
    from .kernels import AdditiveStructureRBFKernel, ScaledProjectionKernel, ScaleKernel
    from .likelihoods import ConstantMean, GaussianLikelihood, SmoothedBoxPrior
    from .models import ExactGPModel
    if type(model) is not ExactGPModel or not isinstance(getattr(model, "mean_module", None), ConstantMean):
        return False
    lik = model.likelihood if likelihood is None else likelihood
    if lik is not model.likelihood or type(lik) is not GaussianLikelihood:
        return False
    if lik.noise_prior is not None and type(lik.noise_prior) is not SmoothedBoxPrior:
        return False
    sk = model.covar_module
    if type(sk) is not ScaleKernel or (sk.shard is not None and getattr(sk.shard, "world_size", 1) > 1):
        return False
    pk = sk.base_kernel
    if type(pk) is not ScaledProjectionKernel or pk.learn_proj:
        return False
    bk = pk.base_kernel
    from .kernels import MemoryEfficientGamKernel
    if type(bk) not in (AdditiveStructureRBFKernel, MemoryEfficientGamKernel) or bk.kernel_type != "RBF" or bk.group != 1:
        return False
    if bk.input_scale_factor() is None or any(p.requires_grad for p in bk.parameters()):
        return False
    if not isinstance(pk.projection_module, torch.nn.Linear) or pk.projection_module.bias is not None:
        return False
    x = model.train_inputs
    return torch.is_tensor(x) and x.dim() == 2 and x.dtype in (torch.float32, torch.float64)


def _prior(likelihood, noise_f):
    """(log p(sigma^2), d log p / d sigma^2) of the SmoothedBoxPrior in host arithmetic (likelihoods.SmoothedBoxPrior)."""
    pr = likelihood.noise_prior
    if pr is None:
        return 0.0, 0.0
    center, radius = 0.5 * (pr.a + pr.b), 0.5 * (pr.b - pr.a)
    dist = max(abs(noise_f - center) - radius, 0.0)
    m = 1.0 + (pr.b - pr.a) / (math.sqrt(2.0 * math.pi) * pr.sigma)
    lp = -0.5 * (dist / pr.sigma) ** 2 - math.log(pr.sigma) - 0.5 * LOG2PI - math.log(m)
    dlp = 0.0 if dist == 0.0 else -(dist / pr.sigma ** 2) * (1.0 if noise_f > center else -1.0)
    return lp, dlp


def _native_step_ok(model, target, raw_ls):
    """May the step run on the step kernels (csrc/rpgp_step.hip)?  The HIP library, float32 device tensors, the CG regime,
    probes drawn from the Woodbury preconditioner, the log-det wanted: the configuration of every timed step.  Everything else
    (float64, the Cholesky regime, a test double as backend, no preconditioner) runs the same arithmetic as torch operations."""
    from .inv_quad_logdet import use_cholesky
    be = _backend.get_backend()
    X = model.train_inputs
    return (hasattr(be, "step_hyper") and X.is_cuda and X.dtype == torch.float32 and target.dtype == torch.float32
            and raw_ls.dtype == torch.float32 and target.dim() == 1 and target.is_contiguous()
            and not use_cholesky(X.shape[0]) and not settings.skip_logdet_forward.on()
            and X.shape[0] >= settings.min_preconditioning_size.value() and settings.max_preconditioner_size.value() > 0
            and settings.num_trace_samples.value() <= 15 and settings.step_kernels.on())


class _FusedMLL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw_ls, raw_os, raw_noise, mean_c, model, likelihood, target, sign=1.0, eager=False):
        # sign = -1: the training LOSS -mll as the node's value (fitting/optimizing.py:70 negates the objective; as a separate
        # tensor operation that is a launch each way plus two autograd nodes per step: exact either way, -(a x + c) = (-a) x - c)
        ctx.sign = sign
        ctx.eager = None
        ctx.want_eager = bool(eager)
        if _native_step_ok(model, target, raw_ls):
            done = _FusedMLL._forward_native(ctx, raw_ls, raw_os, raw_noise, mean_c, model, likelihood, target)
            if done is not None:
                return done
        ctx.native = False
        from .inv_quad_logdet import InvQuadLogDet
        pk = model.covar_module.base_kernel
        bk = pk.base_kernel
        X = model.train_inputs
        n = X.shape[0]
        with torch.no_grad():
            ls = F.softplus(raw_ls).reshape(-1)
            os_ = F.softplus(raw_os).reshape(())
            noise = (F.softplus(raw_noise) + likelihood.MIN_NOISE).reshape(())
            prefetch(os_, noise)                                   # the two scalars the kernels take by value: ONE copy
            noise_f = host_float(noise)
            P = pk.projection_module.weight.t()                    # d x J  (Linear.weight is J x d)
            col = ls.reshape(-1, 1) if (pk.prescale and ls.numel() > 1) else (ls.reshape(1, -1) if ls.numel() > 1 else ls)
            Peff = (P / col).contiguous()
            Z = _backend.get_backend().project(X.contiguous(), Peff)
            op = bk.operator(Z, None, outputscale=os_, shard=None)
            op._noise_host = noise_f
            r = target - mean_c
            st = _Ctx()
            inv_quad, logdet = InvQuadLogDet.forward(st, Z, os_, noise, r, op, None)
            lp, dlp = _prior(likelihood, noise_f)
            # mll = (-0.5 (inv_quad + logdet + n log 2 pi) + log p(sigma^2)) / n      (models.ExactMarginalLogLikelihood)
            value = (inv_quad + logdet) * (sign * -0.5 / n) + (sign * (-0.5 * n * LOG2PI + lp) / n)
        ctx.st, ctx.n, ctx.dlp, ctx.prescale = st, n, dlp, pk.prescale
        ctx.zfac = bk.input_scale_factor()            # the operator acts on zfac * Z (inner lengthscale of the base kernel)
        ctx.X, ctx.P, ctx.ls, ctx.col = X, P, ls, col
        ctx.save_for_backward(raw_ls, raw_os, raw_noise)
        return value.to(raw_ls.dtype)

    @staticmethod
    def _forward_native(ctx, raw_ls, raw_os, raw_noise, mean_c, model, likelihood, target):
        """The same objective with the stretches around the solve as single launches (csrc/rpgp_step.hip)."""
        from .inv_quad_logdet import AddedDiagOperator, _probe_generator, slq_logdet, solve_operator
        from .linear_cg import linear_cg
        from .precond import WoodburyPreconditioner, build_preconditioner
        be = _backend.get_backend()
        pk = model.covar_module.base_kernel
        bk = pk.base_kernel
        X = model.train_inputs.contiguous()
        n = X.shape[0]
        W = pk.projection_module.weight                            # J x d
        p = settings.num_trace_samples.value()
        with torch.no_grad():
            with trace_range("rpgp:project+prepare"):
                Peff, hyp, os_f, noise_f, _ = be.step_hyper(raw_ls.reshape(-1), raw_os.reshape(-1), raw_noise.reshape(-1),
                                                            mean_c.reshape(-1), W, pk.prescale, likelihood.MIN_NOISE)
                os_, noise = hyp[0], hyp[1]                        # 0-dim views; their host values are known
                remember(os_, os_f)
                remember(noise, noise_f)
                Z = be.project(X, Peff)
            op = bk.operator(Z, None, outputscale=os_, shard=None)
            op._noise_host = noise_f
            khat = AddedDiagOperator(op, noise, noise_value=noise_f)
            # (building the packed cache on a side stream beside the preconditioner build was measured a draw in round 5 and
            #  is parked in tools/experiments/r6_removed_forms.patch)
            with trace_range("rpgp:preconditioner"):
                pre = build_preconditioner(op, noise_f, settings)
            if not isinstance(pre, WoodburyPreconditioner) or pre.L.dtype != torch.float32 or pre.k > 64:
                return None
            gen = _probe_generator(Z.device)
            e1 = torch.randn(pre.k, p, generator=gen, device=Z.device, dtype=Z.dtype)      # (the draws of pre.sample, same order)
            e2 = torch.randn(n, p, generator=gen, device=Z.device, dtype=Z.dtype)
            full_rhs = be.step_probes(pre.L, e1, e2, math.sqrt(noise_f), target, hyp[2:3])      # [z | y - c], unnormalised
            matmul, native_op, _ = solve_operator(op, khat, Z, noise_f, p + 1)
            solves, hist = linear_cg(matmul, full_rhs, n_tridiag=p, operator=native_op,
                                     tolerance=settings.cg_tolerance.value(), max_iter=settings.max_cg_iterations.value(),
                                     max_tridiag_iter=settings.max_lanczos_quadrature_iterations.value(),
                                     preconditioner=pre, lanczos="history")
            # The backward pass opens with M^-1 [probes] (two launches): queued HERE, they run while the host does the SLQ
            # quadrature below, instead of behind the autograd engine's start-up at the head of the backward pass.
            ctx.pre_probes = pre.solve(full_rhs[:, :p])
            logdet = float(slq_logdet(hist, n)) + pre.logdet()
            lp, dlp = _prior(likelihood, noise_f)
            # mll = (-0.5 (inv_quad + logdet + n log 2 pi) + log p(sigma^2)) / n      (models.ExactMarginalLogLikelihood)
            sign = ctx.sign
            # (the value is also posted to pinned host memory: `loss_value` below reads it without draining the stream)
            global _posted
            if hasattr(be, "step_value_wait"):
                out, ticket = be.step_value(full_rhs, solves, p, logdet, sign * -0.5 / n, sign * (-0.5 * n * LOG2PI + lp) / n,
                                            post=True)
                _posted = (out.data_ptr(), ticket)
            else:
                out = be.step_value(full_rhs, solves, p, logdet, sign * -0.5 / n, sign * (-0.5 * n * LOG2PI + lp) / n)
        ctx.native = True
        ctx.n, ctx.dlp, ctx.prescale, ctx.zfac = n, dlp, pk.prescale, bk.input_scale_factor()
        ctx.X, ctx.W, ctx.hyp, ctx.op, ctx.pre = X, W, hyp, op, pre
        ctx.solves, ctx.probes = solves, full_rhs[:, :p]          # (solves[:, :p] = Khat^-1 z: the probes were not normalised)
        ctx.shapes = (raw_ls.shape, raw_os.shape, raw_noise.shape, mean_c.shape)
        if ctx.want_eager:
            # settings.eager_gradients: the derivative queued right behind the value, with the incoming gradient 1
            ctx.eager = _FusedMLL._native_grads(ctx, _unit_gradient(X.device))
            ctx.solves = ctx.probes = ctx.pre_probes = None
        return out[0]

    @staticmethod
    def _backward_native(ctx, g):
        if ctx.eager is not None:
            # the derivative ran behind the forward pass with g = 1: scale by the real incoming gradient (x * 1.0 = x: the same
            # bits as the in-kernel form whenever the loss is differentiated directly)
            grads = list(ctx.eager)
            gd = g.reshape(()).to(grads[0].dtype)
            try:
                scaled = torch._foreach_mul(grads, gd)
            except (RuntimeError, TypeError):
                scaled = [t * gd for t in grads]
            g_ls, g_os, g_nz, g_mu = scaled
        else:
            g_ls, g_os, g_nz, g_mu = _FusedMLL._native_grads(ctx, g)
        s_ls, s_os, s_nz, s_mu = ctx.shapes
        return g_ls.reshape(s_ls), g_os.reshape(s_os), g_nz.reshape(s_nz), g_mu.reshape(s_mu), None, None, None, None, None

    @staticmethod
    def _native_grads(ctx, g):
        be = _backend.get_backend()
        n = ctx.n
        with torch.no_grad(), trace_range("rpgp:derivative"):
            g = g.reshape(1).contiguous()
            pre_probes = ctx.pre_probes if ctx.pre_probes is not None else ctx.pre.solve(ctx.probes)
            sign = ctx.sign
            left, right, part, nparts = be.step_lr(ctx.solves, pre_probes, g, sign * -0.5 / n)
            op, gs_scale = ctx.op, 1.0
            from .operators import AdditiveRPOperator
            if type(op) is AdditiveRPOperator and (op.shard is None or op.shard.world_size <= 1):
                # (the plain operator's `_bilinear_derivative` is this call followed by `gs * weight`: the weight rides into the
                #  chain-rule kernel instead of being one more launch)
                gZ, gs = be.bilinear_grad(op.Z1.detach(), left, right, op._scale)
                gs_scale = op.weight
            else:
                gZ, gs = op._bilinear_derivative(left, right)
            dPeff = be.project_grad(ctx.X, gZ.contiguous())                              # d x J:  Z = X Peff
            n_ls = (ctx.hyp.numel() - 8) // 2
            g_ls, g_os, g_nz, g_mu = be.step_hyper_backward(dPeff, ctx.W, n_ls, ctx.prescale, ctx.zfac, ctx.hyp,
                                                            gs.reshape(1), part, nparts, g, sign * -0.5 / n,
                                                            sign * ctx.dlp / n, gs_scale=gs_scale)
        return g_ls, g_os, g_nz, g_mu

    @staticmethod
    def backward(ctx, g):
        if ctx.native:
            return _FusedMLL._backward_native(ctx, g)
        from .inv_quad_logdet import InvQuadLogDet
        raw_ls, raw_os, raw_noise = ctx.saved_tensors
        n = ctx.n
        with torch.no_grad():
            sign = ctx.sign
            gq = g.reshape(()) * (sign * -0.5 / n)                 # d (sign mll) / d inv_quad = d (sign mll) / d logdet
            gZ, gs, gn, gr, _, _ = InvQuadLogDet.backward(ctx.st, gq, gq)
            be = _backend.get_backend()
            if ctx.zfac != 1.0:
                gZ = gZ * ctx.zfac
            dPeff = be.project_grad(ctx.X.contiguous(), gZ.contiguous())             # d x J:  Z = X Peff
            # Peff = P / l (rows for the prescale form, columns for the postscale form):  dl = -sum(dPeff * P) / l^2
            t = dPeff * ctx.P
            if ctx.ls.numel() > 1:
                g_ls = -(t.sum(dim=1) if ctx.prescale else t.sum(dim=0)) / (ctx.ls * ctx.ls)
            else:
                g_ls = -t.sum().reshape(1) / (ctx.ls * ctx.ls)
            g_raw_ls = (g_ls * torch.sigmoid(raw_ls.reshape(-1))).reshape(raw_ls.shape)
            g_raw_os = (gs.reshape(()) * torch.sigmoid(raw_os.reshape(()))).reshape(raw_os.shape)
            g_noise = gn.reshape(()) + g.reshape(()) * (sign * ctx.dlp / n) if ctx.dlp != 0.0 else gn.reshape(())
            g_raw_noise = (g_noise * torch.sigmoid(raw_noise.reshape(()))).reshape(raw_noise.shape)
            g_mean = (-gr.sum()).reshape(1)                         # r = y - c
        return g_raw_ls, g_raw_os, g_raw_noise, g_mean, None, None, None, None, None


_posted = None            # (data pointer of the latest native value, its host ticket): consumed by `evaluate`
_unit = {}


def _unit_gradient(device):
    t = _unit.get(device)
    if t is None:
        t = _unit[device] = torch.ones(1, dtype=torch.float32, device=device)
    return t


def evaluate(model, likelihood, target, negate=False):
    """mll(model(X), y) per datum (ExactMarginalLogLikelihood's value) as one autograd node; `negate`: the loss -mll."""
    global _posted
    pk = model.covar_module.base_kernel
    params = (pk.raw_lengthscale, model.covar_module.raw_outputscale, likelihood.raw_noise, model.mean_module.constant)
    eager = settings.eager_gradients.on() and torch.is_grad_enabled() and any(p.requires_grad for p in params)
    _posted = None
    res = _FusedMLL.apply(*params, model, likelihood, target, -1.0 if negate else 1.0, eager)
    if _posted is not None and _posted[0] == res.data_ptr():
        try:
            res._rpgp_ticket = _posted[1]             # read by `loss_value`
        except Exception:
            pass
    _posted = None
    return res


def value_and_grad(model, likelihood, target, negate=True):
    """The training loop's `loss = -mll(model(X), y); loss.backward()` (fitting/optimizing.py:70-72) without the autograd
    engine: the fused node's forward pass with the derivative queued behind it (`settings.eager_gradients`), the four gradients
    accumulated into `.grad` of the raw parameters exactly as `backward()` of the one-node graph would.  Returns the (detached)
    loss, or None when the step kernels do not serve this model / configuration — the caller then takes the autograd path.
    What it saves is host time only: a Function.apply, the engine's hand-off to its thread and back, one multi-tensor launch —
    ~120 us per optimiser step in which the device has nothing to do."""
    global _posted
    if not (settings.eager_gradients.on() and torch.is_grad_enabled() and applicable(model, likelihood)):
        return None
    pk = model.covar_module.base_kernel
    params = (pk.raw_lengthscale, model.covar_module.raw_outputscale, likelihood.raw_noise, model.mean_module.constant)
    if not all(p.requires_grad and p.is_leaf for p in params) or not _native_step_ok(model, target, params[0]):
        return None
    ctx = _Ctx()
    ctx.sign = -1.0 if negate else 1.0
    ctx.eager = None
    ctx.want_eager = True
    _posted = None
    with torch.no_grad():
        value = _FusedMLL._forward_native(ctx, params[0].detach(), params[1].detach(), params[2].detach(), params[3].detach(),
                                          model, likelihood, target)
    if value is None or ctx.eager is None:
        _posted = None
        return None
    for p, g in zip(params, ctx.eager):
        g = g.reshape(p.shape)
        p.grad = g if p.grad is None else p.grad + g
    if _posted is not None and _posted[0] == value.data_ptr():
        try:
            value._rpgp_ticket = _posted[1]
        except Exception:
            pass
    _posted = None
    return value


def loss_value(loss):
    """`loss.item()` of the training loop (fitting/optimizing.py:76).  A value of the fused objective was posted to pinned host
    memory by the kernel that formed it, at the END OF THE FORWARD pass: reading it there does not wait for the derivative and
    the optimiser update queued behind it, so the host prepares the next step while they run.  Anything else: `.item()`."""
    ticket = getattr(loss, "_rpgp_ticket", None)
    if ticket is not None:
        be = _backend.get_backend()
        wait = getattr(be, "step_value_wait", None)
        v = wait(ticket) if wait is not None else None
        if v is not None:
            return v
    return loss.item()
