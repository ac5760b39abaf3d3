"""Solver settings with the names the reference sets through `gpytorch.settings` (gp_experiment_runner.py:324-332),
plus the GPyTorch defaults it relies on implicitly (SURVEY.md §5 "Config / flags", Appendix B).

Each setting is a context manager class:  `with settings.cg_tolerance(0.01): ...` ;  `settings.cg_tolerance.value()`.
"""


class _Setting:
    _default = None
    _value = None

    def __init__(self, value):
        self._new = value
        self._old = None

    @classmethod
    def value(cls):
        return cls._default if cls._value is None else cls._value

    @classmethod
    def _set(cls, v):
        cls._value = v

    def __enter__(self):
        self._old = type(self)._value
        type(self)._set(self._new)
        return self

    def __exit__(self, *exc):
        type(self)._set(self._old)
        return False


class _Flag(_Setting):
    def __init__(self, state=True):
        super().__init__(bool(state))

    @classmethod
    def on(cls):
        return bool(cls.value())

    @classmethod
    def off(cls):
        return not cls.on()


def _setting(name, default, flag=False):
    return type(name, (_Flag if flag else _Setting,), {"_default": default, "_value": None})


cg_tolerance = _setting("cg_tolerance", 1.0)                 # gpytorch default 1; the runner sets 0.05 (--cg_tol)
eval_cg_tolerance = _setting("eval_cg_tolerance", 0.01)
max_cg_iterations = _setting("max_cg_iterations", 1000)     # the runner sets 10000
max_cholesky_size = _setting("max_cholesky_size", 800)
num_trace_samples = _setting("num_trace_samples", 10)
max_lanczos_quadrature_iterations = _setting("max_lanczos_quadrature_iterations", 20)
max_preconditioner_size = _setting("max_preconditioner_size", 15)
min_preconditioning_size = _setting("min_preconditioning_size", 2000)
max_root_decomposition_size = _setting("max_root_decomposition_size", 100)
preconditioner_tolerance = _setting("preconditioner_tolerance", 1e-3)
fast_pred_var = _setting("fast_pred_var", False, flag=True)
use_toeplitz = _setting("use_toeplitz", True, flag=True)
skip_logdet_forward = _setting("skip_logdet_forward", False, flag=True)
memory_efficient = _setting("memory_efficient", False, flag=True)
skip_posterior_variances = _setting("skip_posterior_variances", False, flag=True)
deterministic_probes = _setting("deterministic_probes", False, flag=True)   # fixed probe vectors (reproducible SLQ)
# Cached-K mode: materialise K once per hyper-parameter step / prediction strategy so that every
# CG iteration is ONE HBM-bound pass over the stored matrix (T = 11 block at N = 50k: 1.9 ms against 3.4 ms for the fused
# sweep, which recomputes N^2 J / 2 exponentials per iteration).  "auto" (default): cache whenever the stored form fits in
# `cache_kernel_fraction` of the device memory and N >= `cache_kernel_min_size` — the packed symmetric cache needs 2 N^2
# bytes (288 GB HBM3E: N <= ~190k), the dense matrix of the wide prediction solves 4 N^2 (N <= ~134k); True / False
# force it on (when it fits) / off.  Pair-sharded multi-GPU operators cache their own 1/world of the pairs (2 N^2 / world
# bytes per rank); J-sharded, SKI and float64 operators never cache.
cache_kernel = _setting("cache_kernel", "auto")
cache_kernel_fraction = _setting("cache_kernel_fraction", 0.25)
cache_kernel_min_size = _setting("cache_kernel_min_size", 4096)


def use_cached_kernel(N, device, bytes_per_entry=4.0):
    """Decision of the cached-K mode for an N x N exact operator living on `device`.  `bytes_per_entry`: 4 for the
    dense matrix, 2 for the packed symmetric cache (every unordered pair once)."""
    import torch
    mode = cache_kernel.value()
    if mode is False or mode == 0:
        return False
    dev = torch.device(device)
    if dev.type != "cuda":
        return mode is True            # (host tensors only occur under the CPU test double: honour an explicit request)
    fits = bytes_per_entry * N * N <= cache_kernel_fraction.value() * torch.cuda.get_device_properties(dev).total_memory
    if mode is True or mode == 1:
        return fits
    return fits and N >= cache_kernel_min_size.value()
tridiagonal_jitter = _setting("tridiagonal_jitter", 1e-6)
# wide-block CG (torch-op loop): stop when the best mean residual has not improved by 1 % over this many consecutive
# convergence tests (fp32 floor on badly conditioned systems); 0 disables (GPyTorch's behaviour: run to max_cg_iterations)
cg_stagnation_window = _setting("cg_stagnation_window", 200)
# wide (N_test-column) solves of the predictive covariance: when Khat has been materialised in HBM and N is at most this,
# factorise it once in float64 instead of running CG on the wide block (0 keeps CG, GPyTorch's behaviour)
dense_solve_size = _setting("dense_solve_size", 20000)
# floats per N x c block of CG state when the predictive covariance is solved in column blocks of test points (1 GiB)
predictive_block_floats = _setting("predictive_block_floats", 1 << 28)
# ... and above that, up to this size, factorise the fp32 matrix once and use the factor as the preconditioner of the wide
# block's CG (0 disables: plain pivoted-Cholesky-preconditioned CG)
cholesky_precond_size = _setting("cholesky_precond_size", 65536)
# Mixed-precision refinement of the float32 prediction solves (the mean cache Khat^-1 (y - c) and the N_test-wide block of the
# predictive covariance): `solve_refinement` rounds of  x <- x + solve32(b - Khat_64 x)  with the residual taken in float64
# (the float64 twin of the fused operator for thin blocks, a float64 copy of the stored dense matrix for wide ones) and the
# correction by the float32 solver.  float32 CG stalls at a TRUE relative residual of ~1e-4 at N = 50 000 (measured against the
# float64 oracle: predictive mean 1.8e-4, variance 8.7e-4 off); one round brings the solution to 5e-8 and the variance to 1e-5; the
# predictive mean is then summed with the float64 cross-covariance operator as well.  0 = GPyTorch's behaviour.
solve_refinement = _setting("solve_refinement", 1)
# ... the mean cache from this size up (every CG-regime model; the float64 twin product costs 4 ms at N = 15 000, 41 ms at
# 50 000); the wide block additionally only above `dense_solve_size`, where the float64 direct solve stops
solve_refinement_min_size = _setting("solve_refinement_min_size", 0)
# The training objective of the flagship model (ExactGPModel + ScaleKernel(ScaledProjectionKernel(additive RBF)) + Gaussian
# likelihood) as ONE autograd node (fused_mll.py): same native calls, ~60 fewer small launches per optimiser step.  False =
# the generic operator-by-operator autograd path for every model.
fused_training = _setting("fused_training", True, flag=True)
# ... and inside that node the stretches around the solve (hyper-parameter transforms, probe draw and normalisation, the value,
# the two sides of the derivative, the chain rule back to the raw parameters) as single launches (csrc/rpgp_step.hip) instead of
# chains of element-wise torch launches: the step is host-bound there.  False = the torch operations (same arithmetic).
step_kernels = _setting("step_kernels", True, flag=True)
# ... and the derivative launched from the FORWARD pass (with the incoming gradient 1, scaled by the real one in backward): when
# the objective is evaluated with gradients enabled, `backward()` follows — fitting/optimizing.py:70-72 — and the autograd
# engine's start-up (~100 us of host time between the value and the first launch of the derivative) is otherwise idle time of
# the device in every optimiser step.  False = the derivative is launched by backward() (a forward that is never followed by a
# backward then does not pay for it).
eager_gradients = _setting("eager_gradients", True, flag=True)
# the all-reduce of the sharded multi-GPU solve: "rccl" (torch.distributed, backend nccl = RCCL over xGMI) or "ipc" (one-shot
# kernel over IPC-mapped peer buffers, csrc/rpgp_comm.hip); the environment variable RPGP_COMM overrides it
comm_backend = _setting("comm_backend", "rccl")
# the native mBCG executor also serves sharded operators (partial products / row blocks summed by the reducer's hook on
# the launch stream); False sends sharded solves through the torch-op loop of linear_cg.py
native_sharded_cg = _setting("native_sharded_cg", True, flag=True)


class fast_computations:
    """fast_computations(covar_root_decomposition, log_prob, solves): False => dense Cholesky
    (the runner passes `not use_chol` three times, gp_experiment_runner.py:326)."""
    _state = {"covar_root_decomposition": True, "log_prob": True, "solves": True}

    def __init__(self, covar_root_decomposition=True, log_prob=True, solves=True):
        self._new = {"covar_root_decomposition": bool(covar_root_decomposition), "log_prob": bool(log_prob),
                     "solves": bool(solves)}
        self._old = None

    @classmethod
    def log_prob(cls):
        return cls._state["log_prob"]

    @classmethod
    def solves(cls):
        return cls._state["solves"]

    @classmethod
    def covar_root_decomposition(cls):
        return cls._state["covar_root_decomposition"]

    def __enter__(self):
        self._old = dict(type(self)._state)
        type(self)._state = dict(self._new)
        return self

    def __exit__(self, *exc):
        type(self)._state = self._old
        return False


class beta_features:
    checkpoint_kernel = _setting("checkpoint_kernel", 0)   # accepted for CLI parity; the fused kernel never stores K
