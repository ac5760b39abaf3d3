"""The GPyTorch-side binding of the C-ABI (SURVEY.md §0 consequence 2, §8(b); INTEGRATION.md §B as code).

What a maintainer of the reference adds to run its OWN stack — `gpytorch.models.ExactGP`,
`ExactMarginalLogLikelihood`, `training_routines.py`, `gp_experiment_runner.py`, the `model_specs/*.json` files — on the
MI355X kernels: one `LazyTensor` / `LinearOperator` subclass whose protocol methods call `librpgp.so`, and a replacement
for the last line of `ScaledProjectionKernel.forward` (gp_models/kernels/scaled_projection_kernel.py:21-37), the same move
the reference's KeOps kernels make when they return a `KeOpsLazyTensor` from `forward`
(gp_models/kernels/imq_kernel.py:51-58).

GPyTorch is not installable in the build image, so the module is IMPORT-GUARDED: the subclass is created against whatever
base class is present (`linear_operator.LinearOperator` on GPyTorch >= 1.9, `gpytorch.lazy.LazyTensor` before), or, in
tests, against a stub base injected through `make_lazy_tensor_class(base)`.  Nothing here falls back to a CPU / dense
path of its own: every protocol method goes through `rpgp_amd.ops` (ctypes -> include/rpgp.h) and fails loudly without
the HIP library.

    from rpgp_amd.gpytorch_adapter import AdditiveRPLazyTensor, scaled_projection_forward
    # gp_models/kernels/scaled_projection_kernel.py, class ScaledProjectionKernel:
    #     def forward(self, x1, x2, diag=False, last_dim_is_batch=False, **params):
    #         return scaled_projection_forward(self, x1, x2, diag=diag, last_dim_is_batch=last_dim_is_batch, **params)
"""
import math

import torch

from . import ops


def find_lazy_base():
    """(base class, flavour) of the installed GPyTorch stack, or (None, None)."""
    try:
        from linear_operator.operators import LinearOperator          # GPyTorch >= 1.9
        return LinearOperator, "linear_operator"
    except Exception:
        pass
    try:
        from gpytorch.lazy import LazyTensor                           # the reference's GPyTorch (1.0 / 1.1)
        return LazyTensor, "gpytorch.lazy"
    except Exception:
        return None, None


def make_lazy_tensor_class(base):
    """`AdditiveRPLazyTensor` as a subclass of `base` (GPyTorch's LazyTensor / LinearOperator, or a test stub with the
    same constructor contract: `base.__init__(self, *representation_tensors)`)."""

    class AdditiveRPLazyTensor(base):
        """K = scale * sum_j exp(-0.5 (z1_ij - z2_i'j)^2) on projected inputs (fp32, HIP device); K is never stored.

        Z1: N x J; Z2: M x J or None for the symmetric train-train kernel; scale: 0-dim tensor (the inner ScaleKernel's
        outputscale 1/J of training_routines.py:148-159, or 1 for MemoryEfficientGamKernel).  The outer
        `gpytorch.kernels.ScaleKernel` of training_routines.py:406 multiplies by its own outputscale as usual."""

        def __init__(self, Z1, Z2, scale):
            self.symmetric = Z2 is None
            if self.symmetric:
                super().__init__(Z1, scale)
            else:
                super().__init__(Z1, Z2, scale)
            self.Z1, self.Z2, self.scale = Z1, Z2, scale
            self._prep = None

        # ---- protocol (names of both GPyTorch generations) ---------------------------------------------------------
        def _size(self):
            m = self.Z1.shape[0] if self.symmetric else self.Z2.shape[0]
            return torch.Size((self.Z1.shape[0], m))

        def _transpose_nonbatch(self):
            if self.symmetric:
                return self
            return type(self)(self.Z2, self.Z1, self.scale)

        def _diagonal(self):
            if not self.symmetric:
                raise RuntimeError("diagonal of a rectangular cross-covariance requested")
            n, J = self.Z1.shape
            return (self.scale * J).expand(n).to(self.Z1.dtype)        # k(x, x) = scale * J

        diag = _diagonal

        def _matmul(self, rhs):
            s = float(self.scale)
            z1 = self.Z1.detach()
            rhs = rhs.detach()
            if not self.symmetric:
                return ops.mvm_rect(z1, self.Z2.detach(), rhs, s)
            if self._prep is None:                # tables of the factorised fast path, once per kernel evaluation
                self._prep = ops.Prepared(z1)
            if self._prep.fast_ok:
                return ops.mvm_sym_prepared(self._prep, rhs, s, 0.0)
            return ops.mvm_sym(z1, rhs, s, 0.0)

        def _quad_form_derivative(self, left_vecs, right_vecs):
            """d/d(representation) of sum((left right^T) * K): (gZ, gscale) — what `loss.backward()` reaches through
            GPyTorch's InvQuadLogDet (fitting/optimizing.py:72)."""
            if not self.symmetric:
                raise NotImplementedError("derivatives are only needed for the train-train kernel")
            gZ, gs = ops.bilinear_grad(self.Z1.detach(), left_vecs.detach(), right_vecs.detach(), float(self.scale))
            return gZ, gs.reshape(self.scale.shape)

        _bilinear_derivative = _quad_form_derivative

        def _get_indices(self, row_index, col_index):
            """K[row_index, col_index] (the row access of GPyTorch's pivoted-Cholesky preconditioner)."""
            z2 = self.Z1 if self.symmetric else self.Z2
            uniq, inv = torch.unique(row_index, return_inverse=True)
            rows = ops.dense(self.Z1.detach().index_select(0, uniq).contiguous(), z2.detach(), float(self.scale))
            return rows[inv, col_index]

        def to_dense(self):
            z2 = self.Z1 if self.symmetric else self.Z2
            return ops.dense(self.Z1.detach(), z2.detach(), float(self.scale))

        evaluate = to_dense

    AdditiveRPLazyTensor.__qualname__ = "AdditiveRPLazyTensor"
    return AdditiveRPLazyTensor


_BASE, FLAVOUR = find_lazy_base()
AdditiveRPLazyTensor = make_lazy_tensor_class(_BASE) if _BASE is not None else None


def available():
    """True when a GPyTorch stack is importable and the subclass exists."""
    return AdditiveRPLazyTensor is not None


def describe_base_kernel(base_kernel):
    """(weight, inner_lengthscale) when `base_kernel` is one of the two additive RBF bases the reference builds —
    `AdditiveStructureKernel(ScaleKernel(RBFKernel), J)` with outputscale 1/J and lengthscale 1
    (training_routines.py:148-159,169-171) or `MemoryEfficientGamKernel` (weight 1, its own default lengthscale,
    memory_efficient_gam_kernel.py:62-69) — else None (the caller keeps the reference's dense path).  Duck-typed so that it
    works on the reference's classes without importing them."""
    name = type(base_kernel).__name__
    if name == "MemoryEfficientGamKernel":
        ls = getattr(base_kernel, "lengthscale", None)
        return 1.0, (float(ls.reshape(-1)[0]) if ls is not None else math.log(2.0))
    if name == "AdditiveStructureKernel":
        inner = getattr(base_kernel, "base_kernel", None)                 # ScaleKernel(RBFKernel)
        rbf = getattr(inner, "base_kernel", None)
        if inner is None or rbf is None or type(rbf).__name__ != "RBFKernel" or not hasattr(inner, "outputscale"):
            return None
        ls = getattr(rbf, "lengthscale", None)
        return float(inner.outputscale), (float(ls.reshape(-1)[0]) if ls is not None else 1.0)
    return None


def scaled_projection_forward(kernel, x1, x2, diag=False, last_dim_is_batch=False, lazy_cls=None, **params):
    """Drop-in body of `ScaledProjectionKernel.forward` (scaled_projection_kernel.py:21-37): the same ARD scaling and
    projection (through the reference's own torch modules, so autograd reaches the lengthscales / a learned projection),
    then the fused operator instead of the dense `base_kernel(...)` call when the base kernel is an additive RBF and a full
    (non-diagonal, non-batch) covariance is requested on a HIP device."""
    eq = x1 is x2 or (x1.shape == x2.shape and torch.equal(x1, x2))
    if kernel.prescale:
        x1 = x1.div(kernel.lengthscale)
    x1 = kernel.projection_module(x1)
    if not kernel.prescale:
        x1 = x1.div(kernel.lengthscale)
    if eq:
        x2 = x1
    else:
        if kernel.prescale:
            x2 = x2.div(kernel.lengthscale)
        x2 = kernel.projection_module(x2)
        if not kernel.prescale:
            x2 = x2.div(kernel.lengthscale)
    cls = lazy_cls if lazy_cls is not None else AdditiveRPLazyTensor
    desc = describe_base_kernel(kernel.base_kernel)
    if cls is None or desc is None or diag or last_dim_is_batch or x1.dim() != 2 or not x1.is_cuda or \
            x1.dtype != torch.float32:
        return kernel.base_kernel(x1, x2, diag=diag, last_dim_is_batch=last_dim_is_batch, **params)
    weight, inner_ls = desc
    if inner_ls != 1.0:
        x1 = x1 / inner_ls
        x2 = x1 if eq else x2 / inner_ls
    scale = torch.tensor(weight, dtype=x1.dtype, device=x1.device)
    return cls(x1.contiguous(), None if eq else x2.contiguous(), scale)
