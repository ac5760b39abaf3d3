"""Kernel modules mirroring the reference's operator/plugin interface for the hot path
(gp_models/kernels/scaled_projection_kernel.py, gp_models/kernels/memory_efficient_gam_kernel.py and the GPyTorch
classes assembled in training_routines.py:131-189,406).  `forward` returns a LinearOperator-shaped object
(rpgp_amd.operators) instead of a dense tensor — the same move KeOps kernels make in the reference
(gp_models/kernels/imq_kernel.py:51-58).
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from . import backend as _backend
from .distributed import JShard, RowShard
from .operators import (AdditiveRPOperator, FamilyAdditiveOperator, MixedGroupOperator, SKIAdditiveOperator,
                        padded_group_size)


def inv_softplus(y):
    """float64 inverse softplus; callers cast to the parameter's dtype (so `--double` models keep full precision)."""
    y = torch.as_tensor(y, dtype=torch.float64)
    return y + torch.log(-torch.expm1(-y))


class _Project(torch.autograd.Function):
    """Z = X @ Peff on the HIP backend; backward dPeff = X^T dZ (X is data: no gradient)."""

    @staticmethod
    def forward(ctx, X, Peff):
        ctx.save_for_backward(X)
        return _backend.get_backend().project(X, Peff.detach().contiguous())

    @staticmethod
    def backward(ctx, gZ):
        (X,) = ctx.saved_tensors
        return None, _backend.get_backend().project_grad(X, gZ.contiguous())


class Kernel(nn.Module):
    """Base with the GPyTorch lengthscale parameterisation: lengthscale = softplus(raw_lengthscale), shape (1, ard)."""

    has_lengthscale = False

    def __init__(self, ard_num_dims=None, **kwargs):
        super().__init__()
        self.ard_num_dims = ard_num_dims
        if self.has_lengthscale:
            n = 1 if ard_num_dims is None else ard_num_dims
            self.raw_lengthscale = nn.Parameter(torch.zeros(1, n))

    @property
    def lengthscale(self):
        return F.softplus(self.raw_lengthscale) if self.has_lengthscale else None

    @lengthscale.setter
    def lengthscale(self, value):
        self._set_lengthscale(value)

    def _set_lengthscale(self, value):
        value = torch.as_tensor(value, dtype=self.raw_lengthscale.dtype).reshape(1, -1)
        value = value.expand_as(self.raw_lengthscale) if value.numel() == 1 else value
        self.raw_lengthscale.data = inv_softplus(value).to(self.raw_lengthscale)

    def initialize(self, **kwargs):
        for name, val in kwargs.items():
            if name == "lengthscale":
                self._set_lengthscale(val)
            elif name == "outputscale":
                self._set_outputscale(val)
            elif hasattr(self, name) and isinstance(getattr(self, name), nn.Parameter):
                getattr(self, name).data = torch.as_tensor(val).to(getattr(self, name)).reshape(getattr(self, name).shape)
            else:
                raise AttributeError("Unknown parameter %s for %s" % (name, type(self).__name__))
        return self

    def __call__(self, x1, x2=None, **params):
        return self.forward(x1, x1 if x2 is None else x2, **params)


class AdditiveStructureRBFKernel(Kernel):
    """`AdditiveStructureKernel(ScaleKernel(RBFKernel(lengthscale=1), outputscale=1/J), J)` of
    training_routines.py:148-159,169-171 as ONE module: K_add = (1/J) sum_j exp(-0.5 (z1_j - z2_j)^2).
    Its parameters (inner lengthscale 1, outputscale 1/J) are frozen by the wrapper
    (scaled_projection_kernel.py:15-17), so they are plain buffers here."""

    def __init__(self, num_dims, weight=None, inner_lengthscale=1.0, ski=False, ski_options=None, kernel_type="RBF",
                 group=1):
        super().__init__()
        self.num_dims = num_dims
        # `kernel_type` of training_routines.py:47-88 (Matern nu=1.5 / InverseMQ / Cosine sub-kernels) and k-dimensional
        # RBF sub-kernels on consecutive column groups (`AdditiveKernel` of training_routines.py:172-174): the same
        # operator with another kernel-function policy (operators.FamilyAdditiveOperator)
        if kernel_type not in ("RBF", "Matern", "InverseMQ", "Cosine"):
            raise ValueError("Unknown kernel type")
        self.kernel_type = kernel_type
        self.group = int(group)
        # float64 buffers: `model.to(torch.double)` (--double) must see 1/J, not float32(1/J)
        self.register_buffer("weight", torch.tensor(1.0 / num_dims if weight is None else float(weight),
                                                    dtype=torch.float64))
        self.register_buffer("inner_lengthscale", torch.tensor(float(inner_lengthscale), dtype=torch.float64))
        # `GridInterpolationKernel(kernel, **ski_options)` wrap of training_routines.py:157-158
        self.ski = bool(ski)
        opts = dict(ski_options or {})
        if self.ski and opts.get("num_dims", 1) != 1:
            raise ValueError("only 1-D grid interpolation per projection is supported (ski_options.num_dims == 1)")
        self.grid_size = int(opts.get("grid_size", 1024))

    def _constants(self):
        """(weight, inner lengthscale) as Python floats: frozen buffers, read from the device ONCE (not once per step)."""
        c = getattr(self, "_const_cache", None)
        key = (self.weight.data_ptr(), self.weight._version, self.inner_lengthscale.data_ptr(), self.inner_lengthscale._version)
        if c is None or c[0] != key:
            c = (key, float(self.weight), float(self.inner_lengthscale))
            self._const_cache = c
        return c[1], c[2]

    def input_scale_factor(self):
        """The constant c with operator(Z) acting on c Z (fused_mll's chain rule: dZ = c d(cZ)); None when it is not one
        frozen scalar."""
        return 1.0 / self._constants()[1]

    def operator(self, Z1, Z2, outputscale=None, shard=None):
        weight_f, il = self._constants()
        if il != 1.0:
            Z1 = Z1 / il
            Z2 = None if Z2 is None else Z2 / il
        if self.ski and self.group != 1:
            raise NotImplementedError("grid interpolation is built for 1-D sub-kernels (ski_options.num_dims == 1, k == 1)")
        if (self.kernel_type != "RBF" or self.group != 1) and not self.ski:
            ncomp = Z1.shape[1] // self.group
            w = torch.full((ncomp,), weight_f, dtype=Z1.dtype, device=Z1.device)
            group = self.group
            if self.kernel_type == "RBF" and group > 1 and padded_group_size(group) != group:
                # ANY k (training_routines.py:172-174; the runner's `--k` ablation, gp_experiment_runner.py:343-354): a
                # k-dimensional RBF is the g-dimensional RBF of the same coordinates padded with g - k zero columns
                # (zero differences contribute exp(0)); g = the next instantiated group size.  The padding is a
                # differentiable torch op, so the derivative w.r.t. the real columns comes back through autograd.
                g = padded_group_size(group)

                def pad(Z):
                    if Z is None:
                        return None
                    return F.pad(Z.reshape(Z.shape[0], ncomp, group), (0, g - group)).reshape(Z.shape[0], ncomp * g)
                Z1, Z2, group = pad(Z1), pad(Z2), g
            return FamilyAdditiveOperator(Z1, Z2, outputscale=outputscale, comp_weights=w, kind=self.kernel_type,
                                          group=group)
        if self.ski:
            return SKIAdditiveOperator(Z1, Z2, outputscale=outputscale, weight=weight_f, kind=self.kernel_type,
                                       grid_size=self.grid_size, row_shard=shard if isinstance(shard, RowShard) else None)
        return AdditiveRPOperator(Z1, Z2, outputscale=outputscale, weight=weight_f,
                                  shard=shard if isinstance(shard, JShard) else None)

    def forward(self, z1, z2, **params):
        return self.operator(z1, None if z2 is z1 else z2)


class MemoryEfficientGamKernel(AdditiveStructureRBFKernel):
    """gp_models/kernels/memory_efficient_gam_kernel.py:62-69: sum of 1-D RBFs, no 1/J weight.
    Constructed bare at training_routines.py:168 (inside ScaledProjectionKernel: the DEFAULT lengthscale
    softplus(0) = ln 2, frozen) or with `ard_num_dims=d` and a trainable lengthscale as the `strictly_additive`
    GAM of training_routines.py:214-218 (then it is called on the raw inputs: Z = X / lengthscale)."""

    has_lengthscale = True

    def __init__(self, num_dims=None, ard_num_dims=None):
        n = ard_num_dims if ard_num_dims else (num_dims if num_dims else 1)
        super().__init__(n, weight=1.0, inner_lengthscale=1.0)
        self.ard_num_dims = ard_num_dims
        self.raw_lengthscale = nn.Parameter(torch.zeros(1, ard_num_dims if ard_num_dims else 1))

    def input_scale_factor(self):
        if self.raw_lengthscale.requires_grad or self.raw_lengthscale.numel() != 1:
            return None
        c = getattr(self, "_ls_cache", None)
        key = (self.raw_lengthscale.data_ptr(), self.raw_lengthscale._version)
        if c is None or c[0] != key:
            c = (key, float(self.lengthscale.detach().reshape(-1)[0]))
            self._ls_cache = c
        return 1.0 / c[1]

    def operator(self, Z1, Z2, outputscale=None, shard=None):
        ls = self.lengthscale
        return super().operator(Z1 / ls, None if Z2 is None else Z2 / ls, outputscale=outputscale, shard=shard)

    def forward(self, x1, x2, outputscale=None, shard=None, **params):
        same = x2 is None or x2 is x1 or (x1.shape == x2.shape and x1.data_ptr() == x2.data_ptr())
        return self.operator(x1, None if same else x2, outputscale=outputscale, shard=shard)


class ScaledProjectionKernel(Kernel):
    """ARD-scale -> project -> base additive kernel (gp_models/kernels/scaled_projection_kernel.py:5-37).

    prescale:  Z = (X / lengthscale) @ P      (ard over the d input dims)
    postscale: Z = (X @ P) / lengthscale      (ard over the J projected dims)
    The base kernel's parameters are frozen and the projection is frozen unless `learn_proj`
    (scaled_projection_kernel.py:10-17; pinned by test.py:597-598,619-621)."""

    has_lengthscale = True

    def __init__(self, projection_module, base_kernel, prescale=False, ard_num_dims=None, learn_proj=False, **kwargs):
        super().__init__(ard_num_dims=ard_num_dims, **kwargs)
        self.projection_module = projection_module
        self.learn_proj = learn_proj
        if not learn_proj:
            for p in self.projection_module.parameters():
                p.requires_grad = False
        self.base_kernel = base_kernel
        for p in self.base_kernel.parameters():
            p.requires_grad = False
        self.prescale = prescale

    def effective_projection(self):
        """Peff (d x J) with the lengthscale folded in."""
        P = self.projection_module.weight.t()           # Linear.weight is J x d  (training_routines.py:144-145)
        ls = self.lengthscale.reshape(-1)
        if self.prescale:
            return P / (ls.reshape(-1, 1) if ls.numel() > 1 else ls)
        return P / (ls.reshape(1, -1) if ls.numel() > 1 else ls)

    def project(self, x):
        return _Project.apply(x.contiguous(), self.effective_projection())

    def float64_operator(self, x1, x2, outputscale):
        """The operator of a FLOAT32 model evaluated in float64 at the hyper-parameters exactly as the model holds them
        (lengthscale / outputscale = float32 softplus values, widened); x2 None = the symmetric train-train operator.  What
        the mixed-precision refinement of the prediction solves takes its residuals with, and what the predictive mean is
        summed with (models.PredictionStrategy).  None when the base kernel has no float64 form (grid interpolation,
        non-RBF / grouped sub-kernels)."""
        bk = self.base_kernel
        if not isinstance(bk, AdditiveStructureRBFKernel) or bk.ski or bk.kernel_type != "RBF" or bk.group != 1 or \
                type(bk).operator is not AdditiveStructureRBFKernel.operator:
            return None
        P = self.projection_module.weight.detach().t().double()
        ls = self.lengthscale.detach().reshape(-1).double()
        if self.prescale:
            Peff = (P / (ls.reshape(-1, 1) if ls.numel() > 1 else ls)).contiguous()
        else:
            Peff = (P / (ls.reshape(1, -1) if ls.numel() > 1 else ls)).contiguous()
        be = _backend.get_backend()
        z1 = be.project(x1.detach().double().contiguous(), Peff)
        z2 = None if x2 is None else be.project(x2.detach().double().contiguous(), Peff)
        return bk.operator(z1, z2, outputscale=outputscale.detach().double())

    def forward(self, x1, x2, outputscale=None, shard=None, **params):
        # the reference decides with torch.equal(x1, x2) (host sync per call, scaled_projection_kernel.py:22);
        # identity of the tensor objects is enough for every call site on the path
        same = x2 is None or x2 is x1 or (x1.shape == x2.shape and x1.data_ptr() == x2.data_ptr())
        z1 = self.project(x1)
        z2 = None if same else self.project(x2)
        return self.base_kernel.operator(z1, z2, outputscale=outputscale, shard=shard)


class GeneralizedProjectionKernel(Kernel):
    """The older projection family (gp_models/kernels/polynomial_projection_kernels.py:19-160):
        k(x, x') = sum_c s_c prod_{m in group c} k1((p_m(x) - p_m(x')) / l_m)
    with one base-kernel type, a projection module, equally sized multiplicative groups (`component_degrees`), one
    lengthscale per projected dimension and one output scale per additive component (trainable iff `weighted`,
    :88-98).  For the RBF the product over a group is the group's multi-dimensional RBF; for the other base kernels it is
    the product form of the runtime-(kind, group) kernels.  `forward` returns operators.FamilyAdditiveOperator."""

    def __init__(self, component_degrees, d, kernel_type, projection_module, learn_proj=False, weighted=False,
                 ski=False, ski_options=None, X=None, **kernel_kwargs):
        super().__init__()
        degrees = list(component_degrees)
        if ski and any(dg != 1 for dg in degrees):
            raise NotImplementedError("grid interpolation is built for 1-D sub-kernels only")
        if ski and dict(ski_options or {}).get("num_dims", 1) != 1:
            raise ValueError("only 1-D grid interpolation per projection is supported (ski_options.num_dims == 1)")
        self.ski = bool(ski)
        self.grid_size = int(dict(ski_options or {}).get("grid_size", 1024))
        # the reference's per-projection grid bounds (polynomial_projection_kernels.py:54-63); `ski_options["grid_rule"] =
        # "shared"` selects this build's single shared grid instead
        self.grid_rule = dict(ski_options or {}).get("grid_rule", "reference")
        if kernel_type not in ("RBF", "Matern", "InverseMQ", "Cosine"):
            raise ValueError("Unknown kernel type")
        # a group of degree k is the ProductKernel of k 1-D sub-kernels (polynomial_projection_kernels.py:70-86); for the RBF that
        # is the group's k-dimensional RBF (the tile kernels), for the other kinds the product form of the runtime-(kind,
        # group) kernels (csrc/rpgp_family_generic.hip, RPGP_KIND_PRODUCT)
        self.product = max(degrees) > 1 and kernel_type != "RBF"
        self.component_degrees = degrees
        # k: the common group size, or None for mixed sizes (general_rp_poly / multi_additive -> MixedGroupOperator)
        self.J, self.k, self.d = len(degrees), (degrees[0] if len(set(degrees)) == 1 else None), d
        self.kernel_type = kernel_type
        self.weighted = weighted
        self.learn_proj = learn_proj
        self.projection_module = projection_module
        for p in self.projection_module.parameters():
            p.requires_grad = bool(learn_proj)
        self.raw_lengthscales = nn.Parameter(torch.zeros(1, sum(degrees)))
        self.raw_outputscales = nn.Parameter(inv_softplus(torch.full((self.J,), 1.0 / self.J)).float(),
                                             requires_grad=bool(weighted))

    @property
    def lengthscales(self):
        return F.softplus(self.raw_lengthscales)

    @property
    def outputscales(self):
        return F.softplus(self.raw_outputscales)

    def initialize(self, mixin_range, lengthscale_range):
        """polynomial_projection_kernels.py:139-156: normalised mixing weights, one lengthscale draw per sub-kernel
        (same order of torch.rand calls)."""
        mixins = torch.rand(self.J) * (mixin_range[1] - mixin_range[0]) + mixin_range[0]
        mixins = mixins / mixins.sum()
        self.raw_outputscales.data = inv_softplus(mixins).to(self.raw_outputscales)
        ls = torch.cat([torch.rand(1) * (lengthscale_range[1] - lengthscale_range[0]) + lengthscale_range[0]
                        for _ in range(sum(self.component_degrees))])
        self.raw_lengthscales.data = inv_softplus(ls).to(self.raw_lengthscales).reshape(1, -1)
        return self

    def effective_projection(self):
        """(d x J*k) projection with the per-dimension lengthscales folded in (a bias cancels in z - z')."""
        return self.projection_module.weight.t() / self.lengthscales.reshape(1, -1)

    def project(self, x):
        return _Project.apply(x.contiguous(), self.effective_projection())

    def forward(self, x1, x2, outputscale=None, shard=None, **params):
        same = x2 is None or x2 is x1 or (x1.shape == x2.shape and x1.data_ptr() == x2.data_ptr())
        z1 = self.project(x1)
        z2 = None if same else self.project(x2)
        if self.ski:
            # per-projection grids by the reference's rule, re-evaluated on the current projections (= its static bounds from X,
            # polynomial_projection_kernels.py:52-63, which scale with 1 / lengthscale like the coordinates do);
            # per-component output scales ride in the grid block
            return SKIAdditiveOperator(z1, z2, outputscale=outputscale, weight=1.0, grid_size=self.grid_size,
                                       comp_weights=self.outputscales, kind=self.kernel_type,
                                       row_shard=shard if isinstance(shard, RowShard) else None, grid_rule=self.grid_rule)
        if self.k is None or (self.kernel_type == "RBF" and padded_group_size(self.k) != self.k):
            # mixed group sizes, or one size the tile kernels are not instantiated for (padded with zero columns there)
            return MixedGroupOperator(z1, z2, outputscale=outputscale, comp_weights=self.outputscales,
                                      kind=self.kernel_type, degrees=self.component_degrees, product=self.product)
        return FamilyAdditiveOperator(z1, z2, outputscale=outputscale, comp_weights=self.outputscales,
                                      kind=self.kernel_type, group=self.k, product=self.product)


class PolynomialProjectionKernel(GeneralizedProjectionKernel):
    """J groups of k projections given as column blocks Ws (polynomial_projection_kernels.py:208-237)."""

    def __init__(self, J, k, d, kernel_type, Ws, bs=None, activation=None, learn_proj=False, weighted=False, ski=False,
                 ski_options=None, X=None, **kernel_kwargs):
        if activation is not None:
            raise ValueError("activation not supported through the normal projection interface. "
                             "Use the GeneralPolynomialProjectionKernel instead.")
        projection_module = nn.Linear(d, J * k, bias=False)
        projection_module.weight = nn.Parameter(torch.cat(Ws, dim=1).t().contiguous())
        super().__init__([k] * J, d, kernel_type, projection_module, learn_proj, weighted, ski, ski_options, X=X,
                         **kernel_kwargs)


class _GroupFeatures(nn.Module):
    """Column selection as a (frozen) projection: weight[i, g_i] = 1 (polynomial_projection_kernels.py:243-262)."""

    def __init__(self, order, d):
        super().__init__()
        W = torch.zeros(len(order), d)
        for i, g in enumerate(order):
            W[i, g] = 1.0
        self.register_buffer("weight", W)


class CustomAdditiveKernel(GeneralizedProjectionKernel):
    """Additive kernel over explicit feature groups (polynomial_projection_kernels.py:240-270)."""

    def __init__(self, groups, d, kernel_type, weighted=False, ski=False, ski_options=None, X=None, **kernel_kwargs):
        order = [i for g in groups for i in g]
        super().__init__([len(g) for g in groups], d, kernel_type, _GroupFeatures(order, d), learn_proj=False,
                         weighted=weighted, ski=ski, ski_options=ski_options, X=X, **kernel_kwargs)
        self.groups = groups


class StrictlyAdditiveKernel(CustomAdditiveKernel):
    """One 1-D sub-kernel per input dimension (polynomial_projection_kernels.py:273-281)."""

    def __init__(self, d, kernel_type, weighted=False, ski=False, ski_options=None, X=None, **kernel_kwargs):
        super().__init__([[i] for i in range(d)], d, kernel_type, weighted=weighted, ski=ski, ski_options=ski_options,
                         X=X, **kernel_kwargs)


class ScaleKernel(Kernel):
    """Outer `gpytorch.kernels.ScaleKernel(kernel)` of training_routines.py:406: K = softplus(raw_outputscale) * K_base."""

    def __init__(self, base_kernel):
        super().__init__()
        self.base_kernel = base_kernel
        self.raw_outputscale = nn.Parameter(torch.zeros(()))
        self.shard = None

    @property
    def outputscale(self):
        return F.softplus(self.raw_outputscale)

    @outputscale.setter
    def outputscale(self, value):
        self._set_outputscale(value)

    def _set_outputscale(self, value):
        self.raw_outputscale.data = inv_softplus(torch.as_tensor(value).reshape(())).to(self.raw_outputscale)

    def forward(self, x1, x2, **params):
        return self.base_kernel.forward(x1, x2, outputscale=self.outputscale, shard=self.shard, **params)

    def float64_operator(self, x1, x2=None):
        """float64 twin of the operator on (x1, x2) (see ScaledProjectionKernel.float64_operator), or None."""
        f = getattr(self.base_kernel, "float64_operator", None)
        if f is None or (self.shard is not None and getattr(self.shard, "world_size", 1) > 1):
            return None
        return f(x1, x2, self.outputscale)


class RBFKernel(Kernel):
    """Plain (non-additive) stationary kernel for `kind: full` (training_routines.py:275-293; BASELINE config 1 = CPU
    plumbing through the runner).  NOT the hot path: dense torch ops, any device.  `kernel_type` selects the reference's
    other full kernels (training_routines.py:47-88): Matern (nu = 1.5), InverseMQ (imq_kernel.py:8-9), Cosine."""

    has_lengthscale = True

    def __init__(self, ard_num_dims=None, kernel_type="RBF", **kwargs):
        super().__init__(ard_num_dims=ard_num_dims, **kwargs)
        if kernel_type not in ("RBF", "Matern", "InverseMQ", "Cosine"):
            raise ValueError("Unknown kernel type")
        self.kernel_type = kernel_type

    def dense(self, x1, x2):
        a = x1 / self.lengthscale
        b = x2 / self.lengthscale
        d2 = (a.pow(2).sum(-1, keepdim=True) - 2.0 * a @ b.t() + b.pow(2).sum(-1).unsqueeze(0)).clamp_min(0.0)
        if self.kernel_type == "RBF":
            return torch.exp(-0.5 * d2)
        if self.kernel_type == "InverseMQ":
            return (d2 + 1.0).rsqrt()
        r = (d2 + 1e-30).sqrt()
        if self.kernel_type == "Matern":
            return (1.0 + math.sqrt(3.0) * r) * torch.exp(-math.sqrt(3.0) * r)
        return torch.cos(math.pi * r)

    def forward(self, x1, x2, outputscale=None, **params):
        from .dense_ops import DenseKernelOperator
        return DenseKernelOperator(self, x1, x2, outputscale)
