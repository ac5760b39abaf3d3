"""rpgp_amd.gpytorch_adapter: the GPyTorch binding as code.  GPyTorch cannot be
installed here, so the `LazyTensor` base is a stub with the constructor contract of the real one (`*representation`), the
trick tests/golden/make_golden.py uses for `gpytorch.kernels.Kernel`.  The GPU tests check every protocol method against
the float64 oracle through ctypes; the CPU tests check the import guard and the host-only logic."""
import math

import numpy as np
import pytest
import torch

from oracle import dense_gp as orc


class StubLazyTensor:
    """Constructor contract of gpytorch.lazy.LazyTensor / linear_operator.LinearOperator: the positional arguments are
    the representation tensors."""

    def __init__(self, *args, **kwargs):
        self._args = args

    def representation(self):
        return tuple(self._args)

    @property
    def shape(self):
        return self._size()


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def test_import_guard_and_host_logic():
    from rpgp_amd import gpytorch_adapter as ga
    base, flavour = ga.find_lazy_base()
    assert ga.available() == (base is not None)          # no GPyTorch in this image: the subclass is simply absent
    cls = ga.make_lazy_tensor_class(StubLazyTensor)
    assert issubclass(cls, StubLazyTensor) and cls.__name__ == "AdditiveRPLazyTensor"
    Z1, Z2 = torch.randn(7, 3), torch.randn(5, 3)
    s = torch.tensor(0.25)
    sq, rect = cls(Z1, None, s), cls(Z1, Z2, s)
    assert sq._size() == (7, 7) and rect._size() == (7, 5)
    assert len(sq.representation()) == 2 and len(rect.representation()) == 3
    assert sq._transpose_nonbatch() is sq
    tr = rect._transpose_nonbatch()
    assert tr._size() == (5, 7) and tr.Z1 is Z2 and tr.Z2 is Z1
    assert torch.allclose(sq.diag(), torch.full((7,), 0.75))
    with pytest.raises(RuntimeError, match="no CPU fallback"):          # the product path never computes on the host
        sq._matmul(torch.randn(7, 2))


class _Stub:
    pass


def _reference_like_kernel(P, ls, prescale, mem_efficient, J):
    """Duck-typed stand-in for the reference's ScaledProjectionKernel instance (attributes read by
    scaled_projection_kernel.py:21-37) with the two base kernels training_routines.py:148-171 builds."""
    k = _Stub()
    k.prescale = prescale
    k.lengthscale = ls
    k.projection_module = torch.nn.Linear(P.shape[0], P.shape[1], bias=False).to(P.device)
    k.projection_module.weight.data = P.t().contiguous()
    if mem_efficient:
        MemoryEfficientGamKernel = type("MemoryEfficientGamKernel", (), {})
        k.base_kernel = MemoryEfficientGamKernel()
        k.base_kernel.lengthscale = torch.tensor([[math.log(2.0)]])
    else:
        RBFKernel = type("RBFKernel", (), {})
        ScaleKernel = type("ScaleKernel", (), {})
        AdditiveStructureKernel = type("AdditiveStructureKernel", (), {})
        rbf = RBFKernel()
        rbf.lengthscale = torch.tensor([[1.0]])
        sk = ScaleKernel()
        sk.base_kernel, sk.outputscale = rbf, torch.tensor(1.0 / J)
        k.base_kernel = AdditiveStructureKernel()
        k.base_kernel.base_kernel = sk
    return k


def test_describe_base_kernel_recognises_the_reference_bases():
    from rpgp_amd import gpytorch_adapter as ga
    k = _reference_like_kernel(torch.randn(4, 6), torch.ones(1, 4), True, False, 6)
    w, ls = ga.describe_base_kernel(k.base_kernel)
    assert abs(w - 1.0 / 6) < 1e-7 and ls == 1.0
    k = _reference_like_kernel(torch.randn(4, 6), torch.ones(1, 4), True, True, 6)
    w, ls = ga.describe_base_kernel(k.base_kernel)
    assert w == 1.0 and abs(ls - math.log(2.0)) < 1e-7
    assert ga.describe_base_kernel(object()) is None


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,J,T", [(700, 333, 20, 3), (2500, 64, 8, 11)])
def test_every_protocol_method_against_the_oracle(gpu_device, N, M, J, T):
    from rpgp_amd import gpytorch_adapter as ga
    cls = ga.make_lazy_tensor_class(StubLazyTensor)
    rng = np.random.default_rng(N + J)
    Z1 = (rng.standard_normal((N, J)) * 0.8).astype(np.float32)
    Z2 = (rng.standard_normal((M, J)) * 0.8).astype(np.float32)
    V = rng.standard_normal((N, T)).astype(np.float32)
    Vm = rng.standard_normal((M, T)).astype(np.float32)
    scale = 1.0 / J
    s = torch.tensor(scale, device=gpu_device)
    z1, z2 = torch.from_numpy(Z1).to(gpu_device), torch.from_numpy(Z2).to(gpu_device)
    sq, rect = cls(z1, None, s), cls(z1, z2, s)
    K, Kr = scale * orc.additive_rbf(Z1, Z1), scale * orc.additive_rbf(Z1, Z2)
    # _matmul (square: prepared fast path; rectangular; transposed rectangular; vector right-hand side)
    assert _rel(sq._matmul(torch.from_numpy(V).to(gpu_device)).cpu().numpy(), K @ V) < 1e-5
    assert _rel(rect._matmul(torch.from_numpy(Vm).to(gpu_device)).cpu().numpy(), Kr @ Vm) < 1e-5
    assert _rel(rect._transpose_nonbatch()._matmul(torch.from_numpy(V).to(gpu_device)).cpu().numpy(), Kr.T @ V) < 1e-5
    assert _rel(sq._matmul(torch.from_numpy(V[:, 0].copy()).to(gpu_device)).cpu().numpy(), K @ V[:, 0]) < 1e-5
    # diag, _size, dense evaluation
    np.testing.assert_allclose(sq.diag().cpu().numpy(), np.diag(K), rtol=1e-6)
    assert sq._size() == (N, N) and rect._size() == (N, M)
    assert np.abs(rect.evaluate().cpu().numpy() - Kr).max() < 2e-6
    # _get_indices: arbitrary (row, col) pairs, repeated rows
    ri = torch.tensor(rng.integers(0, N, 40), device=gpu_device)
    ci = torch.tensor(rng.integers(0, N, 40), device=gpu_device)
    got = sq._get_indices(ri, ci).cpu().numpy()
    np.testing.assert_allclose(got, K[ri.cpu().numpy(), ci.cpu().numpy()], rtol=2e-5, atol=1e-6)
    # _quad_form_derivative = d/dZ, d/dscale of sum((L R^T) * K)
    L = (rng.standard_normal((N, T)) * 0.1).astype(np.float32)
    R = (rng.standard_normal((N, T)) * 0.1).astype(np.float32)
    gZ, gs = sq._quad_form_derivative(torch.from_numpy(L).to(gpu_device), torch.from_numpy(R).to(gpu_device))
    gz_ref, gs_ref = orc.bilinear_grad(Z1, L, R, scale)
    assert _rel(gZ.cpu().numpy(), gz_ref) < 2e-5
    assert abs(float(gs) - gs_ref) < 2e-5 * abs(gs_ref) + 1e-5
    assert gs.shape == s.shape and sq._bilinear_derivative is not None


@pytest.mark.gpu
@pytest.mark.parametrize("prescale,mem_efficient", [(True, False), (False, False), (True, True)])
def test_scaled_projection_forward_replaces_the_dense_call(gpu_device, prescale, mem_efficient):
    """The replacement for scaled_projection_kernel.py:37 on a duck-typed reference kernel: same ARD scaling / projection
    order as :21-36, then the fused operator; against the oracle's (X / l) P resp. (X P) / l kernel matrix."""
    from rpgp_amd import gpytorch_adapter as ga
    cls = ga.make_lazy_tensor_class(StubLazyTensor)
    g = torch.Generator().manual_seed(3)
    N, M, d, J = 900, 120, 6, 20
    X, Xs = torch.randn(N, d, generator=g), torch.randn(M, d, generator=g)
    P = torch.randn(d, J, generator=g)
    ls = torch.rand(1, d if prescale else J, generator=g) + 0.8
    k = _reference_like_kernel(P.to(gpu_device), ls.to(gpu_device), prescale, mem_efficient, J)
    inner = math.log(2.0) if mem_efficient else 1.0
    w = 1.0 if mem_efficient else 1.0 / J

    def zref(A):
        A = A.double().numpy()
        z = (A / ls.double().numpy()) @ P.double().numpy() if prescale else (A @ P.double().numpy()) / ls.double().numpy()
        return z / inner
    V = torch.randn(N, 2, generator=g)
    op = ga.scaled_projection_forward(k, X.to(gpu_device), X.to(gpu_device), lazy_cls=cls)
    assert isinstance(op, cls) and op.symmetric
    ref = w * orc.additive_rbf(zref(X), zref(X)) @ V.double().numpy()
    assert _rel(op._matmul(V.to(gpu_device)).cpu().numpy(), ref) < 2e-5
    cross = ga.scaled_projection_forward(k, Xs.to(gpu_device), X.to(gpu_device), lazy_cls=cls)
    assert isinstance(cross, cls) and not cross.symmetric and cross._size() == (M, N)
    refc = w * orc.additive_rbf(zref(Xs), zref(X)) @ V.double().numpy()
    assert _rel(cross._matmul(V.to(gpu_device)).cpu().numpy(), refc) < 2e-5
    # diag=True keeps the reference's own path (the stub base kernel records the call)
    called = {}
    type(k.base_kernel).__call__ = lambda self, a, b, **kw: called.setdefault("kw", kw) or "dense"
    out = ga.scaled_projection_forward(k, X.to(gpu_device), X.to(gpu_device), diag=True, lazy_cls=cls)
    assert called["kw"]["diag"] is True and not isinstance(out, cls)
