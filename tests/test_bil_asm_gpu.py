"""The hand-scheduled bilinear-derivative sweep (csrc/rpgp_bil_asm.hip; loop generated and CPU-self-tested by
tools/gen_bil_asm.py) through the C-ABI against the float64 oracle (oracle.dense_gp.bilinear_grad: SURVEY.md A.2,
memory_efficient_gam_kernel.py:33-59) and against the compiler-scheduled kernel it replaces (RPGP_BIL_ASM=0): ragged
sizes (rows / columns ending inside a tile, a last subtile of fewer than 64 columns), 5 / 11 / 12 right-hand-side slots,
run-to-run bit identity."""
import os

import numpy as np
import pytest
import torch

from oracle import dense_gp as orc

pytestmark = pytest.mark.gpu


def _with_asm(flag, fn):
    old = os.environ.get("RPGP_BIL_ASM")
    os.environ["RPGP_BIL_ASM"] = "1" if flag else "0"
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop("RPGP_BIL_ASM", None)
        else:
            os.environ["RPGP_BIL_ASM"] = old


@pytest.mark.parametrize("N,T", [(2048, 11), (2300, 5), (3001, 12), (4613, 11)])
def test_bilinear_asm_matches_oracle_and_compiler_kernel(gpu_device, N, T):
    from rpgp_amd import ops
    J = 20
    g = torch.Generator().manual_seed(N + T)
    Z = (torch.randn(N, J, generator=g) * 0.8)
    L = torch.randn(N, T, generator=g) * 0.1
    R = torch.randn(N, T, generator=g) * 0.1
    Zd, Ld, Rd = Z.to(gpu_device), L.to(gpu_device), R.to(gpu_device)
    gz_a, gs_a = _with_asm(True, lambda: ops.bilinear_grad(Zd, Ld, Rd, 0.05))
    gz_b, gs_b = _with_asm(True, lambda: ops.bilinear_grad(Zd, Ld, Rd, 0.05))
    gz_c, gs_c = _with_asm(False, lambda: ops.bilinear_grad(Zd, Ld, Rd, 0.05))
    assert torch.equal(gz_a, gz_b) and float(gs_a) == float(gs_b)              # deterministic
    gz_ref, gs_ref = orc.bilinear_grad(Z.double().numpy(), L.double().numpy(), R.double().numpy(), 0.05)
    for name, gz, gs in (("asm", gz_a, gs_a), ("compiler", gz_c, gs_c)):
        e = np.linalg.norm(gz.double().cpu().numpy() - gz_ref) / np.linalg.norm(gz_ref)
        assert e < 2e-5, (name, N, T, e)
        assert abs(float(gs) - gs_ref) < 2e-5 * abs(gs_ref) + 1e-5, (name, float(gs), gs_ref)
    assert float((gz_a - gz_c).norm() / gz_c.norm()) < 5e-6
