"""Quick GPU check of the matrix-core prepared MVM (rpgp_mfma.hip) against the float64 oracle and the VALU kernel."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rpgp_amd import ops
from oracle import dense_gp as orc

dev = torch.device("cuda:0")
for (N, J, T) in [(2048, 20, 1), (3001, 20, 1), (2500, 10, 1), (4097, 12, 1), (2300, 8, 1), (2200, 4, 1), (2100, 2, 1),
                  (6000, 20, 1)]:
    rng = np.random.default_rng(N)
    Z = (rng.standard_normal((N, J)) * 1.3).astype(np.float32)
    V = rng.standard_normal((N, T)).astype(np.float32)
    Zt, Vt = torch.from_numpy(Z).to(dev), torch.from_numpy(V).to(dev)
    prep = ops.Prepared(Zt)
    out = ops.mvm_sym_prepared(prep, Vt, 0.7 / J, 0.1).cpu().numpy()
    ref = orc.mvm(Z, Z, V, 0.7 / J, 0.1)
    rel = np.linalg.norm(out - ref) / np.linalg.norm(ref)
    print("N=%d J=%d T=%d fast_ok=%s rel err vs fp64 oracle %.3e" % (N, J, T, prep.fast_ok, rel), flush=True)
    assert rel < 1e-5

N, J = 50000, 20
g = torch.Generator().manual_seed(0)
Z = torch.randn(N, J, generator=g).to(dev)
V = torch.randn(N, 1, generator=g).to(dev)
prep = ops.Prepared(Z)
o1 = ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1)
o2 = ops.mvm_sym(Z, V, 1.0 / J, 0.1)
print("N=50k prepared(mfma) vs direct VALU rel diff %.3e" % float((o1 - o2).norm() / o2.norm()))
for name, fn in (("prepared", lambda: ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1)), ("direct", lambda: ops.mvm_sym(Z, V, 1.0 / J, 0.1))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print("%s: %.4f ms / MVM" % (name, (time.perf_counter() - t0) / 20 * 1e3))
