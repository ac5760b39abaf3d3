"""SKI (grid-interpolation) MVM at the headline shape against the exact fused MVM: time and relative difference."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
N, d, J = 50000, 20, 20
X = torch.randn(N, d, generator=torch.Generator().manual_seed(0)).to(dev)
P = torch.randn(d, J, generator=torch.Generator().manual_seed(1)).to(dev)
Z = ops.project(X, (P / math.sqrt(d)).contiguous())
prep = ops.Prepared(Z)
for G in (256, 1024, 4096):
    gp = ops.ski_grid(Z, None, G)
    for T in (1, 11):
        V = torch.randn(N, T, device=dev)
        exact = ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1)
        approx = ops.ski_mvm(Z, Z, gp, V, 1.0 / J, 0.1, G)
        err = float((approx - exact).norm() / exact.norm())
        for _ in range(3): ops.ski_mvm(Z, Z, gp, V, 1.0 / J, 0.1, G)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): ops.ski_mvm(Z, Z, gp, V, 1.0 / J, 0.1, G)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print("G %5d T %2d  %8.1f us  rel diff vs exact %.2e" % (G, T, dt * 1e6, err), flush=True)
