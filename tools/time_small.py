"""Prepared fused MVM across problem sizes: time per MVM and pair-terms/s (efficiency vs the large-N rate)."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
J = 20
sizes = [int(a) for a in sys.argv[1:]] or [1000, 2000, 4000, 7372, 14939, 30000, 50000]
for N in sizes:
    Z = torch.randn(N, J, generator=torch.Generator().manual_seed(0)).to(dev)
    prep = ops.Prepared(Z)
    for T in (1, 11):
        V = torch.randn(N, T, device=dev)
        out = torch.empty_like(V)
        for _ in range(5): ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 50
        for _ in range(reps): ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        print("N %6d T %2d  %8.1f us   pair-terms/s %.3e" % (N, T, dt * 1e6, 0.5 * N * N * J / dt), flush=True)
