"""Fused symmetric MVM of the family members at the headline shape (N = 50k, 20 columns): ms per MVM, pair-terms/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
cols = 20
Z = torch.randn(N, cols, generator=torch.Generator().manual_seed(0)).to(dev)
base_ms = None
for kind, group in (("RBF", 1), ("Matern", 1), ("InverseMQ", 1), ("Cosine", 1), ("RBF", 2), ("RBF", 5), ("RBF", 20)):
    fam = ops.Family(kind, group, torch.full((cols // group,), float(group) / cols, device=dev))
    for T in (1, 11):
        V = torch.randn(N, T, device=dev)
        for _ in range(3): ops.family_mvm_sym(fam, Z, V, 1.0, 0.1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 10
        for _ in range(reps): ops.family_mvm_sym(fam, Z, V, 1.0, 0.1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        print("%-10s group %2d T %2d  %8.3f ms   column-pair terms/s %.3e" % (kind, group, T, dt * 1e3, 0.5 * N * N * cols / dt), flush=True)
V = torch.randn(N, 1, device=dev)
for name, fn in (("hot path direct (R=2)", lambda: ops.mvm_sym(Z, V, 1.0 / cols, 0.1)),):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(name, "%.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
