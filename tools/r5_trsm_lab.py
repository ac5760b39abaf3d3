"""Pieces of the C4 mixed-precision covariance solve (models._mp_solve): library cholesky_solve for 2048 float32 columns,
the float64 residual GEMM, the float64 copy; and a blocked forward/backward substitution whose off-diagonal updates are fp16x3
GEMMs.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
from rpgp_amd.precond import blocked_cholesky
N, C = 50000, 2048
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
Z = torch.randn(N, 20, generator=g).to(dev)
K = ops.dense(Z, Z, 0.05); K.diagonal().add_(0.1)
def timed(f, reps=2):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps, r
res = {}
t, (L, info) = timed(lambda: blocked_cholesky(K), 1); res["blocked_cholesky_s"] = round(t, 3)
B = torch.randn(N, C, generator=g).to(dev)
t, X = timed(lambda: torch.cholesky_solve(B, L)); res["cholesky_solve_2048_s"] = round(t, 3)
t, K64 = timed(lambda: K.double(), 1); res["to_double_s"] = round(t, 3)
S = X.double()
t, _ = timed(lambda: K64 @ S); res["f64_gemm_s"] = round(t, 3)
t, _ = timed(lambda: K @ X); res["f32_gemm_s"] = round(t, 3)
print(json.dumps(res), flush=True)
