"""How long does a float64 Cholesky factorisation + an N_test-wide solve take on the device?  (policy input for
settings.dense_solve_size).  usage: chol_time.py N [T]"""
import sys, time, torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
Z = torch.randn(N, 20, generator=g).to(dev)
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rpgp_amd import ops
K = ops.dense(Z, Z, 0.05).double()
K.diagonal().add_(0.1)
B = torch.randn(N, T, device=dev, dtype=torch.float64)
for dt in (torch.float64, torch.float32):
    Kd = K.to(dt)
    Bd = B.to(dt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    L = torch.linalg.cholesky(Kd)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    X = torch.cholesky_solve(Bd[:, :1024].contiguous(), L)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res = float((Kd @ X - Bd[:, :1024]).norm() / Bd[:, :1024].norm())
    print("N=%d %s: cholesky %.3f s (%.1f TFLOP/s), solve of 1024 columns %.3f s, residual %.2e" % (
        N, str(dt), t1 - t0, N ** 3 / 3 / (t1 - t0) / 1e12, t2 - t1, res), flush=True)
    del L, X, Kd, Bd
