import sys, time, torch
sys.path.insert(0, '/root/repo')
import rpgp_amd
from rpgp_amd import ops
from rpgp_amd.precond import blocked_cholesky
dev = torch.device("cuda:0")
N = 50000
g = torch.Generator().manual_seed(0)
Z = torch.randn(N, 20, generator=g).to(dev)
K = ops.dense(Z, Z, 0.05); K.diagonal().add_(0.1)
torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter(); L, info = blocked_cholesky(K); torch.cuda.synchronize(); print("blocked call", i, round(time.perf_counter() - t0, 3)); del L
for i in range(2):
    t0 = time.perf_counter(); L = torch.linalg.cholesky_ex(K)[0]; torch.cuda.synchronize(); print("library call", i, round(time.perf_counter() - t0, 3)); del L
