import sys, time; sys.path.insert(0, "/root/repo")
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for N, T in ((14939, 14939), (14939, 1660), (50000, 2000)):
    Z = torch.randn(N, 20, generator=torch.Generator().manual_seed(0)).to(dev)
    V = torch.randn(N, T, device=dev)
    gp = ops.ski_grid(Z, None, 1024)
    ops.ski_mvm(Z, Z, gp, V[:, :64].contiguous(), 0.05, 0.1, 1024); torch.cuda.synchronize()
    t0 = time.perf_counter(); ops.ski_mvm(Z, Z, gp, V, 0.05, 0.1, 1024); torch.cuda.synchronize(); t_ski = time.perf_counter() - t0
    K = ops.dense(Z, Z, 0.05); torch.cuda.synchronize()
    t0 = time.perf_counter(); K @ V; torch.cuda.synchronize(); t_d = time.perf_counter() - t0
    t0 = time.perf_counter(); K @ V; torch.cuda.synchronize(); t_d = time.perf_counter() - t0
    print("N", N, "T", T, "ski wide %.1f ms" % (t_ski * 1e3), "dense GEMM %.1f ms" % (t_d * 1e3))
    del K
