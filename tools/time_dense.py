import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
N, J = 50000, 20
Z = (torch.randn(N, J, generator=torch.Generator().manual_seed(0))).to(dev)
t0 = time.perf_counter(); K = ops.dense(Z, Z, 1.0 / J); torch.cuda.synchronize(); print("dense build s", time.perf_counter() - t0)
t0 = time.perf_counter(); K = ops.dense(Z, Z, 1.0 / J); torch.cuda.synchronize(); print("dense build s (2nd)", time.perf_counter() - t0)
for T in (1, 11, 16):
    V = torch.randn(N, T, device=dev)
    for name, fn in (("torch.matmul", lambda: K @ V), ("rpgp_dense_mvm", lambda: ops.dense_mvm(K, V, 0.1))):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): out = fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("T=%d %s: %.3f ms  (%.0f GB/s of K)" % (T, name, ms, 4.0 * N * N / ms / 1e6))
