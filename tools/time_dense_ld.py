import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
J = 20
for N in (7372, 14939, 30000):
    Z = (torch.randn(N, J, generator=torch.Generator().manual_seed(0))).to(dev)
    K0 = ops.dense(Z, Z, 1.0 / J)
    for ld in (N, (N + 3) // 4 * 4, (N + 3) // 4 * 4 + 4, (N + 63) // 64 * 64, (N + 63) // 64 * 64 + 4, (N + 63) // 64 * 64 + 16, (N + 63) // 64 * 64 + 32):
        buf = torch.empty((N, ld), device=dev)
        K = buf[:, :N]
        K.copy_(K0)
        line = "N=%d ld=%d (ld%%64=%d):" % (N, ld, ld % 64)
        for T in (1, 11):
            V = torch.randn(N, T, device=dev)
            ref = K0 @ V + 0.1 * V
            out = ops.dense_mvm(K, V, 0.1)
            err = float((out - ref).norm() / ref.norm())
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.dense_mvm(K, V, 0.1)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            line += "  T=%d %.3f ms (%.0f GB/s, err %.1e)" % (T, ms, 4.0 * N * N / ms / 1e6, err)
        print(line, flush=True)
        del buf, K
