import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
J = 20
for N in (7372, 14939, 30000, 50000):
    Z = (torch.randn(N, J, generator=torch.Generator().manual_seed(0))).to(dev)
    K = ops.dense(Z, Z, 1.0 / J, pad=True); torch.cuda.synchronize()
    t0 = time.perf_counter(); K = ops.dense(Z, Z, 1.0 / J, pad=True); torch.cuda.synchronize(); tb = time.perf_counter() - t0
    prep = ops.Prepared(Z)
    for T in (1, 11):
        V = torch.randn(N, T, device=dev)
        res = {}
        for name, fn in (("torch.matmul", lambda: K @ V), ("rpgp_dense_mvm", lambda: ops.dense_mvm(K, V, 0.1)),
                         ("fused", lambda: ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1))):
            fn(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): out = fn()
            e1.record(); torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / 10
        print("N=%d T=%d build %.3f ms | torch.matmul %.3f ms | rpgp_dense_mvm %.3f ms (%.0f GB/s) | fused %.3f ms" % (
            N, T, tb * 1e3, res["torch.matmul"], res["rpgp_dense_mvm"], 4.0 * N * N / res["rpgp_dense_mvm"] / 1e6, res["fused"]), flush=True)
    del K
