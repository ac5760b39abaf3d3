"""T = 11 block MVM timing at C4 (prepared path), HIP events; checks against the direct kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for (N, J, T) in [(50000, 20, 11), (50000, 20, 1), (14939, 20, 11), (7372, 20, 11)]:
    g = torch.Generator().manual_seed(0)
    Z = (torch.randn(N, 20, generator=g) @ torch.randn(20, J, generator=g) / 20 ** 0.5).to(dev)
    V = torch.randn(N, T, generator=g).to(dev)
    prep = ops.Prepared(Z)
    o1 = ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1)
    o2 = ops.mvm_sym(Z, V, 1.0 / J, 0.1)
    rel = float((o1 - o2).norm() / o2.norm())
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    res = {}
    for name, fn in (("prepared", lambda: ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1, out=o1)), ("direct", lambda: ops.mvm_sym(Z, V, 1.0 / J, 0.1, out=o2))):
        fn(); torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 10
    print("N=%d J=%d T=%d prepared %.4f ms  direct %.4f ms  rel diff %.2e" % (N, J, T, res["prepared"], res["direct"], rel), flush=True)
