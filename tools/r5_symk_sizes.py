"""Packed-cache products over the mid sizes (wide T = 11, thin T = 1 and 4): us per product through the wrapper, error against
the fused sweep.  Used before / after the round-5 layout rule (R = 1 up to N = 16384, one-round chunks).  JSON lines."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for N in [int(a) for a in (sys.argv[1:] or ["4100", "5500", "7372", "9000", "11000", "13000", "14939", "16384", "16385", "20000", "25000", "28000", "28001", "50000"])]:
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, 20, generator=g).to(dev)
    rec = {"N": N}
    for wide, T in ((True, 11), (False, 1), (False, 4)):
        V = torch.randn(N, T, generator=g).to(dev)
        C = ops.SymCache(Z, wide=wide)
        for _ in range(4):
            out = ops.symcache_mvm(C, V, 0.05, 0.1)
        best = 1e30
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30):
                out = ops.symcache_mvm(C, V, 0.05, 0.1)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 30 * 1e6)
        key = ("wide" if wide else "thin") + "_T%d" % T
        rec[key + "_us"] = round(best, 1)
        if N <= 30000:
            ref = ops.mvm_sym(Z, V, 0.05, 0.1)
            rec[key + "_rel_diff_vs_fused"] = float((out - ref).norm() / ref.norm())
        del C
    print(json.dumps(rec), flush=True)
