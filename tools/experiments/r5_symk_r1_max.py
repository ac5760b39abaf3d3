"""EXPERIMENT: 256-row blocks (R = 1) beyond N = 16384 (RPGP_SYMK_R1_MAX knob): thin T = 1 / T = 4 and wide T = 11 products."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for N in [int(a) for a in (sys.argv[1:] or ["18000", "22000", "28000", "36000", "50000"])]:
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, 20, generator=g).to(dev)
    rec = {"N": N}
    for wide, T in ((False, 1), (False, 4), (True, 11)):
        V = torch.randn(N, T, generator=g).to(dev)
        for mx, tag in (("16384", "R2"), ("1000000", "R1")):
            os.environ["RPGP_SYMK_R1_MAX"] = mx
            C = ops.SymCache(Z, wide=wide)
            best = 1e30
            for rep in range(3):
                for _ in range(3): ops.symcache_mvm(C, V, 0.05, 0.1)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20): ops.symcache_mvm(C, V, 0.05, 0.1)
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20 * 1e6)
            rec["%s_T%d_%s_us" % ("wide" if wide else "thin", T, tag)] = round(best, 1)
            del C
    print(json.dumps(rec), flush=True)
