"""EXPERIMENT: one exactly filled round of the wide product sized for 2 or 3 resident workgroups per CU (RPGP_SYMK_SLOTS)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for N in [int(a) for a in (sys.argv[1:] or ["4100", "5500", "7372", "9000", "11000", "13000", "14939", "16384"])]:
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, 20, generator=g).to(dev)
    V = torch.randn(N, 11, generator=g).to(dev)
    C = ops.SymCache(Z, wide=True)
    rec = {"N": N}
    for rep in range(3):
        for slots in ("2", "3"):
            os.environ["RPGP_SYMK_SLOTS"] = slots
            for _ in range(4):
                ops.symcache_mvm(C, V, 0.05, 0.1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30):
                ops.symcache_mvm(C, V, 0.05, 0.1)
            torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 30 * 1e6
            rec["per_cu_%s_us" % slots] = round(min(us, rec.get("per_cu_%s_us" % slots, 1e30)), 1)
    print(json.dumps(rec), flush=True)
