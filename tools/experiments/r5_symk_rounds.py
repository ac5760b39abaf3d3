"""EXPERIMENT: wide packed-cache product (T = 11) with exactly-filled rounds of resident workgroups (1, 2, 3, 4 rounds) against
the N / 28 rule at the large sizes (RPGP_SYMK_ONE_ROUND_MAX / RPGP_SYMK_ROUNDS knobs)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for N in [int(a) for a in (sys.argv[1:] or ["20000", "28001", "33000", "40000", "50000"])]:
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, 20, generator=g).to(dev)
    V = torch.randn(N, 11, generator=g).to(dev)
    C = ops.SymCache(Z, wide=True)
    rec = {"N": N}
    for rep in range(2):
        for cfg in ("rule", "1", "2", "3", "4", "6"):
            if cfg == "rule":
                os.environ["RPGP_SYMK_ONE_ROUND_MAX"] = "28000"; os.environ["RPGP_SYMK_ROUNDS"] = "1"
            else:
                os.environ["RPGP_SYMK_ONE_ROUND_MAX"] = "1000000"; os.environ["RPGP_SYMK_ROUNDS"] = cfg
            for _ in range(3):
                ops.symcache_mvm(C, V, 0.05, 0.1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                ops.symcache_mvm(C, V, 0.05, 0.1)
            torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
            key = "rule_us" if cfg == "rule" else "rounds%s_us" % cfg
            rec[key] = round(min(us, rec.get(key, 1e30)), 1)
    print(json.dumps(rec), flush=True)
