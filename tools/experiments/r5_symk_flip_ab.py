"""Wide packed-cache product (T = 11), caches of 1 - 10 x the Infinity Cache: nontemporal loads / default-policy loads, each
with and without alternating the walking direction of consecutive products (RPGP_SYMK_FLIP).  JSON lines."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for N in [int(a) for a in (sys.argv[1:] or ["11500", "13000", "14939", "18000", "22000", "28000", "36000", "50000"])]:
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, 20, generator=g).to(dev)
    V = torch.randn(N, 11, generator=g).to(dev)
    C = ops.SymCache(Z, wide=True)
    rec = {"N": N, "cache_MB": round(C.nbytes / 1e6, 1)}
    ref = None
    for rep in range(3):
        for nt in ("1", "0"):
            for flip in ("0", "1"):
                os.environ["RPGP_SYMK_NT"] = nt; os.environ["RPGP_SYMK_FLIP"] = flip
                for _ in range(6):
                    out = ops.symcache_mvm(C, V, 0.05, 0.1)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(30):
                    out = ops.symcache_mvm(C, V, 0.05, 0.1)
                torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 30 * 1e6
                key = ("nt" if nt == "1" else "default") + ("_flip" if flip == "1" else "") + "_us"
                rec[key] = round(min(us, rec.get(key, 1e30)), 1)
                if ref is None: ref = out
                rec["bitwise_equal"] = rec.get("bitwise_equal", True) and bool(torch.equal(ref, out))
    del C
    print(json.dumps(rec), flush=True)
