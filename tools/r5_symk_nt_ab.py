"""Wide and thin packed-cache products: nontemporal loads (default) against default-policy loads (RPGP_SYMK_NT=0), same
process, alternating, products back to back on the SAME cache (what a CG solve does).  JSON lines."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for N in [int(a) for a in (sys.argv[1:] or ["4100", "5500", "7372", "9000", "11000", "14939", "25000"])]:
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, 20, generator=g).to(dev)
    for wide, T in ((True, 11), (False, 1)):
        V = torch.randn(N, T, generator=g).to(dev)
        C = ops.SymCache(Z, wide=wide)
        rec = {"N": N, "layout": "wide" if wide else "thin", "T": T, "cache_MB": round(C.nbytes / 1e6, 1)}
        outs = {}
        for rep in range(3):
            for mode in ("1", "0", "auto"):
                if mode == "auto": os.environ.pop("RPGP_SYMK_NT", None)
                else: os.environ["RPGP_SYMK_NT"] = mode
                for _ in range(5):
                    out = ops.symcache_mvm(C, V, 0.05, 0.1)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(30):
                    out = ops.symcache_mvm(C, V, 0.05, 0.1)
                torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 30 * 1e6
                key = {"1": "nt_us", "0": "default_policy_us", "auto": "auto_us"}[mode]
                rec[key] = round(min(us, rec.get(key, 1e30)), 1)
                outs[mode] = out
        rec["bitwise_equal"] = bool(torch.equal(outs["0"], outs["1"]) and torch.equal(outs["0"], outs["auto"]))
        del C
        print(json.dumps(rec), flush=True)
