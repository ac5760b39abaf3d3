"""EXPERIMENT (timing only): the wide packed-cache product with parts of its tile chain removed (RPGP_SYMK_WIDE_V2 = 4..7;
results are wrong by construction for those) — which part costs the bandwidth?"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
names = {"0": "three_barrier", "1": "one_barrier_full", "4": "row_product_only", "5": "loads_only", "6": "loads_and_lds_transposition",
         "7": "full_without_lds_round_trip", "8": "loads_only_no_transposed_slab_store"}
for N in [int(a) for a in (sys.argv[1:] or ["14939", "50000"])]:
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, 20, generator=g).to(dev)
    V = torch.randn(N, 11, generator=g).to(dev)
    C = ops.SymCache(Z, wide=True)
    rec = {"N": N, "cache_GB": round(C.nbytes / 1e9, 3)}
    for rep in range(2):
      for depth in ("8",):
        for mode, name in names.items():
            if depth == "16" and mode == "0":
                continue
            os.environ["RPGP_SYMK_WIDE_V2"] = mode
            os.environ["RPGP_SYMK_WIDE_D"] = depth
            name = name + ("_ring16" if depth == "16" else "")
            for _ in range(3):
                ops.symcache_mvm(C, V, 0.05, 0.1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                ops.symcache_mvm(C, V, 0.05, 0.1)
            torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
            rec[name + "_us"] = round(min(us, rec.get(name + "_us", 1e30)), 1)
    print(json.dumps(rec), flush=True)
