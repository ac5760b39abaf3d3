"""EXPERIMENT: workgroup-count sweep of the wide packed-cache product at C4 — full one-barrier kernel and its loads-only
skeleton (RPGP_SYMK_WIDE_V2 = 1 / 5), RPGP_SYMK_WGS knob (build + product)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
g = torch.Generator().manual_seed(N)
Z = torch.randn(N, 20, generator=g).to(dev)
V = torch.randn(N, 11, generator=g).to(dev)
for wgs in ("default", "512", "1024", "1280", "1536", "1792", "2048", "2304", "2560", "3072", "3584", "4096", "4608"):
    if wgs == "default": os.environ.pop("RPGP_SYMK_WGS", None)
    else: os.environ["RPGP_SYMK_WGS"] = wgs
    C = ops.SymCache(Z, wide=True)
    rec = {"N": N, "wgs": wgs}
    for rep in range(2):
        for mode, name in (("1", "full"), ("5", "loads_only")):
            os.environ["RPGP_SYMK_WIDE_V2"] = mode
            for _ in range(3):
                ops.symcache_mvm(C, V, 0.05, 0.1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                ops.symcache_mvm(C, V, 0.05, 0.1)
            torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
            rec[name + "_us"] = round(min(us, rec.get(name + "_us", 1e30)), 1)
    del C
    print(json.dumps(rec), flush=True)
