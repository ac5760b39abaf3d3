"""EXPERIMENT: packed-cache products with one row-tile set per wave (R = 1, 256-row workgroups, three waves per SIMD in the wide
kernel) against R = 2 at the mid sizes, over workgroup counts (needs the RPGP_SYMK_R1 / RPGP_SYMK_WGS knobs in symk_plan)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
def bench(C, V):
    for _ in range(4):
        ops.symcache_mvm(C, V, 0.05, 0.1)
    best = 1e30
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            ops.symcache_mvm(C, V, 0.05, 0.1)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 30 * 1e6)
    return round(best, 1)
for N in [int(a) for a in (sys.argv[1:] or ["4100", "5500", "7372", "9000", "11000", "14939"])]:
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, 20, generator=g).to(dev)
    for wide, T in ((True, 11), (False, 1)):
        V = torch.randn(N, T, generator=g).to(dev)
        rec = {"N": N, "layout": "wide" if wide else "thin", "T": T}
        os.environ["RPGP_SYMK_R1"] = "0"; os.environ.pop("RPGP_SYMK_WGS", None)
        C = ops.SymCache(Z, wide=wide); rec["R2_default"] = bench(C, V); del C
        os.environ["RPGP_SYMK_R1"] = "1"
        for wgs in ("default", "384", "448", "512", "640", "768", "1024", "1536"):
            if wgs == "default": os.environ.pop("RPGP_SYMK_WGS", None)
            else: os.environ["RPGP_SYMK_WGS"] = wgs
            C = ops.SymCache(Z, wide=wide); rec["R1_wgs" + wgs] = bench(C, V); del C
        print(json.dumps(rec), flush=True)
