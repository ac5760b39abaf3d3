"""EXPERIMENT: wide packed-cache product with one row-tile set per wave (R = 1, BR = 256) at three waves per SIMD against the
R = 2 kernels.  Knobs are process-wide for the build (RPGP_SYMK_R1), so each configuration builds its own cache."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
def bench(C, V):
    for _ in range(3):
        out = ops.symcache_mvm(C, V, 0.05, 0.1)
    best = 1e30
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            out = ops.symcache_mvm(C, V, 0.05, 0.1)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20 * 1e6)
    return best, out
for N in [int(a) for a in (sys.argv[1:] or ["7372", "14939", "50000"])]:
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, 20, generator=g).to(dev)
    V = torch.randn(N, 11, generator=g).to(dev)
    rec = {"N": N}
    os.environ["RPGP_SYMK_R1"] = "0"; os.environ.pop("RPGP_SYMK_WGS", None)
    C = ops.SymCache(Z, wide=True)
    os.environ["RPGP_SYMK_WIDE_V2"] = "0"; rec["R2_three_barrier_us"], ref = bench(C, V); rec["R2_three_barrier_us"] = round(rec["R2_three_barrier_us"], 1)
    os.environ["RPGP_SYMK_WIDE_V2"] = "1"; t, o = bench(C, V); rec["R2_one_barrier_us"] = round(t, 1)
    del C
    os.environ["RPGP_SYMK_R1"] = "1"
    for wgs in ("default", "750", "1500", "0.04", "0.06", "0.09"):
        if wgs == "default": os.environ.pop("RPGP_SYMK_WGS", None)
        else: os.environ["RPGP_SYMK_WGS"] = wgs
        C = ops.SymCache(Z, wide=True)
        for mode in ("1", "2"):
            os.environ["RPGP_SYMK_WIDE_V2"] = mode
            t, o = bench(C, V)
            rec["R1_waves%s_wgs%s_us" % ("2" if mode == "1" else "3", wgs)] = round(t, 1)
            rec["maxdiff"] = max(rec.get("maxdiff", 0.0), float((o - ref).abs().max() / ref.abs().max()))
        del C
    print(json.dumps(rec), flush=True)
