"""EXPERIMENT: wide packed-cache product with 1024-row blocks (four row-tile sets per wave: RPGP_SYMK_R4=1, whole caches above
N = 16384) against the 512-row layout: time and error against the fused sweep."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for N in [int(a) for a in (sys.argv[1:] or ["20000", "28000", "36001", "50000"])]:
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, 20, generator=g).to(dev)
    V = torch.randn(N, 11, generator=g).to(dev)
    ref = ops.mvm_sym(Z, V, 0.05, 0.1)
    rec = {"N": N}
    for r4 in ("0", "1"):
        os.environ["RPGP_SYMK_R4"] = r4
        t0 = time.perf_counter(); C = ops.SymCache(Z, wide=True); torch.cuda.synchronize()
        best = 1e30
        for rep in range(3):
            for _ in range(3): out = ops.symcache_mvm(C, V, 0.05, 0.1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): out = ops.symcache_mvm(C, V, 0.05, 0.1)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20 * 1e6)
        bb = 1e30
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); C2 = ops.SymCache(Z, wide=True); torch.cuda.synchronize(); bb = min(bb, (time.perf_counter() - t0) * 1e6); del C2
        key = "R4" if r4 == "1" else "R2"
        rec[key + "_us"] = round(best, 1); rec[key + "_build_us"] = round(bb, 1)
        rec[key + "_rel_diff_vs_fused"] = float((out - ref).norm() / ref.norm())
        del C
    print(json.dumps(rec), flush=True)
