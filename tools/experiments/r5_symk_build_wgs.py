"""EXPERIMENT: workgroup count of the packed-cache BUILD (RPGP_SYMK_BUILD_WGS knob in rpgp_symcache_build), both layouts."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for N in [int(a) for a in (sys.argv[1:] or ["7372", "14939", "28000", "50000"])]:
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, 20, generator=g).to(dev)
    V = torch.randn(N, 11, generator=g).to(dev)
    for wide in (True, False):
        rec = {"N": N, "layout": "wide" if wide else "thin"}
        ref = None
        for wgs in ("default", "256", "512", "768", "1024", "1536", "2048", "3072", "4096", "6144", "8192"):
            if wgs == "default": os.environ.pop("RPGP_SYMK_BUILD_WGS", None)
            else: os.environ["RPGP_SYMK_BUILD_WGS"] = wgs
            best = 1e30
            for rep in range(3):
                C = ops.SymCache(Z, wide=wide); torch.cuda.synchronize(); del C
                t0 = time.perf_counter()
                for _ in range(5):
                    C = ops.SymCache(Z, wide=wide)
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 5 * 1e6)
            rec["wgs_%s_us" % wgs] = round(best, 1)
            out = ops.symcache_mvm(C, V if wide else V[:, :1].contiguous(), 0.05, 0.1)
            if ref is None: ref = out
            rec["bitwise_equal"] = rec.get("bitwise_equal", True) and bool(torch.equal(ref, out))
            del C
        print(json.dumps(rec), flush=True)
