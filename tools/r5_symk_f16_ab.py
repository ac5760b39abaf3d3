"""(Needs tools/experiments/r5_symk_fp16_pair_cache.patch applied to csrc/rpgp_kernels.hip, RPGP_SYMCACHE_WIDE16 = 2 added to
include/rpgp.h / _lib.py and `wide="packed16"` to ops.SymCache — the experiment is not part of the tree.)
Wide packed-cache product (T = 11): exact-fp32 matrix instructions on the float32 cache against the fp16x3 form on the
fp16-pair cache (RPGP_SYMCACHE_WIDE16), same process, alternating; error of both against a float64 dense product on a row sample.  JSON lines."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for N in [int(a) for a in (sys.argv[1:] or ["7372", "14939", "50000"])]:
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, 20, generator=g).to(dev)
    C = ops.SymCache(Z, wide=True)
    C16 = ops.SymCache(Z, wide="packed16")
    V = (torch.randn(N, 11, generator=g) * torch.logspace(-3, 1, 11)).to(dev)       # columns of very different scales
    V[::7, 3] *= 1e-4
    rows = torch.randperm(N, generator=g)[:512].to(dev)
    ref = (ops.dense(Z[rows], Z, 0.05).double() @ V.double()) + 0.1 * V[rows].double()
    rec = {"N": N, "T": 11}
    outs = {}
    for rep in range(2):
        for f16 in ("0", "1"):
            cc = C16 if f16 == "1" else C
            for _ in range(3):
                out = ops.symcache_mvm(cc, V, 0.05, 0.1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                out = ops.symcache_mvm(cc, V, 0.05, 0.1)
            torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
            key = "fp16x3_us" if f16 == "1" else "fp32_us"
            rec[key] = round(min(us, rec.get(key, 1e30)), 1)
            outs[f16] = out
    for f16, name in (("0", "fp32"), ("1", "fp16x3")):
        d = outs[f16][rows].double() - ref
        rec[name + "_rel_err_per_column_max"] = float((d.norm(dim=0) / ref.norm(dim=0)).max())
    rec["rel_diff_between_forms"] = float((outs["0"] - outs["1"]).norm() / outs["0"].norm())
    t0 = time.perf_counter(); ops.SymCache(Z, wide="packed16"); torch.cuda.synchronize(); rec["build16_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    t0 = time.perf_counter(); ops.SymCache(Z, wide=True); torch.cuda.synchronize(); rec["build32_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    print(json.dumps(rec), flush=True)
