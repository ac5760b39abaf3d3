"""Wide packed-cache product (T = 11): the three-barrier kernel (RPGP_SYMK_WIDE_V2=0) against the one-barrier kernel
(default), same process, alternating; bitwise comparison of the two results and error against a float64 dense product on a
row sample.  JSON lines.   python tools/r5_symk_wide_ab.py [N ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for N in [int(a) for a in (sys.argv[1:] or ["4100", "7372", "14939", "25001", "50000"])]:
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, 20, generator=g).to(dev)
    C = ops.SymCache(Z, wide=True)
    rec = {"N": N}
    for T in (11, 16, 5):
        V = (torch.randn(N, T, generator=g) * torch.logspace(-2, 1, T)).to(dev)
        rows = torch.randperm(N, generator=g)[:256].to(dev)
        ref = (ops.dense(Z[rows], Z, 0.05).double() @ V.double()) + 0.1 * V[rows].double()
        outs = {}
        for rep in range(3):
            for mode in ("0", "1"):
                os.environ["RPGP_SYMK_WIDE_V2"] = mode
                for _ in range(3):
                    out = ops.symcache_mvm(C, V, 0.05, 0.1)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20):
                    out = ops.symcache_mvm(C, V, 0.05, 0.1)
                torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
                key = "T%d_%s_us" % (T, {"0": "three_barrier", "1": "one_barrier"}[mode])
                rec[key] = round(min(us, rec.get(key, 1e30)), 1)
                outs[mode] = out
        rec["T%d_bitwise_equal" % T] = bool(torch.equal(outs["0"], outs["1"]))
        d = outs["1"][rows].double() - ref
        rec["T%d_rel_err_vs_f64" % T] = float(d.norm() / ref.norm())
    print(json.dumps(rec), flush=True)
