"""Rank-15 pivoted Cholesky per-step launches: the flagship fast kernel (own-row operands requested before the pivot is known;
default) against the general per-step kernel (RPGP_PIVCHOL_FAST=0); same process, alternating; JSON lines."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for N, J in ((4100, 20), (7372, 20), (14939, 20), (50000, 20), (131072, 20)):
    g = torch.Generator().manual_seed(N)
    Z = (torch.randn(N, J, generator=g) * 0.8).to(dev)
    rec = {"N": N, "J": J, "rank": 15}
    outs = {}
    for rep in range(3):
        for mode in ("0", "1"):
            os.environ["RPGP_PIVCHOL_FAST"] = mode
            for _ in range(3):
                L = ops.pivoted_cholesky(Z, 0.05, 15)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                L = ops.pivoted_cholesky(Z, 0.05, 15)
            torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
            key = "fast_us" if mode == "1" else "general_us"
            rec[key] = round(min(us, rec.get(key, 1e30)), 1)
            outs[mode] = L
    rec["bitwise_equal"] = bool(torch.equal(outs["0"], outs["1"]))
    print(json.dumps(rec), flush=True)
