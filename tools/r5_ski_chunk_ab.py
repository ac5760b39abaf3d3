"""Chunked vs cell-sorted planned SKI product at the C5 shape (N = 391 386, J = d = 3, G = 1024), rows in file order and in
training.locality_order; HIP-event time per product, per stage, and of the plan.  One JSON line per case."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
from rpgp_amd.training import locality_order
dev = torch.device("cuda:0")
N, J, G = int(os.environ.get("N", 391386)), 3, 1024
g = torch.Generator().manual_seed(0)
X = torch.randn(N, J, generator=g)
Q, _ = torch.linalg.qr(torch.randn(J, J, generator=g))


def timed(fn, reps=50, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 2)


orders = os.environ.get("ORDERS", "locality,file").split(",")
for name, rows in (("locality order", locality_order(X)), ("file order", torch.arange(N))):
    if name.split()[0] not in orders:
        continue
    Z = (X[rows] @ Q).contiguous().to(dev)
    gp = ops.ski_grid(Z, None, G)
    ops.ski_chunk_mode(False)
    plan_us_default = timed(lambda: ops.SkiPlan(Z, gp, G), reps=10)
    ops.ski_chunk_mode(True)
    plan = ops.SkiPlan(Z, gp, G)                    # carries the tables of both forms
    plan_us = timed(lambda: ops.SkiPlan(Z, gp, G), reps=10)
    for T in (11, 1):
        V = torch.randn(N, T, generator=g).to(dev)
        ref = None
        for mode in (True, False):
            ops.ski_chunk_mode(mode)
            out = ops.ski_mvm(Z, Z, gp, V, 1.0 / J, 0.1, G, plan=plan)
            hist = ops.ski_scatter(Z, gp, V, G, plan=plan)
            H = ops.ski_grid_product(hist, gp, G)
            rec = {"rows": name, "T": T, "form": "chunked" if mode else "cell-sorted", "plan_us": plan_us if mode else plan_us_default,
                   "us_per_mvm": timed(lambda: ops.ski_mvm(Z, Z, gp, V, 1.0 / J, 0.1, G, plan=plan)),
                   "scatter_us": timed(lambda: ops.ski_scatter(Z, gp, V, G, plan=plan)),
                   "toeplitz_us": timed(lambda: ops.ski_grid_product(hist, gp, G)),
                   "gather_us": timed(lambda: ops.ski_gather(Z, gp, H, V, 1.0 / J, 0.1, G, plan=plan))}
            if ref is None:
                ref = out
            else:
                rec["rel_diff_vs_chunked"] = float((out - ref).norm() / ref.norm())
            print(json.dumps(rec), flush=True)
        ops.ski_chunk_mode(True)
ops.ski_chunk_mode(False)
