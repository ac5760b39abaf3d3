"""Queue-ahead form vs graph form (rpgp_mbcg_graph_mode) of the native mBCG executor: wall time per solve (the solve's own
synchronisation included) at small and training-size systems.  One JSON line per case."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import linear_cg as lcg, ops
from rpgp_amd.operators import AdditiveRPOperator, SKIAdditiveOperator, AddedDiagOperator, SymCachedOperator
from rpgp_amd.precond import pivoted_cholesky, WoodburyPreconditioner
dev = torch.device("cuda:0")
cases = [("fused", 1000, 20), ("fused", 2000, 20), ("symcache", 3000, 20), ("symcache", 7372, 20), ("symcache", 14939, 20), ("ski", 391386, 3)]
for kind, N, J in cases:
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, J, generator=g).to(dev)
    noise = 0.1
    cls = SKIAdditiveOperator if kind == "ski" else AdditiveRPOperator
    base = cls(Z, None, torch.tensor(0.9, device=dev), 1.0 / J)
    op = AddedDiagOperator(base, torch.tensor(noise, device=dev))
    if kind == "symcache":
        op = SymCachedOperator(base.to_symcache(wide=True), base._scale, noise, diag_value=base._scale * base.num_projections)
    rhs = torch.randn(N, 11, generator=g).to(dev)
    pre = WoodburyPreconditioner(pivoted_cholesky(base._diagonal(), base._get_rows, 15), noise)
    kw = dict(n_tridiag=10, tolerance=0.05, max_iter=200, max_tridiag_iter=20, preconditioner=pre, operator=op, lanczos="history")
    rec = {"operator": kind, "N": N, "J": J, "T": 11}
    for graph in (False, True, False, True):
        ops.mbcg_graph_mode(graph)
        for _ in range(3):
            lcg.linear_cg(op._matmul, rhs, **kw)
        torch.cuda.synchronize()
        reps = 30
        t0 = time.perf_counter()
        for _ in range(reps):
            lcg.linear_cg(op._matmul, rhs, **kw)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
        key = "graph_us" if graph else "queue_ahead_us"
        rec[key] = round(min(us, rec.get(key, 1e30)), 1)
        rec["iterations"] = lcg.stats["last_iterations"]
    ops.mbcg_graph_mode(False)
    rec["us_per_iteration"] = {k: round(rec[k] / max(rec["iterations"], 1), 2) for k in ("queue_ahead_us", "graph_us")}
    print(json.dumps(rec), flush=True)
