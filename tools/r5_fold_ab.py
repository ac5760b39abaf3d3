"""The executor's pass A folded into the SKI gather (default) against the separate pass (RPGP_CG_FOLD_A=0) at the C5 shape:
wall time per solve and per iteration; one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import linear_cg as lcg
from rpgp_amd.operators import SKIAdditiveOperator, AddedDiagOperator
from rpgp_amd.precond import pivoted_cholesky, WoodburyPreconditioner
dev = torch.device("cuda:0")
N, J = 391386, 3
g = torch.Generator().manual_seed(0)
Z = torch.randn(N, J, generator=g).to(dev)
noise = 0.1
base = SKIAdditiveOperator(Z, None, torch.tensor(0.9, device=dev), 1.0 / J)
op = AddedDiagOperator(base, torch.tensor(noise, device=dev))
rhs = torch.randn(N, 11, generator=g).to(dev)
pre = WoodburyPreconditioner(pivoted_cholesky(base._diagonal(), base._get_rows, 15), noise)
kw = dict(n_tridiag=10, tolerance=0.05, max_iter=200, max_tridiag_iter=20, preconditioner=pre, operator=op, lanczos="history")
rec = {"shape": "C5", "N": N, "J": J, "T": 11}
for fold in ("1", "0", "1", "0"):
    os.environ["RPGP_CG_FOLD_A"] = fold
    for _ in range(3):
        lcg.linear_cg(op._matmul, rhs, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        lcg.linear_cg(op._matmul, rhs, **kw)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 30 * 1e6
    key = "folded_us" if fold == "1" else "separate_pass_us"
    rec[key] = round(min(us, rec.get(key, 1e30)), 1)
    rec["iterations_" + ("folded" if fold == "1" else "separate")] = lcg.stats["last_iterations"]
rec["us_per_iteration"] = {"folded": round(rec["folded_us"] / rec["iterations_folded"], 2),
                           "separate": round(rec["separate_pass_us"] / rec["iterations_separate"], 2)}
print(json.dumps(rec))
