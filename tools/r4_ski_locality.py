"""Effect of the row order on the planned SKI product at the C5 shape (N = 391 386, J = d = 3, G = 1024): rows in random
(file) order against training.locality_order (Morton code of the principal coordinates).  HIP-event time per product."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
from rpgp_amd.training import locality_order
dev = torch.device("cuda:0")
N, J = 391386, 3
g = torch.Generator().manual_seed(0)
X = torch.randn(N, J, generator=g)
Q, _ = torch.linalg.qr(torch.randn(J, J, generator=g))
for name, rows in (("file order", torch.arange(N)), ("locality order", locality_order(X))):
    Z = (X[rows] @ Q).contiguous().to(dev)
    gp = ops.ski_grid(Z, None, 1024)
    plan = ops.SkiPlan(Z, gp, 1024)
    for T in (1, 11):
        V = torch.randn(N, T, generator=g).to(dev)
        for _ in range(3):
            ops.ski_mvm(Z, Z, gp, V, 1.0 / J, 0.1, 1024, plan=plan)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.ski_mvm(Z, Z, gp, V, 1.0 / J, 0.1, 1024, plan=plan)
        e1.record(); torch.cuda.synchronize()
        print(json.dumps({"rows": name, "T": T, "us_per_mvm": round(e0.elapsed_time(e1) / 50 * 1e3, 2)}), flush=True)
