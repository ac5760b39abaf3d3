"""SKI product at the C5 shape (N = 391 386, J = 3, grid 1024): the cell-sorted scatter's workgroups dispatched centre-out
(default) against storage order (RPGP_SKI_CELL_ORDER=0); Gaussian and uniform coordinates; same process, alternating; JSON lines."""
import json, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd.operators import AddedDiagOperator, SKIAdditiveOperator
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
N, J = 391386, 3
for dist in ("gaussian", "uniform", "skewed"):
    g = torch.Generator().manual_seed(0)
    if dist == "gaussian": Z = torch.randn(N, J, generator=g)
    elif dist == "uniform": Z = torch.rand(N, J, generator=g) * 6 - 3
    else: Z = torch.randn(N, J, generator=g).exp()          # log-normal: the mass sits at one end of the grid
    Z = Z.to(dev)
    base = SKIAdditiveOperator(Z, None, torch.tensor(1.0, device=dev), 1.0 / J, grid_size=1024)
    khat = AddedDiagOperator(base, torch.tensor(0.5, device=dev))
    for T in (11, 1):
        rhs = torch.randn(N, T, generator=g).to(dev)
        rec = {"coordinates": dist, "T": T}
        outs = {}
        for rep in range(3):
            for mode in ("0", "1"):
                os.environ["RPGP_SKI_CELL_ORDER"] = mode
                for _ in range(5): out = khat._matmul(rhs)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(40): out = khat._matmul(rhs)
                torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 40 * 1e6
                key = "centre_out_us" if mode == "1" else "storage_order_us"
                rec[key] = round(min(us, rec.get(key, 1e30)), 1)
                outs[mode] = out
        rec["bitwise_equal"] = bool(torch.equal(outs["0"], outs["1"]))
        print(json.dumps(rec), flush=True)
