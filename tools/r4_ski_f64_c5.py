"""The float64 parity path of the SKI operator at the full C5 size (N = 391 386, J = 3, grid 1024): product time (T = 1, 11),
agreement with the float32 product and with the float64 sparse oracle on a row sample.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rpgp_amd import ops
from oracle import ski as sko
dev = torch.device("cuda:0")
N, J, G = 391386, 3, 1024
g = torch.Generator().manual_seed(0)
Z = torch.randn(N, J, generator=g, dtype=torch.float64)
res = {"N": N, "J": J, "G": G}
Zd = Z.to(dev); Zf = Zd.float()
gp64, gp32 = ops.ski_grid(Zd, None, G), ops.ski_grid(Zf, None, G)
for T in (1, 11):
    V = torch.randn(N, T, generator=g, dtype=torch.float64)
    Vd = V.to(dev)
    out = ops.ski_mvm(Zd, Zd, gp64, Vd, 1.0 / J, 0.1, G)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        out = ops.ski_mvm(Zd, Zd, gp64, Vd, 1.0 / J, 0.1, G)
    torch.cuda.synchronize(); res["f64_ms_T%d" % T] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    o32 = ops.ski_mvm(Zf, Zf, gp32, Vd.float(), 1.0 / J, 0.1, G).double()
    res["f32_vs_f64_rel_T%d" % T] = float((o32 - out).norm() / out.norm())
    if T == 11:
        grid = (float(gp64[0]), float(gp64[1]))
        ref = sko.mvm_sparse(Z.numpy(), Z.numpy(), V.numpy(), 1.0 / J, G, grid, noise=0.1)
        res["f64_vs_oracle_rel_T11"] = float(np.linalg.norm(out.cpu().numpy() - ref) / np.linalg.norm(ref))
print(json.dumps(res))
