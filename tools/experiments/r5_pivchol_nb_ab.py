"""Rank-15 pivoted Cholesky per-step launches: at most 512 workgroups (RPGP_PIVCHOL_NB=512: the round-2 cap) against at most
2048 (default), exact operator (J = 20) and SKI operator (J = 3, grid 1024) at the large sizes; same process, alternating."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
def bench(fn):
    for _ in range(3): L = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): L = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e6, L
for N, J, ski in ((50000, 20, False), (131072, 20, False), (200000, 20, False), (391386, 3, True), (131072, 3, True)):
    g = torch.Generator().manual_seed(N)
    Z = (torch.randn(N, J, generator=g) * 0.8).to(dev)
    rec = {"N": N, "J": J, "ski": ski}
    if ski:
        grid = ops.ski_grid(Z, None, 1024)
        fn = lambda: ops.ski_pivoted_cholesky(Z, grid, 0.3, 15, 1024)
    else:
        fn = lambda: ops.pivoted_cholesky(Z, 0.05, 15)
    outs = {}
    for rep in range(3):
        for nb in ("512", "1024", "2048"):
            os.environ["RPGP_PIVCHOL_NB"] = nb
            us, L = bench(fn)
            rec["nb%s_us" % nb] = round(min(us, rec.get("nb%s_us" % nb, 1e30)), 1)
            outs[nb] = L
    rec["bitwise_equal"] = bool(torch.equal(outs["512"], outs["2048"]) and torch.equal(outs["512"], outs["1024"]))
    print(json.dumps(rec), flush=True)
