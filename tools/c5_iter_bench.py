"""One native mBCG solve with a FIXED number of iterations at a BASELINE shape (default C5: N = 391 386, J = d = 3, SKI grid
1024, T = 11 right-hand sides, rank-15 preconditioner) — run under rocprofv3 --kernel-trace for the per-kernel times of an
iteration (tools/r3_prof_iter.sh).  Prints wall time per iteration."""
import os, sys, time, warnings, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops, settings, linear_cg as lcg
from rpgp_amd.operators import AdditiveRPOperator, AddedDiagOperator, SKIAdditiveOperator, SymCachedOperator
from rpgp_amd.precond import build_preconditioner
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
shape = sys.argv[1] if len(sys.argv) > 1 else "C5"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 11
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
g = torch.Generator().manual_seed(0)
if shape == "C5":
    N, J = 391386, 3
    Z = torch.randn(N, J, generator=g).to(dev)
    base = SKIAdditiveOperator(Z, None, torch.tensor(1.0, device=dev), 1.0 / J, grid_size=1024)
    khat = AddedDiagOperator(base, torch.tensor(0.5, device=dev))
    noise = 0.5
else:
    N, J = {"C2": (7372, 20), "C3": (14939, 20), "C2cache": (7372, 20), "C3cache": (14939, 20), "N11cache": (11000, 20),
            "N20cache": (20000, 20), "N25cache": (25000, 20), "N35cache": (35000, 20)}.get(shape, (50000, 20))
    Z = (torch.randn(N, J, generator=g)).to(dev)
    base = AdditiveRPOperator(Z, None, torch.tensor(1.0, device=dev), 1.0 / J)
    noise = 0.1
    if shape.endswith("cache"):
        khat = SymCachedOperator(base.to_symcache(wide=T > 4), base._scale, noise, diag_value=base._scale * J)
    else:
        khat = AddedDiagOperator(base, torch.tensor(noise, device=dev))
rhs = torch.randn(N, T, generator=g).to(dev)
pre = build_preconditioner(base, noise, settings)
with settings.cg_stagnation_window(0):
    lcg.linear_cg(khat._matmul, rhs, tolerance=1e-30, max_iter=5, preconditioner=pre, operator=khat)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lcg.linear_cg(khat._matmul, rhs, tolerance=1e-30, max_iter=iters, preconditioner=pre, operator=khat)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
out = khat._matmul(rhs); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    khat._matmul(rhs)
torch.cuda.synchronize(); mv = (time.perf_counter() - t0) / 20
print(json.dumps({"shape": shape, "N": N, "J": J, "T": T, "iterations": iters, "us_per_iteration": round(dt / iters * 1e6, 1),
                  "us_per_mvm_alone": round(mv * 1e6, 1), "us_vector_part": round((dt / iters - mv) * 1e6, 1)}))
