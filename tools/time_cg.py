import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math
from rpgp_amd import ops, settings, linear_cg as lcg
from rpgp_amd.operators import AdditiveRPOperator, AddedDiagOperator
from rpgp_amd.precond import build_preconditioner
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
N, d, J = (int(sys.argv[1]) if len(sys.argv) > 1 else 50000), 20, 20
X = torch.randn(N, d, generator=torch.Generator().manual_seed(0)).to(dev)
P = torch.randn(d, J, generator=torch.Generator().manual_seed(1)).to(dev)
Z = ops.project(X, (P / math.sqrt(d)).contiguous())
base = AdditiveRPOperator(Z, None, torch.tensor(1.0, device=dev), 1.0 / J)
khat = AddedDiagOperator(base, torch.tensor(0.1, device=dev))
for T in (1, 11):
    rhs = torch.randn(N, T, device=dev)
    for pre in (None, build_preconditioner(base, 0.1, settings)):
        for iters in (20, 60):
            lcg.linear_cg(khat._matmul, rhs, tolerance=1e-30, max_iter=5, preconditioner=pre, operator=khat)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            lcg.linear_cg(khat._matmul, rhs, tolerance=1e-30, max_iter=iters, preconditioner=pre, operator=khat)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print("T", T, "precond", pre is not None, "iters", iters, "ms/iter", dt / iters * 1e3)
    out = khat._matmul(rhs); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): khat._matmul(rhs)
    torch.cuda.synchronize(); print("T", T, "plain MVM ms", (time.perf_counter() - t0) / 20 * 1e3)
