"""C4-size (N = 50 000) float32 Cholesky: the library routine against a blocked right-looking factorisation whose trailing
updates are plain GEMMs (where the flops are); plus the raw GEMM / TRSM rates.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
Z = torch.randn(N, 20, generator=g).to(dev)
K = ops.dense(Z, Z, 0.05)
K.diagonal().add_(0.1)
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return time.perf_counter() - t0, r
res = {"N": N}
# raw rates
A = torch.randn(8192, 8192, device=dev); B = torch.randn(8192, 8192, device=dev)
timed(lambda: A @ B)
t, _ = timed(lambda: A @ B); res["gemm_8k_tflops"] = 2 * 8192 ** 3 / t / 1e12
A2 = torch.randn(40000, 4096, device=dev)
t, _ = timed(lambda: A2 @ A2.t()); res["syrk_as_gemm_40k_x_4k_tflops"] = 2 * 40000 * 40000 * 4096 / t / 1e12
Lb = torch.linalg.cholesky(K[:4096, :4096])
t, _ = timed(lambda: torch.linalg.solve_triangular(Lb, K[:4096, 4096:4096 + 40000], upper=False)); res["trsm_4k_x_40k_tflops"] = 4096 * 4096 * 40000 / t / 1e12
t, _ = timed(lambda: torch.linalg.cholesky(K[:4096, :4096])); res["potrf_4k_s"] = t
del A, B, A2
t, Lref = timed(lambda: torch.linalg.cholesky(K)); res["library_potrf_s"] = t; res["library_tflops"] = N ** 3 / 3 / t / 1e12
def blocked(K, nb):
    L = K.clone()
    n = L.shape[0]
    for j0 in range(0, n, nb):
        j1 = min(j0 + nb, n)
        D = torch.linalg.cholesky(L[j0:j1, j0:j1])
        L[j0:j1, j0:j1] = D
        if j1 < n:
            # panel: L21 = A21 D^-T
            P = torch.linalg.solve_triangular(D, L[j1:, j0:j1].t(), upper=False).t()
            L[j1:, j0:j1] = P
            # trailing update (full GEMM on the lower-right block; only its lower triangle is read later)
            L[j1:, j1:].addmm_(P, P.t(), alpha=-1.0)
    return L
for nb in (2048, 4096):
    t, Lb_ = timed(lambda: blocked(K, nb)); res["blocked_nb%d_s" % nb] = t
    err = float((Lb_.tril() - Lref).abs().max()); res["blocked_nb%d_maxdiff" % nb] = err
    del Lb_
print(json.dumps(res))
