"""C4-size float32 Cholesky for the mixed-precision covariance solve: the library routine against a blocked right-looking
factorisation whose trailing updates are bf16x3 GEMMs (three bf16 matrix products with float32 accumulation on a hi / lo split
of the panel: ~fp32 accuracy at several times the fp32 matrix rate on MI355X) on the LOWER block triangle only.  JSON lines."""
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
# does addmm take bf16 operands into a float32 accumulator in one call?
C = torch.zeros(4096, 4096, device=dev); a = torch.randn(4096, 4096, device=dev).bfloat16()
try:
    torch.addmm(C, a, a.t(), alpha=-1.0, out_dtype=torch.float32)
    res["addmm_out_dtype"] = True
except Exception as e:
    res["addmm_out_dtype"] = repr(e)[:120]
t, Lref = timed(lambda: torch.linalg.cholesky(K)); res["library_potrf_s"] = round(t, 4)
D = torch.linalg.cholesky(K[:4096, :4096])
t, _ = timed(lambda: torch.linalg.cholesky(K[:4096, :4096])); res["potrf_4k_ms"] = round(t * 1e3, 2)
I = torch.eye(4096, device=dev)
t, _ = timed(lambda: torch.linalg.solve_triangular(D, I, upper=False)); res["trinv_4k_ms"] = round(t * 1e3, 2)
print(json.dumps(res), flush=True)


PANEL = os.environ.get("PANEL", "inv")


def blocked(K, nb, mode):
    L = K.clone()
    n = L.shape[0]
    eye = torch.eye(nb, device=dev)
    for j0 in range(0, n, nb):
        j1 = min(j0 + nb, n)
        D, info = torch.linalg.cholesky_ex(L[j0:j1, j0:j1])
        L[j0:j1, j0:j1] = D
        if j1 >= n:
            break
        if PANEL == "trsm":
            P = torch.linalg.solve_triangular(D, L[j1:, j0:j1].t(), upper=False).t().contiguous()
        else:
            Dinv = torch.linalg.solve_triangular(D, eye[: j1 - j0, : j1 - j0], upper=False)
            P = L[j1:, j0:j1] @ Dinv.t()                       # panel: A21 D^-T as a GEMM
        L[j1:, j0:j1] = P
        if mode == "fp32":
            for c0 in range(j1, n, nb):
                c1 = min(c0 + nb, n)
                L[c0:, c0:c1].addmm_(P[c0 - j1:], P[c0 - j1:c1 - j1].t(), alpha=-1.0)
        elif mode == "fp16x3":
            hi = P.half(); lo = ((P - hi.float()) * 2048.0).half()
            for c0 in range(j1, n, nb):
                c1 = min(c0 + nb, n)
                blk = L[c0:, c0:c1]
                ah, al = hi[c0 - j1:], lo[c0 - j1:]
                bh, bl = hi[c0 - j1:c1 - j1].t(), lo[c0 - j1:c1 - j1].t()
                t1 = torch.mm(ah, bl, out_dtype=torch.float32)
                t1 = torch.addmm(t1, al, bh, out_dtype=torch.float32)
                t1 = torch.addmm(t1, ah, bh, beta=1.0 / 2048.0, out_dtype=torch.float32)
                blk.sub_(t1)
        else:
            hi = P.bfloat16(); lo = (P - hi.float()).bfloat16()
            for c0 in range(j1, n, nb):
                c1 = min(c0 + nb, n)
                blk = L[c0:, c0:c1]
                ah, al = hi[c0 - j1:], lo[c0 - j1:]
                bh, bl = hi[c0 - j1:c1 - j1].t(), lo[c0 - j1:c1 - j1].t()
                t1 = torch.mm(ah, bh, out_dtype=torch.float32)
                t1 += torch.mm(ah, bl, out_dtype=torch.float32)
                t1 += torch.mm(al, bh, out_dtype=torch.float32)
                blk.sub_(t1)
    return L


for mode in ("fp16x3",):
    for nb in (2048,):
        blocked(K[:8192, :8192].contiguous(), nb, mode)
        t, Lb = timed(lambda: blocked(K, nb, mode))
        rec = {"mode": mode, "nb": nb, "seconds": round(t, 4), "tflops_equiv": round(N ** 3 / 3 / t / 1e12, 1)}
        d = (Lb.tril() - Lref)
        rec["maxdiff_vs_library"] = float(d.abs().max())
        # backward error of the factor on a probe: | K x - L L^T x | / |K x|
        x = torch.randn(N, 4, device=dev, dtype=torch.float64)
        Lt = Lb.tril().double()
        kx = K.double() @ x
        rec["factor_backward_err"] = float((kx - Lt @ (Lt.t() @ x)).norm() / kx.norm())
        Lr = Lref.double()
        rec["library_backward_err"] = float((kx - Lr @ (Lr.t() @ x)).norm() / kx.norm())
        del Lt, Lr, Lb, d
        print(json.dumps(rec), flush=True)
