"""First-use cost of the library calls blocked_cholesky makes, in a fresh process (each line: first call, second call)."""
import sys, time, torch
dev = torch.device("cuda:0")
torch.zeros(1, device=dev); torch.cuda.synchronize()
def t2(name, f):
    out = []
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); out.append(round(time.perf_counter() - t0, 3))
    print(name, out, flush=True)
A = torch.randn(20000, 2048, device=dev); S = torch.randn(2048, 2048, device=dev); S = S @ S.t() / 2048 + torch.eye(2048, device=dev)
t2("fp32 mm 2048", lambda: S @ S)
t2("cholesky_ex 2048", lambda: torch.linalg.cholesky_ex(S))
D = torch.linalg.cholesky_ex(S)[0]
t2("solve_triangular 2048 x 20000", lambda: torch.linalg.solve_triangular(D, A.t(), upper=False))
h = A.half()
t2("mm fp16 -> fp32 (20000x2048)(2048x2048)", lambda: torch.mm(h, h[:2048].t(), out_dtype=torch.float32))
c = torch.mm(h, h[:2048].t(), out_dtype=torch.float32)
t2("addmm fp16 -> fp32", lambda: torch.addmm(c, h, h[:2048].t(), out_dtype=torch.float32))
t2("addmm fp16 -> fp32 beta", lambda: torch.addmm(c, h, h[:2048].t(), beta=1.0 / 2048, out_dtype=torch.float32))
h2 = A[:7000].half()
t2("mm fp16 -> fp32 (7000x2048)(2048x2048)", lambda: torch.mm(h2, h2[:2048].t(), out_dtype=torch.float32))
t2("half()", lambda: A.half())
