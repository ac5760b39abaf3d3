import torch, time
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, K = 24000, 4096
A = torch.randn(M, K, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
flops = 2.0 * M * M * K
ref = None
try:
    ts = t(lambda: torch.mm(A, A.t()))
    print("fp32 mm  %.1f ms  %.1f TFLOP/s" % (ts * 1e3, flops / ts / 1e12))
    ref64 = (A[:2048].double() @ A[:2048].double().t())
    c32 = (A[:2048] @ A[:2048].t()).double()
    print("fp32 mm rel err vs f64: %.2e" % float((c32 - ref64).norm() / ref64.norm()))
except Exception as e:
    print("fp32 mm failed", e)
hi = A.bfloat16(); lo = (A - hi.float()).bfloat16()
try:
    out = torch.mm(hi, hi.t(), out_dtype=torch.float32)
    print("bf16 -> fp32 out_dtype OK", out.dtype)
    def x3():
        c = torch.mm(hi, hi.t(), out_dtype=torch.float32)
        c += torch.mm(hi, lo.t(), out_dtype=torch.float32)
        c += torch.mm(lo, hi.t(), out_dtype=torch.float32)
        return c
    ts = t(x3)
    print("bf16x3 mm  %.1f ms  %.1f TFLOP/s fp32-equivalent" % (ts * 1e3, flops / ts / 1e12))
    h2, l2 = hi[:2048], lo[:2048]
    c = torch.mm(h2, h2.t(), out_dtype=torch.float32) + torch.mm(h2, l2.t(), out_dtype=torch.float32) + torch.mm(l2, h2.t(), out_dtype=torch.float32)
    print("bf16x3 rel err vs f64: %.2e" % float((c.double() - ref64).norm() / ref64.norm()))
    ts = t(lambda: torch.mm(hi, hi.t(), out_dtype=torch.float32))
    print("bf16 single mm %.1f ms  %.1f TFLOP/s" % (ts * 1e3, flops / ts / 1e12))
except Exception as e:
    print("out_dtype failed:", repr(e)[:300])
try:
    ts = t(lambda: torch.mm(hi, hi.t()))
    print("bf16 mm (bf16 out) %.1f ms %.1f TFLOP/s" % (ts * 1e3, flops / ts / 1e12))
except Exception as e:
    print("bf16 mm failed", e)
# potrf timing at 24000
S = A @ A.t() / K + torch.eye(M, device=dev)
ts = t(lambda: torch.linalg.cholesky(S), reps=2)
print("potrf N=%d  %.1f ms  %.1f TFLOP/s" % (M, ts * 1e3, M ** 3 / 3 / ts / 1e12))
