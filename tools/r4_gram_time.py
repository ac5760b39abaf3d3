"""rpgp_gram_f64 at the step's shapes (L^T L, L^T probes, X^T gZ) — time per call."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
res = {}
for name, N, K, T in (("C2 LtL", 7372, 15, 15), ("C2 Ltprobes", 7372, 15, 10), ("C2 XtG", 7372, 8, 20), ("C5 LtL", 391386, 15, 15),
                      ("C5 Ltprobes", 391386, 15, 10), ("C5 XtG", 391386, 3, 3)):
    A = torch.randn(N, K, device=dev); B = torch.randn(N, T, device=dev)
    ref = A.double().t() @ B.double()
    out = ops.gram_f64(A, B)
    err = float((out - ref).abs().max() / ref.abs().max())
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.gram_f64(A, B)
    e1.record(); torch.cuda.synchronize()
    res[name] = {"us": round(e0.elapsed_time(e1) / 50 * 1e3, 2), "rel_err": err}
print(json.dumps(res))
