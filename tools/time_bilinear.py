"""Bilinear derivative (the backward pass of one training step) at the C2 / C3 / C4 shapes: symmetric sweep vs the full
sweep (RPGP_BILINEAR_FULL=1), HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for (N, d, J, T) in [(7372, 8, 20, 11), (14939, 18, 20, 11), (50000, 20, 20, 11), (50000, 20, 20, 1)]:
    g = torch.Generator().manual_seed(0)
    Z = (torch.randn(N, d, generator=g) @ torch.randn(d, J, generator=g) / d ** 0.5).to(dev)
    L = (torch.randn(N, T, generator=g) * 0.1).to(dev)
    R = (torch.randn(N, T, generator=g) * 0.1).to(dev)
    res, outs = {}, {}
    for mode in ("sym", "full"):
        os.environ["RPGP_BILINEAR_FULL"] = "1" if mode == "full" else "0"
        gZ, gs = ops.bilinear_grad(Z, L, R, 0.05)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gZ, gs = ops.bilinear_grad(Z, L, R, 0.05)
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / 5
        outs[mode] = (gZ.clone(), float(gs))
    rel = float((outs["sym"][0] - outs["full"][0]).norm() / outs["full"][0].norm())
    print("N=%d J=%d T=%d: symmetric %.3f ms  full sweep %.3f ms  (%.2fx)  rel diff gZ %.2e  gscale %.6g vs %.6g" % (
        N, J, T, res["sym"], res["full"], res["full"] / res["sym"], rel, outs["sym"][1], outs["full"][1]), flush=True)
