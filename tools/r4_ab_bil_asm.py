"""A/B of the bilinear derivative in ONE process: hand-scheduled loop (RPGP_BIL_ASM=1) against the compiler-scheduled
bilinear_sym_kernel<20,12,true> (=0) at the C2 / C3 / C4 shapes with the training block (T = 11), HIP events, alternating."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
out = []
for (N, d, J, T) in [(7372, 8, 20, 11), (14939, 18, 20, 11), (50000, 20, 20, 11)]:
    g = torch.Generator().manual_seed(0)
    Z = (torch.randn(N, d, generator=g) @ torch.randn(d, J, generator=g) / d ** 0.5).to(dev)
    L = (torch.randn(N, T, generator=g) * 0.1).to(dev)
    R = (torch.randn(N, T, generator=g) * 0.1).to(dev)
    res = {"N": N, "T": T, "pairs": []}
    outs = {}
    for rep in range(3):
        pair = {}
        for flag in ("1", "0"):
            os.environ["RPGP_BIL_ASM"] = flag
            gZ, gs = ops.bilinear_grad(Z, L, R, 0.05)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gZ, gs = ops.bilinear_grad(Z, L, R, 0.05)
            e1.record(); torch.cuda.synchronize()
            pair["asm_ms" if flag == "1" else "compiler_ms"] = round(e0.elapsed_time(e1) / 5, 4)
            outs[flag] = (gZ.clone(), float(gs))
        res["pairs"].append(pair)
    res["rel_diff_gZ"] = float((outs["1"][0] - outs["0"][0]).norm() / outs["0"][0].norm())
    res["gscale"] = [outs["1"][1], outs["0"][1]]
    out.append(res)
    print(json.dumps(res), flush=True)
