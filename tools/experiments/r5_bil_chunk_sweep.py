"""EXPERIMENT: bilinear derivative (hand-scheduled symmetric sweep, T = 11) over the column-chunk size of its workgroups
(RPGP_BIL_CHUNK knob; default = the fused sweep's ~4600-workgroup target) at the C2 / C3 / C4 shapes; one process, HIP events."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
CH = {7372: ["default", "64", "128", "192", "256", "320", "384"], 14939: ["default", "64", "128", "192", "256", "320", "384", "448", "512"],
      50000: ["default", "448", "512", "576", "640", "704", "768", "832", "896", "1024", "1216"]}
for (N, d, J, T) in [(7372, 8, 20, 11), (14939, 18, 20, 11), (50000, 20, 20, 11)]:
    g = torch.Generator().manual_seed(0)
    Z = (torch.randn(N, d, generator=g) @ torch.randn(d, J, generator=g) / d ** 0.5).to(dev)
    L = (torch.randn(N, T, generator=g) * 0.1).to(dev)
    R = (torch.randn(N, T, generator=g) * 0.1).to(dev)
    res = {"N": N, "T": T}
    ref = None
    for rep in range(2):
        for c in CH[N]:
            if c == "default": os.environ.pop("RPGP_BIL_CHUNK", None)
            else: os.environ["RPGP_BIL_CHUNK"] = c
            gZ, gs = ops.bilinear_grad(Z, L, R, 0.05)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gZ, gs = ops.bilinear_grad(Z, L, R, 0.05)
            e1.record(); torch.cuda.synchronize()
            key = "chunk_%s_ms" % c
            res[key] = round(min(e0.elapsed_time(e1) / 5, res.get(key, 1e9)), 4)
            if ref is None: ref = gZ.clone()
            res["max_rel_diff"] = max(res.get("max_rel_diff", 0.0), float((gZ - ref).norm() / ref.norm()))
    print(json.dumps(res), flush=True)
