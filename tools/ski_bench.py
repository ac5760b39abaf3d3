"""SKI MVM at the C5 shape (N = 391 386, J = d = 3, grid 1024) and at C4-with-ski (N = 50k, J = 20): wall time per MVM
and the HBM roofline line  bytes = 4 N (J + 2 T)  (read Z, read V, write out) / time  vs 8 TB/s."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
reps = int(os.environ.get("SKI_REPS", "50"))
for (N, J) in [(391386, 3), (50000, 20)][:int(os.environ.get("SKI_SHAPES", "2"))]:
    g = torch.Generator().manual_seed(0)
    Z = torch.randn(N, J, generator=g).to(dev)
    gp = ops.ski_grid(Z, None, 1024)
    plan = ops.SkiPlan(Z, gp, 1024) if os.environ.get("SKI_PLAN", "1") == "1" else None
    if plan is not None:                      # cost of the per-step plan (sort by interpolation cell)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.SkiPlan(Z, gp, 1024)
        e1.record(); torch.cuda.synchronize()
        print(json.dumps({"workload": "SKI plan N=%d J=%d G=1024" % (N, J), "ms_per_plan": round(e0.elapsed_time(e1) / 5, 4)}), flush=True)
    for T in [int(t) for t in os.environ.get("SKI_TS", "1,11").split(",")]:
        V = torch.randn(N, T, generator=g).to(dev)
        out = ops.ski_mvm(Z, Z, gp, V, 1.0 / J, 0.1, 1024, plan=plan)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.ski_mvm(Z, Z, gp, V, 1.0 / J, 0.1, 1024, plan=plan)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        b = 4.0 * N * (J + 2 * T)
        print(json.dumps({"workload": "SKI MVM N=%d J=%d T=%d G=1024%s" % (N, J, T, " planned" if plan is not None else ""), "ms_per_mvm": round(ms, 4),
                          "algorithmic_bytes": b, "achieved_GBps": round(b / ms / 1e6, 1), "frac_of_8TBps": round(b / ms / 1e6 / 8000, 4)}), flush=True)
