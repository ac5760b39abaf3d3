"""Runs the wide packed-cache product (T = 11) with each slab-reduce width (RPGP_RED_OUTS) and both product kernels; meant
to run under `rocprofv3 --kernel-trace` (tools/r5_symk_red_prof.sh), which attributes the time to the kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for N in [int(a) for a in (sys.argv[1:] or ["7372", "14939", "50000"])]:
    g = torch.Generator().manual_seed(N)
    Z = torch.randn(N, 20, generator=g).to(dev)
    V = torch.randn(N, 11, generator=g).to(dev)
    C = ops.SymCache(Z, wide=True)
    for outs in ("32", "64", "128", "256"):
        os.environ["RPGP_RED_OUTS"] = outs
        for mode in ("0", "1"):
            os.environ["RPGP_SYMK_WIDE_V2"] = mode
            for _ in range(12):
                ops.symcache_mvm(C, V, 0.05, 0.1)
            torch.cuda.synchronize()
    del C
