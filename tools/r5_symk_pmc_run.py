"""Eight wide (T = 11) and eight thin (T = 1) packed-cache products at the C4 shape (for tools/r5_symk_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
g = torch.Generator().manual_seed(N)
Z = torch.randn(N, 20, generator=g).to(dev)
for wide, T in ((True, 11), (False, 1)):
    V = torch.randn(N, T, generator=g).to(dev)
    C = ops.SymCache(Z, wide=wide)
    for _ in range(8):
        ops.symcache_mvm(C, V, 0.05, 0.1)
    torch.cuda.synchronize()
    del C
