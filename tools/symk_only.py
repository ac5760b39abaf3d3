import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 12
Z = torch.randn(N, 20, generator=torch.Generator().manual_seed(0)).to(dev)
WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"
C = ops.SymCache(Z, wide=WIDE)
V = torch.randn(N, T, device=dev)
for _ in range(6):
    ops.symcache_mvm(C, V, 0.05, 0.1)
torch.cuda.synchronize()
