import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 14939
T = int(sys.argv[2]) if len(sys.argv) > 2 else 11
Z = torch.randn(N, 20, generator=torch.Generator().manual_seed(0)).to(dev)
K = ops.dense(Z, Z, 0.05, pad=True)
V = torch.randn(N, T, device=dev)
for _ in range(12):
    ops.dense_mvm(K, V, 0.1)
torch.cuda.synchronize()
