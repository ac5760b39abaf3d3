"""The bilinear derivative at the C4 shape with the training block (T = 11), N launches (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
N, d, J, T = int(os.environ.get("BIL_N", 50000)), 20, 20, 11
g = torch.Generator().manual_seed(0)
Z = (torch.randn(N, d, generator=g) @ torch.randn(d, J, generator=g) / d ** 0.5).to(dev)
L = (torch.randn(N, T, generator=g) * 0.1).to(dev)
R = (torch.randn(N, T, generator=g) * 0.1).to(dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    ops.bilinear_grad(Z, L, R, 0.05)
torch.cuda.synchronize()
