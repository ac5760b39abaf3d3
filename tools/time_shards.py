"""Per-rank kernel time of the J-sharded MVM (what each rank runs at 1/2/4/8 GPUs), measured on one GPU."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
from rpgp_amd.distributed import j_partition
dev = torch.device("cuda:0")
N, d, J = 50000, 20, 20
X = torch.randn(N, d, generator=torch.Generator().manual_seed(0)).to(dev)
P = torch.randn(d, J, generator=torch.Generator().manual_seed(1)).to(dev)
Z = ops.project(X, (P / math.sqrt(d)).contiguous())
V = torch.randn(N, 1, device=dev)
prep = ops.Prepared(Z)
base = None
for world in (1, 2, 4, 8):
    worst = 0.0
    for (j0, j1) in sorted(set(j_partition(J, world)), key=lambda t: t[0]):
        ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.0, j0=j0, j1=j1); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.0, j0=j0, j1=j1)
        e1.record(); torch.cuda.synchronize()
        worst = max(worst, e0.elapsed_time(e1) / 10)
    base = base or worst
    print("world %d: slowest rank %.3f ms  -> ideal-comm speedup %.2fx (efficiency %.0f%%)" % (world, worst, base / worst, 100 * base / worst / world))

print("pair-sharding (equal shares of the tile pairs, all J terms per rank):")
base = None
for world in (1, 2, 4, 8):
    worst = 0.0
    for r in range(world):
        ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.0, shard=(world, r)); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.0, shard=(world, r))
        e1.record(); torch.cuda.synchronize()
        worst = max(worst, e0.elapsed_time(e1) / 10)
    base = base or worst
    print("world %d: slowest rank %.3f ms  -> ideal-comm speedup %.2fx (efficiency %.0f%%)" % (world, worst, base / worst, 100 * base / worst / world))
