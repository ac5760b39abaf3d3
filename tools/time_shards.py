"""ESTIMATE, ONE GPU: the kernel time each rank of a 1 / 2 / 4 / 8-GPU run would spend on ITS share of the sharded C4 MVM
(N = 50 000, J = 20, T = 1), every share run ALONE on the one device of the test box, for both splits
(distributed.JShard: north_star's J-slices; equal shares of the tile pairs).  No collective, no second device: what a node
adds is the all-reduce of the 200 KB partial (bench.py --gpus N measures it).  Prints the per-rank times."""
import json, math, os, sys
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


def ms(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("ESTIMATE, ONE GPU — per-rank kernel time of each rank's share run alone (ms); C4 N=50000 J=20 T=1")
base = {}
for split in ("j", "pairs"):
    print("%s-split:" % split)
    for world in (1, 2, 4, 8):
        if split == "j":
            per = [ms(lambda a=a, b=b: ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.0, j0=a, j1=b)) for a, b in j_partition(J, world)]
            what = " slices " + ",".join(str(b - a) for a, b in j_partition(J, world))
        else:
            per = [ms(lambda r=r: ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.0, shard=(world, r))) for r in range(world)]
            what = ""
        base.setdefault(split, max(per))
        print("  world %d:%s per-rank %s  slowest %.3f -> ideal-comm speedup %.2fx (efficiency %.0f%%)" % (
            world, what, " ".join("%.3f" % p for p in per), max(per), base[split] / max(per), 100 * base[split] / max(per) / world))
