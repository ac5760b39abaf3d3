"""ONE GPU: for the pairs split at world 1 / 2 / 4 / 8, rank 0's and the last rank's share of the C4 MVM run alone — the whole
call (slab zeroing + tile kernel + slab reduce) against the tile kernel alone (rpgp_profile_*: HIP events around its launch)."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
N, d, J = 50000, 20, 20
X = torch.randn(N, d, generator=torch.Generator().manual_seed(0)).to(dev)
P = torch.randn(d, J, generator=torch.Generator().manual_seed(1)).to(dev)
Z = ops.project(X, (P / math.sqrt(d)).contiguous())
V = torch.randn(N, 1, device=dev)
prep = ops.Prepared(Z)


def both(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.rpgp_profile_begin()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    avg, cnt = ctypes.c_float(), ctypes.c_int()
    lib.rpgp_profile_end(ctypes.byref(avg), ctypes.byref(cnt))
    return e0.elapsed_time(e1) / reps, avg.value


for world in (1, 2, 4, 8):
    for r in sorted({0, world // 2, world - 1}):
        tot, ker = both(lambda: ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.0, shard=(world, r)) if world > 1 else
                        ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.0))
        print("world %d rank %d: call %.3f ms, tile kernel %.3f ms, around it %.1f us" % (world, r, tot, ker, (tot - ker) * 1e3))
