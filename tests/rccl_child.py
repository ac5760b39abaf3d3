"""Child process of tests/test_rccl_gpu.py: one rank per GPU, process group on RCCL ("nccl") exactly as bench.py and
runner.py set it up; exercises every collective the multi-GPU paths issue, through the real HIP backend.
Launched as a FRESH process (the parent test process never initialises a process group)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
world, rank = dist.get_world_size(), dist.get_rank()
dev = torch.device("cuda", local_rank)

t = torch.ones(1000, device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
assert float(t[0]) == world

from rpgp_amd import ops, linear_cg as lcg
from rpgp_amd.distributed import JShard, RowShard
from rpgp_amd.operators import (AdditiveRPOperator, AddedDiagOperator, RowShardedSKIOperator, SKIAdditiveOperator,
                                row_sharded_preconditioner)

g = torch.Generator().manual_seed(0)
# (1) pair- and J-sharded exact MVM: one all-reduce of the N x T partial per product
N, J, T = 3000, 20, 2
Z = torch.randn(N, J, generator=g).to(dev)
V = torch.randn(N, T, generator=g).to(dev)
s = torch.tensor(0.9, device=dev)
full = AddedDiagOperator(AdditiveRPOperator(Z, None, s, 1.0 / J), torch.tensor(0.1, device=dev))._matmul(V)
for mode in ("pairs", "j"):
    sh = JShard(J, mode=mode)
    got = AddedDiagOperator(AdditiveRPOperator(Z, None, s, 1.0 / J, shard=sh), torch.tensor(0.1, device=dev))._matmul(V)
    rel = float((got - full).norm() / full.norm())
    assert rel < 1e-6, (mode, rel)

# (1b) cached-K mode under pair-sharding: every rank's packed symmetric cache holds its share of the pairs, the partial
#      products meet in the same all-reduce (world = 1: the whole cache, no collective)
from rpgp_amd.operators import SymCachedOperator
V11 = torch.randn(3000, 11, generator=g).to(dev)
sh = JShard(20, mode="pairs")
base = AdditiveRPOperator(Z, None, s, 1.0 / J, shard=sh)
full11 = AddedDiagOperator(AdditiveRPOperator(Z, None, s, 1.0 / J), torch.tensor(0.1, device=dev))._matmul(V11)
for wide in (False, True):
    cache = base.to_symcache(wide=wide)
    assert cache is not None and cache.world == world
    got = SymCachedOperator(cache, base._scale, 0.1, shard=sh)._matmul(V11)
    rel = float((got - full11).norm() / full11.norm())
    assert rel < 2e-6, ("symcache", wide, rel)

# (2) row-sharded SKI: float64 histogram all-reduce (SUM), grid-range all-reduce (MIN), preconditioner pivots (MAX /
#     MIN all-reduces + broadcasts), CG scalars
N, J, T, G = 40000, 3, 11, 1024
Z = torch.randn(N, J, generator=g).to(dev)
V = torch.randn(N, T, generator=g).to(dev)
noise = 0.2
ref_op = SKIAdditiveOperator(Z, None, s, 1.0 / J, grid_size=G)
ref = AddedDiagOperator(ref_op, torch.tensor(noise, device=dev))._matmul(V)
rs = RowShard(N)
op = RowShardedSKIOperator(Z[rs.r0:rs.r1], s, 1.0 / J, rs, grid_size=G, noise=noise)
out = op._matmul(V[rs.r0:rs.r1])
rel = float((out - ref[rs.r0:rs.r1]).norm() / ref[rs.r0:rs.r1].norm())
assert rel < 2e-6, rel
pre = row_sharded_preconditioner(op, 15)
x = lcg.linear_cg(op._matmul, V[rs.r0:rs.r1].clone(), tolerance=1e-4, max_iter=500, preconditioner=pre,
                  reduce=rs.all_reduce_, global_size=N)
res = op._matmul(x) - V[rs.r0:rs.r1]
sq = torch.stack([res.double().pow(2).sum(), V[rs.r0:rs.r1].double().pow(2).sum()])
rs.all_reduce_(sq)
assert float((sq[0] / sq[1]).sqrt()) < 5e-4, float((sq[0] / sq[1]).sqrt())
# (3) the native executor's sharded modes with the RCCL hook (a ctypes callback that issues dist.all_reduce on the launch
#     stream between the executor's enqueue-only phases).  `force=True` keeps the hook in the loop at world size 1.
from rpgp_amd import backend as _be
from rpgp_amd.distributed import Reducer
from rpgp_amd.precond import build_preconditioner
from rpgp_amd import settings
forced = Reducer(backend="rccl", force=True)
be = _be.get_backend()
Nn, Jn, Tn = 5000, 20, 11
Zn = (torch.randn(Nn, Jn, generator=g) * 0.7).to(dev)
Bn = torch.randn(Nn, Tn, generator=g).to(dev)
base_n = AdditiveRPOperator(Zn, None, s, 1.0 / Jn)
full_n = AddedDiagOperator(base_n, torch.tensor(0.05, device=dev))
pre_n = build_preconditioner(base_n, 0.05, settings)
x_ref = lcg.linear_cg(full_n._matmul, Bn, tolerance=1e-5, max_iter=400, preconditioner=pre_n, operator=full_n)
for mode in ("pairs", "j"):
    shn = JShard(Jn, mode=mode)
    shard_op = AddedDiagOperator(AdditiveRPOperator(Zn, None, s, 1.0 / Jn, shard=shn), torch.tensor(0.05, device=dev))
    desc, keep = shard_op.base.native_descriptor(0.05)
    xs, _, _, its, mres = be.mbcg_solve(desc, Bn, 1e-5, 400, L=pre_n.L, Cinv=pre_n.cinv(), sigma2=pre_n.noise,
                                        sharding=("partial", forced, Nn))
    assert float((xs - x_ref).norm() / x_ref.norm()) < 2e-4, (mode, "hooked solve differs")
    if world > 1:
        xa = lcg.linear_cg(shard_op._matmul, Bn, tolerance=1e-5, max_iter=400, preconditioner=pre_n, operator=shard_op)
        assert lcg.stats.get("native_sharded_calls", 0) > 0
        assert float((xa - x_ref).norm() / x_ref.norm()) < 2e-4
# row mode through the hook
forced_rows = ("rows", forced, N)
desc, keep = op.native_descriptor()
xr, _, _, its, mres = be.mbcg_solve(desc, V[rs.r0:rs.r1].contiguous(), 1e-4, 500, L=pre.L, Cinv=pre.cinv(), sigma2=pre.noise,
                                    sharding=forced_rows)
res = op._matmul(xr) - V[rs.r0:rs.r1]
sq = torch.stack([res.double().pow(2).sum(), V[rs.r0:rs.r1].double().pow(2).sum()])
rs.all_reduce_(sq)
assert float((sq[0] / sq[1]).sqrt()) < 5e-4, float((sq[0] / sq[1]).sqrt())
dist.barrier()
if rank == 0:
    print("RCCL_CHILD_OK world=%d iters=%d sharded_rel=%.2e" % (world, lcg.stats["last_iterations"], rel))
dist.destroy_process_group()
