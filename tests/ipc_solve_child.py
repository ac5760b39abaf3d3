"""Child process of tests/test_ipc_comm_gpu.py: the SHARDED native mBCG executor with W ranks on device 0, all-reduces
through rpgp_comm (RPGP_COMM=ipc), bootstrap over gloo.  Every sharded solve is compared with the unsharded native
solve of the same system computed by the same process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["RPGP_COMM"] = "ipc"
import torch
import torch.distributed as dist

dist.init_process_group(backend="gloo")
world, rank = dist.get_world_size(), dist.get_rank()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)

from rpgp_amd import linear_cg as lcg, settings
from rpgp_amd.distributed import JShard, RowShard, get_reducer
from rpgp_amd.operators import (AdditiveRPOperator, AddedDiagOperator, RowShardedSKIOperator, SKIAdditiveOperator,
                                SymCachedOperator, row_sharded_preconditioner)
from rpgp_amd.precond import build_preconditioner

assert get_reducer().backend == "ipc"


def same_on_all_ranks(t, what):
    h = t.detach().cpu().contiguous()
    allv = [torch.zeros_like(h) for _ in range(world)]
    dist.all_gather(allv, h)
    assert all(torch.equal(a, h) for a in allv), "ranks hold different " + what


g = torch.Generator().manual_seed(0)
# ---- (1) replicated vectors, partial products (pair- and J-sharded fused operator, pair-sharded packed cache) --------
N, J, T = 6000, 20, 11
Z = (torch.randn(N, J, generator=g) * 0.7).to(dev)
B = torch.randn(N, T, generator=g).to(dev)
s, noise = torch.tensor(0.9, device=dev), 0.5
base = AdditiveRPOperator(Z, None, s, 1.0 / J)
full = AddedDiagOperator(base, torch.tensor(noise, device=dev))
pre = build_preconditioner(base, noise, settings)
n0 = lcg.stats.get("native_calls", 0)
x_ref, t_ref = lcg.linear_cg(full._matmul, B, n_tridiag=10, tolerance=1e-5, max_iter=400, preconditioner=pre, operator=full)
it_ref = lcg.stats["last_iterations"]
assert lcg.stats.get("native_calls", 0) == n0 + 1 and lcg.stats.get("native_sharded_calls", 0) == 0
true_res = float((full._matmul(x_ref) - B).norm() / B.norm())
assert true_res < 2e-4, true_res        # (fp32 floor ~ cond * 6e-8)
for mode in ("pairs", "j"):
    sh = JShard(J, mode=mode)
    op = AddedDiagOperator(AdditiveRPOperator(Z, None, s, 1.0 / J, shard=sh), torch.tensor(noise, device=dev))
    ns = lcg.stats.get("native_sharded_calls", 0)
    x, t = lcg.linear_cg(op._matmul, B, n_tridiag=10, tolerance=1e-5, max_iter=400, preconditioner=pre, operator=op)
    assert lcg.stats.get("native_sharded_calls", 0) == ns + 1, "the sharded solve did not run in the native executor"
    rel = float((x - x_ref).norm() / x_ref.norm())
    assert rel < 5e-4, (mode, rel)
    assert abs(lcg.stats["last_iterations"] - it_ref) <= 1, (mode, lcg.stats["last_iterations"], it_ref)
    assert torch.allclose(t[:, :8, :8], t_ref[:, :8, :8], rtol=2e-3, atol=1e-4), mode
    res = float((full._matmul(x) - B).norm() / B.norm())
    assert res < 2e-4, (mode, res)
    same_on_all_ranks(x, "solutions (%s)" % mode)
    # a single right-hand side (the mean-cache solve)
    x1 = lcg.linear_cg(op._matmul, B[:, :1].contiguous(), tolerance=1e-5, max_iter=400, preconditioner=pre, operator=op)
    assert float((full._matmul(x1) - B[:, :1]).norm() / B[:, :1].norm()) < 2e-4
sh = JShard(J, mode="pairs")
shard_base = AdditiveRPOperator(Z, None, s, 1.0 / J, shard=sh)
for wide in (False, True):
    cache = shard_base.to_symcache(wide=wide)
    assert cache is not None and cache.world == world
    op = SymCachedOperator(cache, shard_base._scale, noise, diag_value=shard_base._scale * J, shard=sh)
    ns = lcg.stats.get("native_sharded_calls", 0)
    x = lcg.linear_cg(op._matmul, B, tolerance=1e-5, max_iter=400, preconditioner=pre, operator=op)
    assert lcg.stats.get("native_sharded_calls", 0) == ns + 1
    assert float((x - x_ref).norm() / x_ref.norm()) < 5e-4, wide
    same_on_all_ranks(x, "cached solutions")
    del cache, op

# ---- (2) row-sharded SKI: local rows, histogram + inner products all-reduced inside the executor ------------------------
for (N, J, T, G, tol) in ((40000, 3, 11, 1024, 1e-4), (9001, 5, 1, 256, 1e-5), (world + 1, 2, 2, 64, 1e-5), (2, 2, 1, 64, 1e-5)):     # the last: an empty rank at world 3
    Z = torch.randn(N, J, generator=g).to(dev)
    B = torch.randn(N, T, generator=g).to(dev)
    noise = 0.2
    ref_base = SKIAdditiveOperator(Z, None, s, 1.0 / J, grid_size=G)
    ref_op = AddedDiagOperator(ref_base, torch.tensor(noise, device=dev))
    rs = RowShard(N)
    op = RowShardedSKIOperator(Z[rs.r0:rs.r1], s, 1.0 / J, rs, grid_size=G, noise=noise)
    if N >= 2000:
        pre_ref = build_preconditioner(ref_base, noise, settings)
        pre_sh = row_sharded_preconditioner(op, 15)
    else:
        pre_ref = pre_sh = None
    x_ref = lcg.linear_cg(ref_op._matmul, B, tolerance=tol, max_iter=500, preconditioner=pre_ref, operator=ref_op)
    it_ref = lcg.stats["last_iterations"]
    ns = lcg.stats.get("native_sharded_calls", 0)
    x = lcg.linear_cg(op._matmul, B[rs.r0:rs.r1].contiguous(), tolerance=tol, max_iter=500, preconditioner=pre_sh,
                      reduce=rs.all_reduce_, global_size=N, operator=op)
    assert lcg.stats.get("native_sharded_calls", 0) == ns + 1, "the row-sharded solve did not run in the native executor"
    assert x.shape == (rs.local_rows, T)
    its = torch.tensor([float(lcg.stats["last_iterations"])])
    it_all = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(it_all, its)
    assert all(float(a) == float(its) for a in it_all), "ranks stopped at different iterations"
    assert abs(float(its) - it_ref) <= 2, (float(its), it_ref)
    sq = torch.stack([(x - x_ref[rs.r0:rs.r1]).double().pow(2).sum(), x_ref[rs.r0:rs.r1].double().pow(2).sum()])
    rs.all_reduce_(sq)
    rel = float((sq[0] / sq[1]).sqrt())
    assert rel < 30 * tol, (N, rel)
    # true residual of the assembled solution
    xg = torch.zeros(N, T, device=dev)
    xg[rs.r0:rs.r1] = x
    rs.all_reduce_(xg)
    res = float((ref_op._matmul(xg) - B).norm() / B.norm())
    assert res < max(3 * tol, 2e-4), (N, res)          # (true residual: the fp32 floor of the system, not the tolerance)
get_reducer().check()
dist.barrier()
if rank == 0:
    print("IPC_SOLVE_CHILD_OK world=%d" % world)
dist.destroy_process_group()
