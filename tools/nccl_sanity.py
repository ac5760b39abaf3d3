"""RCCL sanity on the GPU box (world size 1): process-group init exactly as bench.py / runner do it, one all-reduce,
and JShard.sharded_mvm through the real backend."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
t = torch.ones(1000, device="cuda")
dist.all_reduce(t)
torch.cuda.synchronize()
from rpgp_amd import ops
from rpgp_amd.distributed import JShard
Z = torch.randn(3000, 20, device="cuda"); V = torch.randn(3000, 2, device="cuda")
sh = JShard(20)
full = ops.mvm_sym(Z, V, 0.05, 0.1)
got = sh.sharded_mvm(lambda j0, j1, nz: ops.mvm_sym(Z, V, 0.05, nz, shard=(dist.get_world_size(), dist.get_rank())), V, 0.1)
print("world", dist.get_world_size(), "allreduce ok", float(t[0]), "sharded rel diff", float((got - full).norm() / full.norm()))
dist.barrier(); dist.destroy_process_group()
