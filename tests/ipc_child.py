"""Child process of tests/test_ipc_comm_gpu.py: W ranks that ALL use device 0 (the test box has one GPU; RCCL refuses
two ranks on one device, rpgp_comm does not care), bootstrap over a gloo process group, data path entirely through
rpgp_comm (IPC-mapped peer buffers).  Launched as FRESH processes through torch.distributed.run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

dist.init_process_group(backend="gloo")
world, rank = dist.get_world_size(), dist.get_rank()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)

from rpgp_amd.distributed import Reducer

red = Reducer(backend="ipc", max_bytes=8 << 20)
assert red.backend == "ipc" and red.world_size == world

# (1) the collective itself: sizes around the one-shot / two-shot switch, both dtypes, odd counts, repeated calls (the
#     staging buffers alternate and are reused), against gloo's all-reduce of host copies
g = torch.Generator().manual_seed(100 + rank)
for dtype in (torch.float32, torch.float64):
    for count in (1, 7, 1000, 50000, 131072, 131073, 550000, 1000003):
        if count * (8 if dtype == torch.float64 else 4) > red.max_bytes:
            continue
        for rep in range(3):
            x = torch.randn(count, generator=g, dtype=dtype)
            ref = x.clone()
            dist.all_reduce(ref)
            y = x.to(dev)
            red.all_reduce_(y)
            torch.cuda.synchronize()
            got = y.cpu()
            err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
            assert err < (1e-12 if dtype == torch.float64 else 2e-6), (dtype, count, rep, err)
            # bit-identical on every rank
            allv = [torch.zeros_like(got) for _ in range(world)]
            dist.all_gather(allv, got)
            assert all(torch.equal(a, got) for a in allv), ("ranks differ", dtype, count)
red.check()
dist.barrier()
if rank == 0:
    print("IPC_CHILD_OK world=%d" % world)
red.close()
dist.destroy_process_group()
