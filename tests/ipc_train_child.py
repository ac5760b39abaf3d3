"""Child process of tests/test_ipc_comm_gpu.py: `train_exact_gp` END TO END under a process group with W ranks on device
0 — the row-sharded SKI spec (BASELINE config 5's additive_spread_prescale_Jd_ski) and the pair-sharded
additive_rp_prescale_J20 — all-reduces through rpgp_comm (RPGP_COMM=ipc), the solves in the sharded native executor.
Rank 0 of a world-1 launch writes the single-process reference; the multi-rank launches compare against it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["RPGP_COMM"] = "ipc"
import numpy as np
import torch
import torch.distributed as dist

dist.init_process_group(backend="gloo")
world, rank = dist.get_world_size(), dist.get_rank()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ref_path = sys.argv[1]

from rpgp_amd import linear_cg as lcg, settings, specs, training
from rpgp_amd.distributed import JShard, RowShard


def fit(case):
    g = torch.Generator().manual_seed(7)
    double = case == "ski64"          # `--double` (training_routines.py:481) for the row-sharded operator: round 6
    if case in ("ski", "ski64"):
        spec = specs.get("additive_spread_prescale_Jd_ski")
        mk = dict(spec["model_kwargs"], ski_options={"grid_size": 256, "num_dims": 1})
        X = torch.randn(6000, 3, generator=g)
    else:
        spec = specs.get("additive_rp_prescale_J20")
        mk = dict(spec["model_kwargs"])
        X = torch.randn(3000, 6, generator=g)
    y = torch.sin(X).sum(1) + 0.05 * torch.randn(X.shape[0], generator=g)
    losses = []
    tk = dict(spec["train_kwargs"], max_iter=4, init_iters=1, loss_log=losses)
    torch.manual_seed(11)
    np.random.seed(11)
    with settings.cg_tolerance(1e-4), settings.eval_cg_tolerance(1e-6), settings.deterministic_probes(True):
        metrics, pred, model = training.train_exact_gp(X[:-200], y[:-200], X[-200:], y[-200:], spec["kind"], mk, tk,
                                                       devices=["cuda:0"], skip_random_restart=True,
                                                       skip_posterior_variances=True, double=double)
    return losses, metrics, pred, model


out = {}
for case in ("ski", "rp", "ski64"):
    n_sh = lcg.stats.get("native_sharded_calls", 0)
    losses, metrics, pred, model = fit(case)
    params = torch.cat([p.detach().reshape(-1).double().cpu() for p in model.parameters()])
    if world == 1:
        assert model.covar_module.shard is None
        out[case] = {"losses": losses, "nmll": metrics["prior_train_nmll"], "pred": pred, "params": params}
        continue
    ref = torch.load(ref_path)[case]
    shard = model.covar_module.shard
    assert isinstance(shard, RowShard if case in ("ski", "ski64") else JShard) and shard.world_size == world
    if case == "ski64":
        # float64 parity kernels, row-sharded: the torch CG loop with all-reduced inner products (the native executor is
        # float32) — the same model as the single-process float64 fit to the float64 summation order
        assert all(p.dtype == torch.float64 for p in model.parameters())
        assert lcg.stats.get("row_sharded_calls", 0) > 0
        ltol, ptol = 1e-6, 1e-5
    else:
        assert lcg.stats.get("native_sharded_calls", 0) > n_sh, "the sharded solves did not run in the native executor"
        ltol, ptol = 1e-3, 5e-3
    for a, b in zip(losses, ref["losses"]):
        assert abs(a - b) < ltol * max(1.0, abs(b)), (case, losses, ref["losses"])   # fp32 + SLQ on differently ordered sums
                                                                                    # (the 1e-5 check is the float64 gloo test)
    assert abs(metrics["prior_train_nmll"] - ref["nmll"]) < ltol * max(1.0, abs(ref["nmll"]))
    rel_pred = float((pred - ref["pred"]).norm() / ref["pred"].norm())
    assert rel_pred < ptol, (case, rel_pred)       # (two fp32 CG solves of the mean cache, different summation orders)
    # (Adam turns a near-zero gradient component of either sign into an lr-sized step: the parameters are only loosely
    #  comparable in fp32; the losses and predictions above are the tight checks)
    assert float((params - ref["params"]).abs().max()) < 0.1, (case, float((params - ref["params"]).abs().max()))
    allp = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(allp, params)
    assert all(torch.equal(a, params) for a in allp), "ranks trained different models"
if world == 1:
    torch.save(out, ref_path)
dist.barrier()
if rank == 0:
    print("IPC_TRAIN_CHILD_OK world=%d" % world)
dist.destroy_process_group()
