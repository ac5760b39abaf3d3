"""J-sharding across ranks with a gloo process group on CPU (world_size 2 and 3): partition logic, the sharded MVM
(one all-reduce, noise added once) and sharded gradients equal the single-process results."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_j_partition():
    from rpgp_amd.distributed import j_partition
    assert j_partition(20, 8) == [(0, 3), (3, 6), (6, 9), (9, 12), (12, 14), (14, 16), (16, 18), (18, 20)]
    assert j_partition(20, 4) == [(0, 5), (5, 10), (10, 15), (15, 20)]
    assert j_partition(3, 8)[3:] == [(3, 3)] * 5
    assert j_partition(7, 1) == [(0, 7)]
    with pytest.raises(ValueError):
        j_partition(0, 2)


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rpgp_amd import backend, settings
        from rpgp_amd.distributed import JShard
        from rpgp_amd.operators import AdditiveRPOperator, AddedDiagOperator
        from tests.oracle_backend import OracleBackend
        from tests.test_host_stack import _build_model, _problem
        backend.set_backend(OracleBackend())
        torch.manual_seed(0)
        N, J, T = 90, 7, 3
        Z = torch.randn(N, J)
        V = torch.randn(N, T)
        s = torch.tensor(0.8)
        shard = JShard(J)
        assert shard.world_size == world and shard.rank == rank
        full = AddedDiagOperator(AdditiveRPOperator(Z, None, s, 1.0 / J), torch.tensor(0.25))._matmul(V)
        shd = AddedDiagOperator(AdditiveRPOperator(Z, None, s, 1.0 / J, shard=shard), torch.tensor(0.25))._matmul(V)
        assert torch.allclose(full, shd, atol=1e-5), "sharded MVM differs"
        L, R = torch.randn(N, T), torch.randn(N, T)
        g_full = AdditiveRPOperator(Z, None, s, 1.0 / J)._bilinear_derivative(L, R)
        g_shd = AdditiveRPOperator(Z, None, s, 1.0 / J, shard=shard)._bilinear_derivative(L, R)
        assert torch.allclose(g_full[0], g_shd[0], atol=1e-4) and torch.allclose(g_full[1], g_shd[1], atol=1e-4)
        # a full MLL evaluation + backward with a sharded kernel equals the unsharded one on every rank
        X, y, P, ls, noise, sc = _problem(N=70, d=4, J=5, seed=1)
        vals = []
        for use_shard in (False, True):
            model, lik, mll = _build_model(X, y, P, ls, noise, sc)
            if use_shard:
                model.covar_module.shard = JShard(5)
            model.train()
            with settings.max_cholesky_size(0), settings.cg_tolerance(1e-7), settings.deterministic_probes(True):
                v = mll(model(X), y)
                v.backward()
            vals.append((v.item(), model.covar_module.base_kernel.raw_lengthscale.grad.clone()))
        assert abs(vals[0][0] - vals[1][0]) < 1e-5 * abs(vals[0][0])
        assert torch.allclose(vals[0][1], vals[1][1], rtol=1e-3, atol=1e-6)
        gathered = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([vals[1][0]], dtype=torch.float64))
        assert all(abs(float(g) - vals[1][0]) < 1e-12 for g in gathered), "ranks disagree"
        open(os.path.join(tmpdir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_operator_gloo(tmp_path, world):
    port = 29600 + world + (os.getpid() % 200)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert os.path.exists(tmp_path / ("ok%d" % r))
