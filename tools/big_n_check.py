import sys, time; sys.path.insert(0, "/root/repo")
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
for N, J, T in ((300000, 3, 1), (200000, 20, 2)):
    Z = torch.randn(N, J, generator=torch.Generator().manual_seed(0)).to(dev)
    u = torch.randn(N, T, generator=torch.Generator().manual_seed(1)).to(dev)
    v = torch.randn(N, T, generator=torch.Generator().manual_seed(2)).to(dev)
    prep = ops.Prepared(Z)
    t0 = time.perf_counter(); Kv = ops.mvm_sym_prepared(prep, v, 1.0 / J, 0.1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    Ku = ops.mvm_sym_prepared(prep, u, 1.0 / J, 0.1)
    a, b = float((u.double() * Kv.double()).sum()), float((v.double() * Ku.double()).sum())
    rows = torch.arange(0, N, N // 257, device=dev)[:256]
    blk = ops.mvm_rect(Z[rows].contiguous(), Z, v, 1.0 / J) + 0.1 * v[rows]
    d = ops.mvm_sym(Z, v, 1.0 / J, 0.1)
    print("N", N, "J", J, "fast_ok", prep.fast_ok, "time %.1f ms" % (dt * 1e3), "symmetry rel", abs(a - b) / abs(a),
          "rows vs rect", float((Kv[rows] - blk).norm() / blk.norm()), "direct vs prepared", float((d - Kv).norm() / Kv.norm()))
