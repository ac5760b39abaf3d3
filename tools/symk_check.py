"""Packed symmetric cache: correctness against the fused / dense products and timing.  usage: symk_check.py [N ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops
dev = torch.device("cuda:0")
J = 20
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
sizes = [int(a) for a in sys.argv[1:]] or [1000, 4097, 7372, 14939, 30000, 50000]
for N in sizes:
    Z = torch.randn(N, J, generator=torch.Generator().manual_seed(0)).to(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    WIDE = os.environ.get("SYMK_WIDE", "0") == "1"
    C = ops.SymCache(Z, wide=WIDE); torch.cuda.synchronize(); tb1 = time.perf_counter() - t0
    tb = timeit(lambda: ops.SymCache(Z, wide=WIDE), 3)
    for T in (1, 4, 11, 12, 16, 20):
        V = torch.randn(N, T, generator=torch.Generator().manual_seed(T)).to(dev)
        ref = ops.mvm_sym(Z.double(), V.double(), 1.0 / J, 0.1) if N <= 8000 else ops.mvm_sym(Z, V, 1.0 / J, 0.1).double()
        out = ops.symcache_mvm(C, V, 1.0 / J, 0.1)
        err = float((out.double() - ref).norm() / ref.norm())
        # pair-sharded halves sum to the whole
        ts = timeit(lambda: ops.symcache_mvm(C, V, 1.0 / J, 0.1))
        print("N=%d T=%d  err %.2e  symcache %.4f ms (%.2f TB/s dense-equivalent, %.2f TB/s of its own bytes)  build %.3f ms (first %.1f), cache %.2f GB" % (
            N, T, err, ts, 4.0 * N * N / ts / 1e9, C.nbytes / ts / 1e9, tb, tb1 * 1e3, C.nbytes / 1e9), flush=True)
        assert err < 5e-6, err
    if N <= 20000:
        V = torch.randn(N, 11, generator=torch.Generator().manual_seed(5)).to(dev)
        whole = ops.symcache_mvm(C, V, 1.0 / J, 0.1)
        parts = None
        for r in range(3):
            Cr = ops.SymCache(Z, shard=(3, r), wide=WIDE)
            o = ops.symcache_mvm(Cr, V, 1.0 / J, 0.1 if r == 0 else 0.0)
            parts = o if parts is None else parts + o
        print("   3-way pair shards: rel diff %.2e" % float((parts - whole).norm() / whole.norm()))
    del C
