"""Sweep of the cached-K product's resident-round sizing (one child process per setting: the occupancy answer is cached
per process).  usage: gemv_sweep.py"""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from rpgp_amd import ops
    dev = torch.device("cuda:0")
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    res = []
    for N in (7372, 14939, 30000, 50000):
        Z = torch.randn(N, 20, generator=torch.Generator().manual_seed(0)).to(dev)
        K = ops.dense(Z, Z, 0.05, pad=True)
        for T in (1, 4, 11):
            V = torch.randn(N, T, device=dev)
            ref = (K[:, :N].double() @ V.double() + 0.1 * V.double())
            out = ops.dense_mvm(K, V, 0.1)
            err = float((out.double() - ref).norm() / ref.norm())
            assert err < 1e-5, err
            res.append("%d/%d %.4f" % (N, T, timeit(lambda: ops.dense_mvm(K, V, 0.1))))
        del K
    print("percu=%s | %s" % (os.environ.get("RPGP_GEMV_PERCU", "auto"), "  ".join(res)), flush=True)
else:
    for pc in ("auto", "3", "4", "5", "6", "8"):
        env = dict(os.environ)
        if pc != "auto": env["RPGP_GEMV_PERCU"] = pc
        subprocess.run([sys.executable, __file__, "child"], env=env)
