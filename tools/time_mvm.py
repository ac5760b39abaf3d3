"""Quick timing of the fused MVM at a given N (HIP events on the current stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpgp_amd import ops

def run(N, J, T, d=20, reps=10):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, d, generator=g)
    P = torch.randn(d, J, generator=torch.Generator().manual_seed(1))
    Z = ((X / (d ** 0.5)) @ P).to(dev)
    V = torch.randn(N, T, generator=torch.Generator().manual_seed(3)).to(dev)
    out = ops.mvm_sym(Z, V, 1.0 / J, 0.1)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.mvm_sym(Z, V, 1.0 / J, 0.1, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    balg = 4.0 * N * N + 4.0 * N * (d + 2 * T)
    prep = ops.Prepared(Z)
    outp = ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1)
    rel = float((outp - out).norm() / out.norm())
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        ops.mvm_sym_prepared(prep, V, 1.0 / J, 0.1, out=outp)
    e1.record(); torch.cuda.synchronize()
    msp = e0.elapsed_time(e1) / reps
    print("   prepared/factorised: %.3f ms/MVM (%.2fx), rel diff vs direct %.2e, max|a|=%.2f" % (msp, ms / msp, rel, prep.max_abs))
    print("N=%d J=%d T=%d: %.3f ms/MVM  %.1f MVM/s  pair-terms/s=%.3e  dense-equiv %.1f GB/s (%.1f%% of 8 TB/s)"
          % (N, J, T, ms, 1e3 / ms, 0.5 * N * N * J / (ms * 1e-3), balg / (ms * 1e-3) / 1e9, balg / (ms * 1e-3) / 8e12 * 100))

if __name__ == "__main__":
    for N, J, T in [(8192, 20, 1), (16599, 20, 1), (50000, 20, 1), (50000, 20, 11), (50000, 8, 1), (50000, 4, 1)]:
        run(N, J, T)
