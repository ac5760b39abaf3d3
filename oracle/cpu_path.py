"""ORACLE / CPU BASELINE — test and measurement infrastructure only (never imported by the product package).

A torch-CPU float32 restatement ("port") of the tensor-op sequence the reference runs for the additive-RP
kernel MVM on CPU through GPyTorch.  GPyTorch itself cannot be
installed in this image, so this is what `bench.py` times as `cpu_baseline` (kind "port") and what tests
compare with the float64 oracle (oracle/dense_gp.py).

Op sequence restated:
  scaled_projection_kernel.py:21-27   x.div(lengthscale) -> Linear projection            (project)
  AdditiveStructureKernel(ScaleKernel(RBF)) built at training_routines.py:148-159,169-171, called with
  last_dim_is_batch=True: inputs reshaped J x N x 1, squared distance, div(-2).exp(), x outputscale(1/J), sum over J
  outer ScaleKernel (training_routines.py:406) and K @ V in linear_cg (from fitting/optimizing.py:67-71).
When J*N^2*4 B does not fit in RAM the reference needs gpytorch.beta_features.checkpoint_kernel(B)
(gp_experiment_runner.py:250,330): the same build on B-row slices inside every MVM — `row_chunk` below.
"""
import time

import torch


def project(X, P, lengthscale, prescale=True):
    if prescale:
        return (X / lengthscale.reshape(1, -1)) @ P
    return (X @ P) / lengthscale.reshape(1, -1)


def kernel_rows(Z1, Z2, outputscale, weight):
    """Dense (rows of Z1) x (all of Z2) block, built the way the batch kernel does: J x B x N, exp, sum."""
    z1 = Z1.t().unsqueeze(-1)            # J x B x 1
    z2 = Z2.t().unsqueeze(-2)            # J x 1 x N
    sq = (z1 - z2).pow_(2)               # J x B x N
    k = sq.div_(-2).exp_().mul_(weight)  # ScaleKernel(outputscale=1/J) on each batch member
    return k.sum(dim=0).mul_(outputscale)


def mvm(Z, V, outputscale, noise, weight=None, row_chunk=1024, rows=None):
    """out[rows] = (s * K_add(Z,Z) + noise I)[rows] @ V, chunked over rows (checkpoint_kernel semantics).
    `rows` = (start, stop) restricts the computation to a slice of output rows (bounded CPU sample)."""
    N, J = Z.shape
    w = (1.0 / J) if weight is None else weight
    r0, r1 = (0, N) if rows is None else rows
    out = torch.empty((r1 - r0, V.shape[1]), dtype=Z.dtype)
    for s in range(r0, r1, row_chunk):
        e = min(s + row_chunk, r1)
        Kc = kernel_rows(Z[s:e], Z, outputscale, w)
        out[s - r0:e - r0] = Kc @ V + noise * V[s:e]
    return out


def time_mvm_sample(Z, V, outputscale, noise, budget_s=15.0, row_chunk=1024):
    """Time a bounded sample of one N x N MVM (whole row chunks until ~budget_s of CPU work) and extrapolate to
    the full MVM.  Returns dict(mvm_per_s, sample_rows, sample_s, threads, out_sample)."""
    N = Z.shape[0]
    # warm-up chunk (allocator, thread pool)
    mvm(Z, V, outputscale, noise, row_chunk=row_chunk, rows=(0, min(row_chunk, N)))
    t0 = time.perf_counter()
    done = 0
    outs = []
    while done < N:
        e = min(done + row_chunk, N)
        outs.append(mvm(Z, V, outputscale, noise, row_chunk=row_chunk, rows=(done, e)))
        done = e
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    full_s = dt * N / done
    return {"mvm_per_s": 1.0 / full_s, "sample_rows": done, "sample_s": dt, "threads": torch.get_num_threads(),
            "out_sample": torch.cat(outs, dim=0)}
