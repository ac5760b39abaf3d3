#!/usr/bin/env python3
"""bench.py — headline benchmark: additive-RP kernel MVMs/sec at N=50k, J=20 (BASELINE.json metric).

A "step" is ONE application  out = (s*K_add + sigma^2 I) v  of the fused HIP kernel to one vector (T=1) with all
inputs (Z = projected inputs, v) resident in HBM.  With --gpus N>1 (launched through torch.distributed.run, one rank
per GPU) the J=20 additive terms are sharded across ranks and the length-N partials are summed with one RCCL
all-reduce per step (north_star): total work is fixed -> "scaling": "strong".

`--comm ipc` sums the partials with rpgp_comm (one-shot kernel over IPC-mapped peer buffers) instead of RCCL, and
`--all-ranks-on-device 0` puts every rank on ONE GPU (gloo bootstrap + rpgp_comm — RCCL refuses two ranks per device): the
complete N>1 path (rank launch, sharded kernels, all-reduce, MAX-reduced timings, the JSON line with per-rank kernel and
all-reduce times) runs on the one-GPU test box (tests/test_bench_multirank_gpu.py), labelled "not a scaling measurement".

One JSON line is printed by rank 0.  Extra objects:
  roofline     : dense-equivalent algorithmic bytes B_alg = 4N^2 + 4N(d+2T) / mean duration of the
                 dominant kernel (mvm_tile_kernel), measured with HIP events on the launch stream (rpgp_profile_*).
  cpu_baseline : oracle/cpu_path.py (torch-CPU restatement of the reference's op sequence) on a bounded row sample of
                 the same MVM, on rank 0 at N=1 only.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def make_inputs(N, d, J, T, device):
    """SURVEY.md §8(d) primary synthetic input (config C4)."""
    X = torch.randn(N, d, generator=torch.Generator().manual_seed(0))
    P = torch.randn(d, J, generator=torch.Generator().manual_seed(1))   # gen_rp(d,1,'gaussian') x J  (rp.py:12-13)
    ls = torch.full((d,), math.sqrt(d))                                  # prescale lengthscale sqrt(d): Var(Z) ~ 1
    V = torch.randn(N, T, generator=torch.Generator().manual_seed(3))
    return X.to(device), P.to(device), ls.to(device), V.to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=50000)
    ap.add_argument("--d", type=int, default=20)
    ap.add_argument("--J", type=int, default=20)
    ap.add_argument("--T", type=int, default=1)
    ap.add_argument("--shard", choices=["pairs", "j"], default="pairs", help="multi-GPU split of the MVM")
    ap.add_argument("--one-split", action="store_true", help="with --gpus N > 1 time only the split --shard names (default: both)")
    ap.add_argument("--direct", action="store_true", help="use the exact direct kernel instead of the factorised path")
    ap.add_argument("--no-extras", action="store_true", help="skip the T=11 block / full-solve context numbers")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU work for cpu_baseline (0 = skip)")
    ap.add_argument("--comm", choices=["rccl", "ipc"], default=None,
                    help="all-reduce of the sharded step: rccl (torch.distributed 'nccl' = RCCL over xGMI; default on a "
                         "multi-GPU node) or ipc (rpgp_comm: one-shot kernel over IPC-mapped peer buffers; the default — and "
                         "the only choice — with --all-ranks-on-device, because RCCL refuses two ranks per device)")
    ap.add_argument("--all-ranks-on-device", type=int, default=None, metavar="D",
                    help="run every rank on device D (one-GPU box): exercises the complete --gpus N code path — rank launch, "
                         "sharded kernels, all-reduce, MAX-reduced timings — but the ranks SHARE one GPU, so the line is marked "
                         "'not a scaling measurement'")
    ap.add_argument("--dump-result", default=None, metavar="PATH",
                    help="rank 0 saves the (all-reduced) product of the last step as .npy (tests/ compare it with the oracle)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only (gloo, no GPU work): every rank joins, one all-reduce, rank 0 prints the world "
                         "size; used by tests/ to cover the --gpus launcher on machines without GPUs")
    # (ranks started by this script's own launcher below get their flags through the environment: torch.distributed.run's
    #  parser abbreviation-matches flags such as --n against its own --nnodes / --nproc-per-node before it reaches the script)
    argv = json.loads(os.environ["RPGP_BENCH_ARGV"]) if ("RPGP_BENCH_ARGV" in os.environ and len(sys.argv) == 1) else None
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU) through
        # torch.distributed.run.  This parent has not touched the GPU (no HIP call so far) and never will; it waits
        # for the ranks and exits with their status (replaces the `--device cuda:0,cuda:1,...` single-command form of
        # gp_experiment_runner.py:263).
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env["RPGP_BENCH_ARGV"] = json.dumps(sys.argv[1:])
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
        sys.exit(subprocess.call(cmd, env=env))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or drop the launcher and let "
                         "bench.py start the ranks itself)" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_check:
        dist.init_process_group(backend="gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": dist.get_world_size(), "allreduce": float(t[0])}))
        dist.destroy_process_group()
        return
    one_dev = args.all_ranks_on_device
    comm = args.comm or ("ipc" if one_dev is not None else "rccl")
    if one_dev is not None and comm != "ipc":
        raise SystemExit("bench.py: --all-ranks-on-device needs --comm ipc (RCCL refuses several ranks on one device)")
    dev_index = one_dev if one_dev is not None else (local_rank if world > 1 else 0)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if one_dev is not None:
            # bootstrap / barrier / MAX-reduce of the timings over gloo (host); the data path is rpgp_comm on the device
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)

    from rpgp_amd import ops, _lib
    from rpgp_amd.distributed import JShard, Reducer

    N, d, J, T = args.n, args.d, args.J, args.T
    X, P, ls, V = make_inputs(N, d, J, T, device)
    Peff = (P / ls[:, None]).contiguous()
    Z = ops.project(X, Peff)
    outputscale, noise = 1.0, 0.1
    scale = outputscale / J
    shard = JShard(J)
    out = torch.empty_like(V)
    # tables of the factorised fast path: built once per Z (= once per hyper-parameter step, outside the timed region,
    # like Z itself); falls back to the exact direct kernel if the coordinate range is unsafe or --direct is given
    prep = None if args.direct else ops.Prepared(Z)
    fast = prep is not None and prep.fast_ok
    # what a hyper-parameter step pays ONCE before its MVMs and the timed region leaves out: the projection Z = X Peff and the
    # tables of the factorised form (range pass, midpoints, row / column records, the status word read back) — wall time per
    # call between synchronisations, mean of 20 after one warm call
    prepare_us = None
    if prep is not None and world == 1:
        ops.Prepared(ops.project(X, Peff))
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for _ in range(20):
            Zt = ops.project(X, Peff)
            ops.Prepared(Zt)
        torch.cuda.synchronize()
        prepare_us = (time.perf_counter() - tp) / 20 * 1e6
        del Zt

    # multi-GPU split: "pairs" (default) gives every rank an equal share of the (i,i') tile pairs with all J terms;
    # "j" is north_star's J-slice split.  Both end in ONE all-reduce of the length-N partial result per step.  With more
    # than one rank BOTH splits are timed in the same run (K steps each, same fences): `value` is the split --shard names,
    # the other one is reported beside it (multi_gpu.other_split), so the first run on a node carries north_star's
    # "J sharded across N GPUs" number whichever split is the headline.
    # the all-reduce of the sharded step: RCCL through torch.distributed, or rpgp_comm (csrc/rpgp_comm.hip)
    reducer = None
    if world > 1:
        reducer = Reducer(backend=comm, max_bytes=max(1 << 22, 4 * N * T), device=device)
    lib = _lib.load()
    import ctypes

    def fence():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    def run_split(mode):
        """W untimed + exactly K timed steps of the sharded MVM under split `mode`; MAX over ranks."""
        ps = (world, rank) if (world > 1 and mode == "pairs") else None
        ar_events = []

        def local(j0, j1, nz, o=None):
            if world > 1 and mode == "pairs":
                j0, j1 = 0, J
            if fast:
                return ops.mvm_sym_prepared(prep, V, scale, nz, j0=j0, j1=j1, out=o, shard=ps)
            return ops.mvm_sym(Z, V, scale, nz, j0=j0, j1=j1, out=o, shard=ps)

        def step(timed=False):
            if world == 1:
                return local(0, J, noise, out)
            # this rank's partial (its J-slice or its share of the tile pairs; the noise term on rank 0 only), then ONE
            # all-reduce of the N x T partial on the same stream
            partial = local(shard.j0, shard.j1, noise if rank == 0 else 0.0)
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                reducer.all_reduce_(partial)
                e1.record()
                ar_events.append((e0, e1))
            else:
                reducer.all_reduce_(partial)
            return partial

        for _ in range(args.warmup):
            res = step()
        fence()
        _lib.check(lib.rpgp_profile_begin(), "rpgp_profile_begin")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step(timed=True)
        fence()
        elapsed = time.perf_counter() - t0
        avg_ms, cnt = ctypes.c_float(0), ctypes.c_int(0)
        _lib.check(lib.rpgp_profile_end(ctypes.byref(avg_ms), ctypes.byref(cnt)), "rpgp_profile_end")
        per_rank = None
        if world > 1:
            if reducer is not None:
                reducer.check()
            ar_us = sum(a.elapsed_time(b) for a, b in ar_events) / max(len(ar_events), 1) * 1e3
            on = device if one_dev is None else torch.device("cpu")       # (gloo group: host tensors)
            mine = torch.tensor([elapsed, avg_ms.value, ar_us], device=on, dtype=torch.float64)
            allv = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allv, mine)
            tt = torch.stack(allv).cpu()
            elapsed, kernel_ms = float(tt[:, 0].max()), float(tt[:, 1].max())         # MAX over ranks
            per_rank = {"kernel_ms": [round(float(v), 4) for v in tt[:, 1]], "allreduce_us": [round(float(v), 1) for v in tt[:, 2]],
                        "elapsed_s": [round(float(v), 5) for v in tt[:, 0]]}
        else:
            kernel_ms = avg_ms.value
        return {"mode": mode, "elapsed": elapsed, "kernel_ms": kernel_ms, "per_rank": per_rank, "res": res}

    def kernel_name_of(mode):
        """The dominant kernel the split launches on THIS rank (what rocprofv3 --kernel-trace will name)."""
        tt = 1 if T == 1 else (4 if T <= 4 else 12)
        if not fast:
            return "mvm_tile_kernel<%d,%d,2,sym>" % (J if (world == 1 or mode == "pairs") else shard.j1 - shard.j0, tt)
        if world == 1 or mode == "pairs":
            # (the hand-scheduled kernel serves a rank's share of the tile pairs as it serves the whole sweep)
            kid = lib.rpgp_prepared_kernel_id(N, J, T)
            return {1: "mvm_fact_asm_kernel", 2: "mvm_fact_asm_thin_kernel<%d>" % J}.get(kid, "mvm_fact_kernel<%d,%d,2>" % (J, tt))
        pieces, left = [], shard.j1 - shard.j0         # greedy pieces of this rank's slice (kJPieces, csrc/rpgp_kernels.hip)
        while left > 0:
            pieces.append(next(q for q in (20, 10, 8, 5, 4, 3, 2, 1) if q <= left))
            left -= pieces[-1]
        # (a piece of 2 / 3 / 4 / 5 / 8 / 10 projections runs the hand-scheduled loop generated for that width)
        return " + ".join("mvm_fact_asm_thin_kernel<%d>" % q if lib.rpgp_prepared_kernel_id(N, q, T) == 2 else
                          "mvm_fact_kernel<%d,%d,2>" % (q, tt) for q in pieces) + " (rank 0's J-slice)"

    primary = run_split(args.shard if world > 1 else "single")
    other = None
    if world > 1 and not args.one_split:
        other = run_split("j" if args.shard == "pairs" else "pairs")
    elapsed, kernel_ms, per_rank, res = primary["elapsed"], primary["kernel_ms"], primary["per_rank"], primary["res"]
    if args.dump_result and rank == 0:
        import numpy as np
        np.save(args.dump_result, res.detach().cpu().numpy())
        if other is not None:
            np.save(args.dump_result + ".other.npy", other["res"].detach().cpu().numpy())

    ms_per_step = elapsed / args.steps * 1e3
    value = args.steps / elapsed
    b_alg = 4.0 * N * N + 4.0 * N * (d + 2 * T)
    peak = 8.0e12
    # per launch: at N GPUs each rank's launch covers its share of the J terms; the job-level algorithmic bytes are
    # those of one dense-equivalent MVM, attributed to the slowest rank's kernel time
    achieved = b_alg / (kernel_ms * 1e-3)

    kernel_name = kernel_name_of(args.shard)
    # literal HBM bytes per launch from the committed rocprofv3 --pmc passes (cannot be collected live); reported only
    # when the profile was taken on the same kernel + workload as this run AND on the kernel source this run was built
    # from (tools/collect_pmc.py stamps the sha256 of the kernel's source files into the file; a stale profile gives null)
    traffic = None
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "pmc_counters_current.json")))
        import hashlib
        csrc = os.path.join(ROOT, "randomly-projected-additive-gps_amd", "csrc")
        files = ["rpgp_fact_asm.hip", "rpgp_fact_asm_loop.inc"] if kernel_name == "mvm_fact_asm_kernel" else ["rpgp_kernels.hip"]
        hsh = hashlib.sha256()
        for f in files:
            hsh.update(open(os.path.join(csrc, f), "rb").read())
        same_source = prof.get("kernel_source_sha256") == hsh.hexdigest()
        if prof.get("N") == N and prof.get("J") == J and prof.get("T") == T and world == 1 and \
                prof.get("fast") == fast and same_source and prof.get("kernel") == kernel_name:
            traffic = prof["hbm_bytes_high"]
    except Exception:
        traffic = None

    result = {
        "metric": "additive-RP kernel MVMs/sec at N=50k J=20; achieved HBM GB/s vs peak",
        "value": round(value, 3),
        "unit": "MVM/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "C4 synthetic N=%d d=%d J=%d T=%d additive_rp_prescale fused symmetric MVM" % (N, d, J, T),
                   "N": N, "d": d, "J": J, "T": T, "parallelism": ("%s-shard x%d + all-reduce" % (args.shard, world)) if world > 1 else "single GPU",
                   "lengthscale": "sqrt(d)", "outputscale": outputscale, "noise": noise},
        # `frac` is the judged number: dense-equivalent algorithmic bytes per second against the 8 TB/s HBM peak (BASELINE's
        # metric).  `bound` names what actually limits the kernel — transcendental + VALU issue — and `issue_bound_frac` is the
        # kernel's efficiency against ITS bound: the exact-fp32 instruction-mix floor of 12.6 issue cycles per 64 pair-terms
        # (t-FMA 2.2 + v_exp_f32 8.2 + accumulate-FMA 2.2) at the 2.35 GHz the kernel holds.
        "roofline": {"bound": "valu_transcendental", "priced_against": "hbm", "achieved": round(achieved / 1e9, 2),
                     "peak": peak / 1e9, "unit": "GB/s",
                     "frac": round(achieved / peak, 4), "traffic": traffic,
                     "ceiling_frac": 0.604,
                     "ceiling": "zero-overhead issue floor of any exact-fp32 one-exp-per-pair-term formulation: per 64 pair-terms "
                                "t-FMA 2.2 + v_exp_f32 8.2 + accumulate-FMA 2.2 = 12.6 cycles (measured instruction costs, "
                                "profiles/r1_microbench_valu_mfma_rates.txt) x N^2 J / 2 / 64 wave-terms / (1024 SIMDs x 2.35 GHz) "
                                "= 2.07 ms = B_alg / 2.07 ms / 8 TB/s = 0.604; the north_star target 0.60 sits 1 % under it, the "
                                "kernel's own stream is 12.93 cycles",
                     "issue_bound_frac": round((0.5 * N * N * J / 64.0 * 12.6 / (1024 * 2.35e9)) / (kernel_ms * 1e-3), 4),
                     "traffic_source": ("committed rocprofv3 --pmc passes (profiles/pmc_counters_current.json), not measured "
                                        "in this run") if traffic is not None else None,
                     "limiter": "VALU + transcendental issue (v_exp_f32 at quarter rate); neither HBM nor the matrix "
                                "pipe is saturated",
                     "kernel": kernel_name, "kernel_ms": round(kernel_ms, 4),
                     "algorithmic_bytes": b_alg,
                     "pair_terms_per_s": round(0.5 * N * N * J / (kernel_ms * 1e-3), 1),
                     "note": "dense-equivalent bytes (4N^2+4N(d+2T)); the fused kernel is VALU/transcendental-bound, "
                             "its literal HBM traffic is ~MBs (see DESIGN.md)"},
    }

    if world > 1:
        from rpgp_amd.distributed import j_partition
        jtab = ", ".join("%d:%d" % ab for ab in j_partition(J, world))
        split_text = {"pairs": "pairs: every rank an equal share of the (row block, column chunk) tiles, all J terms",
                      "j": "j: rank r owns projections [%s] (north_star's split)" % jtab}
        result["config"]["split"] = split_text[args.shard]
        result["config"]["north_star_split"] = "j-shard x%d [%s] + all-reduce: %s" % (
            world, jtab, "this line's value" if args.shard == "j" else "multi_gpu.other_split of this line")
        result["config"]["comm"] = ("rpgp_comm one-shot all-reduce over IPC-mapped peer buffers" if comm == "ipc"
                                    else "RCCL all-reduce (torch.distributed nccl)")
        result["multi_gpu"] = {"per_rank_kernel_ms": per_rank["kernel_ms"], "per_rank_allreduce_us": per_rank["allreduce_us"],
                               "allreduce_us": max(per_rank["allreduce_us"]), "allreduce_bytes": 4 * N * T,
                               "per_rank_elapsed_s": per_rank["elapsed_s"],
                               "allreduce_timing": "HIP events around the collective on the launch stream (includes waiting "
                                                   "for the slowest peer's partial)"}
        if other is not None:
            # the second split, timed in the same run under the same fences (K steps after W warm-ups)
            om = other["mode"]
            result["multi_gpu"]["other_split"] = {
                "split": split_text[om], "parallelism": "%s-shard x%d + all-reduce" % (om, world),
                "value": round(args.steps / other["elapsed"], 3), "unit": "MVM/s",
                "ms_per_step": round(other["elapsed"] / args.steps * 1e3, 4), "kernel": kernel_name_of(om),
                "kernel_ms": round(other["kernel_ms"], 4), "per_rank_kernel_ms": other["per_rank"]["kernel_ms"],
                "per_rank_allreduce_us": other["per_rank"]["allreduce_us"],
                "rel_diff_vs_value_split": float((other["res"] - res).norm() / res.norm())}
        if one_dev is not None:
            result["multi_gpu"]["all_ranks_on_device"] = one_dev
            result["multi_gpu"]["note"] = ("NOT a scaling measurement: the %d ranks share ONE GPU (their kernels time-slice "
                                           "it); this run proves the N>1 code path end to end, nothing about speed-up" % world)
            result["scaling"] = "strong"

    if world == 1 and not args.no_extras:
        # context numbers (not part of `value`): the T=11 block the training solve uses (10 probes + residual), and one
        # full solve Khat^-1 y with the reference's eval tolerance (SURVEY.md §8(d) "unit of work")
        from rpgp_amd import settings, linear_cg as lcg
        from rpgp_amd.operators import AdditiveRPOperator, AddedDiagOperator
        V11 = torch.randn(N, 11, generator=torch.Generator().manual_seed(4)).to(device)
        blk = (lambda: ops.mvm_sym_prepared(prep, V11, scale, noise)) if fast else (lambda: ops.mvm_sym(Z, V11, scale, noise))
        blk()
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for _ in range(5):
            blk()
        torch.cuda.synchronize()
        t11 = (time.perf_counter() - tb) / 5
        y = torch.sin(X).sum(1)
        y = (y - y.mean()) / y.std()
        from rpgp_amd.precond import build_preconditioner
        base_op = AdditiveRPOperator(Z, None, torch.tensor(outputscale, device=device), 1.0 / J)
        khat = AddedDiagOperator(base_op, torch.tensor(noise, device=device))
        torch.cuda.synchronize()
        ts = time.perf_counter()
        pre = build_preconditioner(base_op, noise, settings)          # rank-15 pivoted Cholesky (one launch)
        alpha = lcg.linear_cg(khat._matmul, y.reshape(-1, 1), tolerance=0.01, max_iter=10000, preconditioner=pre,
                              operator=khat)                          # native mBCG executor on the fused operator
        torch.cuda.synchronize()
        t_solve = time.perf_counter() - ts
        it_fused = lcg.stats["last_iterations"]
        resid = float((khat._matmul(alpha) - y.reshape(-1, 1)).norm() / y.norm())
        # cached-K mode (K materialised once per hyper-parameter step, then a genuinely HBM-bound MFMA thin GEMM)
        cached = None
        if 4.0 * N * N < 0.3 * torch.cuda.get_device_properties(device).total_memory:
            torch.cuda.synchronize()
            tk = time.perf_counter()
            Kd = ops.dense(Z, Z, scale, pad=True)
            torch.cuda.synchronize()
            t_build_cold = time.perf_counter() - tk      # includes the first 10 GB allocation of the process
            del Kd
            torch.cuda.synchronize()
            tk = time.perf_counter()
            Kd = ops.dense(Z, Z, scale, pad=True)        # the caching allocator hands the block back: the kernel itself
            torch.cuda.synchronize()
            t_build = time.perf_counter() - tk
            ops.dense_mvm(Kd, V, noise)
            torch.cuda.synchronize()
            tk = time.perf_counter()
            for _ in range(10):
                oc = ops.dense_mvm(Kd, V, noise)
            torch.cuda.synchronize()
            t_c = (time.perf_counter() - tk) / 10
            ops.dense_mvm(Kd, V11, noise)
            torch.cuda.synchronize()
            tk = time.perf_counter()
            for _ in range(10):
                ops.dense_mvm(Kd, V11, noise)
            torch.cuda.synchronize()
            t_c11 = (time.perf_counter() - tk) / 10
            cached = {"mvm_ms": round(t_c * 1e3, 4), "mvm_per_s": round(1.0 / t_c, 1), "build_ms": round(t_build * 1e3, 3),
                      "build_cold_ms": round(t_build_cold * 1e3, 3),
                      "block_T11_ms": round(t_c11 * 1e3, 4),
                      "hbm_GBps": round(4.0 * N * N / t_c / 1e9, 1), "hbm_frac_of_8TBps": round(4.0 * N * N / t_c / 8e12, 4),
                      "rel_diff_vs_fused": float((oc - res).norm() / res.norm()),
                      "note": "literal HBM stream of the 4N^2-byte matrix (rpgp_dense_mvm); what the training / prediction solves "
                              "run automatically when K fits a quarter of HBM (settings.cache_kernel = 'auto'); not the headline"}
            del Kd
        # packed symmetric cache: the same cached-K product from half the bytes (what training / thin prediction solves
        # take automatically when 2 N^2 bytes fit a quarter of HBM)
        torch.cuda.synchronize()
        tk = time.perf_counter()
        sc = ops.SymCache(Z)
        torch.cuda.synchronize()
        t_sbuild = time.perf_counter() - tk

        def _time_sc(Vx, reps=10):
            ops.symcache_mvm(sc, Vx, scale, noise)
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(reps):
                o_ = ops.symcache_mvm(sc, Vx, scale, noise)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0_) / reps, o_
        t_s1, os1 = _time_sc(V)
        nb = sc.nbytes
        # the same mean-cache solve as above on the packed cache — the path the prediction strategy takes when K fits
        from rpgp_amd.operators import SymCachedOperator
        kc = SymCachedOperator(sc, scale, noise, diag_value=scale * J)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        pre_c = build_preconditioner(base_op, noise, settings)
        alpha_c = lcg.linear_cg(kc._matmul, y.reshape(-1, 1), tolerance=0.01, max_iter=10000, preconditioner=pre_c, operator=kc)
        torch.cuda.synchronize()
        t_solve_c = time.perf_counter() - ts
        it_c = lcg.stats["last_iterations"]
        resid_c = float((khat._matmul(alpha_c) - y.reshape(-1, 1)).norm() / y.norm())
        del sc, kc
        torch.cuda.synchronize()
        tk = time.perf_counter()
        sc = ops.SymCache(Z, wide=True)              # matrix-core tile layout: what the T = 11 training block streams
        torch.cuda.synchronize()
        t_wbuild = time.perf_counter() - tk
        t_s11, os11 = _time_sc(V11)
        ref11 = blk()
        symc = {"mvm_ms": round(t_s1 * 1e3, 4), "mvm_per_s": round(1.0 / t_s1, 1), "build_ms": round(t_sbuild * 1e3, 3),
                "block_T11_ms": round(t_s11 * 1e3, 4), "block_T11_layout": "wide (16x16 MFMA tiles)",
                "wide_build_ms": round(t_wbuild * 1e3, 3),
                "block_T11_rel_diff_vs_fused": float((os11 - ref11).norm() / ref11.norm()),
                "block_T11_own_bytes_GBps": round(nb / t_s11 / 1e9, 1), "cache_GB": round(nb / 1e9, 3),
                "own_bytes_GBps": round(nb / t_s1 / 1e9, 1), "own_bytes_frac_of_8TBps": round(nb / t_s1 / 8e12, 4),
                "dense_equivalent_GBps": round(4.0 * N * N / t_s1 / 1e9, 1),
                "rel_diff_vs_fused": float((os1 - res).norm() / res.norm()),
                "note": "rpgp_symcache_mvm: every unordered pair stored once; thin layout (rotation order of the fused sweep) for "
                        "T <= 4, wide layout (exact-fp32 MFMA on 16x16 tiles, transposed through LDS) for the T = 11 training "
                        "block; HBM-bound on N^2/2 stored values; not the headline"}
        del sc
        # SKI mode (the reference's `ski: true` specs, e.g. additive_spread_prescale_J20_ski.json): grid interpolation of
        # the same operator, O(N (J + T)) per MVM; an approximation (difference reported), never the headline
        gp = ops.ski_grid(Z, None, 1024)
        skp = ops.SkiPlan(Z, gp, 1024)              # per-step plan (points sorted by interpolation cell): what the solves use
        ops.ski_mvm(Z, Z, gp, V, scale, noise, 1024, plan=skp)
        torch.cuda.synchronize()
        tk = time.perf_counter()
        for _ in range(20):
            osk = ops.ski_mvm(Z, Z, gp, V, scale, noise, 1024, plan=skp)
        torch.cuda.synchronize()
        t_ski = (time.perf_counter() - tk) / 20
        ops.ski_mvm(Z, Z, gp, V11, scale, noise, 1024, plan=skp)
        torch.cuda.synchronize()
        tk = time.perf_counter()
        for _ in range(20):
            ops.ski_mvm(Z, Z, gp, V11, scale, noise, 1024, plan=skp)
        torch.cuda.synchronize()
        t_ski11 = (time.perf_counter() - tk) / 20
        ski = {"grid_size": 1024, "mvm_ms": round(t_ski * 1e3, 4), "mvm_per_s": round(1.0 / t_ski, 1),
               "block_T11_ms": round(t_ski11 * 1e3, 4), "rel_diff_vs_fused": float((osk - res).norm() / res.norm()),
               "note": "cubic interpolation onto a 1024-point grid + Toeplitz RBF (rpgp_ski_mvm_planned: cell-sorted scatter, no "
                       "atomics); approximate, not the headline"}
        # backward pass of one training step: the bilinear derivative with the 10 probe solves + the residual solve
        Lb = (torch.randn(N, 11, generator=torch.Generator().manual_seed(5)) * 0.1).to(device)
        Rb = (torch.randn(N, 11, generator=torch.Generator().manual_seed(6)) * 0.1).to(device)
        ops.bilinear_grad(Z, Lb, Rb, scale)
        torch.cuda.synchronize()
        tk = time.perf_counter()
        for _ in range(3):
            ops.bilinear_grad(Z, Lb, Rb, scale)
        torch.cuda.synchronize()
        t_bil = (time.perf_counter() - tk) / 3
        result["extras"] = {"cached_k": cached, "symcache": symc, "ski": ski, "bilinear_derivative_T11_ms": round(t_bil * 1e3, 4),
                            "block_T11_ms": round(t11 * 1e3, 4), "block_T11_mvm_equiv_per_s": round(11.0 / t11, 1),
                            "solve_Khat_inv_y": {"what": "mean-cache solve, rank-15 pivoted-Cholesky preconditioner, native mBCG, fused MVM",
                                                 "tolerance": 0.01, "cg_iterations": it_fused,
                                                 "seconds": round(t_solve, 4), "relative_residual": resid},
                            "solve_Khat_inv_y_packed_cache": {"what": "the same solve on the packed symmetric cache (what the "
                                                              "prediction strategy runs when K fits); cache build "
                                                              "%.1f ms not included" % (t_sbuild * 1e3),
                                                              "tolerance": 0.01, "cg_iterations": it_c,
                                                              "seconds": round(t_solve_c, 4), "relative_residual": resid_c}}

        # one optimiser step (mBCG T = 11 + SLQ + backward + Adam) of the flagship model at the BASELINE shapes C2 / C3 / C5 —
        # context for the solve the headline kernel serves, not part of `value`
        try:
            result["extras"]["optimiser_step_ms"] = _step_times(device)
        except Exception as e:                       # (never let a context number take the benchmark line down)
            result["extras"]["optimiser_step_ms"] = {"error": repr(e)[:200]}

    if world == 1 and prepare_us is not None:
        # the per-hyper-parameter-step work outside the timed step (VERDICT r5 #6b), spread over the MVMs one solve takes: the CG
        # iteration count of the mean-cache solve measured above (extras), else one MVM (the worst case: nothing to spread over)
        its = None
        if "extras" in result:
            its = int(result["extras"]["solve_Khat_inv_y"]["cg_iterations"])
        result["prepare_us"] = round(prepare_us, 1)
        result["prepare_what"] = ("Z = X Peff (project_kernel) + rpgp_prepare (range pass, midpoints, row / column tables) + the status "
                                  "read-back; once per hyper-parameter step, outside `value`'s timed region")
        result["ms_per_step_incl_prepare_amortised_over"] = {
            "cg_iterations": its if its is not None else 1,
            "source": ("iterations of the K^-1 y solve in extras.solve_Khat_inv_y (tolerance 0.01)" if its is not None
                       else "no solve in this run (--no-extras): charged to ONE MVM"),
            "ms_per_step": round(ms_per_step + prepare_us * 1e-3 / max(its or 1, 1), 4),
            "mvm_per_s": round(1e3 / (ms_per_step + prepare_us * 1e-3 / max(its or 1, 1)), 3)}

    if rank == 0 and world == 1 and args.cpu_budget > 0:
        from oracle import cpu_path
        torch.set_num_threads(os.cpu_count() or 1)
        Zc, Vc = Z.cpu(), V.cpu()
        cb = cpu_path.time_mvm_sample(Zc, Vc, outputscale, noise, budget_s=args.cpu_budget)
        gpu_rows = res[:cb["sample_rows"]].cpu()
        rel = float((gpu_rows - cb["out_sample"]).norm() / cb["out_sample"].norm())
        result["cpu_baseline"] = {
            "value": round(cb["mvm_per_s"], 5), "unit": "MVM/s", "cores": cb["threads"], "kind": "port",
            "sample": "first %d of %d output rows of the same MVM (row-chunked J x B x N build + matmul, fp32 torch-CPU, "
                      "%.1f s), extrapolated to N rows" % (cb["sample_rows"], N, cb["sample_s"]),
            "gpu_vs_cpu_rel_err": rel,
        }
        # a competent CPU loop beside the port of the reference's op sequence: oracle/cmvm.c (plain C + OpenMP, FLOAT64, one
        # projection at a time over column tiles, libm exp) on a bounded block of output rows of the same product
        from oracle import cmvm
        import numpy as np
        Zd, Vd = Zc.double().numpy(), Vc.double().numpy()
        nthr = os.cpu_count() or 1
        cmvm.mvm(Zd[:256], Zd, Vd, outputscale / J)                    # (build / load / thread pool)
        tb = time.perf_counter()
        rows_done, blocks = 0, []
        while rows_done < N and time.perf_counter() - tb < max(args.cpu_budget * 0.5, 2.0):
            e = min(rows_done + 2048, N)
            blocks.append(cmvm.mvm(Zd[rows_done:e], Zd, Vd, outputscale / J) + noise * Vd[rows_done:e])
            rows_done = e
        dtb = time.perf_counter() - tb
        ref_rows = np.concatenate(blocks, axis=0)
        relb = float(np.linalg.norm(res[:rows_done].double().cpu().numpy() - ref_rows) / np.linalg.norm(ref_rows))
        result["cpu_baseline_best"] = {
            "value": round(1.0 / (dtb * N / rows_done), 5), "unit": "MVM/s", "cores": nthr, "kind": "port",
            "sample": "first %d of %d output rows of the same MVM by oracle/cmvm.c (C + OpenMP, float64, K never stored; %.1f s), "
                      "extrapolated to N rows" % (rows_done, N, dtb),
            "gpu_vs_cpu_rel_err": relb,
            "note": "the honest CPU comparison: a plain fused loop, not GPyTorch's J x B x N temporaries (cpu_baseline)"}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        if reducer is not None:
            reducer.close()
        dist.destroy_process_group()


def _step_times(device):
    """Median time of one optimiser step (`-mll(model(X), y)`, backward, Adam, `loss.item()` — fitting/optimizing.py:65-76) of the
    additive RP model at the C2 / C3 / C5 shapes on synthetic data, 3 rounds of 20 steps each after 8 untimed ones."""
    import numpy as np
    from rpgp_amd import fused_mll, settings
    from rpgp_amd.training import create_exact_gp, make_optimizer
    from rpgp_amd.models import ExactMarginalLogLikelihood
    out = {"what": "ms per step: preconditioned mBCG on [10 probes | y - c] at cg_tolerance 0.05, SLQ log-det, fused derivative, "
                   "Adam; median of 3 x 20 steps (tools/r4_step_time.py is the longer form)"}
    shapes = {"C2 N=7372 d=8 J=20": (7372, 8, 20, False, False), "C3 N=14939 d=18 J=20 spread": (14939, 18, 20, True, False),
              "C5 N=391386 d=3 J=3 spread ski": (391386, 3, 3, True, True)}
    for name, (N, d, J, sp, ski) in shapes.items():
        g = torch.Generator().manual_seed(0)
        X = torch.randn(N, d, generator=g)
        y = torch.sin(X).sum(1) + 0.05 * torch.randn(N, generator=g)
        X, y = X.to(device), ((y - y.mean()) / y.std()).to(device)
        torch.manual_seed(0)
        np.random.seed(0)
        model, lik = create_exact_gp(X, y, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False,
                                     prescale=True, space_proj=sp, ski=ski,
                                     ski_options={"grid_size": 1024, "num_dims": 1} if ski else None)
        model = model.to(device)
        mll = ExactMarginalLogLikelihood(lik, model)
        opt = make_optimizer(torch.optim.Adam, [p for p in model.parameters() if p.requires_grad], 0.0)

        def run(n):
            for _ in range(n):
                opt.zero_grad()
                loss = mll.negative_and_backward(model(X), y)   # (the closure of training.train_to_convergence)
                opt.step()
                fused_mll.loss_value(loss)                      # (... and its per-step read of the loss)
        ts = []
        with settings.cg_tolerance(0.05), settings.max_cg_iterations(10000):
            model.train()
            run(8)
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(20)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 20 * 1e3)
        out[name] = round(sorted(ts)[1], 3)
        del model, lik, mll, opt, X, y
    return out


if __name__ == "__main__":
    main()
