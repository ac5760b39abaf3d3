"""Experiment harness + CLI: counterpart of gp_experiment_runner.py (load_dataset :17-33, fold logic :65-102,
run_experiment :105-219, CLI :222-381).  Same flags, same model-spec JSON schema, same result columns
(SURVEY.md §5, Appendix D).  UCI `.mat` files are not shipped with the reference and cannot be downloaded here, so
`synthetic:<name>` datasets with the standard (N, d) of the BASELINE configs are provided as stand-ins.

    python -m rpgp_amd.runner -m additive_rp_prescale_J20 -d synthetic:kin8nm -o out.csv --device cuda:0
    (-m takes the path of a reference model_specs/*.json file, or the name of a built-in spec: rpgp_amd.specs)
"""
import argparse
import datetime
import json
import os
import time
import traceback

import numpy as np
import pandas as pd
import torch

from . import settings
from . import training as training_routines
from .training import mean_squared_error

# (N, d) of the datasets the BASELINE configs name (SURVEY.md §8 header) for the synthetic stand-ins
SYNTHETIC_SHAPES = {"yacht": (308, 6), "kin8nm": (8192, 8), "elevators": (16599, 18), "synthetic50k": (50000, 20),
                    "3droad": (434874, 3), "tiny": (120, 4)}


def data_base_path():
    return os.environ.get("RPGP_DATA_BASE_PATH", os.path.join(os.path.expanduser("~"), "data"))


def make_synthetic(name, seed=0):
    """X ~ N(0,1), y = sum_d sin(x_d) + 0.01 eps (the `additive` target of synthetic_test_script.py:63-65,97)."""
    if name not in SYNTHETIC_SHAPES:
        raise ValueError("unknown synthetic dataset '%s' (known: %s)" % (name, sorted(SYNTHETIC_SHAPES)))
    n, d = SYNTHETIC_SHAPES[name]
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(n, d, generator=g, dtype=torch.float64)
    y = torch.sin(X).sum(dim=1) + 0.01 * torch.randn(n, generator=g, dtype=torch.float64)
    return np.concatenate([X.numpy(), y.numpy()[:, None]], axis=1)


def _frame_from_array(data):
    n, d = data.shape
    df = pd.DataFrame(data, columns=list(range(d - 1)) + ["target"])
    df.columns = [str(c) for c in df.columns]
    df = df.reset_index()
    df["target"] = df["target"] - df["target"].mean()
    df["target"] = df["target"] / (df["target"].std())
    df = df.dropna(axis=1, how="all")
    return df


def load_dataset(name):
    """`<data_base_path>/uci/<name>/<name>.mat` with mat['data'] (last column = target), target z-scored over the
    whole set (gp_experiment_runner.py:17-33); `synthetic:<name>` builds a stand-in of the same shape."""
    if name.startswith("synthetic:"):
        return _frame_from_array(make_synthetic(name.split(":", 1)[1]))
    from scipy.io import loadmat
    mat = loadmat(os.path.join(data_base_path(), "uci", name, "{}.mat".format(name)))
    return _frame_from_array(mat["data"])


def get_small_datasets():
    return ["challenger", "fertility", "concreteslump", "autos", "servo", "breastcancer", "machine", "yacht",
            "autompg", "housing", "forest", "stock", "pendulum", "energy"]


def get_medium_datasets():
    return ["concrete", "solar", "airfoil", "wine", "gas", "skillcraft", "sml", "parkinsons", "pumadyn32nm"]


def get_big_datasets():
    return ["pol", "elevators", "bike", "kin40k", "protein", "tamielectric", "keggdirected", "slice",
            "keggundirected", "3droad", "song", "buzz", "houseelectric"]


def get_datasets():
    return get_small_datasets() + get_medium_datasets() + get_big_datasets()


def format_timedelta(delta):
    d = delta.days
    h = delta.seconds // 3600
    m = (delta.seconds - h * 3600) // 60
    s = delta.seconds - h * 3600 - m * 60
    return "{}d {}h {}m {}s".format(d, h, m, s)


def _determine_folds(split, dataset):
    """Fold boundaries (gp_experiment_runner.py:65-76): round(1/split) folds of floor(n*split) rows, the first
    `remaining` folds one row longer.  Pinned by test.py:510-516 ([0, 2, 3, 4] for split=1/3, n=4)."""
    n = len(dataset)
    n_per_fold = int(np.floor(n * split))
    n_folds = int(round(1 / split))
    remaining = n - n_per_fold * n_folds
    fold_starts = [0]
    for i in range(n_folds):
        fold_starts.append(fold_starts[i] + n_per_fold + (1 if i < remaining else 0))
    return fold_starts


def _access_fold(dataset, fold_starts, fold):
    """Test = rows of fold `fold`; train = everything else (gp_experiment_runner.py:79-84; test.py:518-529)."""
    a, b = fold_starts[fold], fold_starts[fold + 1]
    test = dataset.iloc[a:b]
    train = pd.concat([dataset.iloc[0:a], dataset.iloc[b:]])
    return train, test


def _normalize_by_train(train, test):
    """Centre and scale features + target by TRAIN statistics (pandas std, ddof=1; columns with sigma=0 are only
    centred) — gp_experiment_runner.py:87-102; test.py:495-508."""
    train = train.copy()
    test = test.copy()
    cols = [c for c in train.columns if c.lower() != "index"]
    feats = [c for c in cols if c != "target"] + ["target"]
    train[feats] = train[feats].astype(float)
    test[feats] = test[feats].astype(float)
    mu = train[feats].mean()
    train.loc[:, feats] = train[feats] - mu
    test.loc[:, feats] = test[feats] - mu
    for f in feats:
        sigma = train[f].std()
        if sigma > 0:
            train.loc[:, f] = train[f] / sigma
            test.loc[:, f] = test[f] / sigma
    return train, test


class _Progress:
    """Console line per finished fit with the remaining-time estimate (the reference prints the same fields)."""

    def __init__(self, total_fits, enabled):
        self.total, self.enabled, self.t0 = total_fits, enabled, time.time()

    def fit_done(self, fold, repeat, done, row):
        if not self.enabled:
            return
        eta = datetime.timedelta(seconds=(time.time() - self.t0) / done * (self.total - done))
        shown = {k: v for k, v in row.items() if k != "test_pred_z_score"}
        print("{}, fold={}, rep={}, eta={} \n{}".format(datetime.datetime.now(), fold, repeat, format_timedelta(eta), shown))


def _fold_tensors(train, test, features):
    """float32 contiguous tensors of one fold (gp_experiment_runner.py:164-167)."""
    def t(frame, cols):
        return torch.tensor(frame[cols].values, dtype=torch.float).contiguous()
    return t(train, features), t(train, "target"), t(test, features), t(test, "target")


def _fit_row(training_routine, training_options, tensors, addl_metrics, base):
    """One fit -> one result row: the routine's metrics next to mse / rmse / wall time and any additional metric."""
    row = dict(base)
    tic = time.perf_counter()
    model_metrics, ypred = training_routine(*tensors, **training_options)[:2]
    row["train_time"] = time.perf_counter() - tic
    testY = tensors[3]
    row["mse"] = mean_squared_error(ypred, testY)
    row["rmse"] = np.sqrt(row["mse"])
    # column order of the reference's rows: mse, rmse, train_time, then the routine's metrics, then the additional ones
    row = {**{k: row[k] for k in base}, "mse": row["mse"], "rmse": row["rmse"], "train_time": row["train_time"]}
    row.update(model_metrics)
    row.update({name: fn(ypred, testY) for name, fn in addl_metrics.items()})
    return row


def run_experiment(training_routine, training_options, dataset, split, cv, addl_metrics={}, repeats=1,
                   error_repeats=10, normalize_using_train=True, chosen_fold=0, print_to_console=True):
    """Run a training routine over the folds of a dataset (counterpart of gp_experiment_runner.py:105-219; same columns,
    retry rule and console output).

    Folds are contiguous blocks (`_determine_folds`); without `cv` only `chosen_fold` is run.  A fold is fitted `repeats`
    times on the same tensors.  An exception anywhere inside a fold's fits becomes a row of its own (traceback under
    `error`; `d` two less than the feature count and NaN scores, as the reference records it) and a fold without a finished
    fit starts over, at most `error_repeats` times.  Returns the rows as a DataFrame."""
    if isinstance(dataset, str):
        dataset = load_dataset(dataset)
    features = [c for c in dataset.columns if c != "target" and c.lower() != "index"]
    fold_starts = _determine_folds(split, dataset)
    n_folds = len(fold_starts) - 1
    folds = range(n_folds) if cv else [f for f in range(n_folds) if f == chosen_fold]
    progress = _Progress(n_folds * repeats, print_to_console)
    rows = []
    for fold in folds:
        train, test = _access_fold(dataset, fold_starts, fold)
        if normalize_using_train:
            train, test = _normalize_by_train(train, test)
        failures, fitted = 0, False
        while failures < error_repeats and not fitted:
            try:
                tensors = _fold_tensors(train, test, features)
                for repeat in range(repeats):
                    row = _fit_row(training_routine, training_options, tensors, addl_metrics,
                                   {"fold": fold, "repeat": repeat, "n": len(dataset), "d": len(features)})
                    rows.append(row)
                    fitted = True                       # (as in the reference: one finished fit ends the fold's retries,
                    progress.fit_done(fold, repeat, fold * repeats + repeat + 1, row)   # even if a later repeat raises)
            except Exception:                           # ANY exception: record it; an unfitted fold runs again
                failures += 1
                row = dict(error=traceback.format_exc(), fold=fold, n=len(dataset), d=len(features) - 2, mse=np.nan,
                           rmse=np.nan)
                print(row)
                traceback.print_exc()
                rows.append(row)
                print("errors: ", failures)
    results = pd.DataFrame(rows)
    if print_to_console and "rmse" in results:
        print("Mean RMSE = {}".format(results["rmse"].mean()))
    return results


def build_parser():
    p = argparse.ArgumentParser(description="Utility to run a suite of experiments with a GP model on UCI regression "
                                            "datasets (MI355X-native counterpart of gp_experiment_runner.py).")
    p.add_argument("-m", "--model_spec", type=str, required=True,
                   help="path to model specification json file (or the name of a built-in spec, see rpgp_amd.specs)")
    p.add_argument("-d", "--datasets", type=str, nargs="+", required=True,
                   help="UCI dataset name(s), a predefined set (all|small|small-med|med|large|<int>) or synthetic:<name>")
    p.add_argument("-o", "--output", type=str, required=True, help="path to output csv file")
    p.add_argument("-s", "--split", type=float, default=0.1, help="fraction of data in test set")
    p.add_argument("-r", "--repeats", type=int, default=1, help="number of times to repeat each fold")
    p.add_argument("--no_cv", action="store_false", dest="cv")
    p.add_argument("--cg_tol", type=float, default=0.05)
    p.add_argument("--eval_cg_tol", type=float, default=0.01)
    p.add_argument("--fast_pred", dest="fast_pred", action="store_true")
    p.add_argument("--use_chol", action="store_true")
    p.add_argument("--no_toeplitz", dest="use_toeplitz", action="store_false")
    p.add_argument("--memory_efficient", dest="memory_efficient", action="store_true")
    p.add_argument("--device", type=str, default="cpu", help="device string; a comma-separated list means multi-GPU")
    p.add_argument("--skip_posterior_variances", action="store_true")
    p.add_argument("--ablation", action="store_true")
    p.add_argument("--J", type=int, nargs="+", help="Js to use in ablation to overwrite the ablation Js")
    p.add_argument("--k", type=int, nargs="+", help="If used, do ablation on k with given k values instead of J.")
    p.add_argument("--fold", type=int, default=0)
    p.add_argument("--error_repeats", type=int, default=10)
    p.add_argument("--max_cg_iterations", type=int, default=10_000)
    p.add_argument("--skip_evaluate_on_train", action="store_true")
    p.add_argument("--skip_random_restart", action="store_true")
    p.add_argument("--skip_log_det_forward", action="store_true", help="Apply skip log det forward option.")
    p.add_argument("--checkpoint_kernel", type=int, default=0, help="accepted for parity; the fused kernel never stores K")
    p.add_argument("--record_pred_unc", action="store_true", help="Record predictive uncertainty metrics.")
    p.add_argument("--double", action="store_true", help="double precision (only for kinds outside the fused fp32 path)")
    p.add_argument("--cache_kernel", dest="cache_kernel", action="store_const", const=True, default="auto",
                   help="(rpgp) always materialise K once per hyper-parameter step when it fits in HBM (default: auto)")
    p.add_argument("--no_cache_kernel", dest="cache_kernel", action="store_const", const=False,
                   help="(rpgp) never materialise K: every CG iteration runs the fused recompute-in-kernel MVM")
    return p


def resolve_datasets(names):
    """Dataset groups of gp_experiment_runner.py:268-289."""
    names = list(names)
    first = names[0]
    try:
        first = int(first)
    except Exception:
        pass
    if len(names) == 1:
        if first == "all":
            return get_datasets()
        if first == "small":
            return get_small_datasets()
        if first == "small-med":
            return get_datasets()[:18]
        if first == "med":
            return get_datasets()[18:24]
        if first == "large":
            return get_datasets()[24:]
        if isinstance(first, int):
            return get_datasets()[:first]
    return names


def launch_ranks(argv, devices, entry=None):
    """The reference's one-command multi-device form, `--device cuda:0,cuda:1,cuda:2` (gp_experiment_runner.py:263 →
    training_routines.py:407-408; run_scripts/additive_spread_prescale_Jd.sh:6): start one rank per listed device through a
    fresh `torch.distributed.run` child and return its exit status.  This parent has made no HIP call and never will (the
    ranks bind their devices themselves, `init_distributed`); the flags travel through the environment because
    torch.distributed.run's own parser abbreviation-matches script flags such as `--no_cv` / `-m` before the script sees
    them.  `entry` replaces the rank program (default `-m rpgp_amd.runner`); tests/ use it to start ranks that install
    their CPU test double first."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["RPGP_RUNNER_ARGV"] = json.dumps(list(argv))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))         # where the `rpgp_amd` import shim lives
    env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(len(devices)),
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + list(entry or ["-m", "rpgp_amd.runner"])
    return subprocess.call(cmd, env=env)


def init_distributed(devices, backend=None):
    """One process per GPU — started by `launch_ranks` for the reference's `--device cuda:0,cuda:1,...` form, or by the user
    as `python -m torch.distributed.run --nproc-per-node N -m rpgp_amd.runner ... --device cuda` (gp_experiment_runner.py:263
    + training_routines.py:407-408): bind this rank to its GPU and join the RCCL process group BEFORE any GPU call, and map
    `--device` to this rank's device (the LOCAL_RANK-th listed device when the list names one per rank, else
    `cuda:LOCAL_RANK`).  Returns (rank, world); (0, 1) when not launched by a distributed launcher."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = devices[0].startswith("cuda")
    if on_gpu:
        mine = torch.device(devices[local_rank]) if len(devices) == world else torch.device("cuda", local_rank)
        index = local_rank if mine.index is None else mine.index
    if not dist.is_initialized():
        if on_gpu and len(devices) == world and len(set(devices)) < world:
            # the same device listed more than once (`--device cuda:0,cuda:0`): a rehearsal of the multi-rank path on a
            # one-GPU box.  RCCL refuses two ranks per device, so the group is bootstrapped over gloo and the data-path
            # all-reduce runs through rpgp_comm's IPC-mapped buffers (distributed.Reducer backend "ipc").
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.environ["RPGP_COMM"] = "ipc"
            torch.cuda.set_device(index)
            dist.init_process_group(backend or "gloo")
        elif on_gpu:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(index)
            dist.init_process_group(backend or "nccl", device_id=torch.device("cuda", index))
        else:
            dist.init_process_group(backend or "gloo")
    if on_gpu:
        devices[:] = ["cuda:%d" % index]
    else:
        devices[:] = devices[:1]
    if rank != 0:
        import sys
        sys.stdout = open(os.devnull, "w")     # one voice on stdout (every rank computes identical metrics)
    return rank, world


def main(argv=None, rank_entry=None):
    import sys
    if argv is None and len(sys.argv) == 1 and "RPGP_RUNNER_ARGV" in os.environ and "WORLD_SIZE" in os.environ:
        argv = json.loads(os.environ["RPGP_RUNNER_ARGV"])          # a rank started by launch_ranks
    args = build_parser().parse_args(argv)
    if len(args.device.split(",")) > 1 and "WORLD_SIZE" not in os.environ:
        # `--device cuda:0,cuda:1,...` in ONE command: one rank per listed device, started before anything touches a GPU
        status = launch_ranks(sys.argv[1:] if argv is None else argv, args.device.split(","), entry=rank_entry)
        if status != 0:
            raise SystemExit(status)
        return pd.read_csv(args.output, index_col=0) if os.path.exists(args.output) else None
    print("Parser arguments", args)
    if os.path.exists(args.model_spec):
        with open(args.model_spec, "r") as f:
            options = json.load(f)
    else:
        # a bare spec name (or a path whose file does not exist) resolves against the built-in table of rpgp_amd.specs
        from . import specs
        try:
            options = specs.get(args.model_spec)
        except KeyError:
            raise FileNotFoundError("no such model specification file, and not a built-in spec name: %s" % args.model_spec)
    print("Loaded options", options)
    devices = args.device.split(",")
    rank, world = init_distributed(devices)
    print("Using device(s) {}".format(devices))
    datasets = resolve_datasets(args.datasets)

    if options["kind"] in ("ppr_gp", "cgp"):
        raise NotImplementedError("model kind '%s' is outside the MI355X hot path (SURVEY.md §8)" % options["kind"])
    ma = options["kind"] == "model_average"
    if ma:
        # the overloaded "kind" key (gp_experiment_runner.py:299-304): the base model's options with the varying lists inside
        ma_args = options.pop("varying_params")
        options = options["base_model_kwargs"]
        options["model_kwargs"]["varying_params"] = ma_args
        if world > 1:
            raise ValueError("model averaging runs on one device")
    else:
        options["skip_random_restart"] = args.skip_random_restart
    options["devices"] = devices
    options["skip_posterior_variances"] = args.skip_posterior_variances
    options["evaluate_on_train"] = not args.skip_evaluate_on_train
    options["record_pred_unc"] = args.record_pred_unc
    if args.double:
        options["double"] = args.double
    if options["record_pred_unc"] and options["skip_posterior_variances"]:
        raise ValueError("Can't record predictive uncertainty while skipping posterior variances.")

    df = pd.DataFrame()
    use_fast = not args.use_chol
    for dataset in datasets:
        print("Starting dataset {}".format(dataset))
        with settings.cg_tolerance(args.cg_tol), settings.eval_cg_tolerance(args.eval_cg_tol), \
                settings.fast_computations(use_fast, use_fast, use_fast), settings.fast_pred_var(args.fast_pred), \
                settings.use_toeplitz(args.use_toeplitz), settings.max_cg_iterations(args.max_cg_iterations), \
                settings.beta_features.checkpoint_kernel(args.checkpoint_kernel), \
                settings.skip_logdet_forward(args.skip_log_det_forward), \
                settings.memory_efficient(args.memory_efficient), settings.cache_kernel(args.cache_kernel):
            if args.ablation:
                if args.k is not None:
                    abl_vars = args.k
                elif args.J is not None:
                    abl_vars = args.J
                else:
                    abl_vars = [1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377]
            else:
                abl_vars = [-1]
            for abl_val in abl_vars:
                if args.ablation:
                    options["model_kwargs"]["J" if args.k is None else "k"] = abl_val
                routine = training_routines.train_exact_gp_model_average if ma else training_routines.train_exact_gp
                results = run_experiment(routine, options, dataset, split=args.split,
                                         cv=args.cv, repeats=args.repeats, normalize_using_train=True,
                                         chosen_fold=args.fold, error_repeats=args.error_repeats)
                if args.ablation:
                    results["J" if args.k is None else "k"] = abl_val
                results["dataset"] = dataset
                results["options"] = json.dumps(options)
                results["cg_tol"] = args.cg_tol
                results["eval_cg_tol"] = args.eval_cg_tol
                results["use_chol"] = args.use_chol
                results["max_cg_iterations"] = args.max_cg_iterations
                results["use_toeplitz"] = args.use_toeplitz
                results["fast_pred_var"] = args.fast_pred
                results["checkpoint_kernel"] = args.checkpoint_kernel
                results["skip_log_det_forward"] = args.skip_log_det_forward
                results["memory_efficient"] = args.memory_efficient
                df = pd.concat([df, results])
                if rank == 0:                     # every rank holds the same metrics; one writer
                    df.to_csv(args.output)
    return df


if __name__ == "__main__":
    main()
