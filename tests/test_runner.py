"""Harness logic: the literal answers of the reference's TestExperimentHelpers (test.py:494-529) and the CLI
end to end on CPU (BASELINE config 1: RBF_model_spec.json on a yacht-shaped stand-in, Cholesky regime)."""
import json
import os

import numpy as np
import pandas as pd
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spec(name):
    from rpgp_amd import specs
    return specs.get(name)


def test_normalize_by_train_literal():
    from rpgp_amd.runner import _normalize_by_train
    df = pd.DataFrame({"index": [0, 1, 2], "0": [1, 2, 2], "target": [0, 0, 1]})
    new_train, new_test = _normalize_by_train(df.iloc[:2, :], df.iloc[2:, :])
    mean = 0.5
    std = np.std([-0.5, 0.5], ddof=1)
    assert new_train["0"].iloc[0] == (0 - mean) / std
    assert new_test["0"].iloc[0] == (0 + mean) / std
    assert new_test["target"].iloc[0] == 1
    assert new_train["target"].iloc[0] == 0


def test_determine_folds_literal():
    from rpgp_amd.runner import _determine_folds
    df = pd.DataFrame({"index": [0, 1, 2, 3], "0": [1, 2, 2, 4], "target": [0, 0, 1, 1]})
    assert _determine_folds(1 / 3, df) == [0, 2, 3, 4]
    # fold sizes quoted in SURVEY.md §8: yacht 308 -> 277/31, kin8nm 8192 -> 7372/820 (fold 0 of 10)
    for n, ntest in [(308, 31), (8192, 820), (16599, 1660)]:
        fs = _determine_folds(0.1, range(n))
        assert fs[1] - fs[0] == ntest and fs[-1] == n and len(fs) == 11


def test_access_fold_literal():
    from rpgp_amd.runner import _access_fold
    df = pd.DataFrame({"index": [0, 1, 2, 3], "0": [1, 2, 2, 4], "target": [0, 0, 1, 1]})
    train, test = _access_fold(df, [0, 2, 3, 4], 0)
    assert test["index"].values.tolist() == [0, 1] and train["index"].values.tolist() == [2, 3]
    train, test = _access_fold(df, [0, 2, 3, 4], 1)
    assert test["index"].values.tolist() == [2] and train["index"].values.tolist() == [0, 1, 3]


def test_dataset_groups_and_loader():
    from rpgp_amd import runner
    assert len(runner.get_datasets()) == 36
    assert runner.resolve_datasets(["small-med"])[-1] == "wine"
    assert runner.resolve_datasets(["med"])[-1] == "pol"
    assert runner.resolve_datasets(["3"]) == ["challenger", "fertility", "concreteslump"]
    assert runner.resolve_datasets(["yacht", "energy"]) == ["yacht", "energy"]
    df = runner.load_dataset("synthetic:yacht")
    assert df.shape == (308, 8) and list(df.columns) == ["index"] + [str(i) for i in range(6)] + ["target"]
    assert abs(df["target"].mean()) < 1e-12 and abs(df["target"].std() - 1) < 1e-12


def test_mat_loader_layout(tmp_path, monkeypatch):
    """mat['data'] with the target in the last column (gp_experiment_runner.py:21-24)."""
    from scipy.io import savemat
    from rpgp_amd import runner
    data = np.random.default_rng(0).standard_normal((50, 4))
    os.makedirs(tmp_path / "uci" / "toy")
    savemat(str(tmp_path / "uci" / "toy" / "toy.mat"), {"data": data})
    monkeypatch.setenv("RPGP_DATA_BASE_PATH", str(tmp_path))
    df = runner.load_dataset("toy")
    assert df.shape == (50, 5)
    np.testing.assert_allclose(df["0"].values, data[:, 0])
    z = (data[:, 3] - data[:, 3].mean()) / data[:, 3].std(ddof=1)
    np.testing.assert_allclose(df["target"].values, z)


def test_cli_full_rbf_on_cpu(tmp_path):
    """BASELINE config 1 plumbing: spec JSON -> settings -> folds -> fit -> metrics -> CSV, on CPU."""
    from rpgp_amd import runner
    spec = _spec("RBF_model_spec.json")
    spec["train_kwargs"]["max_iter"] = 5
    spec["train_kwargs"]["init_iters"] = 2
    sp = tmp_path / "spec.json"
    json.dump(spec, open(sp, "w"))
    out = tmp_path / "res.csv"
    torch.manual_seed(0)
    df = runner.main(["-m", str(sp), "-d", "synthetic:tiny", "-o", str(out), "--no_cv", "--fold", "1"])
    assert os.path.exists(out)
    row = df.iloc[0]
    for col in ["fold", "repeat", "n", "d", "mse", "rmse", "train_time", "trained_epochs", "prior_train_nmll",
                "train_mse", "train_nll", "test_nll", "test_pred_frac_in_cr", "training_warnings", "testing_warning",
                "state_dict_file", "dataset", "options", "cg_tol", "eval_cg_tol", "use_chol", "max_cg_iterations",
                "use_toeplitz", "fast_pred_var", "checkpoint_kernel", "skip_log_det_forward", "memory_efficient"]:
        assert col in df.columns, col
    assert row["fold"] == 1 and row["n"] == 120 and row["d"] == 4 and np.isfinite(row["rmse"])
    assert row["cg_tol"] == 0.05 and row["max_cg_iterations"] == 10000


def test_cli_additive_spec_through_oracle_backend(tmp_path, oracle_backend):
    from rpgp_amd import runner
    spec = _spec("additive_spread_prescale_J20.json")
    spec["train_kwargs"]["max_iter"] = 3
    spec["train_kwargs"]["init_iters"] = 1
    spec["model_kwargs"]["J"] = 6
    sp = tmp_path / "spec.json"
    json.dump(spec, open(sp, "w"))
    out = tmp_path / "res.csv"
    df = runner.main(["-m", str(sp), "-d", "synthetic:tiny", "-o", str(out), "--no_cv", "--skip_random_restart"])
    assert np.isfinite(df.iloc[0]["rmse"]) and "error" not in df.columns
    assert oracle_backend.calls["dense"] > 0


@pytest.mark.parametrize("name", ["GAM_spec.json", "additive_rp_J20_K1.json", "additive_rp_prescale_J1_K20.json",
                                  "additive_rp_postscale_J20.json", "additive_rp_prescale_J20_matern.json",
                                  "additive_rp_J20_K1_ski.json", "additive_deterministic_spec_unweighted_ski.json",
                                  "additive_spread_prescale_J20_ski.json", "additive_rp_postscale_J20_ski.json"])
def test_cli_family_specs_through_oracle_backend(tmp_path, oracle_backend, name):
    """The other family members' spec files run end to end through the same CLI."""
    from rpgp_amd import runner
    spec = _spec(name)
    spec["train_kwargs"]["max_iter"] = 2
    spec["train_kwargs"]["init_iters"] = 1
    if spec["model_kwargs"].get("J") == 20:
        spec["model_kwargs"]["J"] = 5
    if spec["model_kwargs"].get("k") == 20:
        spec["model_kwargs"]["k"] = 2
    if spec["model_kwargs"].get("ski"):
        spec["model_kwargs"]["ski_options"]["grid_size"] = 64        # keep the dense float64 test double small
    spec["train_kwargs"]["verbose"] = False
    sp = tmp_path / "spec.json"
    json.dump(spec, open(sp, "w"))
    df = runner.main(["-m", str(sp), "-d", "synthetic:tiny", "-o", str(tmp_path / "res.csv"), "--no_cv",
                      "--skip_random_restart"])
    assert np.isfinite(df.iloc[0]["rmse"]) and "error" not in df.columns


def test_run_experiment_records_errors_and_retries():
    from rpgp_amd import runner
    calls = {"n": 0}

    def flaky(trainX, trainY, testX, testY, **kw):
        calls["n"] += 1
        if calls["n"] < 3:
            raise RuntimeError("boom")
        return {"trained_epochs": 1}, torch.zeros_like(testY)

    df = runner.run_experiment(flaky, {}, "synthetic:tiny", split=0.5, cv=False, error_repeats=5,
                               print_to_console=False)
    assert calls["n"] == 3 and len(df) == 3
    assert df["error"].notna().sum() == 2 and "boom" in df["error"].iloc[0]
    assert np.isfinite(df["rmse"].iloc[2])


def test_specs_match_reference_keys():
    """The five in-scope spec files carry the reference's keys/values."""
    for name, J in [("additive_rp_prescale_J20.json", 20), ("additive_spread_prescale_J20.json", 20),
                    ("additive_spread_prescale_Jd.json", "d"), ("additive_spread_prescale_Jd_ski.json", "d")]:
        spec = _spec(name)
        assert spec["kind"] == "additive_rp" and spec["model_kwargs"]["J"] == J
        assert spec["model_kwargs"]["prescale"] is True and spec["model_kwargs"]["noise_prior"] is True
        assert spec["train_kwargs"] == {"verbose": False, "optimizer": "adam", "max_iter": 1000, "lr": 0.1,
                                        "patience": 20, "smooth": True}
    assert _spec("RBF_model_spec.json")["kind"] == "full"


def test_spec_table_files_and_names(tmp_path, oracle_backend):
    """`-m` takes the path of a spec file (a user's copy of the reference's model_specs/*.json) or a built-in name;
    `specs.write_all` materialises the table as files that round-trip."""
    from rpgp_amd import runner, specs
    paths = specs.write_all(str(tmp_path / "model_specs"))
    assert len(paths) == len(specs.names()) >= 16
    for pth in paths:
        assert json.load(open(pth)) == specs.get(os.path.basename(pth))
    spec = specs.get("additive_rp_prescale_J20")
    spec["train_kwargs"].update(max_iter=2, init_iters=1)
    spec["model_kwargs"]["J"] = 4
    f = tmp_path / "mine.json"
    json.dump(spec, open(f, "w"))
    df = runner.main(["-m", str(f), "-d", "synthetic:tiny", "-o", str(tmp_path / "a.csv"), "--no_cv", "--skip_random_restart"])
    assert np.isfinite(df.iloc[0]["rmse"])
    with pytest.raises(FileNotFoundError):
        runner.main(["-m", "no_such_spec", "-d", "synthetic:tiny", "-o", str(tmp_path / "b.csv"), "--no_cv"])
    with pytest.raises(KeyError):
        specs.get("nope")
