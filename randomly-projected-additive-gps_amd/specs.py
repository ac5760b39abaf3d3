"""The model specifications this build serves, as a table (the reference ships them as model_specs/*.json; a user's own
copies of those files work unchanged with `rpgp_amd.runner -m <file>`).  `get(name)` returns the dict the runner would
read from `<name>.json`; `write_all(dir)` materialises the files (done by `__graft_entry__.build()` and by the tests'
conftest, so nothing here is a copy of the reference's files)."""
import copy
import json
import os

_ADAM = {"verbose": False, "optimizer": "adam", "max_iter": 1000, "lr": 0.1, "patience": 20, "smooth": True}
_SKI = {"grid_size": 1024, "num_dims": 1}


def _spec(kind, train=None, **model_kwargs):
    return {"kind": kind, "model_kwargs": model_kwargs, "train_kwargs": dict(_ADAM if train is None else train)}


def _rp(J, prescale, **kw):
    return _spec("additive_rp", J=J, noise_prior=True, kernel_type=kw.pop("kernel_type", "RBF"), learn_proj=False,
                 prescale=prescale, **kw)


SPECS = {
    # the hot path (BASELINE.json configs 2-5) and the CPU plumbing config 1
    "RBF_model_spec": _spec("full", noise_prior=True, kernel_type="RBF", ard=False),
    "additive_rp_prescale_J20": _rp(20, True),
    "additive_spread_prescale_J20": _rp(20, True, space_proj=True, batch_kernel=True),
    "additive_spread_prescale_Jd": _rp("d", True, space_proj=True, mem_efficient=True, batch_kernel=False),
    "additive_spread_prescale_Jd_ski": _rp("d", True, space_proj=True, batch_kernel=True, ski=True, ski_options=dict(_SKI)),
    # other members of the family behind the same operator
    "additive_rp_postscale_J20": _rp(20, False),
    "additive_rp_prescale_J1_K20": _spec("additive_rp", J=1, k=20, noise_prior=True, kernel_type="RBF", learn_proj=False,
                                         prescale=True, batch_kernel=False),
    "additive_rp_prescale_J20_matern": _rp(20, True, kernel_type="Matern"),
    "GAM_spec": _spec("strictly_additive", noise_prior=True, kernel_type="RBF", weighted=False, memory_efficient=True),
    "additive_rp_J20_K1": _spec("rp_poly", k=1, J=20, noise_prior=True, kernel_type="RBF", learn_proj=False, weighted=True),
    "ARD_model_spec": _spec("full", noise_prior=True, kernel_type="RBF", ard=True),
    "Inverse_MQ_ARD_model_spec": _spec("full", noise_prior=True, kernel_type="InverseMQ", ard=True),
    # grid-interpolation (`ski: true`) variants
    "additive_spread_prescale_J20_ski": _rp(20, True, space_proj=True, batch_kernel=False, ski=True, ski_options=dict(_SKI)),
    "additive_rp_postscale_J20_ski": _spec("additive_rp", train=dict(_ADAM, verbose=True, max_iter=100, check_conv=False),
                                           J=20, noise_prior=True, kernel_type="RBF", learn_proj=False, space_proj=False,
                                           prescale=False, ski=True, ski_options={"grid_size": 512, "num_dims": 1}),
    "additive_rp_J20_K1_ski": _spec("rp_poly", k=1, J=20, noise_prior=True, kernel_type="RBF", learn_proj=False,
                                    weighted=True, ski=True, ski_options=dict(_SKI)),
    "additive_deterministic_spec_unweighted_ski": _spec("strictly_additive", noise_prior=True, kernel_type="RBF",
                                                        weighted=False, ski=True, ski_options=dict(_SKI)),
    # the rest of the reference's exact-GP specifications that the built kinds serve
    "additive_rp_prescale_J1": _rp(1, True),
    "additive_rp_prescale_on_sphere": _rp(20, True, space_proj=False, proj_dist="sphere"),
    "additive_rp_J1_K1": _spec("rp_poly", k=1, J=1, noise_prior=True, kernel_type="RBF", learn_proj=False, weighted=True),
    "additive_rp_Jd_K1": _spec("rp_poly", k=1, J="d", noise_prior=True, kernel_type="RBF", learn_proj=False, weighted=True),
    "additive_rp_Jd_spread": _spec("rp_poly", k=1, J="d", noise_prior=True, kernel_type="RBF", learn_proj=False,
                                   weighted=True, space_proj=True),
    "additive_spread_projections": _spec("rp_poly", k=1, J=20, noise_prior=True, kernel_type="RBF", learn_proj=False,
                                         weighted=True, space_proj=True),
    "additive_spread_projections_RO": _spec("rp_poly", train=dict(_ADAM, checkpoint=True, random_restarts=10),
                                            k=1, J=20, noise_prior=True, kernel_type="RBF", learn_proj=False, weighted=True,
                                            space_proj=True, init_lengthscale_range=[0.4, 0.8], init_mixin_range=[0.4, 0.8]),
    "best_of_single_proj": _spec("rp_poly", train=dict(_ADAM, max_iter=0, random_restarts=20, init_iters=1000,
                                                        rr_check_conv=True),
                                 k=1, J=1, noise_prior=True, kernel_type="RBF", learn_proj=False, weighted=True),
    "additive_deterministic_spec_unweighted": _spec("strictly_additive", noise_prior=True, kernel_type="RBF", weighted=False),
    "ARD_RO": _spec("full", train=dict(_ADAM, checkpoint=True, random_restarts=10), noise_prior=True, kernel_type="RBF",
                    ard=True, init_lengthscale_range=[0.4, 0.8]),
    # multiplicative groups of different sizes (operators.MixedGroupOperator)
    "polynomial_rp": _spec("general_rp_poly", degrees=[1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6], noise_prior=True,
                           kernel_type="RBF", learn_proj=False, weighted=True),
    "polynomial_rp_smaller": _spec("general_rp_poly", degrees=[1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3], noise_prior=True,
                                   kernel_type="RBF", learn_proj=False, weighted=True),
    # Bayesian model average over J (training.train_exact_gp_model_average)
    "ma_dpa_gp_ard": {"kind": "model_average", "varying_params": {"J": [1, 2, 3, 5, 8, 13, 21, 34, 55, 89]},
                      "base_model_kwargs": _spec("additive_rp", train=dict(_ADAM, max_iter=300, patience=15), J=20,
                                                 noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True)},
}


def names():
    return sorted(SPECS)


def get(name):
    """Spec dict by name (with or without the .json suffix)."""
    key = os.path.basename(name)
    key = key[:-5] if key.endswith(".json") else key
    if key not in SPECS:
        raise KeyError("unknown model spec '%s' (known: %s)" % (name, ", ".join(names())))
    return copy.deepcopy(SPECS[key])


def write_all(directory):
    """Write every spec as <directory>/<name>.json; returns the list of paths."""
    os.makedirs(directory, exist_ok=True)
    out = []
    for name in names():
        path = os.path.join(directory, name + ".json")
        with open(path, "w") as f:
            json.dump(SPECS[name], f, indent=1, sort_keys=True)
            f.write("\n")
        out.append(path)
    return out
