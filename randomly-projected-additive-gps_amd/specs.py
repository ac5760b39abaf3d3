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
