"""The C-ABI library builds, loads and exports every symbol include/rpgp.h declares (no compute without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rpgp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rpgp_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_bound_and_exported():
    from rpgp_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), "librpgp.so does not export %s" % name
        assert name in _lib.SIGNATURES, "ctypes binding missing for %s" % name
    assert sorted(_lib.SIGNATURES) == declared
    assert lib.rpgp_version() == 4
    assert b"invalid argument" in lib.rpgp_error_string(10001)


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rpgp_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.mvm_sym(torch.zeros(4, 2), torch.zeros(4, 1), 1.0)
    from rpgp_amd import _lib
    assert _lib.load().rpgp_init() != 0           # RPGP_ENODEVICE (or a HIP error): never silently succeeds


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "randomly-projected-additive-gps_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
