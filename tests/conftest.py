import importlib.util
import os
import subprocess
import sys

import pytest

# the library reads its OVM_* test hooks once per thread; tests switch them between proofs of one process (csrc/tuning.h)
os.environ.setdefault("OVM_TUNING_LIVE", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """Imports stark-backend_amd/ (hyphenated directory) as module `stark_backend_amd`."""
    if "stark_backend_amd" in sys.modules:
        return sys.modules["stark_backend_amd"]
    path = os.path.join(ROOT, "stark-backend_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(
        "stark_backend_amd", path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["stark_backend_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def orc():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def dev(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return pkg.Device(0)
