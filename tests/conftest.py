import importlib.util
import os
import subprocess
import sys

import pytest

# the TEST TWIN of the library reads its OVM_* test hooks once per thread; tests switch them between proofs of one process
# (csrc/tuning.h; the product library reads none of them)
os.environ.setdefault("OVM_TUNING_LIVE", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


HOOKS_LIB = os.path.join(ROOT, "stark-backend_amd", "build_hooks", "libopenvm_hip_hooks.so")


def load_pkg(hooks=False):
    """Imports stark-backend_amd/ (hyphenated directory) as module `stark_backend_amd`. hooks=True: a SECOND instance of
    the binding, `stark_backend_amd_hooks`, over the TEST TWIN of the library (tools/build_hooks.sh, -DOVM_TEST_HOOKS=1):
    the only build that reads the OVM_* test hooks of csrc/tuning.h. Both libraries live side by side in one process."""
    name = "stark_backend_amd_hooks" if hooks else "stark_backend_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "stark-backend_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    saved = os.environ.get("OVM_HIP_LIB")
    try:
        if hooks:
            if not os.path.exists(HOOKS_LIB):
                subprocess.run([os.path.join(ROOT, "tools", "build_hooks.sh")], check=True)
            os.environ["OVM_HIP_LIB"] = HOOKS_LIB
        spec.loader.exec_module(mod)
    finally:
        if hooks:
            if saved is None:
                os.environ.pop("OVM_HIP_LIB", None)
            else:
                os.environ["OVM_HIP_LIB"] = saved
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def pkg_hooks():
    """The binding over the test twin: tests that switch an OVM_* hook take this (and dev_hooks) instead of pkg / dev."""
    return load_pkg(hooks=True)


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


@pytest.fixture(scope="session")
def dev_hooks(pkg_hooks):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return pkg_hooks.Device(0)
