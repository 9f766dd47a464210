import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "orb-slam2-dualcam_amd")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """Import the hyphenated package directory as module `orb_slam2_dualcam_amd`."""
    name = "orb_slam2_dualcam_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def synth():
    load_pkg()
    import importlib
    return importlib.import_module("orb_slam2_dualcam_amd.synth")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    O.lib()
    return O


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture
def opts(pkg):
    """opts(name, value): set a library option (include/dcs_abi.h "options") for the rest of the test; the previous values come back afterwards.
    Options replaced the environment switches of rounds 1-4: they are read per call (matcher, solver, tracking) or when an extractor handle
    is created, so one process can exercise every path -- no child processes."""
    old = {}

    def set_(name, value):
        if name not in old:
            old[name] = pkg.abi.get_option(name)
        pkg.abi.set_option(name, int(value))
    yield set_
    for k, v in old.items():
        pkg.abi.set_option(k, v)
