import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

PKG = "structured-light-calculation_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(name):
    return importlib.import_module(PKG + "." + name)


@pytest.fixture(scope="session")
def synth():
    return pkg("synth")


_built = False


def _ensure_built():
    """The library and the C++ test programs are build products (git-ignored).  `make` runs once per session -- a no-op when
    everything is newer than its sources, a rebuild when a source changed or the tree is a fresh checkout (hipcc and gcc are
    on the image) -- so a stale libslx.so is never what gets tested."""
    global _built
    if _built:
        return
    import __graft_entry__
    __graft_entry__.build()
    _built = True


@pytest.fixture(scope="session")
def api():
    _ensure_built()
    return pkg("api")


@pytest.fixture(scope="session")
def shard():
    return pkg("shard")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O   # oracle/oracle.py -- test infrastructure
    O.build()
    return O


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
