import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "diag: diagnostic builds kept for the record (rejected kernels); skipped unless MC_RUN_DIAG=1")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    return entry.load_oracle()


@pytest.fixture(scope="session")
def B():
    """ctypes bindings of the product library; built on demand (no GPU needed to load it)."""
    pkg = entry.load_package()
    if not os.path.exists(pkg.bindings.LIB_PATH):
        entry.build()
    return pkg.bindings


@pytest.fixture(scope="session")
def ctx(B):
    """One mc_context on device 0 for the whole GPU session (fails loudly without a GPU)."""
    c = B.Context(0)
    yield c
    c.close()


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REFERENCE = "/root/reference"
