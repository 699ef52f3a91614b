import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def lib():
    """libcdx.so, built on demand (cross-compiles on CPU)."""
    import cdx
    if not os.path.exists(cdx._abi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return cdx._abi.lib()
