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


@pytest.fixture(scope="session")
def record():
    """record(name, **numbers): appends measured error levels to gpurun_out/metrics.jsonl (kept as evidence)."""
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)

    def _rec(name, **kw):
        with open(os.path.join(out, "metrics.jsonl"), "a") as f:
            f.write(json.dumps({"test": name, **kw}) + "\n")
    return _rec
