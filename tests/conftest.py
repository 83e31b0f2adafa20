import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native libraries are built once per session (idempotent make)."""
    import jello_amd
    from oracle import oracle_engine
    paths = list(jello_amd.lib_paths().values()) + [os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in paths):
        import __graft_entry__ as g
        g.build()
    return True


@pytest.fixture(scope="session")
def engine(built):
    import jello_amd
    e = jello_amd.Engine(0)  # raises if there is no GPU / no HIP extension: GPU tests must not pass on a fallback
    yield e
    e.close()
