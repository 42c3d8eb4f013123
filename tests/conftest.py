import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle binding (test infrastructure)."""
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def rt():
    """The product binding (librtcore.so through its C ABI)."""
    return importlib.import_module("rust-raytracer_amd")


@pytest.fixture(scope="session")
def scenes():
    return importlib.import_module("rust-raytracer_amd.scenes")


@pytest.fixture(scope="session")
def ctx(rt):
    """Context in its exact configuration: the reference's binary tree in the
    reference's visit order (traversal forms 0 and 1) and the cooperative
    straggler path off (a ray finished by its whole wave keeps its hit distance
    but not its visit counters).  These tests compare distances, ids and
    counters with the oracle exactly.  tests/test_gpu_coop.py and
    tests/test_gpu_wide.py cover the other forms, ctx_default the defaults."""
    c = rt.Context(0)
    c.set_option("coop_steps", 0)
    c.set_option("traversal_depth0", 0)
    c.set_option("traversal_deep", 1)
    c.set_option("traversal_stream", 1)
    yield c
    c.close()


@pytest.fixture(scope="session")
def ctx_default(rt):
    """Context with the library defaults."""
    c = rt.Context(0)
    yield c
    c.close()
