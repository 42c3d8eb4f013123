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


# Library defaults of every option a test may change, and what the exact configuration sets on top.
OPTION_DEFAULTS = {"coop_steps": 1024, "coop_steps_binary": 512, "min_active": 40, "min_active_depth0": 20, "traversal_depth0": 3, "traversal_deep": 2,
                   "traversal_stream": 1, "fuse_primary": 1, "sort_extend_from": 0xFFFFFFFF,
                   "sort_connect_from": 1, "sort_bits": 21, "sort_octant": 0, "timing": 1, "overlap_connect": 0,
                   "traversal_deep_connect": -1, "fast_tree_max_leaf": 2, "fast_tree_align": 4, "fast_tree_device_build": 1, "workspace_limit_mb": 0}
EXACT_OPTIONS = {"coop_steps": 0, "coop_steps_binary": 0, "traversal_depth0": 0, "traversal_deep": 1, "traversal_stream": 1}
_LIVE = []  # (context, options it must have at the start of every test)


@pytest.fixture(autouse=True)
def _restore_context_options():
    """The session contexts are shared; whatever a test sets on them (and however it ends) is put
    back before the next test runs."""
    yield
    for c, opts in _LIVE:
        if c.h:
            for k, v in opts.items():
                c.set_option(k, v)


@pytest.fixture(scope="session")
def ctx(rt):
    """Context in its exact configuration: the reference's binary tree in the
    reference's visit order (traversal forms 0 and 1) and the cooperative
    straggler path off (a ray finished by its whole wave keeps its hit distance
    but not its visit counters).  These tests compare distances, ids and
    counters with the oracle exactly.  tests/test_gpu_coop.py and
    tests/test_gpu_wide.py cover the other forms, ctx_default the defaults."""
    c = rt.Context(0)
    opts = dict(OPTION_DEFAULTS, **EXACT_OPTIONS)
    for k, v in opts.items():
        c.set_option(k, v)
    _LIVE.append((c, opts))
    yield c
    _LIVE[:] = [e for e in _LIVE if e[0] is not c]
    c.close()


@pytest.fixture(scope="session")
def ctx_default(rt):
    """Context with the library defaults."""
    c = rt.Context(0)
    _LIVE.append((c, dict(OPTION_DEFAULTS)))
    yield c
    _LIVE[:] = [e for e in _LIVE if e[0] is not c]
    c.close()
