"""The C-ABI library loads and exports every symbol include/rtcore.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest


def declared_symbols(header_path):
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(rt):
    names = declared_symbols(rt.HEADER_PATH)
    assert len(names) >= 18
    lib = ctypes.CDLL(rt.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in rtcore.h but not exported: %s" % missing
    # and the binding binds exactly the declared set
    assert sorted(rt.EXPORTS) == names


def test_struct_sizes(rt):
    assert rt.RAY_DTYPE.itemsize == 48      # src/ray.rs:6-12, repr(C, align(16))
    assert rt.NODE_DTYPE.itemsize == 64
    assert rt.TRI_DTYPE.itemsize == 48
    assert ctypes.sizeof(rt.TraceStats) == 56
    assert ctypes.sizeof(rt.RenderParams) == 48
    assert ctypes.sizeof(rt.CameraC) == 40


def test_version_and_error_string(rt):
    assert rt.version().startswith("rtcore-gfx950")


def test_host_builder_rejects_bad_input(rt):
    pos = np.zeros((3, 3), np.float32)
    idx = np.array([[0, 1, 7]], np.uint32)
    with pytest.raises(rt.RtError) as e:
        rt.build_bvh_host(pos, idx)
    assert e.value.code == 1 and "out of range" in str(e.value)


def test_no_oracle_in_product():
    """The product tree must not reference the oracle."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "rust-raytracer_amd")
    bad = []
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hpp", ".h", ".hip", "Makefile")):
                s = open(os.path.join(dp, fn), errors="replace").read()
                if "rt_oracle" in s or "librt_oracle" in s or "import oracle" in s:
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad
