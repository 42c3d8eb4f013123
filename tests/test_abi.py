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


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
@pytest.mark.parametrize("n_tris", [40, 300000])
def test_host_builders_reject_non_finite_vertices(rt, scenes, bad, n_tris):
    """A NaN or infinite coordinate puts every triangle in bucket 0 and makes every SAH cost NaN; the
    reference indexes outside its bucket array and panics (src/bvh/build.rs:118-120).  Both builders
    refuse such a mesh with RT_ERR_INVALID, single- and multi-threaded path alike."""
    sc = scenes.load("test-mesh", n_tris=n_tris, seed=2)
    pos = sc.positions.copy()
    pos[len(pos) // 2, 1] = bad
    for build in (rt.build_bvh_host, rt.build_bvh_search_host):
        with pytest.raises(rt.RtError) as e:
            build(pos, sc.indices)
        assert e.value.code == 1 and "NaN or infinite" in str(e.value)


def test_host_builders_on_denormal_centroid_range(rt):
    """Centroids a denormal apart: 8*0.99/extent would overflow to infinity and the bucket index
    would leave the bucket array, but the squared extent underflows to zero first and the
    zero-extent rule (src/bvh/build.rs:93-101) makes one leaf.  With that denormal extent on the
    first axis and an ordinary one (0.11) on the second, the second is binned
    normally while the search builder, which also tries the tiny axis, clamps its buckets."""
    tiny = np.float32(1e-44)
    pos = []
    for i in range(12):
        x = np.float32(i) * tiny
        pos += [[x, 0, 0], [x, 1, 0], [x, 0, 1]]
    pos = np.array(pos, np.float32)
    idx = np.arange(36, dtype=np.uint32).reshape(12, 3)
    for build in (rt.build_bvh_host, rt.build_bvh_search_host):
        nodes, order, info, st = build(pos, idx)
        assert len(nodes) == 0 and info["root_count"] == 12 and st["max_objs"] == 12
    pos2 = pos.copy()
    pos2[:, 1] += np.repeat(np.arange(12, dtype=np.float32) * np.float32(0.01), 3)
    for build in (rt.build_bvh_host, rt.build_bvh_search_host):
        nodes, order, info, st = build(pos2, idx)
        assert sorted(order.tolist()) == list(range(12)) and st["num_objs"] == 12 and len(nodes) >= 1


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
