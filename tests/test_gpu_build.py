"""The device-side builder of the fast forms' tree (rt_bvh_build_search_device: binned SAH breadth
first on the GPU) against the host builder it restates (rt_bvh_build_search_host): the same tree --
every node's child boxes equal as f32 values, the same leaf ranges holding the same triangles, the
same BvhStats -- up to the numbering of the nodes and the order of the triangles inside a leaf.
And a scene created over it renders what a scene over the host-built tree renders."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LEAF = 0x80000000


def assert_same_tree(a, b, what):
    (na, oa, ia, sa), (nb, ob, ib, sb) = a, b
    assert sa == sb, (what, sa, sb)
    assert len(na) == len(nb), what
    assert sorted(oa.tolist()) == sorted(ob.tolist()) == list(range(len(oa)))
    # walk both trees together from their roots
    stack = [(ia["root_ref"], ia["root_count"], ib["root_ref"], ib["root_count"])]
    visited = 0
    while stack:
        ra, ca, rb, cb = stack.pop()
        assert bool(ra & LEAF) == bool(rb & LEAF), what
        if ra & LEAF:
            fa, fb = ra & 0x7FFFFFFF, rb & 0x7FFFFFFF
            assert (fa, ca) == (fb, cb), (what, fa, ca, fb, cb)
            assert sorted(oa[fa:fa + ca].tolist()) == sorted(ob[fb:fb + cb].tolist()), what
            continue
        x, y = na[ra], nb[rb]
        visited += 1
        for f in ("bx", "by", "bz"):
            assert np.array_equal(x[f], y[f]), (what, f, x[f], y[f])   # equal as values (+0 == -0)
        stack.append((int(x["left"]), int(x["left_count"]), int(y["left"]), int(y["left_count"])))
        stack.append((int(x["right"]), int(x["right_count"]), int(y["right"]), int(y["right_count"])))
    assert visited == len(na), what


@pytest.mark.parametrize("name,kw", [("cornell-class", {}), ("test-mesh", {"n_tris": 5000, "seed": 21}),
                                     ("test-mesh", {"n_tris": 60000, "seed": 4}), ("sponza-class", {})])
@pytest.mark.parametrize("buckets,max_leaf", [(32, 2), (32, 3), (8, 1), (64, 8)])
def test_device_tree_is_the_host_tree(rt, ctx, scenes, name, kw, buckets, max_leaf):
    sc = scenes.load(name, **kw)
    host = rt.build_bvh_search_host(sc.positions, sc.indices, buckets, max_leaf)
    dev = rt.build_bvh_search_device(ctx, sc.positions, sc.indices, buckets, max_leaf)
    assert_same_tree(host, dev, (name, buckets, max_leaf))


def test_device_tree_degenerate_inputs(rt, ctx):
    """Coincident centroids (one leaf beyond max_leaf, as on the host), a mesh smaller than a leaf
    (refused: the caller builds that on the host), duplicated triangles."""
    from util import deep_chain_mesh
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    pos = np.concatenate([tri] * 20)
    idx = np.arange(60, dtype=np.uint32).reshape(20, 3)
    host = rt.build_bvh_search_host(pos, idx, 32, 2)
    dev = rt.build_bvh_search_device(ctx, pos, idx, 32, 2)
    assert host[3] == dev[3] and dev[2]["root_ref"] & LEAF and dev[2]["root_count"] == 20
    with pytest.raises(rt.RtError):
        rt.build_bvh_search_device(ctx, tri, idx[:1], 32, 2)
    p, i, _ = deep_chain_mesh(n=150)      # a chain: a deep, thin tree (many levels with a single open node)
    assert_same_tree(rt.build_bvh_search_host(p, i, 32, 2), rt.build_bvh_search_device(ctx, p, i, 32, 2), "chain")
    # half the triangles twice: pairs with equal centroids end up together in leaves the count rule cannot split
    sc_pos = np.random.default_rng(5).uniform(-1, 1, size=(3000, 3)).astype(np.float32)
    sc_idx = np.arange(3000, dtype=np.uint32).reshape(1000, 3)
    sc_idx = np.concatenate([sc_idx, sc_idx[:500]])
    assert_same_tree(rt.build_bvh_search_host(sc_pos, sc_idx, 32, 1), rt.build_bvh_search_device(ctx, sc_pos, sc_idx, 32, 1),
                     "duplicates")


def test_scene_bounds_and_validation_on_large_meshes(rt, ctx, scenes):
    """The scans of rt_scene_create that run on several threads for large meshes (vertex box, index and
    finiteness checks): the box is numpy's, a bad index or vertex deep inside the arrays is found."""
    sc = scenes.load("sponza-class")
    pos = np.tile(sc.positions, (6, 1))                      # 1.4 M vertices: past the single-thread limit
    pos[len(sc.positions):] += np.float32(0.5)
    idx = np.concatenate([sc.indices + k * len(sc.positions) for k in range(6)]).astype(np.uint32)
    nrm = np.tile(sc.geom_normals, (6, 1))
    assert len(pos) > (1 << 20) // 3 and len(idx) * 3 > (1 << 20)
    Sg = rt.Scene(ctx, pos, idx, nrm)
    lo, hi = Sg.bounds()
    assert np.array_equal(lo, pos.min(axis=0)) and np.array_equal(hi, pos.max(axis=0))
    Sg.close()
    bad = idx.copy()
    bad[len(bad) * 3 // 4, 1] = len(pos)
    with pytest.raises(rt.RtError, match="out of range"):
        rt.Scene(ctx, pos, bad, nrm)
    nan = pos.copy()
    nan[len(nan) * 2 // 3, 2] = np.nan
    with pytest.raises(rt.RtError, match="NaN"):
        rt.Scene(ctx, nan, idx, nrm)


def test_scene_over_device_tree(O, rt, ctx_default, scenes):
    """rt_scene_create with the device build forced: the fast forms walk a tree made on the GPU and find
    what they find on the host-built one (same distances; ids may differ inside a leaf on exact ties)."""
    from util import random_rays
    c = ctx_default
    sc = scenes.load("sponza-class")
    rays = np.concatenate([O.generate_primary(sc.camera, 160, 90, 1), random_rays(O.make_rays, 20000, 3, lo=-8.0, hi=8.0)])
    out = {}
    try:
        for mode in (0, 2):
            c.set_option("fast_tree_device_build", mode)
            Sg = rt.Scene(c, sc.positions, sc.indices, sc.geom_normals)
            assert Sg.fast_tree_on_device() == mode   # 2: tree and traversal items made on the device
            c.set_option("traversal_stream", 2)
            ids, t, _ = Sg.intersect_stream(rays, stats=False)
            occ, _ = Sg.occluded_stream(rays, stats=False)
            img, st = Sg.render(sc.camera, sc.light, 96, 54, 2, max_bounces=2)
            out[mode] = (ids, t, occ, img, st["total_rays"])
            Sg.close()
    finally:
        c.set_option("fast_tree_device_build", 1)
        c.set_option("traversal_stream", 1)
    assert np.array_equal(out[0][1], out[2][1]) and np.array_equal(out[0][2], out[2][2])
    assert (out[0][0] != out[2][0]).mean() < 1e-3
    assert out[0][4] == out[2][4] and np.allclose(out[0][3], out[2][3], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("name,kw", [("sponza-class", {}), ("test-mesh", {"n_tris": 20000, "seed": 3}),
                                     ("cornell-class", {}), ("dragon-class", {})])
@pytest.mark.parametrize("align,max_leaf", [(4, 2), (0, 3), (3, 1)])
def test_device_made_items_are_the_hosts(rt, ctx_default, scenes, name, kw, align, max_leaf):
    """The traversal items of the fast forms -- 4-wide collapse, quantization, placement, packing --
    made on the device (fast_tree_device_build 2) against the host pipeline over the same device-built
    binary tree (3): the unit array is equal bit for bit, for every placement mode and leaf size."""
    c = ctx_default
    sc = scenes.load(name, **kw)
    items, made = {}, {}
    try:
        c.set_option("fast_tree_align", align)
        c.set_option("fast_tree_max_leaf", max_leaf)
        for mode in (2, 3):
            c.set_option("fast_tree_device_build", mode)
            Sg = rt.Scene(c, sc.positions, sc.indices, sc.geom_normals)
            # a tree with a leaf above max_leaf (triangles the binning cannot separate) goes through the host's
            # leaf refinement, whichever mode is asked for
            made[mode] = Sg.fast_tree_on_device()
            assert made[mode] in ((1, 2) if mode == 2 else (1,))
            items[mode] = Sg.wide_items()
            Sg.close()
    finally:
        c.set_option("fast_tree_device_build", 1)
        c.set_option("fast_tree_align", 4)
        c.set_option("fast_tree_max_leaf", 2)
    assert items[2].shape == items[3].shape and items[2].shape[0] > 3
    assert np.array_equal(items[2], items[3])
    if max_leaf >= 2 and name != "cornell-class":
        assert made[2] == 2
