"""Host-side invariants of the 4-wide collapse (rt_bvh_wide_host): it only
regroups the boxes of the binary tree that the reference builds
(src/bvh/build.rs:45-207), so every child box must be a child box of some
binary node, bit for bit, and the leaves must be the binary tree's leaves, each
exactly once.  Inside a wide node the children are ordered largest box first
(the order the any-hit walk visits them in)."""
import numpy as np
import pytest

LEAF = 0x80000000


def pack(ref, count):
    if not ref & LEAF:
        return None
    assert 1 <= count <= 8
    return LEAF | ((ref & 0x7FFFFFFF) << 3) | (count - 1)


def binary_leaves(nodes, root):
    """Leaf references (device packing) left to right, and per binary node its two child boxes."""
    out = []
    stack = [(root, 0)]
    while stack:
        ref, cnt = stack.pop()
        if ref & LEAF:
            out.append(pack(ref, cnt))
            continue
        n = nodes[ref]
        stack.append((int(n["right"]), int(n["right_count"])))
        stack.append((int(n["left"]), int(n["left_count"])))
    return out


def check_quantized(wide, qwide, node_unit, slot_unit, num_units):
    """The device form of the wide tree.  Every decoded box (in double precision) contains the exact
    child box, is not looser than two grid steps plus the builder's margin per side, and unused slots
    decode to near > far on every axis; grid steps carry two mantissa bits; the children block of every
    node holds its children back to back in slot order, and nodes and triangles tile the unit array
    (three units each) without gaps or overlaps."""
    assert len(wide) == len(qwide) == len(node_unit)
    items = sorted(node_unit.tolist() + slot_unit.tolist())
    assert items == list(range(0, num_units, 3)), "nodes and triangles must tile the unit array"
    assert node_unit[0] == 0
    for w, q in zip(wide, qwide):
        origin = q["origin"].astype(np.float64)
        code = np.array([(int(q["exps"]) >> (10 * k)) & 0x3FF for k in range(3)])
        e = code >> 2
        assert ((e >= 1) & (e <= 254)).all() and int(q["exps"]) >> 30 == 0
        scale = np.ldexp(1.0 + 0.25 * (code & 3), e - 127)
        as_f32 = (code.astype(np.uint32) << 21).view(np.float32)          # how the kernel decodes it
        assert np.array_equal(as_f32.astype(np.float64), scale)
        used = [i for i in range(4) if np.isfinite(w["lo"][0][i])]
        nlo = w["lo"][:, used].astype(np.float64).min(axis=1)
        nhi = w["hi"][:, used].astype(np.float64).max(axis=1)
        big = np.maximum(np.maximum(np.abs(nlo), np.abs(nhi)), nhi - nlo)
        # the smallest representable step that reaches the far side: four fifths of it must not
        assert ((origin + 255 * scale * 0.8 < nhi + np.ldexp(big, -17)) | (code == 4)).all()
        base, meta = int(q["base4"]) >> 2, int(q["meta"])
        assert int(q["base4"]) & 3 == 0
        expect = base
        for i in range(4):
            lo_b = np.array([(int(q["lo"][k]) >> (8 * i)) & 0xFF for k in range(3)], dtype=np.float64)
            hi_b = np.array([(int(q["hi"][k]) >> (8 * i)) & 0xFF for k in range(3)], dtype=np.float64)
            byte = (meta >> (8 * i)) & 0xFF
            if not np.isfinite(w["lo"][0][i]):
                assert (lo_b == 255).all() and (hi_b == 0).all() and byte == meta & 0xFF
                continue
            dlo, dhi = origin + lo_b * scale, origin + hi_b * scale
            elo, ehi = w["lo"][:, i].astype(np.float64), w["hi"][:, i].astype(np.float64)
            assert (dlo < elo).all() and (dhi > ehi).all()           # strictly outside: the margin
            slack = 2.0 * scale + np.ldexp(big, -17)               # two grid steps plus twice the margin
            assert (elo - dlo <= slack).all() and (dhi - ehi <= slack).all()
            ref = int(w["child"][i])
            unit, tris = base + (byte >> 2), byte & 3
            assert unit == expect                                      # back to back, in slot order
            if ref & LEAF:
                first, count = (ref >> 3) & ((1 << 27) - 1), (ref & 7) + 1
                assert tris == count <= 3
                assert slot_unit[first:first + count].tolist() == [unit + 3 * t for t in range(count)]
                expect += 3 * count
            else:
                assert tris == 0 and node_unit[ref] == unit
                expect += 3


@pytest.mark.parametrize("name,kw", [("cornell-class", {}), ("test-mesh", {"n_tris": 3000, "seed": 4}),
                                     ("test-mesh", {"n_tris": 9, "seed": 1}), ("test-mesh", {"n_tris": 5, "seed": 2})])
def test_wide_collapse_invariants(rt, scenes, name, kw):
    sc = scenes.load(name, **kw)
    nodes, order, info, stats = rt.build_bvh_host(sc.positions, sc.indices)
    wide, wroot, wdepth = rt.wide_bvh_host(nodes, info["root_ref"], info["root_count"])
    # the device form needs leaves of at most 3 triangles: the reference's tree as it is must be refused
    if stats["max_objs"] > 3:
        with pytest.raises(rt.RtError):
            rt.wide_bvh_host(nodes, info["root_ref"], info["root_count"], len(sc.indices))
    if info["root_ref"] & LEAF:
        assert len(wide) == 0 and wroot == pack(info["root_ref"], info["root_count"])
        return
    assert wroot == 0 and 1 <= len(wide) <= len(nodes)
    assert wdepth <= stats["max_depth"]

    # box of every binary child, keyed by the device reference of that child
    boxes = {}
    for i, n in enumerate(nodes):
        for side, lane in (("left", 0), ("right", 1)):
            ref = int(n[side])
            key = pack(ref, int(n[side + "_count"])) if ref & LEAF else ref
            lo = (n["bx"][lane], n["by"][lane], n["bz"][lane])
            hi = (n["bx"][lane + 2], n["by"][lane + 2], n["bz"][lane + 2])
            boxes[key] = (np.array(lo, np.float32), np.array(hi, np.float32))

    leaves = []
    seen_inner = set()

    def walk(w):
        node = wide[w]
        used = [i for i in range(4) if np.isfinite(node["lo"][0][i])]
        assert used == list(range(len(used))) and len(used) >= 2
        ext = (node["hi"][:, used] - node["lo"][:, used]).astype(np.float32)
        area = np.float32(2) * (ext[0] * ext[1] + ext[0] * ext[2] + ext[1] * ext[2])
        assert all(area[i] >= area[i + 1] for i in range(len(used) - 1))
        for i in range(4):
            ref = int(node["child"][i])
            if i not in used:
                assert np.all(node["lo"][:, i] == np.inf) and np.all(node["hi"][:, i] == -np.inf)
                assert ref == int(node["child"][0]) and ref & LEAF
                continue
            if ref & LEAF:
                key = ref
                leaves.append(ref)
            else:
                key = int(wide[ref]["bin"])  # a wide child stands for the binary node it came from
            lo, hi = boxes[key]
            assert np.array_equal(node["lo"][:, i], lo) and np.array_equal(node["hi"][:, i], hi)
            if not ref & LEAF:
                assert ref not in seen_inner and ref > w  # pre-order numbering
                seen_inner.add(ref)
                walk(ref)
        if len(used) < 4:
            assert all(int(node["child"][i]) & LEAF for i in used)

    walk(0)
    assert int(wide[0]["bin"]) == info["root_ref"]
    assert len(seen_inner) == len(wide) - 1
    assert sorted(leaves) == sorted(binary_leaves(nodes, info["root_ref"])) and len(set(leaves)) == len(leaves)


@pytest.mark.parametrize("name,kw", [("cornell-class", {}), ("test-mesh", {"n_tris": 3000, "seed": 4}),
                                     ("test-mesh", {"n_tris": 9, "seed": 1}), ("test-mesh", {"n_tris": 5, "seed": 2}),
                                     ("coincident", {})])
@pytest.mark.parametrize("builder", ["refined", "search"])
def test_wide_device_form(rt, scenes, name, kw, builder):
    """The 48-byte nodes and the unit layout rt_scene_create uploads, from both trees the fast forms
    can be built over: the reference's tree with its leaves refined to 3 triangles, and the search
    builder's.  'coincident' adds 20 copies of one triangle (one centroid: a leaf of 20 in either
    builder), which the refinement must cut down by slot."""
    if name == "coincident":
        sc = scenes.load("test-mesh", n_tris=300, seed=6)
        base = len(sc.positions)
        positions = np.vstack([sc.positions, [[0.1, 0.1, 5.0], [0.4, 0.1, 5.0], [0.1, 0.4, 5.0]]]).astype(np.float32)
        indices = np.vstack([sc.indices, np.tile(np.array([[base, base + 1, base + 2]], np.uint32), (20, 1))]).astype(np.uint32)
    else:
        sc = scenes.load(name, **kw)
        positions, indices = sc.positions, sc.indices
    nt = len(indices)
    if builder == "refined":
        nodes, order, info, stats = rt.build_bvh_host(positions, indices)
    else:
        nodes, order, info, stats = rt.build_bvh_search_host(positions, indices, 32, 3)
    xn, xorder, xroot, xcount, extra = rt.refine_bvh_host(positions, indices, nodes, info["root_ref"], info["root_count"],
                                                          order, 3)
    assert sorted(xorder.tolist()) == list(range(nt))
    wide, qwide, node_unit, slot_unit, num_units, wroot, wdepth = rt.wide_bvh_host(xn, xroot, xcount, nt)
    if xroot & LEAF:
        assert len(wide) == 0 and num_units == 3 * nt and slot_unit.tolist() == list(range(0, 3 * nt, 3))
        return
    assert num_units == 3 * (len(wide) + nt)
    check_quantized(wide, qwide, node_unit, slot_unit, num_units)
    # every leaf of the wide tree holds at most 3 triangles and every triangle is in exactly one
    seen = np.zeros(nt, dtype=np.int64)
    for w in wide:
        for i in range(4):
            ref = int(w["child"][i])
            if np.isfinite(w["lo"][0][i]) and ref & LEAF:
                assert not ref & 0x40000000
                first, count = (ref >> 3) & ((1 << 27) - 1), (ref & 7) + 1
                seen[xorder[first:first + count]] += 1
    assert (seen == 1).all()


def test_wide_host_rejects_bad_tree(rt):
    nodes = np.zeros(1, dtype=rt.NODE_DTYPE)
    nodes[0]["left"] = 5  # inner reference beyond the array
    nodes[0]["right"] = LEAF
    nodes[0]["right_count"] = 1
    with pytest.raises(rt.RtError):
        rt.wide_bvh_host(nodes, 0, 0)
    with pytest.raises(rt.RtError):
        rt.wide_bvh_host(nodes, 0, 0, 1)


@pytest.mark.parametrize("max_tris", [1, 2, 3, 4])
@pytest.mark.parametrize("name,kw", [("cornell-class", {}), ("test-mesh", {"n_tris": 3000, "seed": 4})])
def test_refined_leaves(rt, scenes, name, kw, max_tris):
    """rt_bvh_refine_host: original nodes keep index and boxes, every reference leaf keeps its slot
    range and its set of triangles, new leaves hold at most max_tris triangles (triangles whose
    centroids coincide are cut by slot), and every new child box is exactly the bound of the
    triangles below it."""
    sc = scenes.load(name, **kw)
    nodes, order, info, stats = rt.build_bvh_host(sc.positions, sc.indices)
    xn, xorder, xroot, xcount, extra = rt.refine_bvh_host(sc.positions, sc.indices, nodes, info["root_ref"],
                                                          info["root_count"], order, max_tris)
    assert len(xn) >= len(nodes) and xroot == info["root_ref"] and extra <= 7
    assert sorted(xorder.tolist()) == sorted(order.tolist())
    for k in ("bx", "by", "bz"):
        assert np.array_equal(xn[k][:len(nodes)], nodes[k])
    P, I = sc.positions, sc.indices

    def bound(first, count):
        v = P[I[xorder[first:first + count]].reshape(-1)]
        return v.min(axis=0), v.max(axis=0)

    def subtree(ref, count):
        """(first slot, triangle count, lo, hi) of the triangles below a refined child."""
        if ref & LEAF:
            first = ref & 0x7FFFFFFF
            lo, hi = bound(first, count)
            assert count <= max_tris
            return first, count, lo, hi
        assert ref >= len(nodes)   # new nodes are appended
        n = xn[ref]
        fl, cl, lol, hil = subtree(int(n["left"]), int(n["left_count"]))
        fr, cr, lor, hir = subtree(int(n["right"]), int(n["right_count"]))
        assert fr == fl + cl       # contiguous slots, left to right
        for lane, lo, hi in ((0, lol, hil), (1, lor, hir)):
            got_lo = np.array([n["bx"][lane], n["by"][lane], n["bz"][lane]])
            got_hi = np.array([n["bx"][lane + 2], n["by"][lane + 2], n["bz"][lane + 2]])
            assert np.array_equal(got_lo, lo) and np.array_equal(got_hi, hi)
        return fl, cl + cr, np.minimum(lol, lor), np.maximum(hil, hir)

    for i in range(len(nodes)):
        for side in ("left", "right"):
            ref, cnt = int(nodes[i][side]), int(nodes[i][side + "_count"])
            xref, xcnt = int(xn[i][side]), int(xn[i][side + "_count"])
            if not ref & LEAF:
                assert xref == ref
                continue
            first, count, _, _ = subtree(xref, xcnt)
            assert first == (ref & 0x7FFFFFFF) and count == cnt
            assert sorted(xorder[first:first + count].tolist()) == sorted(order[first:first + count].tolist())


def subtree_bounds(nodes, order, tri_lo, tri_hi, ref, cnt, seen, leaf_sizes):
    """Exact bounds of the triangles under `ref`, checking on the way that every inner node's two child
    boxes contain their subtrees (iterative post-order)."""
    result = {}
    stack = [(ref, cnt, False)]
    while stack:
        r, c, done = stack.pop()
        if r & LEAF:
            first = r & 0x7FFFFFFF
            ids = order[first:first + c]
            seen[ids] += 1
            leaf_sizes.append(c)
            result[(r, c)] = (tri_lo[ids].min(axis=0), tri_hi[ids].max(axis=0))
            continue
        n = nodes[r]
        kids = [(int(n["left"]), int(n["left_count"])), (int(n["right"]), int(n["right_count"]))]
        if not done:
            stack.append((r, c, True))
            stack.extend((k[0], k[1], False) for k in kids)
            continue
        boxes = []
        for side, k in enumerate(kids):
            lo, hi = result.pop((k[0], k[1]))
            blo = np.array([n["bx"][side], n["by"][side], n["bz"][side]])
            bhi = np.array([n["bx"][2 + side], n["by"][2 + side], n["bz"][2 + side]])
            assert (blo <= lo).all() and (bhi >= hi).all(), "child box does not contain its triangles"
            boxes.append((lo, hi))
        result[(r, c)] = (np.minimum(boxes[0][0], boxes[1][0]), np.maximum(boxes[0][1], boxes[1][1]))
    return result[(ref, cnt)]


def sah_cost(nodes, root):
    """Sum over inner nodes of the two child areas (times 1) -- the quantity binned SAH minimises
    greedily, relative comparisons only."""
    def area(lo, hi):
        d = np.maximum(hi - lo, 0)
        return 2 * (d[0] * d[1] + d[0] * d[2] + d[1] * d[2])
    total = 0.0
    for n in nodes:
        for side in range(2):
            lo = np.array([n["bx"][side], n["by"][side], n["bz"][side]], dtype=np.float64)
            hi = np.array([n["bx"][2 + side], n["by"][2 + side], n["bz"][2 + side]], dtype=np.float64)
            cnt = int(n["left_count" if side == 0 else "right_count"])
            total += area(lo, hi) * max(cnt, 1)
    return total


@pytest.mark.parametrize("name,kw", [("cornell-class", {}), ("test-mesh", {"n_tris": 3000, "seed": 4}),
                                     ("test-mesh", {"n_tris": 20000, "seed": 3})])
@pytest.mark.parametrize("buckets,max_leaf", [(32, 3), (8, 8), (2, 1)])
def test_search_builder_invariants(rt, scenes, name, kw, buckets, max_leaf):
    """The builder of the fast forms' tree: every triangle in exactly one leaf, leaves within the size
    limit (unless all centroids of a leaf coincide), child boxes contain their triangles, the result
    does not depend on the thread count, and the tree is valid input for the 4-wide collapse."""
    sc = scenes.load(name, **kw)
    nodes, order, info, st = rt.build_bvh_search_host(sc.positions, sc.indices, buckets, max_leaf, threads=1)
    nodes4, order4, info4, st4 = rt.build_bvh_search_host(sc.positions, sc.indices, buckets, max_leaf, threads=4)
    assert nodes.tobytes() == nodes4.tobytes() and np.array_equal(order, order4) and info == info4 and st == st4
    nt = len(sc.indices)
    assert sorted(order.tolist()) == list(range(nt))
    tri = sc.positions[sc.indices]                       # [nt, 3, 3]
    tri_lo, tri_hi = tri.min(axis=1), tri.max(axis=1)
    seen = np.zeros(nt, dtype=np.int64)
    sizes = []
    subtree_bounds(nodes, order, tri_lo, tri_hi, info["root_ref"], info["root_count"], seen, sizes)
    assert (seen == 1).all()
    assert st["num_objs"] == nt and st["num_leaves"] == len(sizes) and st["max_objs"] == max(sizes)
    cent = (tri[:, 0] + tri[:, 1] + tri[:, 2]) / np.float32(3)
    if max(sizes) > max_leaf:                            # only leaves whose centroids all coincide may be larger
        stack = [(info["root_ref"], info["root_count"])]
        while stack:
            r, c = stack.pop()
            if r & LEAF:
                if c > max_leaf:
                    ids = order[(r & 0x7FFFFFFF):(r & 0x7FFFFFFF) + c]
                    assert (cent[ids] == cent[ids][0]).all()
                continue
            n = nodes[r]
            stack.append((int(n["left"]), int(n["left_count"])))
            stack.append((int(n["right"]), int(n["right_count"])))
    if max(sizes) <= 8:
        wide, wroot, depth = rt.wide_bvh_host(nodes, info["root_ref"], info["root_count"])
        assert len(wide) >= 1 or nt <= 8


def test_search_builder_is_not_worse_than_the_reference_builder(rt, scenes):
    """Same leaf limit, same bucket count: trying every axis can only lower the greedy cost at each
    split; over a whole tree the summed child areas should not grow."""
    sc = scenes.load("test-mesh", n_tris=20000, seed=3)
    ref_nodes, _, _, _ = rt.build_bvh_host(sc.positions, sc.indices)
    s_nodes, _, _, _ = rt.build_bvh_search_host(sc.positions, sc.indices, 8, 8)
    s32_nodes, _, _, _ = rt.build_bvh_search_host(sc.positions, sc.indices, 32, 8)
    c_ref, c8, c32 = sah_cost(ref_nodes, 0), sah_cost(s_nodes, 0), sah_cost(s32_nodes, 0)
    assert c8 <= c_ref * 1.02 and c32 <= c_ref * 1.02, (c_ref, c8, c32)


def test_builders_do_not_depend_on_the_thread_count(rt, scenes):
    """Above 200,000 triangles both builders run their top levels with all threads scanning large
    ranges together and build the subtrees below as parallel tasks; unions and counts are exact, so
    the tree is the one a single thread builds."""
    sc = scenes.load("sponza-class")
    assert len(sc.indices) > 200000
    for build in (lambda th: rt.build_bvh_host(sc.positions, sc.indices, threads=th),
                  lambda th: rt.build_bvh_search_host(sc.positions, sc.indices, 32, 3, threads=th)):
        n1, o1, i1, s1 = build(1)
        for th in (3, 8):
            n, o, i, s = build(th)
            assert n.tobytes() == n1.tobytes() and np.array_equal(o, o1) and i == i1 and s == s1


def test_parallel_partition_on_skewed_ranges(rt):
    """The top levels partition large ranges on all threads (the reference's in-place partition, element for
    element).  A mesh whose splits are very uneven -- 97 % of the triangles in a small cluster, the rest spread
    thinly, and a run of exact duplicates -- must give the single-threaded tree whatever the thread count."""
    rng = np.random.default_rng(41)
    n = 220000                                                 # above the 200,000 from which the builders use their threads
    c = np.concatenate([rng.normal(0.0, 0.01, (n * 97 // 100, 3)), rng.uniform(-50.0, 50.0, (n - n * 97 // 100, 3))])
    c[1000:3000] = c[1000]                                     # coincident centroids inside the cluster
    rng.shuffle(c)
    tri = np.array([[0, 0, 0], [1e-3, 0, 0], [0, 1e-3, 0]], np.float32)
    pos = (c[:, None, :].astype(np.float32) + tri[None, :, :]).reshape(-1, 3)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    for build in (lambda th: rt.build_bvh_host(pos, idx, threads=th),
                  lambda th: rt.build_bvh_search_host(pos, idx, 32, 2, threads=th)):
        n1, o1, i1, s1 = build(1)
        for th in (2, 7, 16):
            nn, oo, ii, ss = build(th)
            assert nn.tobytes() == n1.tobytes() and np.array_equal(oo, o1) and ii == i1 and ss == s1, th
