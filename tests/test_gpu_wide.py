"""The 4-wide traversal forms (traversal options 2, 3, 5 = two rays per lane and 6 = node walk and triangle tests side by side) against the oracle.

The wide tree regroups the binary tree's boxes, and each box and triangle test
is the arithmetic of the binary form, so a ray tests a superset of the
triangles the reference's walk (src/bvh/intersect_stream.rs:67-181) tests.  The
closest-hit distance is therefore the oracle's, except for a ray that the
binary walk culls at an intermediate box the wide tree does not have (slab
rounding at a box face): such a ray must then agree with the oracle's brute
force over all triangles, unless it starts on the very surface it should hit
(t ~ 0), where either walk can lose that hit to the same rounding.  Any-hit answers follow the same rule.  Ids may differ
on exact-distance ties."""
import numpy as np
import pytest

from util import adversarial_rays, random_rays

pytestmark = pytest.mark.gpu

SCENES = (("test-mesh", {"n_tris": 20000, "seed": 3}), ("sponza-class", {}), ("cornell-class", {}),
          ("test-mesh", {"n_tris": 7, "seed": 5}))


def check_closest(O, sc, So, rays, io, to, ig, tg, rt, n_ordinary):
    """The first n_ordinary rays are camera and random rays.  The rest are axis-parallel
    rays from mesh vertices: with an infinite reciprocal the reference's slab arithmetic
    (plane*inf - origin*inf) gives +-inf or NaN by the sign of the plane coordinate, so
    whether a box passes depends on the far value at that moment and with it on the visit
    order.  For those rays only soundness is required: a reported hit is a real
    intersection of that triangle and nothing closer is claimed than brute force finds."""
    diff = np.flatnonzero(tg != to)
    ordinary = diff[diff < n_ordinary]
    assert ordinary.size <= max(2, 5e-4 * n_ordinary), "%d of %d rays differ" % (ordinary.size, n_ordinary)
    if ordinary.size:  # the wide form found the true closest hit that the binary walk culled
        ib, tb = So.intersect_brute(rays[ordinary])
        assert np.array_equal(tg[ordinary], tb), (tg[ordinary], to[ordinary], tb)
    special = diff[diff >= n_ordinary]
    if special.size:
        ib, tb = So.intersect_brute(rays[special])
        assert (tb <= tg[special]).all()
        P, I = sc.positions, sc.indices
        for i in special:
            if ig[i] == rt.NO_HIT:
                assert np.isinf(tg[i])
                continue
            v0, v1, v2 = P[I[ig[i]]]
            t = O.tri_intersect(v0, v1 - v0, v2 - v0, rays[i]["origin_far"][:3], rays[i]["direction"][:3])
            assert np.float32(t) == tg[i]
    same = tg == to
    assert np.array_equal((io == rt.NO_HIT)[same], (ig == rt.NO_HIT)[same])
    # exact-distance ties only (the Sponza-class stand-in has coplanar overlapping faces, and the
    # refined leaves test their triangles in another order)
    assert (io[same] != ig[same]).mean() < 3e-2
    return int(diff.size)


@pytest.mark.parametrize("form", [2, 3, 5, 6, 7, 8])
@pytest.mark.parametrize("coop", [0, 5, 4096])
def test_wide_trace_parity(O, rt, ctx_default, scenes, coop, form):
    ctx = ctx_default
    ctx.set_option("traversal_stream", form)
    ctx.set_option("coop_steps", coop)
    try:
        for name, kw in SCENES:
            sc = scenes.load(name, **kw)
            So = O.Scene(sc.positions, sc.indices, sc.geom_normals)
            Sg = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
            lo, hi = float(sc.positions.min()), float(sc.positions.max())
            ordinary = np.concatenate([O.generate_primary(sc.camera, 96, 54, 1),
                                       random_rays(O.make_rays, 30000, 5, lo=lo, hi=hi)])
            rays = np.concatenate([ordinary, adversarial_rays(O.make_rays, sc.positions)])
            io, to, _ = So.intersect_stream(rays)
            ig, tg, st = Sg.intersect_stream(rays, stats=False)
            if coop == 5:
                # forced wave-cooperative finish: distances within a few ulps (other cull order)
                n = ordinary.size
                hit = io[:n] != rt.NO_HIT
                assert np.array_equal(hit, ig[:n] != rt.NO_HIT)
                rel = np.abs(tg[:n][hit] - to[:n][hit]) / np.maximum(np.abs(to[:n][hit]), 1e-20)
                assert rel.max() <= 2e-6
            else:
                check_closest(O, sc, So, rays, io, to, ig, tg, rt, ordinary.size)
            sh = random_rays(O.make_rays, 30000, 6, lo=lo, hi=hi, near=1e-5, far=0.4 * (hi - lo))
            oo, _ = So.occluded1(sh)
            og, _ = Sg.occluded_stream(sh, stats=False)
            bad = np.flatnonzero(oo != og)
            assert bad.size <= 2
            if bad.size:  # the wide form may only find an occluder the binary walk culled, never lose one
                ib, _ = So.intersect_brute(sh[bad])
                assert np.array_equal(og[bad].astype(bool), ib != rt.NO_HIT)
            Sg.close()
    finally:
        ctx.set_option("coop_steps", 4096)
        ctx.set_option("traversal_stream", 1)


def test_wide_stats_request_runs_binary_form(O, rt, ctx, scenes):
    """Visit counters are defined by the binary walk: a call that asks for them gets that walk."""
    ctx.set_option("traversal_stream", 2)
    try:
        sc = scenes.load("test-mesh", n_tris=5000, seed=8)
        So = O.Scene(sc.positions, sc.indices, sc.geom_normals)
        Sg = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
        rays = random_rays(O.make_rays, 5000, 2)
        io, to, so = So.intersect_stream(rays)
        ig, tg, sg = Sg.intersect_stream(rays, stats=True)
        assert np.array_equal(to, tg)
        for k in ("total_rays", "inner_rays", "leaf_rays", "obj_intersect"):
            assert so[k] == sg[k], k
        Sg.close()
    finally:
        ctx.set_option("traversal_stream", 1)


@pytest.mark.parametrize("name,kw,W,H,spp,b", [("cornell-class", {}, 96, 54, 4, 2),
                                               ("test-mesh", {"n_tris": 4000, "seed": 9}, 80, 45, 4, 3)])
@pytest.mark.parametrize("forms", [(3, 2), (2, 3), (3, 5), (3, 6), (3, 7), (3, 8)])
def test_wide_render_parity(O, rt, ctx_default, scenes, name, kw, W, H, spp, b, forms):
    ctx = ctx_default
    sc = scenes.load(name, **kw)
    So = O.Scene(sc.positions, sc.indices, sc.geom_normals)
    Sg = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
    ao, sto = So.render(sc.camera, sc.light, W, H, spp, max_bounces=b)
    ctx.set_option("traversal_depth0", forms[0])
    ctx.set_option("traversal_deep", forms[1])
    try:
        ag, stg = Sg.render(sc.camera, sc.light, W, H, spp, max_bounces=b)
    finally:
        ctx.set_option("traversal_depth0", 3)
        ctx.set_option("traversal_deep", 2)
    assert abs(stg["total_rays"] - sto["total_rays"]) <= 1e-3 * sto["total_rays"]
    differs = ~np.isclose(ag, ao, rtol=2e-5, atol=1e-6).all(axis=-1)
    assert differs.mean() < 5e-3, differs.mean()
    rmse = float(np.sqrt(np.mean((ag[..., :3] / spp - ao[..., :3] / spp) ** 2)))
    assert rmse < 2e-3, rmse
    Sg.close()


@pytest.mark.parametrize("name,b", [("sponza-class", 0), ("sponza-class", 2)])
def test_full_size_fast_mode_against_exact_mode(rt, ctx, ctx_default, scenes, name, b):
    """BASELINE-size frame (1920x1080): the library defaults (a search-built tree, 4-wide quantized
    nodes, fused depth 0, wave hand-off) against the exact configuration on the same GPU.  Same rays are
    traced up to the documented differences: ray totals within 0.1 %, fewer than 0.5 % of the pixels
    differ beyond rounding, and the images agree to RMSE < 2e-3."""
    sc = scenes.load(name)
    W, H, spp = 1920, 1080, 2
    Se = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
    ae, ste = Se.render(sc.camera, sc.light, W, H, spp, max_bounces=b)
    Se.close()
    Sf = rt.Scene(ctx_default, sc.positions, sc.indices, sc.geom_normals)
    af, stf = Sf.render(sc.camera, sc.light, W, H, spp, max_bounces=b)
    Sf.close()
    assert (af[..., 3] == spp).all() and np.isfinite(af).all()
    assert abs(stf["total_rays"] - ste["total_rays"]) <= 1e-3 * ste["total_rays"]
    differs = ~np.isclose(af, ae, rtol=2e-5, atol=1e-6).all(axis=-1)
    assert differs.mean() < 5e-3, differs.mean()
    rmse = float(np.sqrt(np.mean((af[..., :3] / spp - ae[..., :3] / spp) ** 2)))
    assert rmse < 2e-3, rmse


@pytest.mark.parametrize("form", [2, 3, 5, 6, 7, 8])
def test_fast_forms_on_degenerate_leaves(O, rt, ctx_default, scenes, form):
    """A mesh whose tree has a root leaf, and one with clusters of 20 coincident triangles inside a
    larger mesh (leaves beyond 8 triangles in both builders): the fast forms still find the oracle's hits."""
    ctx = ctx_default
    ctx.set_option("traversal_stream", form)
    try:
        pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
        Sg = rt.Scene(ctx, pos, np.array([[0, 1, 2]], np.uint32), np.array([[0, 0, 1]], np.float32))
        rays = rt.make_rays([[0.25, 0.25, 1.0], [2.0, 2.0, 1.0]], [[0, 0, -1.0], [0, 0, -1.0]])
        ids, t, _ = Sg.intersect_stream(rays, stats=False)
        assert ids.tolist() == [0, rt.NO_HIT] and t[0] == 1.0
        Sg.close()

        sc = scenes.load("test-mesh", n_tris=3000, seed=4)
        base = len(sc.positions)
        extra_pos = np.array([[0.1, 0.1, 5.0], [0.4, 0.1, 5.0], [0.1, 0.4, 5.0],
                              [-0.5, -0.5, -5.0], [-0.2, -0.5, -5.0], [-0.5, -0.2, -5.0]], np.float32)
        positions = np.vstack([sc.positions, extra_pos]).astype(np.float32)
        dup = np.vstack([np.tile(np.array([[base, base + 1, base + 2]], np.uint32), (20, 1)),
                         np.tile(np.array([[base + 3, base + 4, base + 5]], np.uint32), (20, 1))])
        indices = np.vstack([sc.indices, dup]).astype(np.uint32)
        normals = np.vstack([sc.geom_normals, np.tile(np.array([[0, 0, 1]], np.float32), (40, 1))]).astype(np.float32)
        So = O.Scene(positions, indices, normals)
        Sg = rt.Scene(ctx, positions, indices, normals)
        lo, hi = positions.min(), positions.max()
        rnd = random_rays(O.make_rays, 20000, 5, lo=lo, hi=hi)
        aimed = rt.make_rays([[0.2, 0.2, 9.0], [-0.4, -0.4, -9.0], [0.2, 0.2, 9.0]],
                             [[0, 0, -1.0], [0, 0, 1.0], [0, 0, 1.0]])
        for rays, what in ((rnd, "random"), (aimed, "aimed at the clusters")):
            io, to, _ = So.intersect_stream(rays)
            ig, tg, _ = Sg.intersect_stream(rays, stats=False)
            same = (tg == to) | (np.isinf(tg) & np.isinf(to))
            assert same.mean() > 1 - 5e-4, what
            og, _ = Sg.occluded_stream(rays, stats=False)
            assert (og.astype(bool) == (io != rt.NO_HIT)).mean() > 1 - 5e-4, what
        ig, tg, _ = Sg.intersect_stream(aimed, stats=False)
        assert tg[0] == 4.0 and tg[1] == 4.0 and ig[2] == rt.NO_HIT
        assert len(sc.indices) <= ig[0] < len(sc.indices) + 20 and len(sc.indices) + 20 <= ig[1]
        Sg.close()
    finally:
        ctx.set_option("traversal_stream", 1)


@pytest.mark.parametrize("max_leaf", [1, 2, 3])
@pytest.mark.parametrize("coop", [0, 4096])   # a hand-over to the cooperative finish after 5 iterations comes at
                                              # another item in the two forms, and that walk has its own tie order
def test_form8_is_form2_bit_for_bit(O, rt, ctx_default, scenes, coop, max_leaf):
    """Form 8 takes a leaf and the item that follows it in one iteration; the visits, their order and the
    arithmetic are those of form 2, so ids (ties included), distances and occlusion flags are equal bit for
    bit -- on trees with leaves of one, two and three triangles (the third is fetched in a pass of its own)."""
    ctx = ctx_default
    ctx.set_option("coop_steps", coop)
    ctx.set_option("fast_tree_max_leaf", max_leaf)
    try:
        for name, kw in SCENES:
            sc = scenes.load(name, **kw)
            Sg = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
            lo, hi = float(sc.positions.min()), float(sc.positions.max())
            rays = np.concatenate([O.generate_primary(sc.camera, 96, 54, 1),
                                   random_rays(O.make_rays, 30000, 5, lo=lo, hi=hi),
                                   adversarial_rays(O.make_rays, sc.positions)])
            out = {}
            for form in (2, 8):
                ctx.set_option("traversal_stream", form)
                ids, t, _ = Sg.intersect_stream(rays.copy(), stats=False)
                occ, _ = Sg.occluded_stream(rays.copy(), stats=False)
                out[form] = (ids, t, occ)
            assert np.array_equal(out[2][0], out[8][0]), name
            assert np.array_equal(out[2][1].view(np.uint32), out[8][1].view(np.uint32)), name
            assert np.array_equal(out[2][2], out[8][2]), name
            Sg.close()
    finally:
        ctx.set_option("traversal_stream", 1)
        ctx.set_option("coop_steps", 4096)
        ctx.set_option("fast_tree_max_leaf", 2)
