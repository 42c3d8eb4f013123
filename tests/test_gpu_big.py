"""GPU parity on the large workloads: the dragon-class (7.2 M triangles, binary tree 33 levels
deep) and San-Miguel-class (11 M triangles, 29 levels) stand-ins at the camera, light and bounce
cap of src/scene.rs:95-126, in 1920x1080 and 3840x2160, and the traversal stack beyond its LDS part.

The reference walks such trees with a 32-entry ArrayVec (src/bvh/intersect1.rs:12,115) that panics
when it overflows; here a lane's entries beyond the first RT_LDS_STACK_ENTRIES go to a global
overflow area, which test_deep_stack_* enter on purpose."""
import numpy as np
import pytest

from util import assert_hits_match, deep_chain_mesh, random_rays, surface_rays

pytestmark = pytest.mark.gpu
COUNTERS = ("total_rays", "inner_rays", "leaf_rays", "obj_intersect")
RTOL, ATOL = 2e-5, 1e-6


class Big:
    pass


@pytest.fixture(scope="module", params=["dragon-class", "san-miguel-class"])
def big(request, O, rt, ctx, ctx_default, scenes):
    """One large stand-in at a time: mesh, oracle scene, the GPU scene in the exact configuration and
    the GPU scene with the library defaults (its own search-built 4-wide tree)."""
    b = Big()
    b.name = request.param
    b.sc = scenes.load(b.name)
    b.So = O.Scene(b.sc.positions, b.sc.indices, b.sc.geom_normals)
    b.Se = rt.Scene(ctx, b.sc.positions, b.sc.indices, b.sc.geom_normals)
    b.Sf = rt.Scene(ctx_default, b.sc.positions, b.sc.indices, b.sc.geom_normals)
    yield b
    b.Se.close()
    b.Sf.close()
    b.So.close()


def test_big_tree_is_the_oracles(big):
    so, sg = big.So.bvh_stats(), big.Se.bvh_stats()
    assert sg == so
    assert so["max_depth"] >= 29 and so["num_objs"] == big.sc.num_triangles and so["max_objs"] <= 8
    nodes_g, tris_g = big.Se.flatten()
    nodes_o, tris_o = big.So.flatten()
    assert nodes_g.shape == nodes_o.shape and np.array_equal(nodes_g.view(np.uint8), nodes_o.view(np.uint8))
    assert np.array_equal(tris_g.view(np.uint8), tris_o.view(np.uint8))


def make_ray_sets(O, big):
    sc = big.sc
    prim = O.generate_primary(sc.camera, 320, 180, 1)                       # 57,600 camera rays
    io, to, _ = big.So.intersect_stream(prim)
    lo, hi = sc.positions.min(axis=0), sc.positions.max(axis=0)
    rng = np.random.default_rng(17)
    o = rng.uniform(lo, hi, size=(50000, 3)).astype(np.float32)
    d = rng.normal(size=(50000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rnd = O.make_rays(o, d.astype(np.float32))
    bounce = surface_rays(O.make_rays, np.tile(prim, 2), np.tile(io, 2), np.tile(to, 2), 23)
    shadow = surface_rays(O.make_rays, prim, io, to, 29, far=0.25 * float(np.linalg.norm(hi - lo)))
    return prim, rnd, bounce, shadow


def test_big_trace_exact(O, rt, ctx, big):
    """Closest hit and any hit on camera rays, uniform random rays and bounce-like rays from the
    surface: distances as f32 values, ids and per-ray visit counters equal to the oracle's stream
    traversal; flags and counters equal to its occluded1."""
    prim, rnd, bounce, shadow = make_ray_sets(O, big)
    assert bounce.shape[0] >= 50000 or big.name == "dragon-class"          # the dragon fills a part of the frame
    for rays, what in ((prim, "primary"), (rnd, "random"), (bounce, "bounce-like"), (shadow, "finite far")):
        io, to, sto = big.So.intersect_stream(rays)
        ig, tg, stg = big.Se.intersect_stream(rays)
        assert assert_hits_match(io, to, ig, tg, "%s %s" % (big.name, what)) == 0
        for k in COUNTERS:
            assert stg[k] == sto[k], (what, k, stg[k], sto[k])
        oo, soo = big.So.occluded1(rays)
        og, sog = big.Se.occluded_stream(rays)
        assert np.array_equal(oo, og), what
        for k in COUNTERS:
            assert sog[k] == soo[k], (what, k)
    assert (ig != rt.NO_HIT).any()


def test_big_single_ray_order(O, rt, ctx, big):
    """Bvh::intersect1's own visit order (src/bvh/intersect1.rs:109-210) on the deep trees."""
    prim, rnd, bounce, shadow = make_ray_sets(O, big)
    ctx.set_option("traversal_stream", 4)
    for rays, what in ((prim, "primary"), (bounce[:30000], "bounce-like"), (shadow[:30000], "finite far")):
        io, to, sto = big.So.intersect1(rays)
        ig, tg, stg = big.Se.intersect_stream(rays)
        assert np.array_equal(to, tg) and np.array_equal(io, ig), what
        for k in COUNTERS:
            assert stg[k] == sto[k], (what, k, stg[k], sto[k])


def test_big_trace_fast_forms(O, rt, ctx_default, big):
    """The forms the renderer uses by default (quantized 4-wide tree over a search-built binary
    tree, stepwise and while-while) on the same rays: the oracle's distances except for a ray the
    binary walk culls at a box face, which must then agree with brute force."""
    prim, rnd, bounce, shadow = make_ray_sets(O, big)
    for form in (2, 3, 5, 6, 7, 8):
        ctx_default.set_option("traversal_stream", form)
        for rays, what in ((prim, "primary"), (bounce, "bounce-like"), (shadow, "finite far")):
            io, to, _ = big.So.intersect_stream(rays)
            ig, tg, _ = big.Sf.intersect_stream(rays, stats=False)
            diff = np.flatnonzero(tg != to)
            assert diff.size <= max(2, 5e-4 * rays.shape[0]), (form, what, diff.size)
            if diff.size:
                ib, tb = big.So.intersect_brute(rays[diff[:8]])             # brute force is 11 M tests per ray
                assert np.array_equal(tg[diff[:8]], tb), (form, what)
            same = tg == to
            assert np.array_equal((io == rt.NO_HIT)[same], (ig == rt.NO_HIT)[same])
            oo, _ = big.So.occluded1(rays)
            og, _ = big.Sf.occluded_stream(rays, stats=False)
            bad = np.flatnonzero(oo != og)
            assert bad.size <= max(2, 5e-4 * rays.shape[0]), (form, what, bad.size)
            if bad.size:
                ib, _ = big.So.intersect_brute(rays[bad[:8]])
                assert np.array_equal(og[bad[:8]].astype(bool), ib != rt.NO_HIT)


def test_big_render_against_oracle(O, rt, ctx, ctx_default, big):
    """A small frame of the full scene against the oracle's default integrator: exact configuration
    to rounding of the pixel sums with every counter equal, library defaults within the bounds
    stated for the fast forms."""
    sc = big.sc
    W, H, spp, b = 240, 135, 2, 2
    ao, sto = big.So.render(sc.camera, sc.light, W, H, spp, max_bounces=b)
    ag, stg = big.Se.render(sc.camera, sc.light, W, H, spp, max_bounces=b, visit_stats=True)
    assert np.array_equal(ag[..., 3], ao[..., 3])
    assert np.allclose(ag, ao, rtol=RTOL, atol=ATOL), float(np.abs(ag - ao).max())
    for k in COUNTERS:
        assert stg[k] == sto[k], (k, stg[k], sto[k])
    af, stf = big.Sf.render(sc.camera, sc.light, W, H, spp, max_bounces=b)
    assert abs(stf["total_rays"] - sto["total_rays"]) <= 1e-3 * sto["total_rays"]
    differs = ~np.isclose(af, ao, rtol=RTOL, atol=ATOL).all(axis=-1)
    assert differs.mean() < 5e-3, differs.mean()
    rmse = float(np.sqrt(np.mean((af[..., :3] / spp - ao[..., :3] / spp) ** 2)))
    assert rmse < 2e-3, rmse


def test_big_full_size_defaults_against_exact(rt, big):
    """1920x1080, max_bounces 2 (src/main.rs:103-104, src/args.rs:27): the library defaults against
    the exact configuration on the same GPU, plus the properties that need no oracle."""
    sc = big.sc
    W, H, spp, b = 1920, 1080, 2, 2
    ae, ste = big.Se.render(sc.camera, sc.light, W, H, spp, max_bounces=b)
    af, stf = big.Sf.render(sc.camera, sc.light, W, H, spp, max_bounces=b)
    for a in (ae, af):
        assert (a[..., 3] == spp).all() and np.isfinite(a).all() and a[..., :3].min() >= 0
    assert abs(stf["total_rays"] - ste["total_rays"]) <= 1e-3 * ste["total_rays"]
    differs = ~np.isclose(af, ae, rtol=RTOL, atol=ATOL).all(axis=-1)
    assert differs.mean() < 5e-3, differs.mean()
    rmse = float(np.sqrt(np.mean((af[..., :3] / spp - ae[..., :3] / spp) ** 2)))
    assert rmse < 2e-3, rmse
    # tile split over 8 shards (the multi-GPU plan): disjoint pixels, bit-identical sum
    acc = np.zeros_like(af)
    rays = 0
    for r in range(8):
        _, s = big.Sf.render(sc.camera, sc.light, W, H, spp, max_bounces=b, shard_index=r, shard_count=8, accum=acc)
        rays += s["total_rays"]
    assert np.array_equal(acc, af) and rays == stf["total_rays"]


def test_big_full_size_exact_against_oracle(big):
    """The full 1920x1080 frame, max_bounces 2, one sample per pixel: the exact configuration on the GPU
    against the oracle's default integrator (all 2 M paths, not a reduced frame), accumulator and the
    per-ray counters; the library defaults against the same oracle frame under the fast-mode rules."""
    sc = big.sc
    W, H, spp, b = 1920, 1080, 1, 2
    ao, sto = big.So.render(sc.camera, sc.light, W, H, spp, max_bounces=b)
    ae, ste = big.Se.render(sc.camera, sc.light, W, H, spp, max_bounces=b, visit_stats=True)
    assert np.array_equal(ae[..., 3], ao[..., 3])
    assert np.allclose(ae, ao, rtol=RTOL, atol=ATOL), float(np.abs(ae - ao).max())
    for k in COUNTERS:
        assert ste[k] == sto[k], (k, ste[k], sto[k])
    af, stf = big.Sf.render(sc.camera, sc.light, W, H, spp, max_bounces=b)
    assert abs(stf["total_rays"] - sto["total_rays"]) <= 1e-3 * sto["total_rays"]
    differs = ~np.isclose(af, ao, rtol=RTOL, atol=ATOL).all(axis=-1)
    assert differs.mean() < 5e-3, differs.mean()


def test_big_4k_progressive_sample_ranges(rt, big):
    """3840x2160, progressive: a frame rendered as successive sample ranges (what sample-range
    sharding reduces) equals the frame rendered at once up to the order of the f32 pixel sums, every
    pixel carries its sample count, and any-hit agrees with closest-hit-exists on the frame's
    camera rays."""
    if big.name != "san-miguel-class":
        pytest.skip("the 4K workload is the San-Miguel-class scene")
    sc = big.sc
    W, H, b = 3840, 2160, 2
    whole, st = big.Sf.render(sc.camera, sc.light, W, H, 3, max_bounces=b)
    assert (whole[..., 3] == 3).all() and np.isfinite(whole).all() and whole[..., :3].min() >= 0
    assert 3 * W * H <= st["total_rays"] <= 3 * W * H * 2 * (b + 1)
    parts = np.zeros_like(whole)
    rays = 0
    for off, n in ((0, 1), (1, 2)):
        _, s = big.Sf.render(sc.camera, sc.light, W, H, n, max_bounces=b, sample_offset=off, accum=parts)
        rays += s["total_rays"]
    assert rays == st["total_rays"]
    assert np.allclose(parts, whole, rtol=RTOL, atol=ATOL)
    cam_rays = big.Se.ctx.generate_primary(sc.camera, W, H, 1)
    ids, t, _ = big.Se.intersect_stream(cam_rays, stats=False)
    occ, _ = big.Se.occluded_stream(cam_rays, stats=False)
    assert np.array_equal(occ.astype(bool), ids != rt.NO_HIT)
    idf, tf, _ = big.Sf.intersect_stream(cam_rays, stats=False)            # exact form on the other scene object
    assert np.array_equal(tf, t)


# ---------------------------------------------------------------------------------------------
# the traversal stack beyond its LDS part


def chain_rays(make_rays, pos):
    o, d = [], []
    xs = pos[::3, 0]
    for k in range(0, len(xs), 7):          # starting between triangles k-1 and k
        x0 = 0.0 if k == 0 else 0.5 * float(xs[k - 1] + xs[k])
        for dy in (0.5, 0.56, 0.7, 0.3):    # beside the triangles, grazing their edge, through them, below
            o.append([x0, x0 * dy, x0 * 0.013])  # no exactly-zero component: those are the NaN slab cases
            d.append([1.0, dy, 0.013])
    far_end = float(xs[-1]) * 2.0
    for dy in (0.5, 0.7):                   # and back from beyond the last one
        o.append([far_end, far_end * dy, far_end * 0.013])
        d.append([-1.0, -dy, -0.013])
    d = np.array(d, np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return make_rays(np.array(o, np.float32), d.astype(np.float32))


def test_deep_stack_exact_forms(O, rt, ctx):
    pos, idx, nrm = deep_chain_mesh()
    So, Sg = O.Scene(pos, idx, nrm), rt.Scene(ctx, pos, idx, nrm)
    depth = So.bvh_stats()["max_depth"]
    assert depth > rt.LDS_STACK_ENTRIES + 8 and Sg.bvh_stats() == So.bvh_stats()
    rays = np.concatenate([chain_rays(O.make_rays, pos), random_rays(O.make_rays, 4000, 5, lo=-2.0, hi=50.0)])
    for form in (0, 1, 4):
        ctx.set_option("traversal_stream", form)
        io, to, sto = (So.intersect1 if form == 4 else So.intersect_stream)(rays)
        ig, tg, stg = Sg.intersect_stream(rays)
        assert np.array_equal(to, tg) and np.array_equal(io, ig), form
        for k in COUNTERS:
            assert stg[k] == sto[k], (form, k)
        deepest = ctx.last_max_stack_entries()[0]
        assert rt.LDS_STACK_ENTRIES < deepest <= depth, (form, deepest)      # the overflow area was used
        oo, soo = So.occluded1(rays)
        og, sog = Sg.occluded_stream(rays)
        assert np.array_equal(oo, og)
        for k in COUNTERS:
            assert sog[k] == soo[k], (form, k)
        assert ctx.last_max_stack_entries()[1] > rt.LDS_STACK_ENTRIES
    assert (ig != rt.NO_HIT).sum() > 20 and (ig == rt.NO_HIT).sum() > 20
    # the same launches without counters (other kernel instantiations, same stack code)
    ctx.set_option("traversal_stream", 1)
    io, to, _ = So.intersect_stream(rays)
    ig2, tg2, _ = Sg.intersect_stream(rays, stats=False)
    assert np.array_equal(ig2, io) and np.array_equal(tg2, to)
    Sg.close()


@pytest.mark.parametrize("coop", [0, 4096])
def test_deep_stack_fast_forms(O, rt, ctx_default, coop):
    """The 4-wide forms defer up to three children per level; on the chain their stacks pass the LDS
    part as well (16 entries in the while-while any-hit kernel).  In a render, too: camera at the origin
    looking along the chain."""
    ctx = ctx_default
    pos, idx, nrm = deep_chain_mesh()
    So, Sg = O.Scene(pos, idx, nrm), rt.Scene(ctx, pos, idx, nrm)
    rays = np.concatenate([chain_rays(O.make_rays, pos), random_rays(O.make_rays, 4000, 5, lo=-2.0, hi=50.0)])
    io, to, _ = So.intersect_stream(rays)
    oo, _ = So.occluded1(rays)
    ctx.set_option("coop_steps", coop)
    for form in (2, 3, 5, 6, 7, 8):
        ctx.set_option("traversal_stream", form)
        ig, tg, _ = Sg.intersect_stream(rays, stats=False)
        assert np.array_equal(tg, to), form
        og, _ = Sg.occluded_stream(rays, stats=False)
        assert np.array_equal(og, oo), form
    Sg.close()


def test_pass_size_at_full_size(rt, ctx_default, scenes):
    """The 1920x1080 frame at the reference's 128 samples per pixel as one pass of 2.65e8 paths (the
    library default) and as sixteen passes of 2^24: same ray totals, accumulators equal up to the order of
    the per-pixel f32 additions.  (Slot indices, queue offsets and sort sizes beyond 2^26 entries.)"""
    sc = scenes.load("sponza-class")
    W, H, spp, b = 1920, 1080, 128, 1
    Sg = rt.Scene(ctx_default, sc.positions, sc.indices, sc.geom_normals)
    try:
        one, s1 = Sg.render(sc.camera, sc.light, W, H, spp, max_bounces=b)
        ctx_default.set_max_paths(1 << 24)
        many, s2 = Sg.render(sc.camera, sc.light, W, H, spp, max_bounces=b)
    finally:
        ctx_default.set_max_paths(1 << 28)
        Sg.close()
    assert (one[..., 3] == spp).all() and np.isfinite(one).all()
    assert s1["total_rays"] == s2["total_rays"]
    assert np.allclose(one, many, rtol=2e-5, atol=1e-6)
