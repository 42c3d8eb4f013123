"""GPU parity of the render path (rt_generate_primary, rt_render, rt_resolve
through the C ABI) against the CPU oracle's restatement of the reference's
default integrator."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# counters that are sums over rays (inner_visit, leaf_visit and leaf_objs count per batch in the stream traversal)
COUNTERS = ("total_rays", "inner_rays", "leaf_rays", "obj_intersect")

# Per-pixel sums are added in a different order on the GPU (per path, then per
# pixel) than in the reference (per bounce, in ray order): relative tolerance on
# the float4 accumulator, everything upstream of the sum is bit-identical.
RTOL, ATOL = 2e-5, 1e-6


def test_primary_rays_bit_exact(O, rt, ctx, scenes):
    for name, W, H, spp, ts in (("cornell-class", 96, 54, 3, 16), ("sponza-class", 200, 113, 2, 16),
                                ("dragon-class", 64, 36, 1, 7)):
        cam = scenes.load(name).camera if name != "dragon-class" else scenes.Camera(
            eye=(1.3, -2.1, 0.55), target=(0, 0, 0.35), fov_deg=60.0)
        a = O.generate_primary(cam, W, H, spp, tile_size=ts, seed=99, sample_offset=5)
        b = ctx.generate_primary(cam, W, H, spp, tile_size=ts, seed=99, sample_offset=5)
        assert a.tobytes() == b.tobytes(), name


@pytest.mark.parametrize("name,W,H,spp,b", [("cornell-class", 96, 54, 8, 2), ("cornell-class", 50, 35, 5, 0),
                                            ("test-mesh", 80, 45, 4, 3), ("sponza-class", 128, 72, 4, 2)])
def test_render_parity(O, rt, ctx, scenes, name, W, H, spp, b):
    sc = scenes.load(name)
    So = O.Scene(sc.positions, sc.indices, sc.geom_normals)
    Sg = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
    ao, sto = So.render(sc.camera, sc.light, W, H, spp, max_bounces=b)
    ag, stg = Sg.render(sc.camera, sc.light, W, H, spp, max_bounces=b, visit_stats=True)
    assert np.array_equal(ag[..., 3], ao[..., 3])
    assert np.allclose(ag, ao, rtol=RTOL, atol=ATOL), float(np.abs(ag - ao).max())
    rmse = float(np.sqrt(np.mean((ag[..., :3] / spp - ao[..., :3] / spp) ** 2)))
    assert rmse < 1e-6
    # every ray of every path is the same ray: the counters match exactly
    for k in COUNTERS:
        assert stg[k] == sto[k], (k, stg[k], sto[k])
    # without visit statistics only total_rays is filled
    ag2, st2 = Sg.render(sc.camera, sc.light, W, H, spp, max_bounces=b)
    assert np.array_equal(ag2, ag) and st2["total_rays"] == sto["total_rays"] and st2["inner_rays"] == 0
    # 8-bit image: identical except where rounding of the sum crosses an integer
    ro, rg = O.resolve(ao), ctx.resolve(ag)
    assert np.abs(ro.astype(int) - rg.astype(int)).max() <= 1 and (ro != rg).mean() < 1e-3
    assert np.array_equal(ctx.resolve(ao), ro)
    Sg.close()


def test_sharding_and_passes(rt, ctx, scenes):
    """Tile shards are disjoint (sum is bit-identical to one call); sample ranges
    and pass sizes only reorder f32 additions."""
    sc = scenes.load("cornell-class")
    Sg = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
    W, H, spp = 90, 50, 6   # partial tiles on both edges
    full, st = Sg.render(sc.camera, sc.light, W, H, spp, max_bounces=2)
    acc = np.zeros((H, W, 4), np.float32)
    tot = 0
    for r in range(3):
        _, s = Sg.render(sc.camera, sc.light, W, H, spp, shard_index=r, shard_count=3, accum=acc)
        tot += s["total_rays"]
    assert np.array_equal(acc, full) and tot == st["total_rays"]
    acc = np.zeros((H, W, 4), np.float32)
    Sg.render(sc.camera, sc.light, W, H, 2, sample_offset=0, accum=acc)
    Sg.render(sc.camera, sc.light, W, H, 4, sample_offset=2, accum=acc)
    assert np.allclose(acc, full, rtol=RTOL, atol=ATOL)
    one, s1 = Sg.render(sc.camera, sc.light, W, H, spp, samples_per_pass=1)
    assert np.allclose(one, full, rtol=RTOL, atol=ATOL) and s1["total_rays"] == st["total_rays"]
    # a workspace smaller than the frame forces tile chunking
    ctx.set_max_paths(4096)
    small, s2 = Sg.render(sc.camera, sc.light, W, H, spp)
    ctx.set_max_paths(1 << 28)  # the library default
    assert np.allclose(small, full, rtol=RTOL, atol=ATOL) and s2["total_rays"] == st["total_rays"]
    Sg.close()


def test_golden_render_fixture(rt, ctx, scenes):
    z = np.load(os.path.join(G, "render_cornell.npz"))
    Sg = rt.Scene(ctx, z["positions"], z["indices"], z["geom_normals"])
    c, l = z["camera"], z["light"]
    cam = scenes.Camera(eye=c[0:3], target=c[3:6], up=c[6:9], fov_deg=float(c[9]))

    class L:
        pos = l[0:3]
        intensity = l[3:6]

    W, H, spp, b, ts, seed = [int(x) for x in z["params"]]
    acc, st = Sg.render(cam, L, W, H, spp, max_bounces=b, tile_size=ts, seed=seed, visit_stats=True)
    assert np.allclose(acc, z["accum"], rtol=RTOL, atol=ATOL)
    names = sorted(["total_rays", "inner_visit", "inner_rays", "leaf_visit", "leaf_rays", "leaf_objs", "obj_intersect"])
    want = dict(zip(names, z["stats"].tolist()))
    for k in COUNTERS:
        assert st[k] == want[k], k
    Sg.close()


def test_full_size_properties(rt, ctx, scenes):
    """BASELINE-size run (Sponza-class, 1920x1080) checked through properties
    that need no oracle: sample counts, non-negativity, additivity of
    sample ranges, any-hit == closest-hit-exists on the frame's primary rays."""
    sc = scenes.load("sponza-class")
    Sg = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
    W, H = 1920, 1080
    a, st = Sg.render(sc.camera, sc.light, W, H, 2, max_bounces=0)
    assert (a[..., 3] == 2).all() and a[..., :3].min() >= 0 and np.isfinite(a).all()
    assert W * H * 2 <= st["total_rays"] <= 2 * W * H * 2
    b = np.zeros_like(a)
    Sg.render(sc.camera, sc.light, W, H, 1, sample_offset=0, max_bounces=0, accum=b)
    Sg.render(sc.camera, sc.light, W, H, 1, sample_offset=1, max_bounces=0, accum=b)
    assert np.allclose(a, b, rtol=RTOL, atol=ATOL)
    rays = ctx.generate_primary(sc.camera, W, H, 1)
    ids, t, _ = Sg.intersect_stream(rays, stats=False)
    occ, _ = Sg.occluded_stream(rays, stats=False)
    assert np.array_equal(occ.astype(bool), ids != rt.NO_HIT)
    assert (ids != rt.NO_HIT).mean() > 0.99                      # closed atrium
    hit = ids != rt.NO_HIT
    assert (t[hit] > 0).all() and np.isinf(t[~hit]).all()
    Sg.close()


def test_fused_primary_is_bit_identical(rt, ctx, scenes):
    """Depth 0 without queues (the extend launch makes the camera rays itself) against the
    generate-kernel path: the same accumulator bit for bit, ragged image size and partial last
    pass included."""
    sc = scenes.load("test-mesh", n_tris=4000, seed=9)
    Sg = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
    try:
        for (W, H, spp, k, b) in ((100, 57, 5, 2, 2), (64, 36, 4, 0, 0)):
            ctx.set_option("fuse_primary", 1)
            a1, s1 = Sg.render(sc.camera, sc.light, W, H, spp, max_bounces=b, samples_per_pass=k)
            ctx.set_option("fuse_primary", 0)
            a0, s0 = Sg.render(sc.camera, sc.light, W, H, spp, max_bounces=b, samples_per_pass=k)
            assert s1["total_rays"] == s0["total_rays"]
            assert np.array_equal(a1, a0)
    finally:
        ctx.set_option("fuse_primary", 1)
    Sg.close()


@pytest.mark.parametrize("which", ["ctx", "ctx_default"])
def test_overlapped_connect_is_bit_identical(rt, request, scenes, which):
    """The any-hit launch of depth d on its own stream beside the closest-hit launch of depth d + 1
    (option overlap_connect) against one launch after the other on one stream: the same accumulator bit for bit
    and the same ray counts, with several passes per frame, sorted and unsorted queues."""
    c = request.getfixturevalue(which)
    sc = scenes.load("test-mesh", n_tris=6000, seed=13)
    Sg = rt.Scene(c, sc.positions, sc.indices, sc.geom_normals)
    try:
        for (W, H, spp, k, b, sort_from) in ((100, 57, 6, 2, 3, 1), (64, 36, 4, 0, 2, 0xFFFFFFFF), (80, 45, 3, 1, 0, 0)):
            c.set_option("sort_connect_from", sort_from)
            c.set_option("overlap_connect", 1)
            a1, s1 = Sg.render(sc.camera, sc.light, W, H, spp, max_bounces=b, samples_per_pass=k)
            a1b, _ = Sg.render(sc.camera, sc.light, W, H, spp, max_bounces=b, samples_per_pass=k)
            c.set_option("overlap_connect", 0)
            a0, s0 = Sg.render(sc.camera, sc.light, W, H, spp, max_bounces=b, samples_per_pass=k)
            assert s1["total_rays"] == s0["total_rays"]
            assert np.array_equal(a1, a0) and np.array_equal(a1b, a0)
    finally:
        c.set_option("overlap_connect", 0)
        c.set_option("sort_connect_from", 1)
    Sg.close()


def test_workspace_allocation_fallback(rt, ctx_default, scenes):
    """When the workspace of a pass of max_paths cannot be allocated the call releases what it had
    grown, halves the pass and plans the frame again: forced here through the workspace_limit_mb
    option.  Same ray total, accumulator equal up to the order of the per-pixel additions; a limit
    below the smallest pass is still an RT_ERR_NOMEM."""
    c = ctx_default
    sc = scenes.load("test-mesh", n_tris=5000, seed=3)
    Sg = rt.Scene(c, sc.positions, sc.indices, sc.geom_normals)
    W, H, spp = 640, 360, 8
    try:
        full, st = Sg.render(sc.camera, sc.light, W, H, spp, max_bounces=2)
        assert c.last_pass_paths() >= W * H * spp
        c.set_option("workspace_limit_mb", 64)                      # about 350 k paths
        small, s2 = Sg.render(sc.camera, sc.light, W, H, spp, max_bounces=2)
        assert (1 << 16) <= c.last_pass_paths() < W * H * spp // 4
        assert s2["total_rays"] == st["total_rays"]
        assert np.allclose(small, full, rtol=RTOL, atol=ATOL)
        c.set_option("workspace_limit_mb", 1)                       # less than the floor of 2^16 paths needs
        with pytest.raises(rt.RtError):
            Sg.render(sc.camera, sc.light, W, H, spp, max_bounces=2)
        c.set_option("workspace_limit_mb", 0)
        again, s3 = Sg.render(sc.camera, sc.light, W, H, spp, max_bounces=2)
        assert np.array_equal(again, full) and s3["total_rays"] == st["total_rays"]
    finally:
        c.set_option("workspace_limit_mb", 0)
    Sg.close()


def test_queue_order_does_not_change_the_image(rt, ctx, scenes):
    """Morton-ordered or not, shadow and bounce queues give the same accumulator bit for bit (radiance
    is kept per path slot and summed in a fixed order) and the same ray counts."""
    sc = scenes.load("test-mesh", n_tris=4000, seed=9)
    Sg = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
    never = 0xFFFFFFFF
    out = {}
    try:
        for name, (ext, con) in {"none": (never, never), "default": (never, 1), "both": (1, 0), "late": (2, 2)}.items():
            ctx.set_option("sort_extend_from", ext)
            ctx.set_option("sort_connect_from", con)
            out[name] = Sg.render(sc.camera, sc.light, 160, 90, 4, max_bounces=3)
    finally:
        ctx.set_option("sort_extend_from", never)
        ctx.set_option("sort_connect_from", 1)
    a0, s0 = out["none"]
    for name, (a, s) in out.items():
        assert np.array_equal(a, a0), name
        assert s["total_rays"] == s0["total_rays"], name
    Sg.close()


@pytest.mark.parametrize("which", ["exact", "default"])
def test_refill_threshold_does_not_change_the_image(rt, ctx, ctx_default, scenes, which):
    """How many lanes of a wave may idle before it fetches new rays decides which lane walks which ray
    and nothing else: the accumulator and the ray counts are the same bit for bit for any threshold, in
    the exact forms and in the library's default forms."""
    c = ctx if which == "exact" else ctx_default
    sc = scenes.load("test-mesh", n_tris=4000, seed=9)
    Sg = rt.Scene(c, sc.positions, sc.indices, sc.geom_normals)
    out = []
    for deep, depth0 in ((40, 20), (1, 1), (64, 64), (17, 53)):
        c.set_option("min_active", deep)
        c.set_option("min_active_depth0", depth0)
        out.append(Sg.render(sc.camera, sc.light, 160, 90, 4, max_bounces=3))
    for a, s in out[1:]:
        assert np.array_equal(a, out[0][0])
        assert s["total_rays"] == out[0][1]["total_rays"]
    Sg.close()
