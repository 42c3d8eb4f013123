"""Known-answer and self-consistency tests of the CPU oracle.

The reference ships no tests or golden vectors, so the oracle is pinned by
analytic answers, by a brute-force all-triangles loop that shares only the
triangle arithmetic with the traversals, and by the committed fixtures."""
import math

import numpy as np
import pytest

from util import adversarial_rays, assert_hits_match, random_rays

INF = float("inf")
V0, E1, E2 = (0, 0, 0), (1, 0, 0), (0, 1, 0)


def test_triangle_known_answers(O):
    down = (0, 0, -1)
    assert O.tri_intersect(V0, E1, E2, (0.25, 0.25, 1), down) == 1.0          # interior
    assert O.tri_intersect(V0, E1, E2, (0.25, 0.25, 3.5), down) == 3.5
    for p in [(0, 0, 1), (1, 0, 1), (0, 1, 1), (0.5, 0, 1), (0, 0.5, 1), (0.5, 0.5, 1)]:
        assert O.tri_intersect(V0, E1, E2, p, down) == 1.0, p                   # vertices and edges are inclusive
    for p in [(1.001, 0, 1), (-0.001, 0.5, 1), (0.6, 0.6, 1), (0.5, -1e-4, 1)]:
        assert O.tri_intersect(V0, E1, E2, p, down) == INF, p                   # outside
    assert O.tri_intersect(V0, E1, E2, (0.25, 0.25, 1), (1, 0, 0)) == INF       # parallel: |det| < eps
    assert O.tri_intersect(V0, E1, E2, (0.25, 0.25, -1), down) == -1.0          # behind: negative t is returned
    assert O.tri_intersect(V0, E1, E2, (0.25, 0.25, -1), (0, 0, 1)) == 1.0      # no back-face culling
    # |det| just under / over f32::EPSILON (src/triangle_opt.rs:23)
    eps = np.float32(1.1920929e-7)
    assert O.tri_intersect(V0, (float(eps) * 0.99, 0, 0), (0, 1, 0), (0, 0.5, 1), down) == INF
    assert O.tri_intersect(V0, (float(eps) * 1.01, 0, 0), (0, 1, 0), (0, 0.5, 1), down) == 1.0


def _one_tri_scene(O):
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    return O.Scene(pos, np.array([[0, 1, 2]], np.uint32), np.array([[0, 0, 1]], np.float32))


def test_accept_rule_half_open(O):
    """near <= t < far (src/bvh/intersect1.rs:198): t == near hits, t == far does not."""
    S = _one_tri_scene(O)
    o = np.array([[0.25, 0.25, 1.0]] * 4, np.float32)
    d = np.array([[0, 0, -1.0]] * 4, np.float32)
    rays = O.make_rays(o, d, near=np.array([0, 1.0, 1.0000001, 0], np.float32),
                       far=np.array([np.inf, np.inf, np.inf, 1.0], np.float32))
    for fn in (S.intersect1, S.intersect_stream):
        ids, t, _ = fn(rays)
        assert ids.tolist() == [0, 0, O.NONE, O.NONE]
        assert t[0] == 1.0 and t[1] == 1.0 and np.isinf(t[2]) and t[3] == 1.0
    occ, _ = S.occluded1(rays)
    assert occ.tolist() == [1, 1, 0, 0]
    # a single-leaf tree: root is a leaf reference
    ref, cnt = S.root()
    assert ref == 0x80000000 and cnt == 1 and S.num_inner() == 0


@pytest.mark.parametrize("name,kw", [("cornell-class", {}), ("test-mesh", {}), ("test-mesh", {"n_tris": 4000, "seed": 11})])
def test_traversals_agree_with_brute_force(O, scenes, name, kw):
    sc = scenes.load(name, **kw)
    S = O.Scene(sc.positions, sc.indices, sc.geom_normals)
    rays = np.concatenate([random_rays(O.make_rays, 3000, 1), O.generate_primary(sc.camera, 48, 27, 1)])
    ib, tb = S.intersect_brute(rays)
    i1, t1, st1 = S.intersect1(rays)
    is_, ts, sts = S.intersect_stream(rays)
    assert_hits_match(ib, tb, i1, t1, "brute vs intersect1")
    assert_hits_match(ib, tb, is_, ts, "brute vs intersect_stream")
    # Adversarial rays (axis-parallel, origins exactly on vertices).  The slab
    # test rounds o*(1/d) before the fused multiply-subtract, so a ray whose
    # origin lies exactly on a box face can miss that box: the traversals may
    # lose a hit at distance 0 that the all-triangles loop finds.  Everything
    # else must agree exactly.
    adv = adversarial_rays(O.make_rays, sc.positions)
    ab, atb = S.intersect_brute(adv)
    for fn in (S.intersect1, S.intersect_stream):
        ai, at, _ = fn(adv)
        differs = ~(at == atb)
        assert (np.abs(atb[differs]) < 1e-6).all(), "a mismatch away from distance 0"
        assert differs.mean() < 0.05
    assert (ib != O.NONE).mean() > 0.2
    # any-hit == (closest hit exists) for the same [near, far)
    o1, so1 = S.occluded1(rays)
    os_, sos = S.occluded_stream(rays)
    assert np.array_equal(o1.astype(bool), ib != O.NONE)
    assert np.array_equal(os_.astype(bool), ib != O.NONE)
    # counters: every traversal counts each submitted ray once
    for st in (st1, sts, so1, sos):
        assert st["total_rays"] == rays.shape[0]
    # the stream traversal culls with the shrinking far, the single-ray one does not
    assert sts["inner_rays"] <= st1["inner_rays"]
    # shadow-style rays with finite far
    sh = random_rays(O.make_rays, 2000, 5, near=1e-5, far=0.7)
    ib, tb = S.intersect_brute(sh)
    o1, _ = S.occluded1(sh)
    assert np.array_equal(o1.astype(bool), ib != O.NONE)


def test_bvh_invariants(O, scenes):
    sc = scenes.load("test-mesh", n_tris=3000, seed=5)
    S = O.Scene(sc.positions, sc.indices, sc.geom_normals)
    nodes, tris = S.flatten()
    st = S.bvh_stats()
    nt = sc.num_triangles
    assert st["num_objs"] == nt and st["max_objs"] <= 8 and nodes.shape[0] == st["num_leaves"] - 1
    assert sorted(tris["id"].tolist()) == list(range(nt))          # every triangle in exactly one leaf slot
    pos = sc.positions[sc.indices.astype(np.int64)]               # [nt,3,3]
    tri_lo, tri_hi = pos.min(1), pos.max(1)
    covered = np.zeros(nt, bool)

    def walk(ref, count, lo, hi, depth):
        if ref & 0x80000000:
            first = ref & 0x7FFFFFFF
            assert 1 <= count <= 8
            ids = tris["id"][first:first + count]
            assert not covered[first:first + count].any()
            covered[first:first + count] = True
            assert (tri_lo[ids] >= lo - 0).all() and (tri_hi[ids] <= hi + 0).all()   # child box encloses its triangles
            return depth
        n = nodes[ref]
        b = np.stack([n["bx"], n["by"], n["bz"]])                  # [axis, lane]
        dl = walk(int(n["left"]), int(n["left_count"]), b[:, 0], b[:, 2], depth + 1)
        dr = walk(int(n["right"]), int(n["right_count"]), b[:, 1], b[:, 3], depth + 1)
        return max(dl, dr)

    big = np.full(3, np.inf, np.float32)
    assert walk(0, 0, -big, big, 0) == st["max_depth"]
    assert covered.all()


def test_bvh_stats_snapshot(O, scenes):
    """Tree shape of the stand-in scenes (guards the builder against drift)."""
    snap = {"cornell-class": (30, {"num_leaves": 7, "num_objs": 30, "max_depth": 4, "max_objs": 8}),
            "sponza-class": (262640, {"num_leaves": 44744, "num_objs": 262640, "max_depth": 23, "max_objs": 8})}
    for name, (nt, want) in snap.items():
        sc = scenes.load(name)
        assert sc.num_triangles == nt
        assert O.Scene(sc.positions, sc.indices, sc.geom_normals).bvh_stats() == want


def test_degenerate_centroids_make_one_leaf(O):
    """Zero centroid extent -> a leaf of any size (src/bvh/build.rs:93-101)."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    idx = np.tile(np.array([[0, 1, 2]], np.uint32), (20, 1))
    S = O.Scene(pos, idx, np.tile(np.array([[0, 0, 1]], np.float32), (20, 1)))
    assert S.bvh_stats() == {"num_leaves": 1, "num_objs": 20, "max_depth": 0, "max_objs": 20}
    rays = O.make_rays([[0.2, 0.2, 1]], [[0, 0, -1]])
    ids, t, st = S.intersect_stream(rays)
    assert ids[0] == 0 and t[0] == 1.0 and st["obj_intersect"] == 20   # first found wins on exact ties (strict <)


def test_sincos_and_rng(O):
    us = np.linspace(0, 1, 4001, endpoint=False)
    err = 0.0
    for u in us:
        s, c = O.sincos_2pi(float(np.float32(u)))
        b = float(np.float32(2 * 3.14159274) * np.float32(u))
        err = max(err, abs(s - math.sin(b)), abs(c - math.cos(b)))
    assert err < 4e-7
    x = np.array([O.rng_f32(1235468, p, s, d) for p in range(40) for s in range(10) for d in range(4)])
    assert x.min() >= 0.0 and x.max() < 1.0 and abs(x.mean() - 0.5) < 0.03 and len(set(x.tolist())) > 1590
    assert O.rng_f32(1, 2, 3, 4) == O.rng_f32(1, 2, 3, 4) != O.rng_f32(1 << 32 | 1, 2, 3, 4)


def test_sample_light_known_answer(O, scenes):
    L = scenes.Light(pos=(0, 0, 2), power=10.0)
    ok, wi, dist, w = O.sample_light((0, 0, 1), (0, 0, 1), (0, 0, 0), L)
    assert ok and wi.tolist() == [0, 0, 1] and dist == 2.0
    want = np.float32(10.0) * (np.float32(0.8) / np.float32(3.14159274)) * np.float32(1.0) / np.float32(4.0)
    assert np.allclose(w, want, rtol=1e-7)
    # light and viewer on opposite sides of the surface: rejected (src/integrators/common.rs:45)
    ok, *_ = O.sample_light((0, 0, -1), (0, 0, 1), (0, 0, 0), L)
    assert not ok
    # both below: accepted (two-sided)
    ok, *_ = O.sample_light((0, 0, -1), (0, 0, 1), (0, 0, 4), L)
    assert ok


def test_sample_diffuse_ray_properties(O):
    rng = np.random.default_rng(0)
    for _ in range(300):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        wo = rng.normal(size=3)
        wo /= np.linalg.norm(wo)
        u0, u1 = rng.random(), rng.random()
        wi, w = O.sample_diffuse_ray(wo, n, float(np.float32(u0)), float(np.float32(u1)))
        assert abs(np.linalg.norm(wi) - 1) < 1e-5
        assert np.dot(wi, n) * np.dot(wo, n) > -1e-6           # sampled on wo's side of the surface
        assert np.allclose(w, 0.8, rtol=2e-5)                    # f*cos/pdf = albedo for a cosine-sampled Lambertian
    for n in ([0, 0, 1], [0, 0, -1]):                            # identity and the dot == -1 branch of src/math.rs:53-55
        wi, w = O.sample_diffuse_ray([0, 0, 1.0] if n[2] > 0 else [0, 0, -1.0], n, 0.25, 0.5)
        assert abs(np.linalg.norm(wi) - 1) < 1e-5 and wi[2] * n[2] > 0


def test_camera(O, scenes):
    cam = scenes.Camera(eye=(0, -3, 0), target=(0, 0, 0), up=(0, 0, 1), fov_deg=60)
    W, H = 65, 33
    c = O.camera_sample_dir(cam, W, H, 16, 32, 16, 0.5, 0.5)      # centre pixel, no jitter
    assert np.allclose(c, [0, 1, 0], atol=1e-6)
    l = O.camera_sample_dir(cam, W, H, 16, 0, 16, 0.5, 0.5)       # left edge: half the horizontal fov
    assert abs(math.degrees(math.atan2(-l[0], l[1])) - 30) < 1e-3
    # the x walk is incremental within a tile, so the tile size changes low-order bits only
    a = O.camera_sample_dir(cam, W, H, 16, 47, 5, 0.3, 0.8)
    b = O.camera_sample_dir(cam, W, H, 64, 47, 5, 0.3, 0.8)
    assert np.allclose(a, b, atol=1e-6)
    rays = O.generate_primary(cam, W, H, 2)
    assert rays.shape[0] == W * H * 2 and (rays["origin_far"][:, 3] == np.inf).all()
    assert np.allclose(rays["dir_recip_near"][:, :3] * rays["direction"][:, :3], 1, atol=1e-6)


def test_render_modes_and_sharding(O, scenes):
    sc = scenes.load("cornell-class")
    S = O.Scene(sc.positions, sc.indices, sc.geom_normals)
    W, H, spp = 48, 32, 4
    img, st = S.render(sc.camera, sc.light, W, H, spp, max_bounces=2, nthreads=2)
    assert (img[..., 3] == spp).all() and img[..., :3].min() >= 0 and img[..., :3].max() > 0
    # same estimator, same random numbers -> the three schedulings agree to rounding
    img_s, st_s = S.render(sc.camera, sc.light, W, H, spp, max_bounces=2, mode=O.MODE_STREAM, nthreads=1)
    img_1, st_1 = S.render(sc.camera, sc.light, W, H, spp, max_bounces=2, mode=O.MODE_SINGLE_RAY, nthreads=1)
    assert np.allclose(img, img_s, rtol=1e-5, atol=1e-6) and np.allclose(img, img_1, rtol=1e-5, atol=1e-6)
    assert st["total_rays"] == st_s["total_rays"] == st_1["total_rays"]
    # TraceStats invariant: total = primary + bounce + shadow rays; with b=0 at most 2 per sample
    _, st0 = S.render(sc.camera, sc.light, W, H, spp, max_bounces=0, nthreads=2)
    assert W * H * spp <= st0["total_rays"] <= 2 * W * H * spp
    # tile shards add up to the frame bit for bit; sample ranges to rounding
    acc = np.zeros((H, W, 4), np.float32)
    for r in range(3):
        S.render(sc.camera, sc.light, W, H, spp, shard_index=r, shard_count=3, accum=acc, nthreads=2)
    assert np.array_equal(acc, img)
    acc = np.zeros((H, W, 4), np.float32)
    S.render(sc.camera, sc.light, W, H, 1, sample_offset=0, accum=acc, nthreads=2)
    S.render(sc.camera, sc.light, W, H, 3, sample_offset=1, accum=acc, nthreads=2)
    assert np.allclose(acc, img, rtol=1e-5, atol=1e-6)
    rgb = O.resolve(img)
    assert rgb.shape == (H, W, 3) and rgb.max() > 20


def test_product_builder_matches_oracle(O, rt, scenes):
    """Host logic of the product (its own BVH builder) against the oracle's
    tree, byte for byte, sequential and threaded."""
    for name, kw in (("cornell-class", {}), ("test-mesh", {}), ("sponza-class", {})):
        sc = scenes.load(name, **kw)
        S = O.Scene(sc.positions, sc.indices, sc.geom_normals)
        on, ot = S.flatten()
        for th in (1, 4):
            nodes, order, info, st = rt.build_bvh_host(sc.positions, sc.indices, threads=th)
            assert nodes.tobytes() == on.tobytes()
            assert np.array_equal(order, ot["id"])
            assert st == S.bvh_stats()
            assert (info["root_ref"], info["root_count"]) == S.root()
