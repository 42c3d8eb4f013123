"""The SSE/AVX2 restatement of the reference's traversals (oracle/rt_oracle_simd.c, the reference's
own intrinsics statement by statement) against the scalar restatement (oracle/rt_oracle.c).

Both are restatements, so parity with the reference stays unpinned; what this pins is the scalar
code's emulation of x86 behaviour: the second-operand-on-NaN rule of max_ps/min_ps, the sign-bit
lanes of blendv, the single rounding of fmsub and the pairing of rays in the stream filter.  Hit
ids, distance bits and all seven TraceStats counters must be equal on every ray set, the
adversarial one (axis-parallel rays, origins on vertices: NaN slabs) included."""
import numpy as np
import pytest

from util import adversarial_rays, random_rays, surface_rays

COUNTERS = ("total_rays", "inner_visit", "inner_rays", "leaf_visit", "leaf_rays", "leaf_objs", "obj_intersect")


def ray_sets(O, sc, S):
    prim = O.generate_primary(sc.camera, 64, 36, 2)
    rnd = random_rays(O.make_rays, 3000, 1)
    adv = adversarial_rays(O.make_rays, sc.positions)
    shadow = random_rays(O.make_rays, 2000, 5, near=1e-5, far=0.7)
    ids, t, _ = S.intersect_stream(prim)
    bounce = surface_rays(O.make_rays, prim, ids, t, seed=9)
    odd = rnd[:4097]  # an odd count: the last ray of a node's list is paired with itself
    return (("primary", prim), ("random", rnd), ("adversarial", adv), ("finite far", shadow),
            ("bounce-like", bounce), ("odd count", odd), ("one ray", rnd[:1]), ("none", rnd[:0]))


@pytest.mark.parametrize("name,kw", [("cornell-class", {}), ("test-mesh", {}), ("test-mesh", {"n_tris": 4000, "seed": 11}),
                                     ("sponza-class", {})])
def test_simd_traversals_equal_scalar(O, scenes, name, kw):
    sc = scenes.load(name, **kw)
    S = O.Scene(sc.positions, sc.indices, sc.geom_normals)
    for what, rays in ray_sets(O, sc, S):
        for scalar, simd in ((S.intersect1, S.simd_intersect1), (S.intersect_stream, S.simd_intersect_stream)):
            ia, ta, sa = scalar(rays)
            ib, tb, sb = simd(rays)
            assert np.array_equal(ia, ib), (what, scalar.__name__)
            assert np.array_equal(ta.view(np.uint32), tb.view(np.uint32)), (what, scalar.__name__)
            for k in COUNTERS:
                assert sa[k] == sb[k], (what, scalar.__name__, k, sa[k], sb[k])
        for scalar, simd in ((S.occluded1, S.simd_occluded1), (S.occluded_stream, S.simd_occluded_stream)):
            oa, sa = scalar(rays)
            ob, sb = simd(rays)
            assert np.array_equal(oa, ob), (what, scalar.__name__)
            for k in COUNTERS:
                assert sa[k] == sb[k], (what, scalar.__name__, k, sa[k], sb[k])


def test_simd_nan_slabs_are_exercised(O, scenes):
    """The adversarial set really produces NaN slab distances (plane * inf - origin * inf with both
    products infinite of one sign), i.e. the equality above covers the case it is there for."""
    sc = scenes.load("test-mesh")
    adv = adversarial_rays(O.make_rays, sc.positions)
    rd = adv["dir_recip_near"][:, :3]
    o = adv["origin_far"][:, :3]
    lo, hi = sc.positions.min(axis=0), sc.positions.max(axis=0)
    with np.errstate(invalid="ignore", over="ignore"):
        t = np.stack([lo * rd - o * rd, hi * rd - o * rd])
    assert np.isinf(rd).any() and np.isnan(t).any()


@pytest.mark.parametrize("mode", ["MODE_STREAM", "MODE_STREAM_SHADOW_IMMEDIATE", "MODE_SINGLE_RAY"])
def test_simd_render_equals_scalar(O, scenes, mode):
    sc = scenes.load("test-mesh", n_tris=3000, seed=5)
    S = O.Scene(sc.positions, sc.indices, sc.geom_normals)
    m = getattr(O, mode)
    a, sa = S.render(sc.camera, sc.light, 40, 24, 3, max_bounces=2, mode=m, nthreads=2)
    b, sb = S.render(sc.camera, sc.light, 40, 24, 3, max_bounces=2, mode=m | O.MODE_SIMD, nthreads=2)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for k in COUNTERS:
        assert sa[k] == sb[k], (k, sa[k], sb[k])
