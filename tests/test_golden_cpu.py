"""The oracle reproduces the committed fixtures (tests/golden/make_golden.py)."""
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rays(a, dtype):
    return np.ascontiguousarray(a).view(dtype).reshape(-1)


def test_trace_fixture(O):
    z = np.load(os.path.join(G, "trace_testmesh.npz"))
    S = O.Scene(z["positions"], z["indices"], z["geom_normals"])
    ids, t, st = S.intersect_stream(_rays(z["rays"], O.RAY_DTYPE))
    assert np.array_equal(ids, z["hit_ids"]) and np.array_equal(t.view(np.uint32), z["hit_t"].view(np.uint32))
    assert [st[k] for k in sorted(st)] == z["stats"].tolist()
    occ, so = S.occluded1(_rays(z["shadow_rays"], O.RAY_DTYPE))
    assert np.array_equal(occ, z["occluded"]) and [so[k] for k in sorted(so)] == z["stats_occ"].tolist()
    assert (z["hit_ids"] != O.NONE).sum() > 300 and 100 < z["occluded"].sum() < 1000


def test_render_fixture(O, scenes):
    z = np.load(os.path.join(G, "render_cornell.npz"))
    S = O.Scene(z["positions"], z["indices"], z["geom_normals"])
    c, l = z["camera"], z["light"]
    cam = scenes.Camera(eye=c[0:3], target=c[3:6], up=c[6:9], fov_deg=float(c[9]))

    class L:
        pos = l[0:3]
        intensity = l[3:6]

    W, H, spp, b, ts, seed = [int(x) for x in z["params"]]
    acc, st = S.render(cam, L, W, H, spp, max_bounces=b, tile_size=ts, seed=seed, nthreads=3)
    assert np.array_equal(acc, z["accum"])
    assert [st[k] for k in sorted(st)] == z["stats"].tolist()
    assert np.array_equal(O.resolve(acc), z["rgb"])
