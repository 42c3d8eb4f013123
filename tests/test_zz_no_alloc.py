"""rt_intersect_stream_noalloc: the closest-hit entry point that allocates nothing (its ray-fetch
counters and stack overflow live in a fixed device array of the library) against
rt_intersect_stream_device on the same device buffers, bit for bit, on two calls in a row, the second
with a smaller input; its limits are errors.  Collected last: a fault here must not hide the tests
before it."""
import numpy as np
import pytest

from util import random_rays

pytestmark = pytest.mark.gpu


def run_both(rt, Sg, rays, n):
    import torch
    host = rays[:n].view(np.float32).reshape(n, 12)
    out = []
    for call in ("ref", "noalloc"):
        d_rays = torch.from_numpy(host.copy()).cuda()
        d_hits = torch.full((n,), 0x7BADBAD, dtype=torch.int32, device="cuda")
        if call == "ref":
            Sg.intersect_stream_device(d_rays.data_ptr(), d_hits.data_ptr(), n)
        else:
            Sg.intersect_stream_noalloc(d_rays.data_ptr(), d_hits.data_ptr(), n)
        torch.cuda.synchronize()
        out.append((d_hits.cpu().numpy().view(np.uint32), d_rays.cpu().numpy()))
    return out


@pytest.mark.parametrize("form", [1, 2])
def test_noalloc_matches_the_allocating_entry_point(O, rt, ctx, scenes, form):
    ctx.set_option("traversal_stream", form)
    sc = scenes.load("sponza-class")
    Sg = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
    lo, hi = float(sc.positions.min()), float(sc.positions.max())
    rays = np.concatenate([O.generate_primary(sc.camera, 640, 360, 2),
                           random_rays(O.make_rays, 200000, 31, lo=lo, hi=hi),
                           random_rays(O.make_rays, 100000, 32, lo=lo, hi=hi, near=1e-5, far=0.3 * (hi - lo))])
    for n in (rays.shape[0], 70001):          # the second call is smaller and reuses the scratch as it was left
        (h_ref, r_ref), (h_na, r_na) = run_both(rt, Sg, rays, n)
        assert np.array_equal(h_na, h_ref)
        assert r_na.tobytes() == r_ref.tobytes()          # far distances shrunk identically, nothing else touched
        assert (h_ref != rt.NO_HIT).mean() > 0.5
    if form == 1:                              # and against the oracle itself
        So = O.Scene(sc.positions, sc.indices, sc.geom_normals)
        io, to, _ = So.intersect_stream(rays[:70001])
        assert np.array_equal(io, h_na) and np.array_equal(to, r_na.view(rt.RAY_DTYPE).reshape(-1)["origin_far"][:, 3])
    Sg.close()


def test_noalloc_limits(rt, ctx):
    import torch
    from util import deep_chain_mesh
    C, lib = rt.C, rt._lib
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    Sg = rt.Scene(ctx, pos, np.array([[0, 1, 2]], np.uint32), np.array([[0, 0, 1]], np.float32))
    buf = torch.zeros(64, dtype=torch.float32, device="cuda")
    vp = C.c_void_p
    assert lib.rt_intersect_stream_noalloc(Sg.h, vp(buf.data_ptr()), vp(buf.data_ptr()), (1 << 28) + 1, None) == 4
    assert b"limit" in lib.rt_last_error()
    assert lib.rt_intersect_stream_noalloc(Sg.h, None, None, 5, None) == 1
    assert lib.rt_intersect_stream_noalloc(Sg.h, None, None, 0, None) == 0
    Sg.close()
    # a tree whose rays can defer more nodes than the fixed scratch holds per lane
    p, i, nrm = deep_chain_mesh()   # binary depth 37, wide depth 25: up to 76 deferred nodes
    deep = rt.Scene(ctx, p, i, nrm)
    assert deep.bvh_stats()["max_depth"] >= 30
    assert lib.rt_intersect_stream_noalloc(deep.h, vp(buf.data_ptr()), vp(buf.data_ptr()), 1, None) == 4
    assert b"defer" in lib.rt_last_error()
    deep.close()
