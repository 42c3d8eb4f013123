"""rt_intersect_stream_noalloc: the closest-hit entry point that allocates nothing (its ray-fetch
counters and stack overflow live in a fixed device array of the library) against
rt_intersect_stream_device on the same device buffers, bit for bit, on two calls in a row, the second
with a smaller input; its limits are errors.  Collected last: a fault here must not hide the tests
before it."""
import numpy as np
import pytest

from util import random_rays

pytestmark = pytest.mark.gpu


def run_both(rt, Sg, rays, n, ref_form=None, form=None):
    """(hits, rays) of the allocating entry point and of the allocation-free one.  ref_form / form: the
    traversal_stream option of either call when they differ (the allocation-free entry point runs form 2
    where 5 or 7 is selected)."""
    import torch
    host = rays[:n].view(np.float32).reshape(n, 12)
    out = []
    for call in ("ref", "noalloc"):
        d_rays = torch.from_numpy(host.copy()).cuda()
        d_hits = torch.full((n,), 0x7BADBAD, dtype=torch.int32, device="cuda")
        if call == "ref":
            if ref_form is not None:
                Sg.ctx.set_option("traversal_stream", ref_form)
            Sg.intersect_stream_device(d_rays.data_ptr(), d_hits.data_ptr(), n)
        else:
            if form is not None:
                Sg.ctx.set_option("traversal_stream", form)
            Sg.intersect_stream_noalloc(d_rays.data_ptr(), d_hits.data_ptr(), n)
        torch.cuda.synchronize()
        out.append((d_hits.cpu().numpy().view(np.uint32), d_rays.cpu().numpy()))
    return out


@pytest.mark.parametrize("form", [1, 2, 7])   # 7 (like 5) runs as form 2 here
def test_noalloc_matches_the_allocating_entry_point(O, rt, ctx, scenes, form):
    ctx.set_option("traversal_stream", form)
    sc = scenes.load("sponza-class")
    Sg = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
    lo, hi = float(sc.positions.min()), float(sc.positions.max())
    rays = np.concatenate([O.generate_primary(sc.camera, 640, 360, 2),
                           random_rays(O.make_rays, 200000, 31, lo=lo, hi=hi),
                           random_rays(O.make_rays, 100000, 32, lo=lo, hi=hi, near=1e-5, far=0.3 * (hi - lo))])
    for n in (rays.shape[0], 70001):          # the second call is smaller and reuses the scratch as it was left
        # form 7 selected: the allocation-free call runs form 2, so that is what it must equal
        (h_ref, r_ref), (h_na, r_na) = run_both(rt, Sg, rays, n, ref_form=2 if form == 7 else form, form=form)
        assert np.array_equal(h_na, h_ref)
        assert r_na.tobytes() == r_ref.tobytes()          # far distances shrunk identically, nothing else touched
        assert (h_ref != rt.NO_HIT).mean() > 0.5
    if form == 1:                              # and against the oracle itself
        So = O.Scene(sc.positions, sc.indices, sc.geom_normals)
        io, to, _ = So.intersect_stream(rays[:70001])
        assert np.array_equal(io, h_na) and np.array_equal(to, r_na.view(rt.RAY_DTYPE).reshape(-1)["origin_far"][:, 3])
    Sg.close()


def chain_rays(make_rays, pos):
    """Rays along the chain mesh of util.deep_chain_mesh: through every box beside, through and below the
    triangles, from several starting points and back from the far end."""
    o, d = [], []
    xs = pos[::3, 0]
    for k in range(0, len(xs), 5):
        x0 = 0.0 if k == 0 else 0.5 * float(xs[k - 1] + xs[k])
        for dy in (0.5, 0.56, 0.7, 0.3):
            o.append([x0, x0 * dy, x0 * 0.013])
            d.append([1.0, dy, 0.013])
    far_end = float(xs[-1]) * 2.0
    for dy in (0.5, 0.7):
        o.append([far_end, far_end * dy, far_end * 0.013])
        d.append([-1.0, -dy, -0.013])
    d = np.array(d, np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return make_rays(np.array(o, np.float32), d.astype(np.float32))


@pytest.mark.parametrize("form", [0, 1, 3, 4, 6])
def test_noalloc_overflow_area(O, rt, ctx, form):
    """A chain mesh whose rays defer more nodes than a lane keeps in LDS (24) but no more than the fixed
    scratch holds per lane (56): the part of the scratch array behind the fetch counters -- column and
    stride addressing at the entry point's fixed grid -- against the allocating entry point, bit for bit,
    for every closest-hit form the entry point can run."""
    from util import deep_chain_mesh
    # binary depth 26 at n = 120 (27 deferred nodes at most); the 4-wide forms defer up to three children per
    # level and are within the scratch at n = 80 (wide depth 17)
    pos, idx, nrm = deep_chain_mesh(n=120 if form in (0, 1, 4) else 80)
    Sg = rt.Scene(ctx, pos, idx, nrm)
    rays = np.concatenate([chain_rays(O.make_rays, pos), random_rays(O.make_rays, 3000, 7, lo=-2.0, hi=50.0)])
    if form in (0, 1, 4):   # the binary forms report the deepest stack a ray held
        ctx.set_option("traversal_stream", form)
        Sg.intersect_stream(rays)
        assert ctx.last_max_stack_entries()[0] > rt.LDS_STACK_ENTRIES
    ctx.set_option("traversal_stream", form)
    for n in (rays.shape[0], 1501):
        (h_ref, r_ref), (h_na, r_na) = run_both(rt, Sg, rays, n)
        assert np.array_equal(h_na, h_ref), form
        assert r_na.tobytes() == r_ref.tobytes(), form
    assert (h_ref != rt.NO_HIT).sum() > 10
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
    ctx.set_option("traversal_stream", 2)     # the bound is that of the selected form: here 3 x wide depth + 1
    assert lib.rt_intersect_stream_noalloc(deep.h, vp(buf.data_ptr()), vp(buf.data_ptr()), 1, None) == 4
    assert b"defer" in lib.rt_last_error()
    ctx.set_option("traversal_stream", 1)     # binary depth 37: 38 deferred nodes fit
    assert lib.rt_intersect_stream_noalloc(deep.h, vp(buf.data_ptr()), vp(buf.data_ptr()), 0, None) == 0
    deep.close()
