"""Shared helpers for the parity tests."""
import numpy as np


def random_rays(make_rays, n, seed, lo=-1.5, hi=1.5, near=0.0, far=np.inf):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return make_rays(o, d.astype(np.float32), near=near, far=far)


def adversarial_rays(make_rays, positions, seed=3):
    """Axis-parallel directions (zero components -> infinite reciprocals),
    origins on mesh vertices (box faces) and negative zeros: the NaN slab cases."""
    rng = np.random.default_rng(seed)
    pos = np.asarray(positions, dtype=np.float32)
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1],
                     [1, 1, 0], [0, 1, -1], [-1, 0, 1]], dtype=np.float32)
    axes = axes / np.linalg.norm(axes, axis=1, keepdims=True)
    pick = pos[rng.integers(0, pos.shape[0], size=64)]
    o = []
    d = []
    for a in axes:
        for p in pick:
            o.append(p - a * np.float32(2.0))
            d.append(a)
            o.append(p)
            d.append(a)
    d = np.array(d, dtype=np.float32)
    d[::7] = np.where(d[::7] == 0, np.float32(-0.0), d[::7])
    return make_rays(np.array(o, dtype=np.float32), d)


def assert_hits_match(ids_a, t_a, ids_b, t_b, what=""):
    """Closest-hit parity rule: distances equal as f32 values (no tolerance); ids
    equal except where two traversal orders met an exact-t tie.  Returns the
    number of such ties."""
    assert t_a.dtype == np.float32 and t_b.dtype == np.float32
    same_t = t_a == t_b  # value equality: +0 and -0 are the same distance (a tie at the origin)
    assert same_t.all(), "%s: %d of %d hit distances differ" % (what, (~same_t).sum(), same_t.size)
    none_a = ids_a == 0xFFFFFFFF
    none_b = ids_b == 0xFFFFFFFF
    assert np.array_equal(none_a, none_b), "%s: hit/miss differs" % what
    diff = ids_a != ids_b
    return int(diff.sum())  # all of these are exact-t ties by the check above


def surface_rays(make_rays, rays, ids, t, seed, near=1e-5, far=np.inf):
    """Bounce-like rays: from the hit points of `rays` (ids/t = their closest hits) in
    uniformly random directions, with the reference's SHADOW_RAY_NEAR offset."""
    rng = np.random.default_rng(seed)
    hit = ids != 0xFFFFFFFF
    o = rays["origin_far"][hit, :3] + rays["direction"][hit, :3] * t[hit, None]
    d = rng.normal(size=o.shape)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return make_rays(o.astype(np.float32), d.astype(np.float32), near=near, far=far)


def deep_chain_mesh(n=190, ratio=1.5, first=1e-17):
    """A row of n triangles in the planes x = first * ratio^i, each ratio times larger than the one
    before (1e-17 .. 1e16: surface areas stay inside the f32 range).  Binned SAH peels the few
    triangles of the last buckets off the far end per level, so the tree is a chain about n/5 levels
    deep with the long side on the left (smaller x), and a ray from the origin along (1, 0.5, 0.013)
    passes through every box beside every triangle: it defers one sibling per level, more than a
    traversal stack holds in LDS.  (Triangles below x ~ 1e-4 have |det| < f32::EPSILON and are never
    hit, src/triangle_opt.rs:24-26; their boxes are walked all the same.)"""
    pos, nrm = [], []
    for i in range(n):
        x = np.float32(first) * np.float32(ratio) ** np.float32(i)
        pos += [[x, 0.1 * x, -x], [x, x, -x], [x, x, x]]
        nrm.append([-1.0, 0.0, 0.0])
    pos = np.array(pos, np.float32)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    return pos, idx, np.array(nrm, np.float32)
