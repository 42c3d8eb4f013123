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
