"""Lists rays on which the 4-wide form and the oracle's binary walk disagree."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle as O
from util import adversarial_rays, random_rays
rt = importlib.import_module("rust-raytracer_amd")
scenes = importlib.import_module("rust-raytracer_amd.scenes")
ctx = rt.Context(0)
ctx.set_option("traversal_deep", 2)
ctx.set_option("coop_steps", 0)
np.set_printoptions(precision=9, linewidth=200)
for name, kw in (("test-mesh", {"n_tris": 20000, "seed": 3}), ("sponza-class", {}), ("cornell-class", {})):
    sc = scenes.load(name, **kw)
    So = O.Scene(sc.positions, sc.indices, sc.geom_normals)
    Sg = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
    lo, hi = float(sc.positions.min()), float(sc.positions.max())
    ordinary = np.concatenate([O.generate_primary(sc.camera, 96, 54, 1), random_rays(O.make_rays, 30000, 5, lo=lo, hi=hi)])
    rays = np.concatenate([ordinary, adversarial_rays(O.make_rays, sc.positions)])
    io, to, _ = So.intersect_stream(rays)
    ig, tg, _ = Sg.intersect_stream(rays, stats=False)
    ib, tb = So.intersect_brute(rays)
    diff = np.flatnonzero(tg != to)
    print(name, "rays", rays.size, "ordinary", ordinary.size, "differ", diff.size, "binary!=brute", int((to != tb).sum()), "wide!=brute", int((tg != tb).sum()))
    for i in diff[:40]:
        r = rays[i]
        print("  ray", i, "o", r["origin_far"][:3], "d", r["direction"][:3], "binary", to[i], io[i], "wide", tg[i], ig[i], "brute", tb[i], ib[i])
