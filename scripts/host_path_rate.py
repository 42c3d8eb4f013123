#!/usr/bin/env python3
"""Rate of the host-buffer entry points (rt_intersect_stream / rt_occluded_stream), PCIe copies
included: primary rays of the Sponza-class stand-in in calls of n rays, n = one reference tile
(32,768 rays) up to 8 M."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("rust-raytracer_amd")
scenes = importlib.import_module("rust-raytracer_amd.scenes")

sc = scenes.load("sponza-class")
ctx = rt.Context(0)
scene = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
rays = ctx.generate_primary(sc.camera, 1920, 1080, 4)          # 8.3 M StreamRays
lib, C = rt._lib, rt.C
for n in (32768, 262144, 2097152, 8294400):
    r = np.ascontiguousarray(rays[:n].copy())
    hits = np.empty(n, dtype=np.uint32)
    occ = np.empty(n, dtype=np.uint8)
    for name in ("intersect", "occluded"):
        best = 1e9
        for rep in range(4):
            r[:] = rays[:n]
            t0 = time.perf_counter()
            if name == "intersect":
                rc = lib.rt_intersect_stream(scene.h, rt._p(r), rt._p(hits), n, None)
            else:
                rc = lib.rt_occluded_stream(scene.h, rt._p(r), rt._p(occ), n, None)
            dt = time.perf_counter() - t0
            assert rc == 0
            best = min(best, dt)
        print("%-9s n=%8d  %8.3f ms  %8.1f Mrays/s" % (name, n, best * 1e3, n / best / 1e6), flush=True)
scene.close()
ctx.close()
