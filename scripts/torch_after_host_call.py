"""Probe: does torch still initialise its GPU state after a host-buffer trace call of n rays?"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1])
import torch
rt = importlib.import_module("rust-raytracer_amd")
scenes = importlib.import_module("rust-raytracer_amd.scenes")
sc = scenes.load("test-mesh", n_tris=20000, seed=3)
ctx = rt.Context(0)
Sg = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
rays = ctx.generate_primary(sc.camera, 640, 360, 10)[:n]
r = np.ascontiguousarray(rays.copy())
hits = np.zeros(n, dtype=np.uint32)
st = rt.TraceStats()
use_stats = len(sys.argv) > 2
assert rt._lib.rt_intersect_stream(Sg.h, rt._p(r), rt._p(hits), n, rt.C.byref(st) if use_stats else None) == 0
print("n", n, "stats", use_stats, "hits", int((hits != rt.NO_HIT).sum()), flush=True)
x = torch.zeros(4, device="cuda")
print("torch ok", x.sum().item(), flush=True)
