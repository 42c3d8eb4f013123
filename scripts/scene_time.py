import importlib, sys, time
sys.path.insert(0, '.')
rt = importlib.import_module("rust-raytracer_amd")
scenes = importlib.import_module("rust-raytracer_amd.scenes")
t0=time.time(); sc = scenes.load("san-miguel-class"); print("generate %.1f s" % (time.time()-t0), flush=True)
ctx = rt.Context(0)
t0=time.time(); S = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals); print("rt.Scene %.2f s" % (time.time()-t0))
