import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("rust-raytracer_amd"); scenes = importlib.import_module("rust-raytracer_amd.scenes")
sc = scenes.load("sponza-class")
ctx = rt.Context(0)
S = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
acc = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
for fuse in (0, 1):
    ctx.set_option("fuse_primary", fuse)
    for spp in (1, 4, 8, 16, 32):
        for rep in range(2):
            acc.zero_()
            S.render_device(sc.camera, sc.light, acc.data_ptr(), 1920, 1080, spp, max_bounces=0)
            torch.cuda.synchronize()
        t = ctx.last_launch_ms()
        print("fuse", fuse, "spp", spp, " ".join("%s %.2f" % (k, ms) for k, ms in t), flush=True)
