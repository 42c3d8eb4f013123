"""Per-launch extend times of the pathological sample range under several settings."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("rust-raytracer_amd"); scenes = importlib.import_module("rust-raytracer_amd.scenes")
sc = scenes.load("san-miguel-class")
ctx = rt.Context(0)
S = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
acc = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
def run(tag, off, n, **opts):
    for k, v in opts.items():
        ctx.set_option(k, v)
    acc.zero_()
    S.render_device(sc.camera, sc.light, acc.data_ptr(), 1920, 1080, n, max_bounces=2, sample_offset=off)
    torch.cuda.synchronize()
    t = ctx.last_launch_ms()
    print(tag, "offset", off, "n", n, "extend", " ".join("%.1f" % ms for k, ms in t if k == "extend"),
          "connect", " ".join("%.1f" % ms for k, ms in t if k == "connect"), flush=True)
run("default", 96, 32)
run("coop64", 96, 32, coop_steps=64)
run("coop0", 96, 32, coop_steps=0)
run("exact", 96, 32, coop_steps=256, traversal_depth0=0, traversal_deep=1)
ctx.set_option("traversal_depth0", 3); ctx.set_option("traversal_deep", 2)
for off in range(96, 128, 4):
    run("scan", off, 4)
