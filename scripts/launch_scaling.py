"""Kernel time of one closest-hit launch against its ray count (fixed cost per launch vs cost per ray)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("rust-raytracer_amd"); scenes = importlib.import_module("rust-raytracer_amd.scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "sponza-class"
sc = scenes.load(name)
ctx = rt.Context(0)
form = int(os.environ.get("FORM", "3"))
ctx.set_option("traversal_stream", form)
S = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
full = ctx.generate_primary(sc.camera, 1920, 1080, 16)
d_full = torch.from_numpy(full.view(np.uint8)).cuda()
hits = torch.zeros(full.shape[0], dtype=torch.int32, device="cuda")
for frac in (1 / 32768, 1 / 512, 1 / 64, 1 / 16, 1 / 8, 1 / 4, 1 / 2, 1.0):
    n = int(full.shape[0] * frac)
    ts = []
    for rep in range(4):
        rays = d_full.clone()   # the call shrinks far in place
        torch.cuda.synchronize()
        S.intersect_stream_device(rays.data_ptr(), hits.data_ptr(), n)
        torch.cuda.synchronize()
        ms, _ = ctx.last_kernel_ms()
        ts.append(ms["extend"])
    t = min(ts[1:])
    print("n %9d  %.3f ms  %.1f Mrays/s" % (n, t, n / t / 1e3), flush=True)
