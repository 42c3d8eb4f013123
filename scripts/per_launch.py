#!/usr/bin/env python3
"""Per-launch kernel times and per-depth counters of one render call (debug
aid).  usage: per_launch.py <workload> [spp] [bounces]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
rt = importlib.import_module("rust-raytracer_amd")
scenes = importlib.import_module("rust-raytracer_amd.scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "san-miguel-class"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
bmax = int(sys.argv[3]) if len(sys.argv) > 3 else 2
sc = scenes.load(name)
ctx = rt.Context(0)
S = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
print(name, S.bvh_stats(), S.info())
W, H = 1920, 1080
acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
prev_e = prev_c = None
for b in range(bmax + 1):
    S.render_device(sc.camera, sc.light, acc.data_ptr(), W, H, spp, max_bounces=b, stream=stream, visit_stats=True)
    e, c = ctx.last_stats_split()
    print("  max steps of a single ray up to depth %d: extend %d, connect %d" % ((b,) + ctx.last_max_ray_steps()))
    S.render_device(sc.camera, sc.light, acc.data_ptr(), W, H, spp, max_bounces=b, stream=stream)
    torch.cuda.synchronize()
    times = ctx.last_launch_ms()
    ext = [ms for k, ms in times if k == "extend"]
    con = [ms for k, ms in times if k == "connect"]
    de = {k: e[k] - (prev_e[k] if prev_e else 0) for k in e}
    dc = {k: c[k] - (prev_c[k] if prev_c else 0) for k in c}
    prev_e, prev_c = e, c
    for tag, d, ms in (("extend", de, ext[b]), ("connect", dc, con[b])):
        n = max(1, d["total_rays"])
        print("depth %d %-7s rays %9d  %7.3f ms  %7.1f Mrays/s  nodes/ray %6.2f  tris/ray %6.2f  leaves/ray %5.2f" % (
            b, tag, d["total_rays"], ms, d["total_rays"] / ms / 1e3, d["inner_rays"] / n, d["obj_intersect"] / n,
            d["leaf_rays"] / n))
    if b == bmax:
        for rep in range(4):
            S.render_device(sc.camera, sc.light, acc.data_ptr(), W, H, spp, max_bounces=b, stream=stream,
                            sample_offset=(rep + 1) * spp)
            torch.cuda.synchronize()
            t = ctx.last_launch_ms()
            print("  repeat %d extend" % rep, " ".join("%.2f" % ms for k, ms in t if k == "extend"),
                  " connect", " ".join("%.2f" % ms for k, ms in t if k == "connect"))
        print("other:", " ".join("%.3f" % ms for k, ms in times if k == "other"))
