#!/usr/bin/env python3
"""Per-depth extend/connect kernel times over several frames (min / median),
to separate bulk throughput from straggler tails.  usage: depth_times.py <workload> [spp] [bounces] [repeats]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
rt = importlib.import_module("rust-raytracer_amd")
scenes = importlib.import_module("rust-raytracer_amd.scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "san-miguel-class"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
b = int(sys.argv[3]) if len(sys.argv) > 3 else 2
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
sc = scenes.load(name)
ctx = rt.Context(0)
S = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
W, H = 1920, 1080
acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
ext, con, oth, tot = [], [], [], []
import time
for rep in range(reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = S.render_device(sc.camera, sc.light, acc.data_ptr(), W, H, spp, max_bounces=b, stream=stream, sample_offset=rep * spp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    if rep == 0:
        continue
    t = ctx.last_launch_ms()
    ext.append([ms for k, ms in t if k == "extend"])
    con.append([ms for k, ms in t if k == "connect"])
    oth.append(sum(ms for k, ms in t if k == "other"))
    tot.append((dt, st["total_rays"]))
ext, con = np.array(ext), np.array(con)
env = {k: v for k, v in os.environ.items() if k.startswith("RTCORE_")}
print(name, "spp", spp, "b", b, env)
print("  extend  min", np.round(ext.min(0), 2), "median", np.round(np.median(ext, 0), 2), "max", np.round(ext.max(0), 2))
print("  connect min", np.round(con.min(0), 2), "median", np.round(np.median(con, 0), 2), "max", np.round(con.max(0), 2))
print("  other   median %.2f ms;  frame wall min %.2f median %.2f ms;  Mrays/s best %.0f median %.0f" % (
    np.median(oth), min(d for d, _ in tot), np.median([d for d, _ in tot]),
    max(r / d / 1e3 for d, r in tot), np.median([r / d / 1e3 for d, r in tot])))
