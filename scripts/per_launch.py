#!/usr/bin/env python3
"""Per-launch kernel times of one render call (debug aid): which bounce depth
costs what.  usage: per_launch.py <workload> [spp] [bounces]"""
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
sc = scenes.load(name)
ctx = rt.Context(0)
S = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
W, H = 1920, 1080
acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
for it in range(2):
    st = S.render_device(sc.camera, sc.light, acc.data_ptr(), W, H, spp, max_bounces=b, sample_offset=it * spp,
                         stream=torch.cuda.current_stream().cuda_stream, visit_stats=(it == 1))
    torch.cuda.synchronize()
e, c = ctx.last_stats_split()
print(name, "total", st["total_rays"], "extend rays", e["total_rays"], "connect rays", c["total_rays"])
for k, ms in ctx.last_launch_ms():
    print("  %-8s %8.3f ms" % (k, ms))
