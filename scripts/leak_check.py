#!/usr/bin/env python3
"""Creates and destroys the dragon-class scene five times and prints the time of rt_scene_create and the free
device memory with the scene alive and after rt_scene_destroy (it must come back to the same value).  Needs a GPU."""
import importlib
import sys
import time

sys.path.insert(0, '.')
import torch

rt = importlib.import_module("rust-raytracer_amd")
scenes = importlib.import_module("rust-raytracer_amd.scenes")
sc = scenes.load("dragon-class")
ctx = rt.Context(0)
for i in range(5):
    t0 = time.time()
    S = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
    dt = time.time() - t0
    free1, total = torch.cuda.mem_get_info()
    S.close()
    free2, _ = torch.cuda.mem_get_info()
    print("round %d: create %.2f s, free with the scene %.2f GB, after destroy %.2f GB" % (i, dt, free1 / 1e9, free2 / 1e9),
          flush=True)
