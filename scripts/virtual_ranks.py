#!/usr/bin/env python3
"""Load balance of the one-node split, measured on ONE GPU: every rank's share of a frame rendered
in turn (rt_node_render's plan without a communicator), timed on its own.

    python scripts/virtual_ranks.py [--world 8] [--workload san-miguel-class] [--spp 128]

A node of `world` GPUs finishes a frame when its slowest rank does, so the predicted efficiency of a
split is mean(rank time) / max(rank time); the sum-reduce of the accumulation buffer (33 MB at 1080p)
comes on top and is a fraction of a millisecond over xGMI.  Prints one JSON line."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--workload", default="san-miguel-class")
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--bounces", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    import torch
    rt = importlib.import_module("rust-raytracer_amd")
    scenes = importlib.import_module("rust-raytracer_amd.scenes")
    node = importlib.import_module("rust-raytracer_amd.node_render")
    sc = scenes.load(args.workload)
    ctx = rt.Context(0)
    scene = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
    W, H = args.width, args.height
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def frame(plan, offset):
        t0 = time.perf_counter()
        st = scene.render_device(sc.camera, sc.light, acc.data_ptr(), W, H, plan["samples"], max_bounces=args.bounces,
                                 sample_offset=plan["sample_offset"] + offset, shard_index=plan["shard_index"],
                                 shard_count=plan["shard_count"], stream=stream, stats=True)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, st["total_rays"]

    out = {"workload": args.workload, "world": args.world, "spp": args.spp, "size": [W, H], "bounces": args.bounces}
    whole = node.shard_plan(0, 1, "tile", args.spp)
    frame(whole, 0)
    t1 = min(frame(whole, args.spp * (i + 1))[0] for i in range(args.repeat))
    out["one_gpu_ms"] = round(t1 * 1e3, 2)
    for mode in ("tile", "sample"):
        times, rays = [], []
        for r in range(args.world):
            plan = node.shard_plan(r, args.world, mode, args.spp)
            frame(plan, 0)
            best = min(frame(plan, args.spp * (i + 1)) for i in range(args.repeat))
            times.append(best[0])
            rays.append(best[1])
        mean, worst = sum(times) / len(times), max(times)
        out[mode] = {"rank_ms": [round(t * 1e3, 2) for t in times], "rank_mrays": [round(x / 1e6, 1) for x in rays],
                     "predicted_efficiency": round(mean / worst, 4),
                     # against the one-GPU frame: what the split itself costs (smaller launches, fixed launch costs)
                     "speedup_vs_one_gpu": round(t1 / worst, 2),
                     "strong_scaling_efficiency": round(t1 / worst / args.world, 4)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
