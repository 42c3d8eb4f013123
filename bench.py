#!/usr/bin/env python3
"""Throughput benchmark of the MI355X ray-tracing core.

A step is one pass of the hot path over one batch of synthetic input: the
1920x1080 frame of the workload at `--spp` samples per pixel through generate /
extend / shade / connect / accumulate, plus (N > 1) the RCCL sum-reduce of the
float4 accumulation buffer to rank 0.  Scene, BVH and all queues are resident
in HBM before the timed region.  Ranks render disjoint sample ranges of the same
frame with full scene replicas (weak scaling: per-GPU work is fixed).

Metric: Mrays/s = TraceStats.total_rays / 1e6 / seconds, counting every primary,
bounce and shadow ray (src/trace_stats.rs:28-30, src/main.rs:175).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1]: Sponza primary + shadow rays only on one MI355X
    "sponza-class": dict(scene="sponza-class", kw={}, bounces=0),
    # configs[2]: high-poly mesh, diffuse path tracing
    "dragon-class": dict(scene="dragon-class", kw={}, bounces=2),
    # configs[3]: San Miguel full path tracing
    "san-miguel-class": dict(scene="san-miguel-class", kw={}, bounces=2),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec
# vector L1: 64 B per clock per CU x 256 CUs x 2.4 GHz -- the unit the trace kernels saturate
L1_PEAK_GBS = 64.0 * 256 * 2.4
# fp32 vector peak: 256 CUs x 128 FMA lanes x 2 flops x 2.4 GHz
VALU_PEAK_TFLOPS = 256 * 128 * 2 * 2.4e9 / 1e12


def measured_counters(workload):
    """Per-launch hardware counters of this workload's trace kernels from the newest committed
    rocprofv3 --pmc summary (profiles/rNN_x_pmc_<workload>.json, scripts/gpu_profile_round.sh):
    counters cannot be read from inside the timed run."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % workload)))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        return json.load(fh), os.path.relpath(files[-1], ROOT)



def extend_bytes_per_ray(st):
    """Algorithmic bytes per closest-hit ray: 48 (StreamRay read) + 16 (hit
    record) + 64 per inner-node visit + 48 per triangle test."""
    n = max(1, st["total_rays"])
    return 48.0 + 16.0 + 64.0 * st["inner_rays"] / n + 48.0 * st["obj_intersect"] / n


def flops_per_ray(st):
    """Algorithmic flops per ray of the reference's binary walk: 38 per inner-node visit (12 fma,
    12 min/max, 2 compares: src/bvh/intersect1.rs:160-181) + 46 per triangle test
    (src/triangle_opt.rs:19-42)."""
    n = max(1, st["total_rays"])
    return 38.0 * st["inner_rays"] / n + 46.0 * st["obj_intersect"] / n


def connect_bytes_per_ray(st):
    n = max(1, st["total_rays"])
    return 48.0 + 16.0 + 64.0 * st["inner_rays"] / n + 48.0 * st["obj_intersect"] / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="sponza-class", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=32, help="samples per pixel per step")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=None)
    ap.add_argument("--scene-scale", type=float, default=None, help="shrink the stand-in scene (debug)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--shard", default="sample", choices=["sample", "tile"],
                    help="sample: every rank renders --spp samples of the whole frame per step (weak scaling); "
                         "tile: the step's --spp samples are split by tile (strong scaling)")
    ap.add_argument("--max-paths", type=int, default=None, help="path slots per render pass (library default 2^26)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    n_gpus = world
    # one GPU per rank; ranks wrap around only when a node is rehearsed on fewer GPUs (gloo)
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

    rt = importlib.import_module("rust-raytracer_amd")  # raises if librtcore.so is missing
    scenes = importlib.import_module("rust-raytracer_amd.scenes")
    node = importlib.import_module("rust-raytracer_amd.node_render")

    wl = WORKLOADS[args.workload]
    kw = dict(wl["kw"])
    if args.scene_scale is not None:  # debug: smaller stand-in
        if args.workload == "san-miguel-class":
            kw["scale"] = args.scene_scale
        elif args.workload == "sponza-class":
            kw["detail"] = args.scene_scale
        else:
            kw["n"] = max(8, int(1900 * args.scene_scale ** 0.5))
    bounces = wl["bounces"] if args.bounces is None else args.bounces
    sc = scenes.load(wl["scene"], **kw)
    W, H, S = args.width, args.height, args.spp

    ctx = rt.Context(dev_index)
    if args.max_paths:
        ctx.set_max_paths(args.max_paths)
    t0 = time.time()
    scene = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
    build_s = time.time() - t0

    # Two frame buffers: while frame i is summed across the ranks (the collective runs on the
    # backend's stream) frame i+1 is rendered into the other one.
    step_accums = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(2)]
    pending = [None, None]
    total_accum = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    frame_spp = S * n_gpus if args.shard == "sample" else S   # samples of the whole frame per step
    last = {}

    def retire(slot):
        """Frame in buffer `slot` has been reduced: rank 0 adds it to the running image."""
        if pending[slot] is not None:
            work = pending[slot]
            pending[slot] = None
            if work is not True:
                work.wait()
            if rank == 0:
                total_accum.add_(step_accums[slot])

    def step(i, visit_stats=False):
        slot = i % 2
        retire(slot)
        step_accum = step_accums[slot]
        step_accum.zero_()

        def render_fn(acc, samples, sample_offset, shard_index, shard_count):
            st = scene.render_device(sc.camera, sc.light, acc.data_ptr(), W, H, samples, max_bounces=bounces,
                                     sample_offset=sample_offset, shard_index=shard_index,
                                     shard_count=shard_count, stream=stream, stats=True, visit_stats=visit_stats)
            last["st"] = st
            return st["total_rays"]

        rays, work = node.node_render(render_fn, step_accum, rank, n_gpus, args.shard, frame_spp,
                                      sample_offset=i * frame_spp, wait=False)
        pending[slot] = work if work is not None else True
        return {"total_rays": rays}

    def drain():
        retire(0)
        retire(1)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # counters for the roofline (un-timed, same workload)
    step(0, visit_stats=True)
    drain()
    ext_st, con_st = ctx.last_stats_split()
    for i in range(args.warmup):
        step(i)
    drain()
    total_accum.zero_()

    barrier()
    t_start = time.perf_counter()
    rays = 0
    ms = {"extend": 0.0, "connect": 0.0, "other": 0.0}
    launches = {"extend": 0, "connect": 0, "other": 0}
    ext_rays = con_rays = 0
    for i in range(args.steps):
        st = step(i)
        rays += st["total_rays"]
        m, l = ctx.last_kernel_ms()
        e, c = ctx.last_stats_split()
        ext_rays += e["total_rays"]
        con_rays += c["total_rays"]
        for k in ms:
            ms[k] += m[k]
            launches[k] += l[k]
    drain()
    barrier()
    elapsed = time.perf_counter() - t_start

    if world > 1:
        t = torch.tensor([elapsed, float(rays)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays_all = float(t[1])
    else:
        rays_all = float(rays)

    if rank == 0:
        value = rays_all / 1e6 / elapsed
        b_ext = extend_bytes_per_ray(ext_st)
        b_con = connect_bytes_per_ray(con_st)
        ext_avg_ms = ms["extend"] / max(1, launches["extend"])
        ext_rays_per_launch = ext_rays / max(1, launches["extend"])
        achieved = b_ext * ext_rays_per_launch / (ext_avg_ms * 1e-3) / 1e9 if ext_avg_ms > 0 else 0.0
        con_avg_ms = ms["connect"] / max(1, launches["connect"])
        con_rays_per_launch = con_rays / max(1, launches["connect"])
        con_ach = b_con * con_rays_per_launch / (con_avg_ms * 1e-3) / 1e9 if con_avg_ms > 0 else 0.0
        img = total_accum.cpu().numpy()
        pmc, pmc_file = measured_counters(args.workload)
        pmc_ext = (pmc or {}).get("extend", {})
        traffic = None
        if "hbm_read_bytes_per_launch" in pmc_ext:
            traffic = pmc_ext["hbm_read_bytes_per_launch"] + pmc_ext.get("hbm_write_bytes_per_launch", 0.0)
        out = {
            "metric": "Mrays/s (primary+secondary+shadow rays, whole job)",
            "value": round(value, 2),
            "unit": "Mrays/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak" if args.shard == "sample" else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s stand-in (%d triangles, procedural), %dx%d, %d spp per step per GPU, "
                            "max_bounces %d%s, tile 16, %s sharding + sum-reduce of the accumulation buffer (%s)" % (
                                args.workload, sc.num_triangles, W, H, S, bounces,
                                " (primary + shadow rays)" if bounces == 0 else "", args.shard,
                                args.backend if n_gpus > 1 else "single GPU"),
                "triangles": sc.num_triangles,
                "bvh_build_s": round(build_s, 3),
                "rays_per_step_per_gpu": int(rays / max(1, args.steps)),
                "mean_pixel": float(img[..., :3].sum() / max(1.0, float(img[..., 3].sum()))),
                # the reference's other two console figures (src/main.rs:170-179)
                "samples_per_pixel_per_s": round(frame_spp * args.steps / elapsed, 2),
                "msamples_per_s": round(frame_spp * args.steps * W * H / 1e6 / elapsed, 2),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "extend_kernel (closest-hit traversal)",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": pmc_file,
                "algorithmic_bytes_per_launch": round(b_ext * ext_rays_per_launch, 0),
                # the working set is cache resident: the unit these kernels saturate is the vector L1
                "l1_peak": round(L1_PEAK_GBS, 1),
                "l1_frac": round(achieved / L1_PEAK_GBS, 4),
                "counters": {k: round(v, 4) for k, v in pmc_ext.items()
                             if k in ("l1_cache_busy", "vector_memory_address_unit_busy",
                                      "valu_issue_busy_at_2_cycles", "active_lanes_per_valu_instruction",
                                      "l1_hit_rate", "l2_hit_rate")},
                "bytes_per_ray": round(b_ext, 1),
                # secondary roofline: algorithmic fp32 work of the same walk against the vector ALUs
                "flops_per_ray": round(flops_per_ray(ext_st), 1),
                "achieved_tflops": round(flops_per_ray(ext_st) * ext_rays_per_launch / (ext_avg_ms * 1e-3) / 1e12, 2)
                if ext_avg_ms > 0 else 0.0,
                "valu_peak_tflops": round(VALU_PEAK_TFLOPS, 1),
                "rays_per_launch": int(ext_rays_per_launch),
                "avg_launch_ms": round(ext_avg_ms, 4),
                "launches": launches["extend"],
                "mrays_per_s": round(ext_rays_per_launch / (ext_avg_ms * 1e-3) / 1e6, 1) if ext_avg_ms > 0 else 0.0,
                "inner_visits_per_ray": round(ext_st["inner_rays"] / max(1, ext_st["total_rays"]), 2),
                "tri_tests_per_ray": round(ext_st["obj_intersect"] / max(1, ext_st["total_rays"]), 2),
            },
            "kernels": {
                "connect": {"achieved_gbs": round(con_ach, 1), "bytes_per_ray": round(b_con, 1),
                            "avg_launch_ms": round(con_avg_ms, 4), "launches": launches["connect"],
                            "mrays_per_s": round(con_rays_per_launch / (con_avg_ms * 1e-3) / 1e6, 1) if con_avg_ms > 0 else 0.0,
                            "inner_visits_per_ray": round(con_st["inner_rays"] / max(1, con_st["total_rays"]), 2),
                            "tri_tests_per_ray": round(con_st["obj_intersect"] / max(1, con_st["total_rays"]), 2)},
                "other_ms_per_step": round(ms["other"] / max(1, args.steps), 4),
                "extend_ms_per_step": round(ms["extend"] / max(1, args.steps), 4),
                "connect_ms_per_step": round(ms["connect"] / max(1, args.steps), 4),
            },
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, W, H, bounces, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    scene.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(sc, W, H, bounces, budget_s):
    """The CPU oracle (C restatement of the reference's default integrator,
    integrate_tile_stream_shadow_immediate, tile-parallel like src/main.rs:157-161)
    timed on this box's host cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O  # checker / baseline only
    threads = min(os.cpu_count() or 1, 16)
    So = O.Scene(sc.positions, sc.indices, sc.geom_normals)
    t0 = time.perf_counter()
    _, st = So.render(sc.camera, sc.light, W, H, 1, max_bounces=bounces, nthreads=threads)
    dt = time.perf_counter() - t0
    spp = 1
    rays, secs = st["total_rays"], dt
    more = int(min(64, max(0, (budget_s - dt) // max(dt, 1e-3))))
    if more >= 1:
        t0 = time.perf_counter()
        _, st = So.render(sc.camera, sc.light, W, H, more, max_bounces=bounces, sample_offset=1, nthreads=threads)
        secs = time.perf_counter() - t0
        rays, spp = st["total_rays"], more
    return {"value": round(rays / 1e6 / secs, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "same scene, camera and bounce cap, %dx%d at %d spp (%.1f s of CPU work); C restatement of the "
                      "reference's default integrator, not the Rust binary" % (W, H, spp, secs)}


if __name__ == "__main__":
    main()
