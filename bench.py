#!/usr/bin/env python3
"""Throughput benchmark of the MI355X ray-tracing core.

A step is one pass of the hot path over one batch of synthetic input: one 1920x1080 frame of the
workload at `--spp` samples per pixel (default 128, the reference's default, src/args.rs:24) through
generate / extend / shade / connect / accumulate, plus (N > 1) the RCCL sum-reduce of the float4
accumulation buffer to rank 0.  Scene, both trees and all queues are resident in HBM before the
timed region.

Default workload: the San-Miguel-class stand-in at max_bounces 2 (BASELINE.json's north-star
workload, camera and light of src/scene.rs:119-125).  With N > 1 the frame is split by tile
(`--shard tile`: rank r renders the tiles i with i % N == r at all samples; total work is fixed,
strong scaling); `--shard sample` gives every rank `--spp` samples of the whole frame (weak scaling).

Metric: Mrays/s = TraceStats.total_rays / 1e6 / seconds, counting every primary, bounce and shadow
ray (src/trace_stats.rs:28-30, src/main.rs:175).

At N = 1 the line also carries, under "workloads", the rate of the two other single-GPU workloads
(Sponza-class primary + shadow rays, dragon-class diffuse path tracing) and of every workload in
the exact configuration (the reference's tree walked in the reference's order), each timed for at
least `--min-seconds`.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import glob
import importlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1]: Sponza primary + shadow rays only on one MI355X
    "sponza-class": dict(scene="sponza-class", bounces=0),
    # configs[2]: high-poly mesh, diffuse path tracing
    "dragon-class": dict(scene="dragon-class", bounces=2),
    # configs[3]: San Miguel full path tracing (the north-star workload)
    "san-miguel-class": dict(scene="san-miguel-class", bounces=2),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E
CUS, SIMDS, CLOCK_GHZ = 256, 1024, 2.4
# fp32 vector peak: 256 CUs x 128 FMA lanes x 2 flops x 2.4 GHz
VALU_PEAK_TFLOPS = CUS * 128 * 2 * CLOCK_GHZ * 1e9 / 1e12
EXACT_OPTIONS = {"traversal_depth0": 0, "traversal_deep": 1}
DEFAULT_OPTIONS = {"traversal_depth0": 3, "traversal_deep": 2}


def measured_counters(workload):
    """Per-launch hardware counters of this workload's trace kernels from the newest committed
    rocprofv3 --pmc summary (profiles/rNN_x_pmc_<workload>.json, scripts/gpu_profile_round.sh):
    counters cannot be read from inside the timed run."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % workload)))
    if not files:
        return {}, None
    with open(files[-1]) as fh:
        return json.load(fh), os.path.relpath(files[-1], ROOT)


def bytes_per_ray(inner, tris, rays):
    """Algorithmic bytes per ray of the reference's binary walk: 48 (StreamRay read) + 16 (hit record)
    + 64 per inner-node visit + 48 per triangle test."""
    n = max(1, rays)
    return 48.0 + 16.0 + 64.0 * inner / n + 48.0 * tris / n


def flops_per_ray(inner, tris, rays):
    """Algorithmic flops per ray of the same walk: 38 per inner-node visit (12 fma, 12 min/max,
    2 compares: src/bvh/intersect1.rs:160-181) + 46 per triangle test (src/triangle_opt.rs:19-42)."""
    n = max(1, rays)
    return 38.0 * inner / n + 46.0 * tris / n


def scene_kwargs(workload, scale):
    if scale is None:
        return {}
    if workload == "san-miguel-class":
        return {"scale": scale}
    if workload == "sponza-class":
        return {"detail": scale}
    return {"n": max(8, int(1900 * scale ** 0.5))}


class LaunchTimes:
    """HIP-event durations of the trace launches of render calls, by (kind, depth class)."""

    def __init__(self, nb):
        self.nb = nb
        self.ms = {}
        self.n = {}
        self.other = 0.0

    def add_call(self, launches):
        seen = {"extend": 0, "connect": 0}
        for kind, ms in launches:
            if kind == "other":
                self.other += ms
                continue
            depth = seen[kind] % self.nb   # per pass: depth 0, 1, .. nb-1 in issue order
            seen[kind] += 1
            key = (kind, "depth0" if depth == 0 else "deep")
            self.ms[key] = self.ms.get(key, 0.0) + ms
            self.n[key] = self.n.get(key, 0) + 1


def kernel_name(kind, depth_class, options, fused):
    """Template instantiation the library launches for this kind of launch (kernels.hip)."""
    form = options["traversal_depth0"] if depth_class == "depth0" else options["traversal_deep"]
    io = "PrimaryIO" if (kind == "extend" and depth_class == "depth0" and fused) else "QueueIO"
    return "trace_kernel<%s,false,%d,%s>" % ("true" if kind == "connect" else "false", form, io)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="san-miguel-class", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=128, help="samples per pixel of the frame one step renders")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=None)
    ap.add_argument("--scene-scale", type=float, default=None, help="shrink the stand-in scenes (debug)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--shard", default="tile", choices=["sample", "tile"],
                    help="tile: the frame's tiles are dealt round-robin to the ranks (strong scaling, default); "
                         "sample: every rank renders --spp samples of the whole frame per step (weak scaling)")
    ap.add_argument("--max-paths", type=int, default=None, help="path slots per render pass (library default 2^28)")
    ap.add_argument("--no-extras", action="store_true", help="skip the other workloads and the exact-mode rates")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="timed seconds per extra measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU work of the baseline leg, both rates together")
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
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

    rt = importlib.import_module("rust-raytracer_amd")  # raises if librtcore.so is missing
    scenes = importlib.import_module("rust-raytracer_amd.scenes")
    node = importlib.import_module("rust-raytracer_amd.node_render")

    W, H, S = args.width, args.height, args.spp
    ctx = rt.Context(dev_index)
    if args.max_paths:
        ctx.set_max_paths(args.max_paths)
    stream = torch.cuda.current_stream().cuda_stream

    def load(workload):
        sc = scenes.load(WORKLOADS[workload]["scene"], **scene_kwargs(workload, args.scene_scale))
        t0 = time.time()
        scene = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
        return sc, scene, time.time() - t0

    wl = WORKLOADS[args.workload]
    bounces = wl["bounces"] if args.bounces is None else args.bounces
    nb = bounces + 1
    sc, scene, build_s = load(args.workload)

    # Two frame buffers: while frame i is summed across the ranks (the collective runs on the
    # backend's stream) frame i+1 is rendered into the other one.
    step_accums = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(2)]
    pending = [None, None]
    total_accum = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    frame_spp = S * n_gpus if args.shard == "sample" else S   # samples of the whole frame per step
    reduce_s = [0.0]

    def retire(slot):
        """Frame in buffer `slot` has been reduced: rank 0 adds it to the running image."""
        if pending[slot] is not None:
            work = pending[slot]
            pending[slot] = None
            if work is not True:
                t0 = time.perf_counter()
                work.wait()
                reduce_s[0] += time.perf_counter() - t0
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
            return st["total_rays"]

        rays, work = node.node_render(render_fn, step_accum, rank, n_gpus, args.shard, frame_spp,
                                      sample_offset=i * frame_spp, dist=dist, wait=False)
        pending[slot] = work if work is not None else True
        return rays

    def drain():
        retire(0)
        retire(1)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # visit counters of the reference's walk on this workload's rays, per depth (un-timed; the call
    # runs the exact forms): the algorithmic bytes and flops of the roofline
    step(0, visit_stats=True)
    drain()
    visit_ext, visit_con = ctx.last_depth_stats()
    for i in range(args.warmup):
        step(i)
    drain()
    total_accum.zero_()
    reduce_s[0] = 0.0

    barrier()
    t_start = time.perf_counter()
    rays = 0
    times = LaunchTimes(nb)
    depth_rays = {"extend": [0] * 8, "connect": [0] * 8}
    for i in range(args.steps):
        rays += step(i)
        times.add_call(ctx.last_launch_ms())
        de, dc = ctx.last_depth_stats()
        for d in range(8):
            depth_rays["extend"][d] += de["rays"][d]
            depth_rays["connect"][d] += dc["rays"][d]
    drain()
    barrier()
    elapsed = time.perf_counter() - t_start

    if world > 1:
        t = torch.tensor([elapsed, float(rays), reduce_s[0]], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays_all = float(t[1])
        reduce_wait_s = float(tmax[2])
    else:
        rays_all = float(rays)
        reduce_wait_s = 0.0

    out = None
    if rank == 0:
        value = rays_all / 1e6 / elapsed
        img = total_accum.cpu().numpy()
        options = dict(DEFAULT_OPTIONS)
        pmc, pmc_file = measured_counters(args.workload)

        # per kind of launch: rays, time, algorithmic work of the reference's walk on the same rays
        kernels = {}
        for (kind, dclass), ms in times.ms.items():
            n_launch = times.n[(kind, dclass)]
            depths = [0] if dclass == "depth0" else list(range(1, 8))
            visit = visit_ext if kind == "extend" else visit_con
            r_timed = sum(depth_rays[kind][d] for d in depths)
            r_visit = sum(visit["rays"][d] for d in depths)
            inner = sum(visit["inner_rays"][d] for d in depths)
            tris = sum(visit["obj_intersect"][d] for d in depths)
            name = kernel_name(kind, dclass, options, fused=True)
            avg_ms = ms / max(1, n_launch)
            per_launch = r_timed / max(1, n_launch)
            kernels["%s_%s" % (kind, dclass)] = {
                "kernel": name, "launches": n_launch, "avg_launch_ms": round(avg_ms, 4),
                "ms_per_step": round(ms / max(1, args.steps), 4),
                "rays_per_launch": int(per_launch),
                "mrays_per_s": round(per_launch / (avg_ms * 1e-3) / 1e6, 1) if avg_ms > 0 else 0.0,
                "inner_visits_per_ray": round(inner / max(1, r_visit), 2),
                "tri_tests_per_ray": round(tris / max(1, r_visit), 2),
                "bytes_per_ray": round(bytes_per_ray(inner, tris, r_visit), 1),
                "flops_per_ray": round(flops_per_ray(inner, tris, r_visit), 1),
            }
        dominant = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        roofline = make_roofline(kernels[dominant], (pmc.get("kernels") or {}).get(kernels[dominant]["kernel"]),
                                 pmc_file)
        for k, rec in kernels.items():   # the same fractions for every trace kernel, short form
            r = make_roofline(rec, (pmc.get("kernels") or {}).get(rec["kernel"]), pmc_file)
            rec["bound"] = r["bound"]
            rec["hbm_frac"] = r["hbm"]["frac"]
            rec["address_unit_busy_frac"] = r["units"].get("address_unit_busy_frac")
            rec["valu_issue_frac"] = r["units"].get("valu_issue_frac")
            rec["valu_algorithmic_frac"] = r["valu_algorithmic_frac"]

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
                "workload": "%s stand-in (%d triangles, procedural), %dx%d, %d spp per step%s, max_bounces %d%s, "
                            "tile 16, library defaults (4-wide quantized tree), %s" % (
                                args.workload, sc.num_triangles, W, H, S,
                                " per GPU" if args.shard == "sample" and n_gpus > 1 else "", bounces,
                                " (primary + shadow rays)" if bounces == 0 else "",
                                "single GPU" if n_gpus == 1 else
                                "%s sharding over %d GPUs + sum-reduce of the accumulation buffer (%s)" % (
                                    args.shard, n_gpus, args.backend)),
                "triangles": sc.num_triangles,
                "bvh_build_s": round(build_s, 3),
                "timed_s": round(elapsed, 3),
                "rays_per_step": int(rays_all / max(1, args.steps)),
                "mean_pixel": float(img[..., :3].sum() / max(1.0, float(img[..., 3].sum()))),
                # the reference's other two console figures (src/main.rs:170-179)
                "samples_per_pixel_per_s": round(frame_spp * args.steps / elapsed, 2),
                "msamples_per_s": round(frame_spp * args.steps * W * H / 1e6 / elapsed, 2),
                "reduce_wait_ms_per_step": round(reduce_wait_s / max(1, args.steps) * 1e3, 3),
            },
            "roofline": roofline,
            "kernels": dict(kernels, other_ms_per_step=round(times.other / max(1, args.steps), 4)),
        }

    if n_gpus == 1 and not args.no_extras:
        out["workloads"] = extras(rt, scenes, ctx, args, sc, scene, bounces, value, stream, dev, load)
        scene = None  # closed by extras
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sc, W, H, bounces, args.cpu_seconds)
    if rank == 0:
        check_fractions(out)
        print(json.dumps(out), flush=True)

    if scene is not None:
        scene.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def make_roofline(rec, pmc, pmc_file):
    """Roofline record of one trace kernel.  `rec`: launch statistics measured live (HIP events on the
    launch stream); `pmc`: per-launch hardware counters of the same kernel from the committed
    rocprofv3 --pmc passes (None when there is no profile of this build).

    `bound` names the most-utilised unit the counters show and `frac` is that unit's busy fraction:
    the vector-memory address unit (TA_TA_BUSY / (256 CUs x kernel cycles)), the L1 tag pipeline behind it
    (TCP_TOTAL_CACHE_ACCESSES / (256 CUs x kernel cycles): one line access per clock), VALU issue
    (SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x kernel cycles)) or HBM (measured bytes / time / peak).
    The HBM side is kept under `hbm`: bytes from FETCH_SIZE x the calibrated factor + WRITE_SIZE, the
    part of them the kernel asked for (L2 misses x 48-byte items), and the reminder that FETCH_SIZE
    counts Infinity-Cache hits.  The algorithmic bytes (the reference's walk: 64 B per node visit,
    48 B per triangle test) are mostly served by L1/L2, so they are reported, never divided by a peak."""
    t = rec["avg_launch_ms"] * 1e-3
    alg_bytes = rec["bytes_per_ray"] * rec["rays_per_launch"]
    alg_flops = rec["flops_per_ray"] * rec["rays_per_launch"]
    r = {
        "bound": None, "kernel": rec["kernel"],
        "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
        "traffic_source": pmc_file,
        "avg_launch_ms": rec["avg_launch_ms"], "launches": rec["launches"], "rays_per_launch": rec["rays_per_launch"],
        "algorithmic_bytes_per_launch": round(alg_bytes, 0),
        "algorithmic_gbs": round(alg_bytes / t / 1e9, 1) if t > 0 else None,
        "algorithmic_flops_per_launch": round(alg_flops, 0),
        "valu_peak_tflops": round(VALU_PEAK_TFLOPS, 1),
        "valu_algorithmic_tflops": round(alg_flops / t / 1e12, 2) if t > 0 else None,
        "valu_algorithmic_frac": round(alg_flops / t / 1e12 / VALU_PEAK_TFLOPS, 4) if t > 0 else None,
        "units": {},
        "hbm": {"peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": None, "frac": None, "traffic": None,
                "note": "FETCH_SIZE counts Infinity-Cache hits as well: frac is an upper bound on HBM use"},
    }
    if not (pmc and t > 0):
        r["bound"] = "unknown (no counters of this kernel in profiles/)"
        return r
    # the profiled launches hold the same rays per launch as the timed ones (same pass size); scale by
    # the ray counts in case they do not
    scale = rec["rays_per_launch"] / pmc["rays_per_launch"] if pmc.get("rays_per_launch") else 1.0
    h = r["hbm"]
    traffic = pmc.get("hbm_read_bytes_per_launch", 0.0) + pmc.get("hbm_write_bytes_per_launch", 0.0)
    if traffic > 0:
        h["traffic"] = r["traffic"] = round(traffic * scale, 0)
        h["achieved"] = round(traffic * scale / t / 1e9, 1)
        h["frac"] = round(traffic * scale / t / 1e9 / HBM_PEAK_GBS, 4)
        h["fetch_size_factor"] = pmc.get("fetch_size_factor", 2.0)
        h["fetch_size_factor_source"] = pmc.get("fetch_size_factor_source", "MI355X guide: coalesced 16 B/lane reads")
        r["algorithmic_over_traffic"] = round(alg_bytes / (traffic * scale), 2)
        if "l2_misses_per_launch" in pmc:
            # what the kernel asked the fabric for: one 48-byte item per L2 miss
            useful = pmc["l2_misses_per_launch"] * 48.0 * scale
            h["useful_bytes"] = round(useful, 0)
            h["useful_hbm_frac"] = round(useful / t / 1e9 / HBM_PEAK_GBS, 4)
    u = r["units"]
    if "valu_issue_busy_at_2_cycles" in pmc:
        u["valu_issue_frac"] = round(pmc["valu_issue_busy_at_2_cycles"], 4)
    if "active_lanes_per_valu_instruction" in pmc:
        u["valu_lane_frac"] = round(pmc["active_lanes_per_valu_instruction"] / 64.0, 4)
    if "valu_issue_frac" in u and "valu_lane_frac" in u:
        u["valu_useful_frac"] = round(u["valu_issue_frac"] * u["valu_lane_frac"], 4)
    if "vector_memory_address_unit_busy" in pmc:
        u["address_unit_busy_frac"] = round(pmc["vector_memory_address_unit_busy"], 4)
    if "address_unit_stalled_by_cache" in pmc:
        u["address_unit_stalled_by_cache_frac"] = round(pmc["address_unit_stalled_by_cache"], 4)
    if h["frac"] is not None:
        u["hbm_frac"] = h["frac"]
    if "l1_accesses_per_clock_per_cu" in pmc:
        # one access = one line touched by one vector memory instruction; the L1 takes one per clock
        u["l1_pipeline_frac"] = round(min(1.0, pmc["l1_accesses_per_clock_per_cu"]), 4)
    if "l1_fill_latency_cycles" in pmc:
        u["l1_fill_latency_cycles"] = round(pmc["l1_fill_latency_cycles"], 1)
    for k in ("l1_hit_rate", "l2_hit_rate", "kernel_cycles", "valu_instructions_per_launch",
              "vector_loads_per_launch"):
        if k in pmc:
            u[k] = round(pmc[k], 4)
    # the most-utilised unit bounds the kernel
    cyc = pmc.get("kernel_cycles", 0.0)
    cands = []
    if "address_unit_busy_frac" in u:
        cands.append((u["address_unit_busy_frac"], "vector-memory address unit (TA busy cycles)",
                      u["address_unit_busy_frac"] * cyc, cyc, "cycles per CU per launch"))
    if "valu_issue_frac" in u:
        cands.append((u["valu_issue_frac"], "VALU issue (2 cycles per wave64 instruction)", u["valu_issue_frac"] * cyc, cyc,
                      "cycles per SIMD per launch"))
    if "l1_pipeline_frac" in u:
        cands.append((u["l1_pipeline_frac"], "L1 tag pipeline (one line access per clock and CU)",
                      u["l1_pipeline_frac"] * cyc, cyc, "accesses per CU per launch"))
    if h["frac"] is not None:
        cands.append((h["frac"], "hbm", h["achieved"], HBM_PEAK_GBS, "GB/s"))
    if cands:
        frac, name, achieved, peak, unit = max(cands)
        r.update(bound=name, frac=round(frac, 4), achieved=round(achieved, 1), peak=round(peak, 1), unit=unit)
    return r


def check_fractions(node, path=""):
    """No field named *frac may leave [0, 1]."""
    if isinstance(node, dict):
        for k, v in node.items():
            if k.endswith("frac") and v is not None:
                assert 0.0 <= v <= 1.0, "%s%s = %r is not a fraction" % (path, k, v)
            check_fractions(v, path + k + ".")


def timed_rate(torch, scene, sc, acc, W, H, spp, bounces, stream, min_seconds):
    """Whole-job Mrays/s of repeated frames of `spp` samples, timed for at least min_seconds."""
    def frame(i):
        st = scene.render_device(sc.camera, sc.light, acc.data_ptr(), W, H, spp, max_bounces=bounces,
                                 sample_offset=i * spp, stream=stream, stats=True)
        return st["total_rays"]

    frame(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rays, n = 0, 0
    while True:
        rays += frame(n + 1)
        n += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dt >= min_seconds and n >= 3:
            break
    return {"value": round(rays / 1e6 / dt, 2), "unit": "Mrays/s", "frames": n, "spp_per_frame": spp,
            "ms_per_frame": round(dt / n * 1e3, 3), "timed_s": round(dt, 3)}


def extras(rt, scenes, ctx, args, sc_main, scene_main, bounces_main, value_main, stream, dev, load):
    """The other single-GPU workloads, and every workload in the exact configuration (forms 0/1 on the
    reference's own tree, same queues and shading): the rates a host gets when it needs ids and visit
    order to be the reference's, beside the default (4-wide tree) rates."""
    import torch
    W, H = args.width, args.height
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    out = {}

    def exact_rate(scene, sc, spp, bounces):
        for k, v in EXACT_OPTIONS.items():
            ctx.set_option(k, v)
        try:
            return timed_rate(torch, scene, sc, acc, W, H, spp, bounces, stream, args.min_seconds)
        finally:
            for k, v in DEFAULT_OPTIONS.items():
                ctx.set_option(k, v)

    name = "%s b=%d" % (args.workload, bounces_main)
    out[name] = {"value": round(value_main, 2), "unit": "Mrays/s", "note": "the line's value",
                 "exact_mode": exact_rate(scene_main, sc_main, args.spp, bounces_main)}
    if args.workload == "san-miguel-class" and (W, H) == (1920, 1080):
        # BASELINE config 5: the same scene at 3840x2160, rendered progressively in steps of a sample range
        # (one step of 32 samples per pixel is as many paths as the 1080p frame at 128)
        W4, H4, spp4 = 3840, 2160, max(1, args.spp // 4)
        acc4 = torch.zeros((H4, W4, 4), dtype=torch.float32, device=dev)
        rec = timed_rate(torch, scene_main, sc_main, acc4, W4, H4, spp4, bounces_main, stream, args.min_seconds)
        rec["note"] = "3840x2160, progressive: every frame is the next range of %d samples per pixel" % spp4
        for k, v in EXACT_OPTIONS.items():
            ctx.set_option(k, v)
        try:
            rec["exact_mode"] = timed_rate(torch, scene_main, sc_main, acc4, W4, H4, spp4, bounces_main, stream,
                                           args.min_seconds)
        finally:
            for k, v in DEFAULT_OPTIONS.items():
                ctx.set_option(k, v)
        out["%s 4K b=%d" % (args.workload, bounces_main)] = rec
        del acc4
    scene_main.close()
    for wl in sorted(WORKLOADS):
        if wl == args.workload:
            continue
        b = WORKLOADS[wl]["bounces"]
        sc, scene, build_s = load(wl)
        rec = timed_rate(torch, scene, sc, acc, W, H, args.spp, b, stream, args.min_seconds)
        rec["triangles"] = sc.num_triangles
        rec["bvh_build_s"] = round(build_s, 3)
        rec["exact_mode"] = exact_rate(scene, sc, args.spp, b)
        out["%s b=%d" % (wl, b)] = rec
        scene.close()
    return out


def cpu_baseline(sc, W, H, bounces, budget_s):
    """The CPU oracle (C restatement of the reference's default integrator,
    integrate_tile_stream_shadow_immediate, tile-parallel like src/main.rs:157-161) timed on this
    box's host cores on a bounded sample of the same workload, built for this box's CPU with the
    flags printed in `flags`.  Two rates: the traversal in the reference's own SSE/AVX2 intrinsics
    (rt_oracle_simd.c, kind "port-simd": the fairer figure, and the record's value) and the scalar
    restatement (kind "port")."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O  # checker / baseline only
    path, flags = O.build_native(tempfile.mkdtemp(prefix="rt_oracle_"))
    if path:
        O.use_library(path)
    else:
        flags = "committed build (-O2 -march=x86-64-v3 -ffp-contract=off); native build failed: %s" % flags
    threads = min(os.cpu_count() or 1, 16)
    So = O.Scene(sc.positions, sc.indices, sc.geom_normals)

    def timed(mode, budget):
        t0 = time.perf_counter()
        _, st = So.render(sc.camera, sc.light, W, H, 1, max_bounces=bounces, nthreads=threads, mode=mode)
        dt = time.perf_counter() - t0
        rays, secs, spp = st["total_rays"], dt, 1
        more = int(min(64, max(0, (budget - dt) // max(dt, 1e-3))))
        if more >= 1:
            t0 = time.perf_counter()
            _, st = So.render(sc.camera, sc.light, W, H, more, max_bounces=bounces, sample_offset=1, nthreads=threads,
                              mode=mode)
            secs = time.perf_counter() - t0
            rays, spp = st["total_rays"], more
        return rays / 1e6 / secs, spp, secs

    simd_rate, simd_spp, simd_s = timed(O.MODE_STREAM_SHADOW_IMMEDIATE | O.MODE_SIMD, budget_s / 2)
    scalar_rate, scalar_spp, scalar_s = timed(O.MODE_STREAM_SHADOW_IMMEDIATE, budget_s / 2)
    return {"value": round(simd_rate, 3), "unit": "Mrays/s", "cores": threads, "kind": "port-simd",
            "flags": "gcc " + flags + " -mavx2 -mfma",
            "sample": "same scene, camera and bounce cap, %dx%d at %d spp (%.1f s of CPU work); C restatement of the "
                      "reference's default integrator with its traversals in the reference's own intrinsics: AVX2 "
                      "two-rays-by-two-children node filter, four-ray SSE leaf loop, SSE single-ray shadow rays (the Rust "
                      "binary itself cannot be built here)" % (W, H, simd_spp, simd_s),
            "scalar": {"value": round(scalar_rate, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                       "sample": "the same frame at %d spp (%.1f s) through the scalar restatement" % (scalar_spp, scalar_s)}}


if __name__ == "__main__":
    main()
