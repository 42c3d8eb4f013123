"""Multi-process path (N > 1) on the CPU with the gloo backend, world_size 2
and 3: each rank renders its shard with the CPU oracle standing in for the
GPU, buffers are sum-reduced to rank 0 exactly as bench.py does, and the
result is compared with the one-process frame."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, B = 48, 32, 5, 2


def _worker(rank, world, mode, port, out_path, overlapped=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    nr = importlib.import_module("rust-raytracer_amd.node_render")
    scenes = importlib.import_module("rust-raytracer_amd.scenes")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.load("cornell-class")
    S = O.Scene(sc.positions, sc.indices, sc.geom_normals)
    accum = torch.zeros((H, W, 4), dtype=torch.float32)

    def render_fn(acc, samples, sample_offset, shard_index, shard_count):
        a = acc.numpy()
        _, st = S.render(sc.camera, sc.light, W, H, samples, max_bounces=B, sample_offset=sample_offset,
                         shard_index=shard_index, shard_count=shard_count, accum=a, nthreads=1)
        return st["total_rays"]

    if overlapped:
        # two frames of the same image in flight: the second is rendered while the first is reduced
        other = torch.zeros_like(accum)
        first = SPP // 2
        r0, w0 = nr.node_render(render_fn, accum, rank, world, mode, first, wait=False)
        r1, w1 = nr.node_render(render_fn, other, rank, world, mode, SPP - first, sample_offset=first, wait=False)
        w0.wait()
        w1.wait()
        accum += other
        rays = r0 + r1
    else:
        rays = nr.node_render(render_fn, accum, rank, world, mode, SPP)
    t = torch.tensor([float(rays)], dtype=torch.float64)
    dist.all_reduce(t)
    if rank == 0:
        np.savez(out_path, accum=accum.numpy(), rays=np.array([t.item()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "sample"), (2, "tile"), (3, "tile"), (3, "sample")])
def test_gloo_sharded_render(O, scenes, tmp_path, world, mode):
    out = str(tmp_path / "frame.npz")
    port = 29500 + (os.getpid() % 2000) + world * 7 + (0 if mode == "tile" else 3)
    mp.spawn(_worker, args=(world, mode, port, out), nprocs=world, join=True)
    z = np.load(out)
    sc = scenes.load("cornell-class")
    S = O.Scene(sc.positions, sc.indices, sc.geom_normals)
    ref, st = S.render(sc.camera, sc.light, W, H, SPP, max_bounces=B, nthreads=1)
    assert int(z["rays"][0]) == st["total_rays"]
    assert np.array_equal(z["accum"][..., 3], ref[..., 3])           # every pixel got all its samples exactly once
    if mode == "tile":
        assert np.array_equal(z["accum"], ref)                        # disjoint pixels: bit-identical
    else:
        assert np.allclose(z["accum"], ref, rtol=1e-5, atol=1e-6)     # f32 sums in a different order


def test_gloo_overlapped_reduce(O, scenes, tmp_path):
    """node_render(wait=False): frames reduced while the next one is rendered add up to the same image."""
    out = str(tmp_path / "frame.npz")
    port = 29500 + (os.getpid() % 2000) + 41
    mp.spawn(_worker, args=(2, "sample", port, out, True), nprocs=2, join=True)
    z = np.load(out)
    sc = scenes.load("cornell-class")
    S = O.Scene(sc.positions, sc.indices, sc.geom_normals)
    ref, st = S.render(sc.camera, sc.light, W, H, SPP, max_bounces=B, nthreads=1)
    assert int(z["rays"][0]) == st["total_rays"]
    assert np.array_equal(z["accum"][..., 3], ref[..., 3])
    assert np.allclose(z["accum"], ref, rtol=1e-5, atol=1e-6)


def test_c_abi_shard_plan_matches_python(rt):
    nr = importlib.import_module("rust-raytracer_amd.node_render")
    for world in (1, 2, 3, 8):
        for mode in ("sample", "tile"):
            for samples, off in ((128, 0), (5, 7), (2, 1), (0, 0)):
                for rank in range(world):
                    assert rt.shard_plan(rank, world, mode, samples, off) == nr.shard_plan(rank, world, mode, samples, off)
    with pytest.raises(rt.RtError):
        rt.shard_plan(2, 2, "sample", 8)
    with pytest.raises(rt.RtError):
        rt.shard_plan(0, 1, 5, 8)


@pytest.mark.gpu
def test_rt_node_render_with_rccl(rt, scenes):
    """rt_node_render on a one-rank RCCL communicator: the library opens librccl.so itself and the
    in-place ncclReduce leaves rank 0's frame equal to what rt_render_device produces; virtual ranks
    rendered one after the other add up to the unsharded frame."""
    import ctypes as C
    rccl = C.CDLL("librccl.so")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        sc = scenes.load("test-mesh", n_tris=3000, seed=5)
        ctx = rt.Context(0)
        Sg = rt.Scene(ctx, sc.positions, sc.indices, sc.geom_normals)
        W2, H2, spp = 64, 40, 6
        whole = torch.zeros((H2, W2, 4), dtype=torch.float32, device="cuda:0")
        st = Sg.node_render_device(sc.camera, sc.light, whole.data_ptr(), W2, H2, spp, 0, 1, "sample",
                                   nccl_comm=comm.value)
        torch.cuda.synchronize()
        ref = torch.zeros_like(whole)
        st_ref = Sg.render_device(sc.camera, sc.light, ref.data_ptr(), W2, H2, spp)
        torch.cuda.synchronize()
        assert st["total_rays"] == st_ref["total_rays"]
        assert torch.equal(whole, ref)
        for mode in ("sample", "tile"):
            acc = torch.zeros_like(whole)
            rays = 0
            for rank in range(3):   # three virtual ranks into one buffer, no communicator
                rays += Sg.node_render_device(sc.camera, sc.light, acc.data_ptr(), W2, H2, spp, rank, 3, mode)["total_rays"]
            torch.cuda.synchronize()
            assert rays == st_ref["total_rays"]
            if mode == "tile":
                assert torch.equal(acc, ref)
            else:
                assert torch.allclose(acc, ref, rtol=1e-5, atol=1e-6)
        Sg.close()
        ctx.close()
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_shard_plan():
    nr = importlib.import_module("rust-raytracer_amd.node_render")
    for world in (1, 2, 3, 8):
        for samples in (0, 1, 5, 128, 1024):
            plans = [nr.shard_plan(r, world, "sample", samples, sample_offset=7) for r in range(world)]
            assert sum(p["samples"] for p in plans) == samples
            nxt = 7
            for p in plans:                                           # contiguous, disjoint, in rank order
                assert p["sample_offset"] == nxt
                nxt += p["samples"]
            tiles = [nr.shard_plan(r, world, "tile", samples) for r in range(world)]
            assert [t["shard_index"] for t in tiles] == list(range(world))
            assert all(t["shard_count"] == world and t["samples"] == samples for t in tiles)
    with pytest.raises(ValueError):
        nr.shard_plan(2, 2, "sample", 4)
    with pytest.raises(ValueError):
        nr.shard_plan(0, 1, "rows", 4)
