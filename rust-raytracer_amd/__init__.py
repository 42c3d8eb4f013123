"""Python host binding of librtcore.so, the MI355X ray-tracing core.

Mirrors the reference's trace boundary: `Scene.intersect_stream` /
`Scene.occluded_stream` stand in for `Bvh::intersect_stream` /
`Bvh::occluded_stream` (src/bvh/intersect_stream.rs:10-15, 200-205) and
`Scene.render` for the per-tile integrator call of src/main.rs:133-161.

There is no CPU fallback: importing works without a GPU (so the library's
symbols can be checked), but every compute call goes to the HIP library and
raises `RtError` when that fails.  A missing librtcore.so raises at import.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

# PyTorch wheels carry their own copy of the HIP runtime.  A process that loads librtcore.so (and with
# it the system's libamdhip64) first and imports torch afterwards can end up with two runtimes, and
# the second one finds no GPU ("No HIP GPUs are available").  Importing torch first makes both use the
# same copy; hosts without torch simply use the system runtime.
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    pass

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTCORE_LIB selects another build of the same library (tuning experiments)
LIB_PATH = os.environ.get("RTCORE_LIB") or os.path.join(_HERE, "librtcore.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "rtcore.h")

NO_HIT = 0xFFFFFFFF
LDS_STACK_ENTRIES = 24  # RT_LDS_STACK_ENTRIES
RENDER_VISIT_STATS = 1

RAY_DTYPE = np.dtype([("origin_far", np.float32, 4),
                      ("dir_recip_near", np.float32, 4),
                      ("direction", np.float32, 4)])
NODE_DTYPE = np.dtype([("bx", np.float32, 4), ("by", np.float32, 4), ("bz", np.float32, 4),
                       ("left", np.uint32), ("right", np.uint32),
                       ("left_count", np.uint32), ("right_count", np.uint32)])
WIDE_DTYPE = np.dtype([("lo", np.float32, (3, 4)), ("hi", np.float32, (3, 4)), ("child", np.uint32, 4),
                       ("bin", np.uint32), ("pad", np.uint32, 3)])
QWIDE_DTYPE = np.dtype([("origin", np.float32, 3), ("exps", np.uint32), ("base4", np.uint32), ("meta", np.uint32),
                        ("lo", np.uint32, 3), ("hi", np.uint32, 3)])
TRI_DTYPE = np.dtype([("v0", np.float32, 3), ("v0v1", np.float32, 3), ("v0v2", np.float32, 3),
                      ("id", np.uint32), ("pad", np.uint32, 2)])


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rtcore error %d: %s" % (code, msg))
        self.code = code


class TraceStats(C.Structure):
    """src/trace_stats.rs:4-12"""
    _fields_ = [(n, C.c_uint64) for n in ("total_rays", "inner_visit", "inner_rays", "leaf_visit",
                                          "leaf_rays", "leaf_objs", "obj_intersect")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class DepthStats(C.Structure):
    """rt_depth_stats"""
    _fields_ = [("rays", C.c_uint64 * 8), ("inner_rays", C.c_uint64 * 8), ("obj_intersect", C.c_uint64 * 8)]

    def as_dict(self):
        return {n: [int(x) for x in getattr(self, n)] for n, _ in self._fields_}


class BvhStats(C.Structure):
    """src/bvh/build.rs:5-11"""
    _fields_ = [(n, C.c_uint32) for n in ("num_leaves", "num_objs", "max_depth", "max_objs")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class SceneInfo(C.Structure):
    _fields_ = [("num_triangles", C.c_uint32), ("num_inner_nodes", C.c_uint32),
                ("root_ref", C.c_uint32), ("root_count", C.c_uint32), ("device_bytes", C.c_uint64)]


class CameraC(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("target", C.c_float * 3), ("up", C.c_float * 3),
                ("fov_deg", C.c_float)]


class LightC(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("intensity", C.c_float * 3)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples", C.c_uint32),
                ("sample_offset", C.c_uint32), ("max_bounces", C.c_uint32),
                ("tile_size", C.c_uint32), ("shard_index", C.c_uint32),
                ("shard_count", C.c_uint32), ("samples_per_pass", C.c_uint32),
                ("flags", C.c_uint32), ("seed", C.c_uint64)]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "librtcore.so is missing (%s). Build it with `make -C %s/csrc` or "
        "`python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback."
        % (LIB_PATH, _HERE))

_lib = C.CDLL(LIB_PATH)
_vp = C.c_void_p

EXPORTS = {
    "rt_last_error": (C.c_char_p, []),
    "rt_version": (C.c_char_p, []),
    "rt_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "rt_ctx_destroy": (None, [_vp]),
    "rt_ctx_set_max_paths": (C.c_int, [_vp, C.c_uint64]),
    "rt_ctx_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "rt_ctx_last_kernel_ms": (C.c_int, [_vp, C.POINTER(C.c_float * 3), C.POINTER(C.c_uint32 * 3)]),
    "rt_debug_sort_keys": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp]),
    "rt_ctx_last_launch_ms": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(C.c_uint32)]),
    "rt_ctx_last_max_ray_steps": (C.c_int, [_vp, C.POINTER(C.c_uint64 * 2)]),
    "rt_ctx_last_depth_stats": (C.c_int, [_vp, C.POINTER(DepthStats * 2)]),
    "rt_ctx_last_max_stack_entries": (C.c_int, [_vp, C.POINTER(C.c_uint64 * 2)]),
    "rt_ctx_last_pass_paths": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "rt_ctx_last_stats_split": (C.c_int, [_vp, C.POINTER(TraceStats * 2)]),
    "rt_scene_create": (C.c_int, [_vp, _vp, C.c_uint32, _vp, C.c_uint32, _vp, C.POINTER(_vp)]),
    "rt_scene_destroy": (None, [_vp]),
    "rt_bvh_build_search_host": (C.c_int, [_vp, C.c_uint32, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp,
                                           C.POINTER(SceneInfo), C.POINTER(BvhStats)]),
    "rt_bvh_build_search_device": (C.c_int, [_vp, _vp, C.c_uint32, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp,
                                             C.POINTER(SceneInfo), C.POINTER(BvhStats)]),
    "rt_scene_fast_tree_on_device": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "rt_scene_bounds": (C.c_int, [_vp, C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3)]),
    "rt_scene_get_wide_items": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "rt_bvh_build_host": (C.c_int, [_vp, C.c_uint32, _vp, C.c_uint32, C.c_uint32, _vp, _vp,
                                    C.POINTER(SceneInfo), C.POINTER(BvhStats)]),
    "rt_bvh_wide_host": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp,
                                   C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint32)]),
    "rt_bvh_refine_host": (C.c_int, [_vp, C.c_uint32, _vp, C.c_uint32, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp,
                                     C.c_uint32, _vp, _vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rt_scene_stats": (C.c_int, [_vp, C.POINTER(BvhStats)]),
    "rt_scene_get_info": (C.c_int, [_vp, C.POINTER(SceneInfo)]),
    "rt_scene_get_flat": (C.c_int, [_vp, _vp, _vp]),
    "rt_intersect_stream": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(TraceStats)]),
    "rt_intersect_stream_device": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(TraceStats), _vp]),
    "rt_intersect_stream_noalloc": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp]),
    "rt_occluded_stream": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(TraceStats)]),
    "rt_occluded_stream_device": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(TraceStats), _vp]),
    "rt_intersect1": (C.c_int, [_vp, _vp, C.POINTER(C.c_uint32), C.POINTER(TraceStats)]),
    "rt_occluded1": (C.c_int, [_vp, _vp, C.POINTER(C.c_uint8), C.POINTER(TraceStats)]),
    "rt_render": (C.c_int, [_vp, C.POINTER(CameraC), C.POINTER(LightC), C.POINTER(RenderParams), _vp,
                            C.POINTER(TraceStats)]),
    "rt_render_device": (C.c_int, [_vp, C.POINTER(CameraC), C.POINTER(LightC), C.POINTER(RenderParams),
                                   _vp, C.POINTER(TraceStats), _vp]),
    "rt_shard_plan_make": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, _vp]),
    "rt_node_render": (C.c_int, [_vp, C.POINTER(CameraC), C.POINTER(LightC), C.POINTER(RenderParams), C.c_uint32,
                                 C.c_uint32, C.c_int, _vp, _vp, C.POINTER(TraceStats), _vp]),
    "rt_generate_primary": (C.c_int, [_vp, C.POINTER(CameraC), C.POINTER(RenderParams), _vp]),
    "rt_resolve": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp]),
}
for _name, (_res, _args) in EXPORTS.items():
    _fn = getattr(_lib, _name)  # AttributeError here = the library lacks a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


def _check(rc):
    if rc != 0:
        raise RtError(rc, _lib.rt_last_error().decode("utf-8", "replace"))


def _p(a):
    return a.ctypes.data_as(_vp)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def camera_c(cam):
    return CameraC(_f3(cam.eye), _f3(cam.target), _f3(cam.up), float(cam.fov_deg))


def light_c(light):
    return LightC(_f3(light.pos), _f3(light.intensity))


def version():
    return _lib.rt_version().decode()


def make_rays(origins, directions, near=0.0, far=np.inf):
    """StreamRay::new (src/ray.rs:17-24) on arrays."""
    o = np.asarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.asarray(directions, dtype=np.float32).reshape(-1, 3)
    rays = np.zeros(o.shape[0], dtype=RAY_DTYPE)
    rays["origin_far"][:, :3] = o
    rays["origin_far"][:, 3] = np.float32(far) if np.isscalar(far) else np.asarray(far, np.float32)
    rays["direction"][:, :3] = d
    with np.errstate(divide="ignore"):
        rays["dir_recip_near"][:, :3] = np.float32(1.0) / d
    rays["dir_recip_near"][:, 3] = np.float32(near) if np.isscalar(near) else np.asarray(near, np.float32)
    return rays


def build_bvh_host(positions, indices, threads=0):
    """Bvh::build (src/bvh/build.rs:45-63) on the host only; returns
    (nodes, order, info dict, BvhStats dict)."""
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    nt = idx.shape[0]
    nodes = np.zeros(nt, dtype=NODE_DTYPE)
    order = np.zeros(nt, dtype=np.uint32)
    info, st = SceneInfo(), BvhStats()
    _check(_lib.rt_bvh_build_host(_p(pos), pos.shape[0], _p(idx), nt, int(threads), _p(nodes), _p(order),
                                  C.byref(info), C.byref(st)))
    inf = {n: int(getattr(info, n)) for n, _ in SceneInfo._fields_}
    return nodes[:inf["num_inner_nodes"]].copy(), order, inf, st.as_dict()


def build_bvh_search_host(positions, indices, buckets=32, max_leaf=3, threads=0):
    """rt_bvh_build_search_host: the builder of the fast forms' tree; same returns as build_bvh_host."""
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    nt = idx.shape[0]
    nodes = np.zeros(nt, dtype=NODE_DTYPE)
    order = np.zeros(nt, dtype=np.uint32)
    info, st = SceneInfo(), BvhStats()
    _check(_lib.rt_bvh_build_search_host(_p(pos), pos.shape[0], _p(idx), nt, int(threads), int(buckets), int(max_leaf),
                                         _p(nodes), _p(order), C.byref(info), C.byref(st)))
    inf = {n: int(getattr(info, n)) for n, _ in SceneInfo._fields_}
    return nodes[:inf["num_inner_nodes"]].copy(), order, inf, st.as_dict()


def build_bvh_search_device(ctx, positions, indices, buckets=32, max_leaf=3):
    """rt_bvh_build_search_device: the search builder run on the GPU of `ctx`; same returns as
    build_bvh_search_host (nodes numbered breadth first)."""
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    nt = idx.shape[0]
    nodes = np.zeros(nt, dtype=NODE_DTYPE)
    order = np.zeros(nt, dtype=np.uint32)
    info, st = SceneInfo(), BvhStats()
    _check(_lib.rt_bvh_build_search_device(ctx.h, _p(pos), pos.shape[0], _p(idx), nt, int(buckets), int(max_leaf),
                                           _p(nodes), _p(order), C.byref(info), C.byref(st)))
    inf = {n: int(getattr(info, n)) for n, _ in SceneInfo._fields_}
    return nodes[:inf["num_inner_nodes"]].copy(), order, inf, st.as_dict()


def wide_bvh_host(nodes, root_ref, root_count, num_triangles=None):
    """rt_bvh_wide_host: 4-wide forms of a canonical flat tree.

    Without num_triangles: (exact wide nodes, wide root ref, wide depth).  With it (the tree must
    have leaves of at most 3 triangles): (exact wide nodes, quantized 48-byte nodes, first unit of
    every node, first unit of every leaf slot, units in all, wide root ref, wide depth)."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    out = np.zeros(max(len(nodes), 1), dtype=WIDE_DTYPE)
    n, root, depth, units = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    if num_triangles is None:
        _check(_lib.rt_bvh_wide_host(_p(nodes), len(nodes), int(root_ref), int(root_count), 0, _p(out), None, None,
                                     None, C.byref(n), C.byref(root), C.byref(depth), None))
        return out[:n.value].copy(), int(root.value), int(depth.value)
    qout = np.zeros(max(len(nodes), 1), dtype=QWIDE_DTYPE)
    node_unit = np.zeros(max(len(nodes), 1), dtype=np.uint32)
    slot_unit = np.zeros(max(int(num_triangles), 1), dtype=np.uint32)
    _check(_lib.rt_bvh_wide_host(_p(nodes), len(nodes), int(root_ref), int(root_count), int(num_triangles), _p(out),
                                 _p(qout), _p(node_unit), _p(slot_unit), C.byref(n), C.byref(root), C.byref(depth),
                                 C.byref(units)))
    return (out[:n.value].copy(), qout[:n.value].copy(), node_unit[:n.value].copy(), slot_unit[:int(num_triangles)].copy(),
            int(units.value), int(root.value), int(depth.value))


def refine_bvh_host(positions, indices, nodes, root_ref, root_count, order, max_tris=3):
    """rt_bvh_refine_host -> (refined flat nodes, refined leaf order, root ref, root count, extra depth)."""
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    order = np.ascontiguousarray(order, dtype=np.uint32)
    nt = idx.shape[0]
    xnodes = np.zeros(len(nodes) + nt, dtype=NODE_DTYPE)
    xorder = np.zeros(nt, dtype=np.uint32)
    n, root, count, extra = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    _check(_lib.rt_bvh_refine_host(_p(pos), pos.shape[0], _p(idx), nt, _p(nodes), len(nodes), int(root_ref),
                                   int(root_count), _p(order), int(max_tris), _p(xnodes), _p(xorder), C.byref(n),
                                   C.byref(root), C.byref(count), C.byref(extra)))
    return xnodes[:n.value].copy(), xorder, int(root.value), int(count.value), int(extra.value)


SHARD_SAMPLE, SHARD_TILE = 0, 1


class ShardPlan(C.Structure):
    _fields_ = [("samples", C.c_uint32), ("sample_offset", C.c_uint32), ("shard_index", C.c_uint32),
                ("shard_count", C.c_uint32)]


def shard_plan(rank, world, mode, samples, sample_offset=0):
    """rt_shard_plan_make -> dict(samples, sample_offset, shard_index, shard_count)."""
    out = ShardPlan()
    _check(_lib.rt_shard_plan_make(int(rank), int(world), {"sample": 0, "tile": 1}.get(mode, mode), int(samples),
                                   int(sample_offset), C.byref(out)))
    return {n: int(getattr(out, n)) for n, _ in ShardPlan._fields_}


class Context:
    """One per GPU (rt_ctx)."""

    def __init__(self, device=0):
        h = _vp()
        _check(_lib.rt_ctx_create(int(device), C.byref(h)))
        self.h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "h", None):
            _lib.rt_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        _check(_lib.rt_ctx_set_option(self.h, name.encode(), int(value)))

    def set_max_paths(self, n):
        _check(_lib.rt_ctx_set_max_paths(self.h, int(n)))

    def last_kernel_ms(self):
        """({extend, connect, other} milliseconds, launches) of the last call."""
        ms = (C.c_float * 3)()
        ln = (C.c_uint32 * 3)()
        _check(_lib.rt_ctx_last_kernel_ms(self.h, C.byref(ms), C.byref(ln)))
        names = ("extend", "connect", "other")
        return {n: float(ms[i]) for i, n in enumerate(names)}, {n: int(ln[i]) for i, n in enumerate(names)}

    def debug_sort_keys(self, keys, bits=32):
        """Indices 0..n-1 ordered (stably) by the top `bits` bits of u32 keys."""
        k = np.ascontiguousarray(keys, dtype=np.uint32)
        perm = np.zeros(k.shape[0], dtype=np.uint32)
        _check(_lib.rt_debug_sort_keys(self.h, _p(k), k.shape[0], int(bits), _p(perm)))
        return perm

    def last_launch_ms(self, cap=4096):
        """[(kind, ms)] of the last call's timed launches in issue order; kind is
        'extend', 'connect' or 'other'."""
        ms = np.zeros(cap, np.float32)
        kind = np.zeros(cap, np.uint32)
        n = C.c_uint32()
        _check(_lib.rt_ctx_last_launch_ms(self.h, _p(ms), _p(kind), cap, C.byref(n)))
        names = ("extend", "connect", "other")
        return [(names[int(kind[i])], float(ms[i])) for i in range(min(cap, n.value))]

    def last_max_ray_steps(self):
        out = (C.c_uint64 * 2)()
        _check(_lib.rt_ctx_last_max_ray_steps(self.h, C.byref(out)))
        return int(out[0]), int(out[1])

    def last_depth_stats(self):
        """Per-depth counters of the last render call with statistics: (closest-hit, any-hit)."""
        out = (DepthStats * 2)()
        _check(_lib.rt_ctx_last_depth_stats(self.h, C.byref(out)))
        return out[0].as_dict(), out[1].as_dict()

    def last_pass_paths(self):
        """Path slots per pass the last render call planned with (smaller than max_paths after a
        failed workspace allocation)."""
        out = C.c_uint64(0)
        _check(_lib.rt_ctx_last_pass_paths(self.h, C.byref(out)))
        return int(out.value)

    def last_max_stack_entries(self):
        out = (C.c_uint64 * 2)()
        _check(_lib.rt_ctx_last_max_stack_entries(self.h, C.byref(out)))
        return int(out[0]), int(out[1])

    def last_stats_split(self):
        """Counters of the last call with statistics: (closest-hit, any-hit)."""
        out = (TraceStats * 2)()
        _check(_lib.rt_ctx_last_stats_split(self.h, C.byref(out)))
        return out[0].as_dict(), out[1].as_dict()

    def generate_primary(self, cam, width, height, samples, tile_size=16, seed=1235468, sample_offset=0):
        p = RenderParams(width, height, samples, sample_offset, 0, tile_size, 0, 1, 0, 0, seed)
        rays = np.zeros(width * height * samples, dtype=RAY_DTYPE)
        cc = camera_c(cam)
        _check(_lib.rt_generate_primary(self.h, C.byref(cc), C.byref(p), _p(rays)))
        return rays

    def resolve(self, accum):
        a = np.ascontiguousarray(accum, dtype=np.float32)
        h, w = a.shape[:2]
        rgb = np.zeros((h, w, 3), dtype=np.uint8)
        _check(_lib.rt_resolve(self.h, _p(a), w, h, _p(rgb)))
        return rgb


class Scene:
    """Device-resident BVH + triangles (rt_scene); the `Bvh` of src/bvh.rs:30-34."""

    def __init__(self, ctx, positions, indices, geom_normals):
        self.ctx = ctx
        pos = np.ascontiguousarray(positions, dtype=np.float32)
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        nrm = np.ascontiguousarray(geom_normals, dtype=np.float32)
        if pos.ndim != 2 or pos.shape[1] != 3 or idx.ndim != 2 or idx.shape[1] != 3 or nrm.shape != idx.shape:
            raise ValueError("positions [nv,3], indices [nt,3], geom_normals [nt,3] expected")
        h = _vp()
        _check(_lib.rt_scene_create(ctx.h, _p(pos), pos.shape[0], _p(idx), idx.shape[0], _p(nrm), C.byref(h)))
        self.h = h
        self.num_triangles = idx.shape[0]

    def close(self):
        if getattr(self, "h", None):
            _lib.rt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bounds(self):
        """(lo, hi) of the mesh's vertices."""
        lo, hi = (C.c_float * 3)(), (C.c_float * 3)()
        _check(_lib.rt_scene_bounds(self.h, C.byref(lo), C.byref(hi)))
        return np.array(lo[:], np.float32), np.array(hi[:], np.float32)

    def fast_tree_on_device(self):
        """0: the tree of the fast traversal forms came from the host builder, 1: from the device-side
        builder, 2: and its traversal items were made on the device too."""
        out = C.c_int(0)
        _check(_lib.rt_scene_fast_tree_on_device(self.h, C.byref(out)))
        return int(out.value)

    def wide_items(self):
        """The fast forms' traversal items as [units, 4] uint32 (rt_scene_get_wide_items)."""
        n = C.c_uint64(0)
        _check(_lib.rt_scene_get_wide_items(self.h, None, 0, C.byref(n)))
        out = np.zeros((int(n.value), 4), dtype=np.uint32)
        _check(_lib.rt_scene_get_wide_items(self.h, _p(out), n.value, C.byref(n)))
        return out

    def bvh_stats(self):
        st = BvhStats()
        _check(_lib.rt_scene_stats(self.h, C.byref(st)))
        return st.as_dict()

    def info(self):
        i = SceneInfo()
        _check(_lib.rt_scene_get_info(self.h, C.byref(i)))
        return {n: int(getattr(i, n)) for n, _ in SceneInfo._fields_}

    def flatten(self):
        inf = self.info()
        nodes = np.zeros(inf["num_inner_nodes"], dtype=NODE_DTYPE)
        tris = np.zeros(inf["num_triangles"], dtype=TRI_DTYPE)
        _check(_lib.rt_scene_get_flat(self.h, _p(nodes), _p(tris)))
        return nodes, tris

    # -- host-buffer trace calls ------------------------------------------------
    def intersect_stream(self, rays, stats=True):
        """Returns (hit ids u32, hit distances f32, stats dict or None); `rays` is not modified."""
        r = np.ascontiguousarray(rays.copy())
        hits = np.full(r.shape[0], NO_HIT, dtype=np.uint32)
        st = TraceStats()
        _check(_lib.rt_intersect_stream(self.h, _p(r), _p(hits), r.shape[0], C.byref(st) if stats else None))
        return hits, r["origin_far"][:, 3].copy(), (st.as_dict() if stats else None)

    def occluded_stream(self, rays, stats=True):
        r = np.ascontiguousarray(rays)
        occ = np.zeros(r.shape[0], dtype=np.uint8)
        st = TraceStats()
        _check(_lib.rt_occluded_stream(self.h, _p(r), _p(occ), r.shape[0], C.byref(st) if stats else None))
        return occ, (st.as_dict() if stats else None)

    def intersect1(self, ray):
        """Bvh::intersect1 (src/bvh/intersect1.rs:109-210) on one ray (a length-1 RAY_DTYPE array):
        (hit id, hit distance, stats dict)."""
        r = np.ascontiguousarray(ray.copy()).reshape(1)
        hit = C.c_uint32(NO_HIT)
        st = TraceStats()
        _check(_lib.rt_intersect1(self.h, _p(r), C.byref(hit), C.byref(st)))
        return int(hit.value), np.float32(r["origin_far"][0, 3]), st.as_dict()

    def occluded1(self, ray):
        """Bvh::occluded1 (src/bvh/intersect1.rs:7-106) on one ray: (bool, stats dict)."""
        r = np.ascontiguousarray(ray).reshape(1)
        occ = C.c_uint8(0)
        st = TraceStats()
        _check(_lib.rt_occluded1(self.h, _p(r), C.byref(occ), C.byref(st)))
        return bool(occ.value), st.as_dict()

    def render(self, cam, light, width, height, samples, max_bounces=2, tile_size=16, seed=1235468,
               sample_offset=0, shard_index=0, shard_count=1, samples_per_pass=0, accum=None,
               visit_stats=False):
        if accum is None:
            accum = np.zeros((height, width, 4), dtype=np.float32)
        assert accum.dtype == np.float32 and accum.flags["C_CONTIGUOUS"] and accum.shape == (height, width, 4)
        p = RenderParams(width, height, samples, sample_offset, max_bounces, tile_size, shard_index,
                         shard_count, samples_per_pass, RENDER_VISIT_STATS if visit_stats else 0, seed)
        st = TraceStats()
        cc, lc = camera_c(cam), light_c(light)
        _check(_lib.rt_render(self.h, C.byref(cc), C.byref(lc), C.byref(p), _p(accum), C.byref(st)))
        return accum, st.as_dict()

    def node_render_device(self, cam, light, accum_ptr, width, height, samples, rank, world, mode, nccl_comm=None,
                           max_bounces=2, tile_size=16, seed=1235468, sample_offset=0, samples_per_pass=0,
                           stream=0, stats=True):
        """rt_node_render: this rank's part of the frame, then (nccl_comm given) the sum-reduce to rank 0."""
        p = RenderParams(width, height, samples, sample_offset, max_bounces, tile_size, 0, 1, samples_per_pass, 0, seed)
        st = TraceStats()
        cc, lc = camera_c(cam), light_c(light)
        _check(_lib.rt_node_render(self.h, C.byref(cc), C.byref(lc), C.byref(p), int(rank), int(world),
                                   {"sample": 0, "tile": 1}.get(mode, mode), _vp(nccl_comm), _vp(accum_ptr),
                                   C.byref(st) if stats else None, _vp(stream)))
        return st.as_dict() if stats else None

    # -- device-pointer calls (inputs resident in HBM) --------------------------
    def intersect_stream_device(self, rays_ptr, hits_ptr, n, stream=0, stats=False):
        st = TraceStats()
        _check(_lib.rt_intersect_stream_device(self.h, _vp(rays_ptr), _vp(hits_ptr), int(n),
                                               C.byref(st) if stats else None, _vp(stream)))
        return st.as_dict() if stats else None

    def intersect_stream_noalloc(self, rays_ptr, hits_ptr, n, stream=0):
        """rt_intersect_stream_noalloc: closest hit on device buffers without any allocation."""
        _check(_lib.rt_intersect_stream_noalloc(self.h, _vp(rays_ptr), _vp(hits_ptr), int(n), _vp(stream)))

    def occluded_stream_device(self, rays_ptr, occ_ptr, n, stream=0, stats=False):
        st = TraceStats()
        _check(_lib.rt_occluded_stream_device(self.h, _vp(rays_ptr), _vp(occ_ptr), int(n),
                                              C.byref(st) if stats else None, _vp(stream)))
        return st.as_dict() if stats else None

    def render_device(self, cam, light, accum_ptr, width, height, samples, max_bounces=2, tile_size=16,
                      seed=1235468, sample_offset=0, shard_index=0, shard_count=1, samples_per_pass=0,
                      stream=0, stats=True, visit_stats=False):
        p = RenderParams(width, height, samples, sample_offset, max_bounces, tile_size, shard_index,
                         shard_count, samples_per_pass, RENDER_VISIT_STATS if visit_stats else 0, seed)
        st = TraceStats()
        cc, lc = camera_c(cam), light_c(light)
        _check(_lib.rt_render_device(self.h, C.byref(cc), C.byref(lc), C.byref(p), _vp(accum_ptr),
                                     C.byref(st) if stats else None, _vp(stream)))
        return st.as_dict() if stats else None
