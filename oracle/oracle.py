"""ctypes binding of the CPU oracle (oracle/librt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py.  The product package never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librt_oracle.so")

NONE = 0xFFFFFFFF

RAY_DTYPE = np.dtype([("origin_far", np.float32, 4),
                      ("dir_recip_near", np.float32, 4),
                      ("direction", np.float32, 4)])
NODE_DTYPE = np.dtype([("bx", np.float32, 4), ("by", np.float32, 4), ("bz", np.float32, 4),
                       ("left", np.uint32), ("right", np.uint32),
                       ("left_count", np.uint32), ("right_count", np.uint32)])
TRI_DTYPE = np.dtype([("v0", np.float32, 3), ("v0v1", np.float32, 3), ("v0v2", np.float32, 3),
                      ("id", np.uint32), ("pad", np.uint32, 2)])


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("total_rays", "inner_visit", "inner_rays", "leaf_visit",
                                          "leaf_rays", "leaf_objs", "obj_intersect")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class BvhStats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("num_leaves", "num_objs", "max_depth", "max_objs")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class CameraC(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("target", C.c_float * 3), ("up", C.c_float * 3),
                ("fov_deg", C.c_float)]


class LightC(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("intensity", C.c_float * 3)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples", C.c_uint32),
                ("sample_offset", C.c_uint32), ("max_bounces", C.c_uint32),
                ("tile_size", C.c_uint32), ("shard_index", C.c_uint32),
                ("shard_count", C.c_uint32), ("mode", C.c_uint32), ("nthreads", C.c_uint32),
                ("seed", C.c_uint64)]


MODE_STREAM = 0
MODE_STREAM_SHADOW_IMMEDIATE = 1
MODE_SINGLE_RAY = 3
MODE_SIMD = 0x100  # or-ed into a mode: traversal through the SSE/AVX2 restatement


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in (
                "rt_oracle.c", "rt_oracle_simd.c", "rt_oracle.h", "rt_oracle_internal.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None

# flags of the baseline build: full optimisation for the host it runs on, arithmetic still un-fused
NATIVE_CFLAGS = ["-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=gnu11", "-pthread"]


def build_native(out_dir):
    """Compiles rt_oracle.c for the CPU this runs on (bench.py's cpu_baseline leg: the committed
    library targets x86-64-v3 so that it runs on any box).  Returns (path, flags) or (None, reason)."""
    out = os.path.join(out_dir, "librt_oracle_native.so")
    cmd = ["gcc"] + NATIVE_CFLAGS + ["-mavx2", "-mfma", "-shared", "-o", out, os.path.join(_HERE, "rt_oracle.c"),
                                     os.path.join(_HERE, "rt_oracle_simd.c"), "-lm", "-lpthread"]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError) as e:
        return None, repr(e)
    return out, " ".join(NATIVE_CFLAGS)


def use_library(path):
    """Binds another build of the same source (before the first call into the library)."""
    global _LIB_PATH, _lib
    _LIB_PATH = path
    _lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp = C.c_void_p
        L.orc_scene_create.restype = vp
        L.orc_scene_create.argtypes = [vp, C.c_uint32, vp, C.c_uint32, vp]
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_scene_bvh_stats.argtypes = [vp, C.POINTER(BvhStats)]
        L.orc_scene_num_inner.restype = C.c_uint32
        L.orc_scene_num_inner.argtypes = [vp]
        L.orc_scene_num_triangles.restype = C.c_uint32
        L.orc_scene_num_triangles.argtypes = [vp]
        L.orc_scene_root.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_scene_flatten.argtypes = [vp, vp, vp]
        L.orc_tri_intersect.restype = C.c_float
        L.orc_tri_intersect.argtypes = [vp] * 5
        L.orc_intersect1_n.argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(Stats)]
        L.orc_occluded1_n.argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(Stats)]
        L.orc_intersect_stream.argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(Stats)]
        L.orc_occluded_stream.argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(Stats)]
        L.orc_intersect_brute.argtypes = [vp, vp, vp, C.c_uint32]
        for name in ("orc_simd_intersect1_n", "orc_simd_occluded1_n", "orc_simd_intersect_stream",
                     "orc_simd_occluded_stream"):
            getattr(L, name).argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(Stats)]
        L.orc_sample_light.restype = C.c_int
        L.orc_sample_light.argtypes = [vp, vp, vp, C.POINTER(LightC), vp, C.POINTER(C.c_float), vp]
        L.orc_sample_diffuse_ray.argtypes = [vp, vp, C.c_float, C.c_float, vp, vp]
        L.orc_sincos_2pi.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_rng_u32.restype = C.c_uint32
        L.orc_rng_u32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_rng_f32.restype = C.c_float
        L.orc_rng_f32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_camera_sample_dir.argtypes = [C.POINTER(CameraC), C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.c_uint32, C.c_float, C.c_float, vp]
        L.orc_generate_primary.argtypes = [C.POINTER(CameraC), C.POINTER(RenderParams), vp]
        L.orc_render.argtypes = [vp, C.POINTER(CameraC), C.POINTER(LightC),
                                 C.POINTER(RenderParams), vp, C.POINTER(Stats)]
        L.orc_resolve.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def camera_c(cam):
    return CameraC(_f3(cam.eye), _f3(cam.target), _f3(cam.up), float(cam.fov_deg))


def light_c(light):
    return LightC(_f3(light.pos), _f3(light.intensity))


def make_rays(origins, directions, near=0.0, far=np.inf):
    """StreamRay::new (src/ray.rs:17-24) on arrays."""
    o = np.asarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.asarray(directions, dtype=np.float32).reshape(-1, 3)
    n = o.shape[0]
    rays = np.zeros(n, dtype=RAY_DTYPE)
    rays["origin_far"][:, :3] = o
    rays["origin_far"][:, 3] = np.float32(far) if np.isscalar(far) else np.asarray(far, np.float32)
    rays["direction"][:, :3] = d
    with np.errstate(divide="ignore"):
        rays["dir_recip_near"][:, :3] = np.float32(1.0) / d
    rays["dir_recip_near"][:, 3] = np.float32(near) if np.isscalar(near) else np.asarray(near, np.float32)
    return rays


class Scene:
    def __init__(self, positions, indices, geom_normals):
        self.positions = np.ascontiguousarray(positions, dtype=np.float32)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32)
        self.geom_normals = np.ascontiguousarray(geom_normals, dtype=np.float32)
        self.h = lib().orc_scene_create(_p(self.positions), self.positions.shape[0],
                                        _p(self.indices), self.indices.shape[0],
                                        _p(self.geom_normals))
        if not self.h:
            raise RuntimeError("orc_scene_create failed")

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bvh_stats(self):
        st = BvhStats()
        lib().orc_scene_bvh_stats(self.h, C.byref(st))
        return st.as_dict()

    def num_inner(self):
        return lib().orc_scene_num_inner(self.h)

    def root(self):
        a, b = C.c_uint32(), C.c_uint32()
        lib().orc_scene_root(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def flatten(self):
        nodes = np.zeros(self.num_inner(), dtype=NODE_DTYPE)
        tris = np.zeros(self.indices.shape[0], dtype=TRI_DTYPE)
        lib().orc_scene_flatten(self.h, _p(nodes), _p(tris))
        return nodes, tris

    def _trace(self, fn, rays, out_dtype):
        rays = np.ascontiguousarray(rays.copy())
        out = np.zeros(rays.shape[0], dtype=out_dtype)
        st = Stats()
        fn(self.h, _p(rays), _p(out), rays.shape[0], C.byref(st))
        return out, rays["origin_far"][:, 3].copy(), st.as_dict()

    def intersect1(self, rays):
        return self._trace(lib().orc_intersect1_n, rays, np.uint32)

    def occluded1(self, rays):
        o, _, st = self._trace(lib().orc_occluded1_n, rays, np.uint8)
        return o, st

    def intersect_stream(self, rays):
        return self._trace(lib().orc_intersect_stream, rays, np.uint32)

    def occluded_stream(self, rays):
        o, _, st = self._trace(lib().orc_occluded_stream, rays, np.uint8)
        return o, st

    # the same four traversals through the SSE/AVX2 restatement (rt_oracle_simd.c)
    def simd_intersect1(self, rays):
        return self._trace(lib().orc_simd_intersect1_n, rays, np.uint32)

    def simd_occluded1(self, rays):
        o, _, st = self._trace(lib().orc_simd_occluded1_n, rays, np.uint8)
        return o, st

    def simd_intersect_stream(self, rays):
        return self._trace(lib().orc_simd_intersect_stream, rays, np.uint32)

    def simd_occluded_stream(self, rays):
        o, _, st = self._trace(lib().orc_simd_occluded_stream, rays, np.uint8)
        return o, st

    def intersect_brute(self, rays):
        rays = np.ascontiguousarray(rays.copy())
        out = np.zeros(rays.shape[0], dtype=np.uint32)
        lib().orc_intersect_brute(self.h, _p(rays), _p(out), rays.shape[0])
        return out, rays["origin_far"][:, 3].copy()

    def render(self, cam, light, width, height, samples, max_bounces=2, tile_size=16,
               seed=1235468, sample_offset=0, shard_index=0, shard_count=1,
               mode=MODE_STREAM_SHADOW_IMMEDIATE, nthreads=None, accum=None):
        if nthreads is None:
            nthreads = min(os.cpu_count() or 1, 16)
        if accum is None:
            accum = np.zeros((height, width, 4), dtype=np.float32)
        p = RenderParams(width, height, samples, sample_offset, max_bounces, tile_size,
                         shard_index, shard_count, mode, nthreads, seed)
        st = Stats()
        cc, lc = camera_c(cam), light_c(light)
        lib().orc_render(self.h, C.byref(cc), C.byref(lc), C.byref(p), _p(accum), C.byref(st))
        return accum, st.as_dict()


def generate_primary(cam, width, height, samples, tile_size=16, seed=1235468, sample_offset=0):
    p = RenderParams(width, height, samples, sample_offset, 0, tile_size, 0, 1, 1, 1, seed)
    rays = np.zeros(width * height * samples, dtype=RAY_DTYPE)
    cc = camera_c(cam)
    lib().orc_generate_primary(C.byref(cc), C.byref(p), _p(rays))
    return rays


def resolve(accum):
    h, w = accum.shape[:2]
    rgb = np.zeros((h, w, 3), dtype=np.uint8)
    a = np.ascontiguousarray(accum, dtype=np.float32)
    lib().orc_resolve(_p(a), w, h, _p(rgb))
    return rgb


def tri_intersect(v0, e1, e2, o, d):
    a = [np.ascontiguousarray(x, dtype=np.float32) for x in (v0, e1, e2, o, d)]
    return float(lib().orc_tri_intersect(*[_p(x) for x in a]))


def sample_light(wo, n, p, light):
    wo, n, p = [np.ascontiguousarray(x, dtype=np.float32) for x in (wo, n, p)]
    wi = np.zeros(3, np.float32)
    w = np.zeros(3, np.float32)
    dist = C.c_float()
    lc = light_c(light)
    ok = lib().orc_sample_light(_p(wo), _p(n), _p(p), C.byref(lc), _p(wi), C.byref(dist), _p(w))
    return bool(ok), wi, float(dist.value), w


def sample_diffuse_ray(wo, n, u0, u1):
    wo, n = [np.ascontiguousarray(x, dtype=np.float32) for x in (wo, n)]
    wi = np.zeros(3, np.float32)
    w = np.zeros(3, np.float32)
    lib().orc_sample_diffuse_ray(_p(wo), _p(n), u0, u1, _p(wi), _p(w))
    return wi, w


def sincos_2pi(u):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos_2pi(u, C.byref(s), C.byref(c))
    return float(s.value), float(c.value)


def rng_f32(seed, pixel, sample, dim):
    return float(lib().orc_rng_f32(seed, pixel, sample, dim))


def camera_sample_dir(cam, width, height, tile_size, x, y, u0, u1):
    d = np.zeros(3, np.float32)
    cc = camera_c(cam)
    lib().orc_camera_sample_dir(C.byref(cc), width, height, tile_size, x, y, u0, u1, _p(d))
    return d
