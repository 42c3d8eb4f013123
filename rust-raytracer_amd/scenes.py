"""Procedural stand-in scenes.

The reference loads `scenes/asian_dragon_obj/asian_dragon.obj` and
`scenes/San_Miguel/san-miguel.obj` (src/scene.rs:93-130); in the reference
checkout every asset is a Git-LFS pointer, so no geometry exists.  These
generators build deterministic meshes of matching triangle count and character
(Cornell-class ~36, Sponza-class ~262 K, dragon-class ~7.2 M, San-Miguel-class
~10 M).  Cameras and lights for the dragon and San Miguel stand-ins are the
reference's own (src/scene.rs:103-109, 119-125); Cornell and Sponza have no
definition in the reference and theirs are invented here.

A mesh is (positions f32[nv,3], indices u32[nt,3], geom_normals f32[nt,3]):
the indexed form src/mesh.rs:9-15 holds after loading.  Zero-area triangles are
dropped like src/scene.rs:53-59 does, and the geometric normal is
normalize(v0v1 x v0v2), flipped to the side of the generator's shading normal
(src/mesh.rs:180-188).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

F32 = np.float32


@dataclass
class Camera:
    """Inputs of Camera::new (src/camera.rs:25)."""
    eye: tuple
    target: tuple
    up: tuple = (0.0, 0.0, 1.0)
    fov_deg: float = 60.0


@dataclass
class Light:
    """PointLight::new (src/light.rs:10-15): intensity = color * power."""
    pos: tuple
    color: tuple = (1.0, 1.0, 1.0)
    power: float = 1.0

    @property
    def intensity(self):
        return tuple(np.float32(c) * np.float32(self.power) for c in self.color)


@dataclass
class SceneData:
    name: str
    positions: np.ndarray
    indices: np.ndarray
    geom_normals: np.ndarray
    camera: Camera
    light: Light
    notes: str = ""
    extra: dict = field(default_factory=dict)

    @property
    def num_triangles(self):
        return int(self.indices.shape[0])


class _MeshParts:
    """Accumulates indexed parts; each part carries per-vertex shading normals
    that are only used to orient the geometric normal."""

    def __init__(self):
        self.pos = []
        self.idx = []
        self.nrm = []
        self.nv = 0

    def add(self, pos, idx, nrm):
        pos = np.ascontiguousarray(pos, dtype=F32).reshape(-1, 3)
        nrm = np.ascontiguousarray(nrm, dtype=F32).reshape(-1, 3)
        idx = np.ascontiguousarray(idx, dtype=np.int64).reshape(-1, 3)
        assert pos.shape == nrm.shape
        self.pos.append(pos)
        self.nrm.append(nrm)
        self.idx.append(idx + self.nv)
        self.nv += pos.shape[0]

    def finish(self):
        pos = np.concatenate(self.pos, axis=0)
        nrm = np.concatenate(self.nrm, axis=0)
        idx = np.concatenate(self.idx, axis=0)
        return finalize_mesh(pos, idx, nrm)


def finalize_mesh(pos, idx, vertex_normals=None):
    """Zero-area skip (src/scene.rs:53-59) and geometric normals
    (src/mesh.rs:180-188), all in f32."""
    pos = np.ascontiguousarray(pos, dtype=F32)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    v0 = pos[idx[:, 0]]
    e1 = pos[idx[:, 1]] - v0
    e2 = pos[idx[:, 2]] - v0
    c = np.empty_like(e1)
    c[:, 0] = e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1]
    c[:, 1] = e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2]
    c[:, 2] = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
    magsq = (c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]) + c[:, 2] * c[:, 2]
    keep = (magsq != 0) & np.isfinite(magsq)
    idx = idx[keep]
    c = c[keep]
    ng = c / np.sqrt(magsq[keep])[:, None]
    ng = ng.astype(F32)
    if vertex_normals is not None:
        n0 = np.ascontiguousarray(vertex_normals, dtype=F32)[idx[:, 0]]
        d = (ng[:, 0] * n0[:, 0] + ng[:, 1] * n0[:, 1]) + ng[:, 2] * n0[:, 2]
        ng[d < 0] *= F32(-1)
    return pos, np.ascontiguousarray(idx.astype(np.uint32)), np.ascontiguousarray(ng)


# --------------------------------------------------------------------------
# primitives

def grid_patch(fn, nu, nv, u0=0.0, u1=1.0, v0=0.0, v1=1.0, wrap_u=False):
    """Tessellate P(u,v) -> (pos, nrm) on an nu x nv quad grid (2 tris/quad)."""
    cu = nu if wrap_u else nu + 1
    us = np.linspace(u0, u1, nu + 1, dtype=np.float64)[:cu]
    vs = np.linspace(v0, v1, nv + 1, dtype=np.float64)
    U, V = np.meshgrid(us, vs, indexing="ij")
    pos, nrm = fn(U.reshape(-1), V.reshape(-1))
    i = np.arange(nu, dtype=np.int64)[:, None]
    j = np.arange(nv, dtype=np.int64)[None, :]
    i1 = (i + 1) % cu if wrap_u else i + 1
    a = (i * (nv + 1) + j).reshape(-1)
    b = (i1 * (nv + 1) + j).reshape(-1)
    c = (i1 * (nv + 1) + j + 1).reshape(-1)
    d = (i * (nv + 1) + j + 1).reshape(-1)
    idx = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], axis=0)
    return pos, idx, nrm


def plane_patch(origin, eu, ev, nu, nv, normal):
    origin = np.asarray(origin, dtype=np.float64)
    eu = np.asarray(eu, dtype=np.float64)
    ev = np.asarray(ev, dtype=np.float64)
    normal = np.asarray(normal, dtype=np.float64)

    def fn(u, v):
        p = origin[None, :] + u[:, None] * eu[None, :] + v[:, None] * ev[None, :]
        return p, np.broadcast_to(normal, p.shape)

    return grid_patch(fn, nu, nv)


def box_parts(parts, lo, hi, n=1, inward=False, skip=()):
    """Axis-aligned box from six plane patches (n x n quads per face)."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    s = -1.0 if inward else 1.0
    ex = np.array([hi[0] - lo[0], 0, 0])
    ey = np.array([0, hi[1] - lo[1], 0])
    ez = np.array([0, 0, hi[2] - lo[2]])
    faces = {
        "-x": (lo, ey, ez, (-s, 0, 0)),
        "+x": (lo + ex, ey, ez, (s, 0, 0)),
        "-y": (lo, ex, ez, (0, -s, 0)),
        "+y": (lo + ey, ex, ez, (0, s, 0)),
        "-z": (lo, ex, ey, (0, 0, -s)),
        "+z": (lo + ez, ex, ey, (0, 0, s)),
    }
    for name, (o, eu, ev, nrm) in faces.items():
        if name in skip:
            continue
        parts.add(*plane_patch(o, eu, ev, n, n, nrm))


def cylinder_parts(parts, base, radius, height, nseg, nh, axis=2, bulge=0.0):
    base = np.asarray(base, dtype=np.float64)

    def fn(u, v):
        ang = 2 * np.pi * u
        r = radius * (1.0 + bulge * np.sin(np.pi * v) + 0.02 * np.cos(12 * ang))
        ca, sa = np.cos(ang), np.sin(ang)
        loc = np.stack([r * ca, r * sa, v * height], 1)
        nl = np.stack([ca, sa, np.zeros_like(ca)], 1)
        if axis == 0:
            loc = loc[:, [2, 0, 1]]
            nl = nl[:, [2, 0, 1]]
        elif axis == 1:
            loc = loc[:, [1, 2, 0]]
            nl = nl[:, [1, 2, 0]]
        return base[None, :] + loc, nl

    parts.add(*grid_patch(fn, nseg, nh, wrap_u=True))


def ellipsoid_parts(parts, center, radii, nu, nv, disp=None):
    center = np.asarray(center, dtype=np.float64)
    radii = np.asarray(radii, dtype=np.float64)

    def fn(u, v):
        th = 2 * np.pi * u
        ph = np.pi * v
        d = np.stack([np.sin(ph) * np.cos(th), np.sin(ph) * np.sin(th), np.cos(ph)], 1)
        r = np.ones_like(u) if disp is None else disp(u, v, d)
        p = center[None, :] + d * radii[None, :] * r[:, None]
        return p, d

    parts.add(*grid_patch(fn, nu, nv, wrap_u=True))


def leaf_cloud_parts(parts, rng, n, center, radii, size):
    """n small randomly oriented triangles inside an ellipsoid: foliage-like,
    high depth complexity, incoherent for secondary rays."""
    center = np.asarray(center, dtype=np.float64)
    radii = np.asarray(radii, dtype=np.float64)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = rng.random(n) ** (1.0 / 3.0)
    c = center[None, :] + d * r[:, None] * radii[None, :]
    a = rng.normal(size=(n, 3))
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.normal(size=(n, 3))
    b -= a * np.sum(a * b, axis=1, keepdims=True)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    s = size * (0.6 + 0.8 * rng.random(n))[:, None]
    p0 = c - 0.5 * s * a - 0.3 * s * b
    p1 = c + 0.5 * s * a - 0.3 * s * b
    p2 = c + 0.6 * s * b
    pos = np.stack([p0, p1, p2], 1).reshape(-1, 3)
    nrm = np.repeat(np.cross(a, b), 3, axis=0)
    idx = np.arange(3 * n, dtype=np.int64).reshape(n, 3)
    parts.add(pos, idx, nrm)


# --------------------------------------------------------------------------
# scenes

def cornell_class():
    """Cornell-class box: 5 walls + 2 boxes = 34 triangles, Z-up.
    Camera and light are invented (the reference CLI has no Cornell scene,
    src/args.rs:13-17)."""
    parts = _MeshParts()
    box_parts(parts, (0, 0, 0), (1, 1, 1), n=1, inward=True, skip=("-y",))  # open towards the camera
    # short box and tall box, rotated about Z
    def rot_box(lo, hi, ang, pivot):
        sub = _MeshParts()
        box_parts(sub, lo, hi, n=1)
        pos = np.concatenate(sub.pos, 0).astype(np.float64)
        nrm = np.concatenate(sub.nrm, 0).astype(np.float64)
        idx = np.concatenate(sub.idx, 0)
        ca, sa = np.cos(ang), np.sin(ang)
        R = np.array([[ca, -sa, 0], [sa, ca, 0], [0, 0, 1]])
        pv = np.asarray(pivot, dtype=np.float64)
        pos = (pos - pv) @ R.T + pv
        nrm = nrm @ R.T
        keep = []  # drop the bottom face (it lies in the floor)
        for t in idx:
            keep.append(not np.all(np.abs(pos[t][:, 2]) < 1e-9))
        parts.add(pos, idx[np.array(keep)], nrm)

    rot_box((0.55, 0.15, 0.0), (0.85, 0.45, 0.30), 0.30, (0.7, 0.3, 0.0))
    rot_box((0.18, 0.50, 0.0), (0.48, 0.80, 0.60), -0.32, (0.33, 0.65, 0.0))
    pos, idx, ng = parts.finish()
    return SceneData("cornell-class", pos, idx, ng,
                     Camera(eye=(0.5, -1.6, 0.5), target=(0.5, 0.5, 0.5), fov_deg=45.0),
                     Light(pos=(0.5, 0.5, 0.92), power=0.6),
                     notes="invented camera/light")


def sponza_class(detail=1.0):
    """Atrium stand-in, ~262 K triangles at detail=1: tessellated floor, two
    storeys of colonnades (cylinders), arches, draped curtains, outer walls.
    Camera and light are invented."""
    parts = _MeshParts()
    L, W, H = 30.0, 12.0, 12.0
    s = float(np.sqrt(detail))
    def q(n):
        return max(1, int(round(n * s)))
    # floor / ceiling rim / walls
    parts.add(*plane_patch((-L / 2, -W / 2, 0), (L, 0, 0), (0, W, 0), q(120), q(48), (0, 0, 1)))
    parts.add(*plane_patch((-L / 2, -W / 2, H), (L, 0, 0), (0, W, 0), q(60), q(24), (0, 0, -1)))
    parts.add(*plane_patch((-L / 2, -W / 2, 0), (L, 0, 0), (0, 0, H), q(60), q(24), (0, 1, 0)))
    parts.add(*plane_patch((-L / 2, W / 2, 0), (L, 0, 0), (0, 0, H), q(60), q(24), (0, -1, 0)))
    parts.add(*plane_patch((-L / 2, -W / 2, 0), (0, W, 0), (0, 0, H), q(24), q(24), (1, 0, 0)))
    parts.add(*plane_patch((L / 2, -W / 2, 0), (0, W, 0), (0, 0, H), q(24), q(24), (-1, 0, 0)))
    # upper gallery floors
    for ysign in (-1, 1):
        y0 = ysign * W / 2 if ysign < 0 else W / 2 - 2.5
        box_parts(parts, (-L / 2, y0, 5.6), (L / 2, y0 + 2.5, 6.0), n=q(8))
    # colonnades: 2 sides x 2 storeys x 12 columns
    ncol = 12
    for ysign in (-1, 1):
        y = ysign * (W / 2 - 2.5)
        for k in range(ncol):
            x = -L / 2 + (k + 0.5) * L / ncol
            cylinder_parts(parts, (x, y, 0.0), 0.38, 5.6, q(40), q(40), bulge=0.06)
            cylinder_parts(parts, (x, y, 6.0), 0.28, 4.6, q(36), q(32), bulge=0.04)
            box_parts(parts, (x - 0.5, y - 0.5, 0.0), (x + 0.5, y + 0.5, 0.25), n=q(3))
            box_parts(parts, (x - 0.45, y - 0.45, 5.3), (x + 0.45, y + 0.45, 5.6), n=q(3))
    # arches between ground-floor columns (half-torus tubes)
    for ysign in (-1, 1):
        y = ysign * (W / 2 - 2.5)
        for k in range(ncol - 1):
            xc = -L / 2 + (k + 1.0) * L / ncol
            R = 0.5 * L / ncol - 0.05

            def arch(u, v, xc=xc, y=y, R=R):
                a = np.pi * v
                t = 2 * np.pi * u
                cx = xc + R * np.cos(a)
                cz = 4.2 + 1.2 * np.sin(a)
                rr = 0.16
                nx = np.cos(a) * np.cos(t)
                nz = np.sin(a) * np.cos(t)
                ny = np.sin(t)
                p = np.stack([cx + rr * nx, y + rr * ny, cz + rr * nz], 1)
                return p, np.stack([nx, ny, nz], 1)

            parts.add(*grid_patch(arch, q(16), q(28), wrap_u=True))
    # draped curtains hanging along the atrium axis
    for k, (y, ph) in enumerate(((-1.2, 0.0), (0.0, 1.3), (1.2, 2.1))):
        def drape(u, v, y=y, ph=ph):
            x = -L / 2 + 2.0 + u * (L - 4.0)
            z = 11.0 - 4.5 * v
            yy = y + 0.25 * np.sin(9.0 * x + ph) * (0.3 + v) + 0.1 * np.sin(23.0 * x + 3.0 * z)
            p = np.stack([x, yy, z], 1)
            return p, np.broadcast_to(np.array([0.0, 1.0, 0.0]), p.shape)

        parts.add(*grid_patch(drape, q(150), q(48)))
    # a lion-head-ish lump on the far wall and some urns (dense small objects)
    def lump(u, v, d):
        return 1.0 + 0.15 * np.sin(7 * 2 * np.pi * u) * np.sin(5 * np.pi * v) + 0.08 * np.cos(19 * 2 * np.pi * u + 11 * np.pi * v)
    ellipsoid_parts(parts, (-L / 2 + 0.4, 0.0, 4.0), (0.5, 1.1, 1.1), q(90), q(60), disp=lump)
    for k in range(8):
        x = -L / 2 + 3.0 + k * (L - 6.0) / 7
        ellipsoid_parts(parts, (x, 0.0, 0.55), (0.42, 0.42, 0.55), q(48), q(36), disp=lump)
    pos, idx, ng = parts.finish()
    return SceneData("sponza-class", pos, idx, ng,
                     Camera(eye=(L / 2 - 2.0, 0.6, 2.2), target=(-L / 2, -0.4, 3.6), fov_deg=60.0),
                     Light(pos=(2.0, 0.3, 9.5), power=60.0),
                     notes="invented camera/light")


def dragon_class(n=1900):
    """One dense displaced closed surface, 2*n*n triangles (~7.2 M at n=1900),
    unit scale; camera and light of the reference's AsianDragon scene
    (src/scene.rs:103-109)."""
    parts = _MeshParts()

    def disp(u, v, d):
        th = 2 * np.pi * u
        ph = np.pi * v
        r = 1.0
        r = r + 0.18 * np.sin(3 * th) * np.sin(2 * ph) ** 2
        r = r + 0.07 * np.sin(11 * th + 5 * ph) * np.sin(ph)
        r = r + 0.025 * np.sin(47 * th) * np.sin(31 * ph)
        r = r + 0.006 * np.sin(211 * th + 3.0) * np.sin(173 * ph + 1.0)   # scales
        r = r + 0.002 * np.sin(701 * th) * np.cos(557 * ph)
        return r

    ellipsoid_parts(parts, (0.0, 0.0, 0.42), (0.95, 0.36, 0.40), n, n, disp=disp)
    pos, idx, ng = parts.finish()
    eye = np.array([0.6, -1.0, 0.25], dtype=F32)
    eye = eye / np.sqrt((eye[0] * eye[0] + eye[1] * eye[1]) + eye[2] * eye[2]) * F32(2.5)
    return SceneData("dragon-class", pos, idx, ng,
                     Camera(eye=tuple(float(x) for x in eye), target=(0.0, 0.0, 0.35), fov_deg=60.0),
                     Light(pos=(5.0, -10.0, 5.0), power=300.0),
                     notes="reference camera/light (src/scene.rs:103-109)")


def san_miguel_class(scale=1.0, seed=20240607):
    """Courtyard stand-in, ~10 M triangles at scale=1: tessellated ground and
    arcade walls, columns, furniture, and many dense foliage clusters (small
    randomly oriented triangles).  Camera and light of the reference's SanMiguel
    scene (src/scene.rs:119-125)."""
    rng = np.random.default_rng(seed)
    parts = _MeshParts()
    s = float(np.sqrt(scale))

    def q(n):
        return max(1, int(round(n * s)))

    X0, X1, Y0, Y1, H = 2.0, 38.0, -11.0, 8.0, 9.0
    # ground with gentle cobble displacement
    def ground(u, v):
        x = X0 + u * (X1 - X0)
        y = Y0 + v * (Y1 - Y0)
        z = 0.015 * np.sin(9.1 * x) * np.sin(8.3 * y)
        p = np.stack([x, y, z], 1)
        return p, np.broadcast_to(np.array([0.0, 0.0, 1.0]), p.shape)
    parts.add(*grid_patch(ground, q(720), q(380)))
    # surrounding walls (two storeys)
    parts.add(*plane_patch((X0, Y0, 0), (X1 - X0, 0, 0), (0, 0, H), q(300), q(80), (0, 1, 0)))
    parts.add(*plane_patch((X0, Y1, 0), (X1 - X0, 0, 0), (0, 0, H), q(300), q(80), (0, -1, 0)))
    parts.add(*plane_patch((X0, Y0, 0), (0, Y1 - Y0, 0), (0, 0, H), q(160), q(80), (1, 0, 0)))
    parts.add(*plane_patch((X1, Y0, 0), (0, Y1 - Y0, 0), (0, 0, H), q(160), q(80), (-1, 0, 0)))
    # arcade: columns along both long sides + gallery slab
    ncol = 16
    for yy in (Y0 + 3.0, Y1 - 3.0):
        for k in range(ncol):
            x = X0 + (k + 0.5) * (X1 - X0) / ncol
            cylinder_parts(parts, (x, yy, 0.0), 0.30, 4.2, q(64), q(96), bulge=0.05)
            cylinder_parts(parts, (x, yy, 4.6), 0.22, 3.4, q(48), q(64), bulge=0.04)
        box_parts(parts, (X0, yy - 0.6, 4.2), (X1, yy + 0.6, 4.6), n=q(24))
    # furniture: tables (boxes + cylinder legs) and chairs
    for k in range(14):
        cx = 8.0 + 2.1 * k
        cy = -1.5 + 2.2 * np.sin(1.7 * k)
        cylinder_parts(parts, (cx, cy, 0.72), 0.55, 0.04, q(96), q(4))
        ellipsoid_parts(parts, (cx, cy, 0.76), (0.55, 0.55, 0.02), q(96), q(24))
        cylinder_parts(parts, (cx, cy, 0.0), 0.05, 0.72, q(24), q(24))
        for a in range(3):
            ang = 2.1 * a + 0.4 * k
            px, py = cx + 0.9 * np.cos(ang), cy + 0.9 * np.sin(ang)
            box_parts(parts, (px - 0.2, py - 0.2, 0.42), (px + 0.2, py + 0.2, 0.46), n=q(6))
            box_parts(parts, (px - 0.2, py + 0.17, 0.46), (px + 0.2, py + 0.2, 0.95), n=q(6))
            for lx in (-0.17, 0.17):
                for ly in (-0.17, 0.17):
                    cylinder_parts(parts, (px + lx, py + ly, 0.0), 0.02, 0.42, q(10), q(6))
    # trees: trunk + dense crowns of leaf triangles
    trees = [(12.0, -4.5), (19.0, 1.5), (24.5, -4.0), (31.0, 2.0), (7.0, 2.5), (34.0, -5.5)]
    leaves_per_tree = int(900_000 * scale)
    for (tx, ty) in trees:
        cylinder_parts(parts, (tx, ty, 0.0), 0.28, 3.6, q(64), q(96), bulge=-0.1)
        for c in range(3):
            off = rng.normal(size=3) * np.array([0.8, 0.8, 0.4])
            leaf_cloud_parts(parts, rng, leaves_per_tree // 3,
                             (tx + off[0], ty + off[1], 5.0 + off[2]), (2.4, 2.4, 1.7), 0.09)
    # potted plants and hedges: many small dense clusters near eye level
    nplants = 60
    leaves_per_plant = int(55_000 * scale)
    for k in range(nplants):
        px = X0 + 2.0 + rng.random() * (X1 - X0 - 4.0)
        py = Y0 + 3.8 + rng.random() * (Y1 - Y0 - 7.6)
        ellipsoid_parts(parts, (px, py, 0.25), (0.25, 0.25, 0.25), q(32), q(24))
        leaf_cloud_parts(parts, rng, leaves_per_plant, (px, py, 0.9), (0.45, 0.45, 0.55), 0.05)
    # ivy on the far wall
    leaf_cloud_parts(parts, rng, int(700_000 * scale), (X0 + 0.15, -1.5, 3.5), (0.12, 8.0, 3.2), 0.06)
    pos, idx, ng = parts.finish()
    return SceneData("san-miguel-class", pos, idx, ng,
                     Camera(eye=(28.0, -1.65, 1.8), target=(27.0, -1.65, 1.8), fov_deg=60.0),
                     Light(pos=(20.0, -2.5, 3.0), power=20.0),
                     notes="reference camera/light (src/scene.rs:119-125)")


def test_mesh(n_tris=1000, seed=7):
    """Small random triangle soup inside the unit cube plus a floor, for
    traversal parity tests and committed fixtures."""
    rng = np.random.default_rng(seed)
    parts = _MeshParts()
    parts.add(*plane_patch((-1, -1, -0.5), (2, 0, 0), (0, 2, 0), 4, 4, (0, 0, 1)))
    leaf_cloud_parts(parts, rng, max(1, n_tris - 32), (0, 0, 0.1), (0.8, 0.8, 0.5), 0.15)
    pos, idx, ng = parts.finish()
    return SceneData("test-mesh", pos, idx, ng,
                     Camera(eye=(0.2, -2.4, 0.5), target=(0.0, 0.0, 0.0), fov_deg=50.0),
                     Light(pos=(0.5, -1.0, 2.0), power=8.0))


SCENES = {
    "cornell-class": cornell_class,
    "sponza-class": sponza_class,
    "dragon-class": dragon_class,
    "san-miguel-class": san_miguel_class,
    "test-mesh": test_mesh,
}


def load(name, **kw):
    return SCENES[name](**kw)


def write_obj(scene: SceneData, path):
    """Write positions / one dummy vt / per-triangle vn, the subset of OBJ the
    reference requires (every face vertex needs v/vt/vn, src/scene.rs:46-51)."""
    with open(path, "w") as f:
        f.write("# %s\n" % scene.name)
        for p in scene.positions:
            f.write("v %.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
        f.write("vt 0 0\n")
        for n in scene.geom_normals:
            f.write("vn %.9g %.9g %.9g\n" % (n[0], n[1], n[2]))
        for t, tri in enumerate(scene.indices):
            f.write("f %d/1/%d %d/1/%d %d/1/%d\n" % (tri[0] + 1, t + 1, tri[1] + 1, t + 1, tri[2] + 1, t + 1))
