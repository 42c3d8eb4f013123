/*
 * rt_oracle_internal.h -- TEST INFRASTRUCTURE ONLY (see rt_oracle.h).
 *
 * Types and helpers shared by the two restatements of the reference's traversal: the scalar one
 * (rt_oracle.c) and the literal SSE/AVX2 one (rt_oracle_simd.c).  Not part of the oracle's interface.
 */
#ifndef RT_ORACLE_INTERNAL_H
#define RT_ORACLE_INTERNAL_H

#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_NONE 0xFFFFFFFFu
#define ORC_LEAF_BIT 0x80000000u

/* ------------------------------------------------------------------------ */
/* small vector helpers.  [3P nalgebra 0.34.1] cross = (ay*bz-az*by,
 * az*bx-ax*bz, ax*by-ay*bx); dot = (ax*bx+ay*by)+az*bz; all un-fused.        */

typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_load(const float *p) { return v3_make(p[0], p[1], p[2]); }
static inline void v3_store(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_div(v3 a, float s) { return v3_make(a.x / s, a.y / s, a.z / s); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b)
{
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 v3_normalize(v3 a) { return v3_div(a, sqrtf(v3_dot(a, a))); }

/* ------------------------------------------------------------------------ */
/* src/aabb.rs:11-66                                                         */

typedef struct { v3 min, max; } aabb;

static inline float minf_(float a, float b) { return a < b ? a : b; }
static inline float maxf_(float a, float b) { return a > b ? a : b; }

static inline aabb aabb_empty(void)
{
    aabb b = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
    return b;
}
static inline void aabb_grow(aabb *b, v3 p) /* aabb.rs:50-53 */
{
    b->min = v3_make(minf_(b->min.x, p.x), minf_(b->min.y, p.y), minf_(b->min.z, p.z));
    b->max = v3_make(maxf_(b->max.x, p.x), maxf_(b->max.y, p.y), maxf_(b->max.z, p.z));
}
static inline aabb aabb_join(aabb a, aabb o) /* aabb.rs:56-61 */
{
    aabb r;
    r.min = v3_make(minf_(a.min.x, o.min.x), minf_(a.min.y, o.min.y), minf_(a.min.z, o.min.z));
    r.max = v3_make(maxf_(a.max.x, o.max.x), maxf_(a.max.y, o.max.y), maxf_(a.max.z, o.max.z));
    return r;
}
static inline float aabb_surface_area(aabb b) /* aabb.rs:37-40 */
{
    v3 s = v3_sub(b.max, b.min);
    return 2.f * ((s.x * s.y + s.x * s.z) + s.y * s.z);
}

/* ------------------------------------------------------------------------ */
/* src/triangle_opt.rs:19-42  Moeller-Trumbore, un-fused, no back-face cull. */

static inline float tri_intersect(v3 v0, v3 v0v1, v3 v0v2, v3 o, v3 d)
{
    v3 pvec = v3_cross(d, v0v1);
    float det = v3_dot(v0v2, pvec);
    if (fabsf(det) < 1.1920929e-7f) /* f32::EPSILON */
        return INFINITY;
    float inv_det = 1.f / det;
    v3 tvec = v3_sub(o, v0);
    float u = v3_dot(tvec, pvec) * inv_det;
    if (!(u >= 0.f && u <= 1.f))
        return INFINITY;
    v3 qvec = v3_cross(tvec, v0v2);
    float v = v3_dot(d, qvec) * inv_det;
    if (v < 0.f || u + v > 1.f)
        return INFINITY;
    return v3_dot(v0v1, qvec) * inv_det;
}

/* ------------------------------------------------------------------------ */
/* src/bvh.rs:19-34 pointer tree                                             */

typedef struct node {
    int is_leaf;
    /* inner: lanes {minL,minR,maxL,maxR} per axis (bvh.rs:21-22) */
    float bx[4], by[4], bz[4];
    struct node *child[2];
    /* leaf: triangles_range */
    uint32_t first, count;
} node;

typedef struct { v3 v0v1, v0v2, v0; } tri_opt;

struct orc_scene {
    uint32_t num_vertices, num_triangles;
    float *positions;      /* 3*nv */
    uint32_t *indices;     /* 3*nt */
    float *geom_normals;   /* 3*nt */
    tri_opt *triangles;    /* leaf order (bvh.rs:31) */
    uint32_t *object_indices; /* leaf slot -> mesh triangle (bvh.rs:32) */
    node *root;
    uint32_t num_inner;
    orc_bvh_stats bstats;
};

static inline void stats_inner(orc_stats *st, uint64_t rays)
{
    st->inner_visit += 1;
    st->inner_rays += rays;
}
static inline void stats_leaf(orc_stats *st, uint64_t rays, uint64_t objs)
{
    st->leaf_visit += 1;
    st->leaf_rays += rays;
    st->leaf_objs += objs;
    st->obj_intersect += rays * objs;
}

typedef struct { const node *n; uint32_t list; size_t start; } stream_item;

/* scratch lists are sized 32*n entries each like intersect_stream.rs:22-23. */
typedef struct {
    uint32_t *buf;
    size_t cap;
} stream_scratch;

static uint32_t *scratch_get(stream_scratch *sc, size_t entries)
{
    if (sc->cap < entries) {
        free(sc->buf);
        sc->buf = (uint32_t *)malloc(entries * sizeof(uint32_t));
        sc->cap = entries;
    }
    return sc->buf;
}


/* The four traversal entry points a tile integrator is run with (scalar or SIMD restatement). */
typedef struct orc_trace_fns {
    void (*intersect_stream)(const orc_scene *s, orc_ray *rays, uint32_t *hits, uint32_t n, orc_stats *st,
                             stream_scratch *sc);
    void (*occluded_stream)(const orc_scene *s, orc_ray *rays, uint8_t *occluded, uint32_t n, orc_stats *st,
                            stream_scratch *sc);
    uint32_t (*intersect1)(const orc_scene *s, orc_ray *ray, orc_stats *st);
    int (*occluded1)(const orc_scene *s, const orc_ray *ray, orc_stats *st);
} orc_trace_fns;
extern const orc_trace_fns orc_simd_trace_fns; /* rt_oracle_simd.c */

#endif
