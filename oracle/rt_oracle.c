/*
 * rt_oracle.c -- TEST INFRASTRUCTURE ONLY (see rt_oracle.h).
 *
 * CPU restatement of the reference's hot path.  PARITY UNPINNED: the reference
 * holds no tests or golden vectors; see the header.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (Rust never contracts a*b+c;
 * the only fused operations are the explicit fmaf() calls in the node slab
 * tests, where the reference uses fused_mul_sub).
 */
#define _GNU_SOURCE
#include "rt_oracle_internal.h"

#include <pthread.h>
#include <stdatomic.h>

float orc_tri_intersect(const float v0[3], const float v0v1[3], const float v0v2[3],
                        const float origin[3], const float direction[3])
{
    return tri_intersect(v3_load(v0), v3_load(v0v1), v3_load(v0v2), v3_load(origin),
                         v3_load(direction));
}

/* src/ray.rs:17-24 -- 1/d is NOT clamped: +-inf for +-0. */
void orc_ray_new(orc_ray *r, const float origin[3], const float direction[3], float near_t,
                 float far_t)
{
    r->origin_far[0] = origin[0];
    r->origin_far[1] = origin[1];
    r->origin_far[2] = origin[2];
    r->origin_far[3] = far_t;
    r->direction[0] = direction[0];
    r->direction[1] = direction[1];
    r->direction[2] = direction[2];
    r->direction[3] = 0.f;
    r->dir_recip_near[0] = 1.f / direction[0];
    r->dir_recip_near[1] = 1.f / direction[1];
    r->dir_recip_near[2] = 1.f / direction[2];
    r->dir_recip_near[3] = near_t;
}

typedef struct { v3 centroid; aabb box; } obj_bound;

static inline uint32_t bucket_idx(float k0, float k1, float c) /* build.rs:117-118 */
{
    float f = k1 * (c - k0);
    /* Rust `as usize`: saturating, NaN -> 0 */
    if (!(f > 0.f))
        return 0;
    if (f >= 4294967296.f)
        return 0xFFFFFFFFu;
    return (uint32_t)f;
}

static inline float axis_of(v3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

/* src/bvh/build.rs:65-207 (the rayon::join there is a sequential recursion here;
 * partitioning is per node, so the tree is identical). */
static node *build_recursive(uint32_t *indices, uint32_t n, uint32_t start_idx,
                             const obj_bound *bounds_of, uint32_t depth, orc_scene *s)
{
    node *nd = (node *)calloc(1, sizeof(node));
    int make_leaf = (n >= 1 && n <= 8); /* build.rs:80-81 */
    aabb bounds = aabb_empty(), cb = aabb_empty();
    v3 size = {0, 0, 0};
    if (!make_leaf) {
        for (uint32_t i = 0; i < n; i++) { /* build.rs:83-89 */
            bounds = aabb_join(bounds, bounds_of[indices[i]].box);
            aabb_grow(&cb, bounds_of[indices[i]].centroid);
        }
        size = v3_sub(cb.max, cb.min);
        if (v3_dot(size, size) == 0.f) /* build.rs:93-101 */
            make_leaf = 1;
    }
    if (make_leaf) { /* build.rs:72-78, 18-23 */
        nd->is_leaf = 1;
        nd->first = start_idx;
        nd->count = n;
        s->bstats.num_leaves += 1;
        s->bstats.num_objs += n;
        if (depth > s->bstats.max_depth) s->bstats.max_depth = depth;
        if (n > s->bstats.max_objs) s->bstats.max_objs = n;
        return nd;
    }

    int axis = (size.x > size.y && size.x > size.z) ? 0 : (size.y > size.z ? 1 : 2); /* 103-109 */

    enum { BUCKETS = 8 };
    aabb bbox[BUCKETS];
    uint32_t bcount[BUCKETS];
    for (int i = 0; i < BUCKETS; i++) { bbox[i] = aabb_empty(); bcount[i] = 0; }

    float k0 = axis_of(cb.min, axis); /* build.rs:114-116 */
    float k1 = (float)BUCKETS * (1.f - 0.01f) / (axis_of(cb.max, axis) - axis_of(cb.min, axis));
    for (uint32_t i = 0; i < n; i++) { /* build.rs:119-126 */
        const obj_bound *ob = &bounds_of[indices[i]];
        uint32_t b = bucket_idx(k0, k1, axis_of(ob->centroid, axis));
        bbox[b] = aabb_join(bbox[b], ob->box);
        bcount[b] += 1;
    }

    uint32_t min_bucket = 0; /* build.rs:128-150 */
    float min_cost = INFINITY;
    aabb bounds_l = aabb_empty(), bounds_r = aabb_empty();
    for (int i = 0; i < BUCKETS - 1; i++) {
        aabb l = aabb_empty(), r = aabb_empty();
        uint32_t nl = 0, nr = 0;
        for (int j = 0; j <= i; j++) { l = aabb_join(l, bbox[j]); nl += bcount[j]; }
        for (int j = i + 1; j < BUCKETS; j++) { r = aabb_join(r, bbox[j]); nr += bcount[j]; }
        float cost = (aabb_surface_area(l) * (float)nl + aabb_surface_area(r) * (float)nr) /
                     aabb_surface_area(bounds);
        if (cost < min_cost) {
            min_bucket = (uint32_t)i;
            min_cost = cost;
            bounds_l = l;
            bounds_r = r;
        }
    }

    uint32_t left = 0, right = n - 1; /* build.rs:152-168 */
    for (;;) {
        while (bucket_idx(k0, k1, axis_of(bounds_of[indices[left]].centroid, axis)) <= min_bucket)
            left++;
        while (bucket_idx(k0, k1, axis_of(bounds_of[indices[right]].centroid, axis)) > min_bucket)
            right--;
        if (left >= right)
            break;
        uint32_t t = indices[left];
        indices[left] = indices[right];
        indices[right] = t;
    }
    uint32_t split = left; /* build.rs:170-171 */

    nd->is_leaf = 0;
    s->num_inner += 1;
    nd->bx[0] = bounds_l.min.x; nd->bx[1] = bounds_r.min.x; /* build.rs:196-202 */
    nd->bx[2] = bounds_l.max.x; nd->bx[3] = bounds_r.max.x;
    nd->by[0] = bounds_l.min.y; nd->by[1] = bounds_r.min.y;
    nd->by[2] = bounds_l.max.y; nd->by[3] = bounds_r.max.y;
    nd->bz[0] = bounds_l.min.z; nd->bz[1] = bounds_r.min.z;
    nd->bz[2] = bounds_l.max.z; nd->bz[3] = bounds_r.max.z;
    nd->child[0] = build_recursive(indices, split, start_idx, bounds_of, depth + 1, s);
    nd->child[1] =
        build_recursive(indices + split, n - split, start_idx + split, bounds_of, depth + 1, s);
    return nd;
}

static void free_tree(node *n)
{
    if (!n) return;
    if (!n->is_leaf) { free_tree(n->child[0]); free_tree(n->child[1]); }
    free(n);
}

orc_scene *orc_scene_create(const float *positions, uint32_t num_vertices,
                            const uint32_t *indices, uint32_t num_triangles,
                            const float *geom_normals)
{
    if (num_triangles == 0) return NULL;
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    s->num_vertices = num_vertices;
    s->num_triangles = num_triangles;
    s->positions = (float *)malloc(sizeof(float) * 3 * (size_t)num_vertices);
    s->indices = (uint32_t *)malloc(sizeof(uint32_t) * 3 * (size_t)num_triangles);
    s->geom_normals = (float *)malloc(sizeof(float) * 3 * (size_t)num_triangles);
    memcpy(s->positions, positions, sizeof(float) * 3 * (size_t)num_vertices);
    memcpy(s->indices, indices, sizeof(uint32_t) * 3 * (size_t)num_triangles);
    memcpy(s->geom_normals, geom_normals, sizeof(float) * 3 * (size_t)num_triangles);

    /* build.rs:46-47; centroid mesh.rs:107-110, aabb mesh.rs:97-104 */
    obj_bound *ob = (obj_bound *)malloc(sizeof(obj_bound) * (size_t)num_triangles);
    s->object_indices = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)num_triangles);
    for (uint32_t t = 0; t < num_triangles; t++) {
        v3 a = v3_load(positions + 3 * (size_t)indices[3 * (size_t)t + 0]);
        v3 b = v3_load(positions + 3 * (size_t)indices[3 * (size_t)t + 1]);
        v3 c = v3_load(positions + 3 * (size_t)indices[3 * (size_t)t + 2]);
        ob[t].centroid = v3_div(v3_add(v3_add(a, b), c), 3.f);
        ob[t].box = aabb_empty();
        aabb_grow(&ob[t].box, a);
        aabb_grow(&ob[t].box, b);
        aabb_grow(&ob[t].box, c);
        s->object_indices[t] = t;
    }
    s->root = build_recursive(s->object_indices, num_triangles, 0, ob, 0, s);
    free(ob);

    /* build.rs:53-56, triangle_opt.rs:72-81 */
    s->triangles = (tri_opt *)malloc(sizeof(tri_opt) * (size_t)num_triangles);
    for (uint32_t slot = 0; slot < num_triangles; slot++) {
        uint32_t t = s->object_indices[slot];
        v3 a = v3_load(positions + 3 * (size_t)indices[3 * (size_t)t + 0]);
        v3 b = v3_load(positions + 3 * (size_t)indices[3 * (size_t)t + 1]);
        v3 c = v3_load(positions + 3 * (size_t)indices[3 * (size_t)t + 2]);
        s->triangles[slot].v0v1 = v3_sub(b, a);
        s->triangles[slot].v0v2 = v3_sub(c, a);
        s->triangles[slot].v0 = a;
    }
    return s;
}

void orc_scene_destroy(orc_scene *s)
{
    if (!s) return;
    free_tree(s->root);
    free(s->positions);
    free(s->indices);
    free(s->geom_normals);
    free(s->triangles);
    free(s->object_indices);
    free(s);
}

void orc_scene_bvh_stats(const orc_scene *s, orc_bvh_stats *out) { *out = s->bstats; }
uint32_t orc_scene_num_inner(const orc_scene *s) { return s->num_inner; }
uint32_t orc_scene_num_triangles(const orc_scene *s) { return s->num_triangles; }

static uint32_t flatten_rec(const node *n, orc_flat_node *out, uint32_t *next, uint32_t *count)
{
    if (n->is_leaf) {
        *count = n->count;
        return ORC_LEAF_BIT | n->first;
    }
    uint32_t me = (*next)++;
    *count = 0;
    if (out) {
        memcpy(out[me].bx, n->bx, sizeof(n->bx));
        memcpy(out[me].by, n->by, sizeof(n->by));
        memcpy(out[me].bz, n->bz, sizeof(n->bz));
    }
    uint32_t lc, rc;
    uint32_t l = flatten_rec(n->child[0], out, next, &lc);
    uint32_t r = flatten_rec(n->child[1], out, next, &rc);
    if (out) {
        out[me].left = l;
        out[me].right = r;
        out[me].left_count = lc;
        out[me].right_count = rc;
    }
    return me;
}

void orc_scene_root(const orc_scene *s, uint32_t *root_ref, uint32_t *root_count)
{
    if (s->root->is_leaf) {
        *root_ref = ORC_LEAF_BIT | s->root->first;
        *root_count = s->root->count;
    } else {
        *root_ref = 0;
        *root_count = 0;
    }
}

void orc_scene_flatten(const orc_scene *s, orc_flat_node *nodes, orc_flat_tri *tris)
{
    uint32_t next = 0, cnt = 0;
    if (nodes) flatten_rec(s->root, nodes, &next, &cnt);
    if (tris) {
        for (uint32_t i = 0; i < s->num_triangles; i++) {
            v3_store(tris[i].v0, s->triangles[i].v0);
            v3_store(tris[i].v0v1, s->triangles[i].v0v1);
            v3_store(tris[i].v0v2, s->triangles[i].v0v2);
            tris[i].id = s->object_indices[i];
            tris[i].pad[0] = tris[i].pad[1] = 0;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* x86 min/max semantics: _mm_max_ps(a,b) = a > b ? a : b (second operand on
 * NaN), _mm_min_ps(a,b) = a < b ? a : b.  The reference's slab tests rely on
 * that operand order to stay conservative when bb*inf - o*inf is NaN.         */

static inline float mm_max(float a, float b) { return a > b ? a : b; }
static inline float mm_min(float a, float b) { return a < b ? a : b; }

/* slab test of src/bvh/intersect1.rs:144-179 (same in occluded1 41-76):
 * near/far box planes picked by the SIGN BIT of 1/d (blendv), t = fmsub(bb,
 * 1/d, o*(1/d)); tnear = max(max(tx,ty), max(tz,near)); tfar likewise.       */
static inline void slab1(const node *n, const float o_rd[3], const float rd[3], float near_t,
                         float far_t, int *hit_l, int *hit_r)
{
    const float *bb[3] = {n->bx, n->by, n->bz};
    float tn[2][3], tf[2][3];
    for (int a = 0; a < 3; a++) {
        int neg = signbit(rd[a]) != 0;
        for (int c = 0; c < 2; c++) {
            float bnear = neg ? bb[a][2 + c] : bb[a][c];
            float bfar = neg ? bb[a][c] : bb[a][2 + c];
            tn[c][a] = fmaf(bnear, rd[a], -o_rd[a]);
            tf[c][a] = fmaf(bfar, rd[a], -o_rd[a]);
        }
    }
    float tnear[2], tfar[2];
    for (int c = 0; c < 2; c++) {
        tnear[c] = mm_max(mm_max(tn[c][0], tn[c][1]), mm_max(tn[c][2], near_t));
        tfar[c] = mm_min(mm_min(tf[c][0], tf[c][1]), mm_min(tf[c][2], far_t));
    }
    *hit_l = tnear[0] <= tfar[0];
    *hit_r = tnear[1] <= tfar[1];
}

#define STACK1 32 /* intersect1.rs:12,115 ArrayVec<_,32>; overflow panics there */

/* src/bvh/intersect1.rs:109-210.  The near/far register is loaded once before
 * the loop (128-129), so node culling uses the INITIAL far; only triangle
 * acceptance sees the shrinking far. */
uint32_t orc_intersect1(const orc_scene *s, orc_ray *ray, orc_stats *st)
{
    orc_stats dummy;
    if (!st) { memset(&dummy, 0, sizeof dummy); st = &dummy; }
    st->total_rays += 1;
    uint32_t hit_obj = ORC_NONE;
    const node *stack[STACK1 * 4];
    int sp = 0;
    float rd[3] = {ray->dir_recip_near[0], ray->dir_recip_near[1], ray->dir_recip_near[2]};
    float o_rd[3] = {ray->origin_far[0] * rd[0], ray->origin_far[1] * rd[1],
                     ray->origin_far[2] * rd[2]};
    float near_t = ray->dir_recip_near[3], far0 = ray->origin_far[3];
    v3 o = v3_load(ray->origin_far), d = v3_load(ray->direction);
    const node *cur = s->root;
    while (cur) {
        if (!cur->is_leaf) {
            stats_inner(st, 1);
            int hl, hr;
            slab1(cur, o_rd, rd, near_t, far0, &hl, &hr);
            if (hl) {
                if (hr) stack[sp++] = cur->child[1];
                cur = cur->child[0];
            } else if (hr) {
                cur = cur->child[1];
            } else {
                cur = sp ? stack[--sp] : NULL;
            }
        } else {
            stats_leaf(st, 1, cur->count);
            for (uint32_t i = cur->first; i < cur->first + cur->count; i++) {
                const tri_opt *t = &s->triangles[i];
                float h = tri_intersect(t->v0, t->v0v1, t->v0v2, o, d);
                if (h >= near_t && h < ray->origin_far[3]) { /* intersect1.rs:198 */
                    ray->origin_far[3] = h;
                    hit_obj = s->object_indices[i];
                }
            }
            cur = sp ? stack[--sp] : NULL;
        }
    }
    return hit_obj;
}

/* src/bvh/intersect1.rs:7-106 */
int orc_occluded1(const orc_scene *s, const orc_ray *ray, orc_stats *st)
{
    orc_stats dummy;
    if (!st) { memset(&dummy, 0, sizeof dummy); st = &dummy; }
    st->total_rays += 1;
    const node *stack[STACK1 * 4];
    int sp = 0;
    float rd[3] = {ray->dir_recip_near[0], ray->dir_recip_near[1], ray->dir_recip_near[2]};
    float o_rd[3] = {ray->origin_far[0] * rd[0], ray->origin_far[1] * rd[1],
                     ray->origin_far[2] * rd[2]};
    float near_t = ray->dir_recip_near[3], far_t = ray->origin_far[3];
    v3 o = v3_load(ray->origin_far), d = v3_load(ray->direction);
    const node *cur = s->root;
    while (cur) {
        if (!cur->is_leaf) {
            stats_inner(st, 1);
            int hl, hr;
            slab1(cur, o_rd, rd, near_t, far_t, &hl, &hr);
            if (hl) {
                if (hr) stack[sp++] = cur->child[1];
                cur = cur->child[0];
            } else if (hr) {
                cur = cur->child[1];
            } else {
                cur = sp ? stack[--sp] : NULL;
            }
        } else {
            stats_leaf(st, 1, cur->count);
            for (uint32_t i = cur->first; i < cur->first + cur->count; i++) {
                const tri_opt *t = &s->triangles[i];
                float h = tri_intersect(t->v0, t->v0v1, t->v0v2, o, d);
                if (h >= near_t && h < far_t) /* intersect1.rs:95 */
                    return 1;
            }
            cur = sp ? stack[--sp] : NULL;
        }
    }
    return 0;
}

void orc_intersect1_n(const orc_scene *s, orc_ray *rays, uint32_t *hits, uint32_t n,
                      orc_stats *st)
{
    for (uint32_t i = 0; i < n; i++) hits[i] = orc_intersect1(s, &rays[i], st);
}

void orc_occluded1_n(const orc_scene *s, const orc_ray *rays, uint8_t *occluded, uint32_t n,
                     orc_stats *st)
{
    for (uint32_t i = 0; i < n; i++) occluded[i] = (uint8_t)orc_occluded1(s, &rays[i], st);
}

/* ------------------------------------------------------------------------ */
/* Ray-stream traversal, src/bvh/intersect_stream.rs.                        */

/* slab test of intersect_stream.rs:116-146.  The AVX2 code keeps the far half
 * negated so one max covers both halves; max on negated values equals min on
 * the values themselves with the same second-operand-on-NaN rule, so:
 *   tnear = max(max(max(tx,ty),tz),near)   tfar = min(min(min(tx,ty),tz),far)
 * bit0 hitL = nearL<=farL, bit1 hitR = nearR<=farR, bit2 = nearL<=nearR.     */
static inline unsigned slab_stream(const node *n, const orc_ray *ray)
{
    const float *bb[3] = {n->bx, n->by, n->bz};
    float tn[2][3], tf[2][3];
    for (int a = 0; a < 3; a++) {
        float rd = ray->dir_recip_near[a];
        float o_rd = ray->origin_far[a] * rd;
        int neg = signbit(rd) != 0;
        for (int c = 0; c < 2; c++) {
            float bnear = neg ? bb[a][2 + c] : bb[a][c];
            float bfar = neg ? bb[a][c] : bb[a][2 + c];
            tn[c][a] = fmaf(bnear, rd, -o_rd);
            tf[c][a] = fmaf(bfar, rd, -o_rd);
        }
    }
    float near_t = ray->dir_recip_near[3], far_t = ray->origin_far[3];
    float tnear[2], tfar[2];
    for (int c = 0; c < 2; c++) {
        tnear[c] = mm_max(mm_max(mm_max(tn[c][0], tn[c][1]), tn[c][2]), near_t);
        tfar[c] = mm_min(mm_min(mm_min(tf[c][0], tf[c][1]), tf[c][2]), far_t);
    }
    unsigned m = 0;
    if (tnear[0] <= tfar[0]) m |= 1u;
    if (tnear[1] <= tfar[1]) m |= 2u;
    if (tnear[0] <= tnear[1]) m |= 4u;
    return m;
}

/* intersect_stream.rs:10-198 + intersect_objs 384-426 */
static void intersect_stream_impl(const orc_scene *s, orc_ray *rays, uint32_t *hits, uint32_t n,
                                  orc_stats *st, stream_scratch *sc)
{
    st->total_rays += n;
    if (n == 0) return;
    size_t list_len = (size_t)n * 32;
    uint32_t *base = scratch_get(sc, list_len * 3);
    uint32_t *lists[3] = {base, base + list_len, base + 2 * list_len};
    size_t sizes[3] = {0, 0, 0};
    for (uint32_t i = 0; i < n; i++) lists[0][i] = i;
    sizes[0] = n;
    stream_item stack[64 * 4];
    int sp = 0;
    stack[sp].n = s->root; stack[sp].list = 0; stack[sp].start = 0; sp++;
    while (sp) {
        stream_item it = stack[--sp];
        size_t a = it.start, last = sizes[it.list];
        sizes[it.list] = it.start;
        const node *nd = it.n;
        if (!nd->is_leaf) {
            stats_inner(st, last - a);
            size_t orig[3] = {sizes[0], sizes[1], sizes[2]};
            for (; a < last; a++) {
                uint32_t r = lists[it.list][a];
                unsigned m = slab_stream(nd, &rays[r]);
                unsigned hl = m & 1u, hr = (m >> 1) & 1u, lf = (m >> 2) & 1u;
                lists[0][sizes[0]] = r;
                lists[1][sizes[1]] = r;
                lists[2][sizes[2]] = r;
                sizes[0] += hl & lf;
                sizes[1] += hr;
                sizes[2] += hl & (lf ^ 1u);
            }
            if (sizes[2] - orig[2] > 0) { stack[sp].n = nd->child[0]; stack[sp].list = 2; stack[sp].start = orig[2]; sp++; }
            if (sizes[1] - orig[1] > 0) { stack[sp].n = nd->child[1]; stack[sp].list = 1; stack[sp].start = orig[1]; sp++; }
            if (sizes[0] - orig[0] > 0) { stack[sp].n = nd->child[0]; stack[sp].list = 0; stack[sp].start = orig[0]; sp++; }
        } else {
            stats_leaf(st, last - a, nd->count);
            for (; a < last; a++) {
                uint32_t r = lists[it.list][a];
                orc_ray *ray = &rays[r];
                v3 o = v3_load(ray->origin_far), d = v3_load(ray->direction);
                for (uint32_t i = nd->first; i < nd->first + nd->count; i++) {
                    const tri_opt *t = &s->triangles[i];
                    float h = tri_intersect(t->v0, t->v0v1, t->v0v2, o, d);
                    if (h >= ray->dir_recip_near[3] && h < ray->origin_far[3]) {
                        ray->origin_far[3] = h;
                        hits[r] = s->object_indices[i];
                    }
                }
            }
        }
    }
}

/* intersect_stream.rs:200-382 + occluded_objs 428-469.  The reference skips a
 * ray that is already occluded only when it is the first of a pair (258-261);
 * a pair's second ray is processed regardless.  Pairing is restated so the
 * visit counters match. */
static void occluded_stream_impl(const orc_scene *s, orc_ray *rays, uint8_t *occluded, uint32_t n,
                                 orc_stats *st, stream_scratch *sc)
{
    st->total_rays += n;
    if (n == 0) return;
    size_t list_len = (size_t)n * 32;
    uint32_t *base = scratch_get(sc, list_len * 2);
    uint32_t *lists[2] = {base, base + list_len};
    size_t sizes[2] = {0, 0};
    for (uint32_t i = 0; i < n; i++) lists[0][i] = i;
    sizes[0] = n;
    stream_item stack[64 * 4];
    int sp = 0;
    stack[sp].n = s->root; stack[sp].list = 0; stack[sp].start = 0; sp++;
    while (sp) {
        stream_item it = stack[--sp];
        size_t a = it.start, last = sizes[it.list];
        sizes[it.list] = it.start;
        const node *nd = it.n;
        if (!nd->is_leaf) {
            stats_inner(st, last - a);
            size_t orig[2] = {sizes[0], sizes[1]};
            while (a < last) {
                uint32_t ra = lists[it.list][a];
                if (occluded[ra]) { a += 1; continue; }
                uint32_t rb = (a + 1 < last) ? lists[it.list][a + 1] : ra;
                unsigned m = slab_stream(nd, &rays[ra]);
                lists[0][sizes[0]] = ra;
                lists[1][sizes[1]] = ra;
                sizes[0] += m & 1u;
                sizes[1] += (m >> 1) & 1u;
                if (ra != rb) {
                    unsigned mb = slab_stream(nd, &rays[rb]);
                    lists[0][sizes[0]] = rb;
                    lists[1][sizes[1]] = rb;
                    sizes[0] += mb & 1u;
                    sizes[1] += (mb >> 1) & 1u;
                }
                a += 2;
            }
            if (sizes[1] - orig[1] > 0) { stack[sp].n = nd->child[1]; stack[sp].list = 1; stack[sp].start = orig[1]; sp++; }
            if (sizes[0] - orig[0] > 0) { stack[sp].n = nd->child[0]; stack[sp].list = 0; stack[sp].start = orig[0]; sp++; }
        } else {
            stats_leaf(st, last - a, nd->count);
            for (; a < last; a++) {
                uint32_t r = lists[it.list][a];
                orc_ray *ray = &rays[r];
                v3 o = v3_load(ray->origin_far), d = v3_load(ray->direction);
                for (uint32_t i = nd->first; i < nd->first + nd->count; i++) {
                    const tri_opt *t = &s->triangles[i];
                    float h = tri_intersect(t->v0, t->v0v1, t->v0v2, o, d);
                    if (h >= ray->dir_recip_near[3] && h < ray->origin_far[3]) {
                        ray->origin_far[3] = h;
                        occluded[r] = 1;
                    }
                }
            }
        }
    }
}

void orc_intersect_stream(const orc_scene *s, orc_ray *rays, uint32_t *hits, uint32_t n,
                          orc_stats *st)
{
    orc_stats dummy;
    if (!st) { memset(&dummy, 0, sizeof dummy); st = &dummy; }
    stream_scratch sc = {0, 0};
    for (uint32_t i = 0; i < n; i++) hits[i] = ORC_NONE; /* caller's resize(None), integrate_stream.rs:95-96 */
    /* chunks keep the 3*32*n scratch bounded; rays are independent */
    const uint32_t chunk = 1u << 16;
    for (uint32_t off = 0; off < n; off += chunk) {
        uint32_t m = n - off < chunk ? n - off : chunk;
        intersect_stream_impl(s, rays + off, hits + off, m, st, &sc);
    }
    free(sc.buf);
}

void orc_occluded_stream(const orc_scene *s, orc_ray *rays, uint8_t *occluded, uint32_t n,
                         orc_stats *st)
{
    orc_stats dummy;
    if (!st) { memset(&dummy, 0, sizeof dummy); st = &dummy; }
    stream_scratch sc = {0, 0};
    const uint32_t chunk = 1u << 16;
    for (uint32_t off = 0; off < n; off += chunk) {
        uint32_t m = n - off < chunk ? n - off : chunk;
        occluded_stream_impl(s, rays + off, occluded + off, m, st, &sc);
    }
    free(sc.buf);
}

/* brute force in mesh order; same acceptance rule, same triangle arithmetic. */
void orc_intersect_brute(const orc_scene *s, orc_ray *rays, uint32_t *hits, uint32_t n)
{
    for (uint32_t r = 0; r < n; r++) {
        orc_ray *ray = &rays[r];
        v3 o = v3_load(ray->origin_far), d = v3_load(ray->direction);
        hits[r] = ORC_NONE;
        for (uint32_t t = 0; t < s->num_triangles; t++) {
            v3 a = v3_load(s->positions + 3 * (size_t)s->indices[3 * (size_t)t + 0]);
            v3 b = v3_load(s->positions + 3 * (size_t)s->indices[3 * (size_t)t + 1]);
            v3 c = v3_load(s->positions + 3 * (size_t)s->indices[3 * (size_t)t + 2]);
            float h = tri_intersect(a, v3_sub(b, a), v3_sub(c, a), o, d);
            if (h >= ray->dir_recip_near[3] && h < ray->origin_far[3]) {
                ray->origin_far[3] = h;
                hits[r] = t;
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* Random numbers.  The reference draws from a sequential per-tile
 * SmallRng (src/main.rs:84); a sequential stream cannot be reproduced by a
 * decomposition-independent GPU path, so both sides use this counter-based
 * generator keyed by (seed, global pixel, global sample, dimension).  The
 * u32 -> f32 mapping is rand 0.8.5's Standard for f32: (x >> 8) * 2^-24 [3P]. */

static inline uint32_t pcg_hash(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}

uint32_t orc_rng_u32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim)
{
    uint32_t h = pcg_hash((uint32_t)seed ^ 0x9E3779B9u);
    h = pcg_hash(h ^ (uint32_t)(seed >> 32));
    h = pcg_hash(h ^ pixel);
    h = pcg_hash(h ^ sample);
    h = pcg_hash(h ^ dim);
    return h;
}

float orc_rng_f32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim)
{
    return (float)(orc_rng_u32(seed, pixel, sample, dim) >> 8) * (1.f / 16777216.f);
}

/* sin and cos of 2*pi*u for u in [0,1).  The reference calls libm sinf/cosf
 * (src/brdf.rs:13-16) or `wide`'s polynomial (brdf.rs:82-83); neither is
 * bit-reproducible on a GPU, so both sides evaluate this fixed sequence of
 * un-fused f32 operations (Cody-Waite reduction by pi/2, Cephes minimax
 * polynomials on [-pi/4, pi/4]). */
void orc_sincos_2pi(float u, float *s_out, float *c_out)
{
    float b = (2.f * 3.14159274f) * u; /* brdf.rs:14 */
    float kf = floorf(b * 0.636619772f + 0.5f);
    int k = (int)kf;
    float r = b - kf * 1.5703125f;
    r = r - kf * 4.83751297e-4f;
    r = r - kf * 7.54978995e-8f;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) *
                   z * z -
               0.5f * z + 1.f;
    switch (k & 3) {
    case 0: *s_out = sp; *c_out = cp; break;
    case 1: *s_out = cp; *c_out = -sp; break;
    case 2: *s_out = -sp; *c_out = -cp; break;
    default: *s_out = -cp; *c_out = sp; break;
    }
}

/* ------------------------------------------------------------------------ */
/* Shading, src/integrators/common.rs                                        */

#define ORC_PI 3.14159274f
#define SHADOW_RAY_NEAR 1e-5f /* common.rs:12 */

/* common.rs:37-60 */
int orc_sample_light(const float dir_out[3], const float normal[3], const float hit_pos[3],
                     const orc_light *light, float dir_in[3], float *dist, float weight[3])
{
    v3 n = v3_load(normal), wo = v3_load(dir_out), p = v3_load(hit_pos);
    v3 lv = v3_sub(v3_load(light->pos), p);
    float ndotl = v3_dot(n, lv);
    if (ndotl * v3_dot(n, wo) < 0.f)
        return 0;
    float d2 = v3_dot(lv, lv);
    float d = sqrtf(d2);
    v3 wi = v3_div(lv, d);
    ndotl = fabsf(ndotl / d);
    float f = 0.8f / ORC_PI; /* brdf.rs:24-26 */
    v3 w = v3_div(v3_scale(v3_mul(v3_load(light->intensity), v3_make(f, f, f)), ndotl), d2);
    if (w.x == 0.f && w.y == 0.f && w.z == 0.f)
        return 0;
    v3_store(dir_in, wi);
    *dist = d;
    v3_store(weight, w);
    return 1;
}

typedef struct { float w; v3 v; } quat;

/* src/math.rs:48-58.  [3P nalgebra] new_normalize divides the four
 * coordinates by sqrt(((i*i+j*j)+k*k)+w*w).  For dot == -1 the reference
 * builds a rotation by pi about Y through exp(); restated as
 * (w = cos(pi/2 in f32), v = (0,1,0)) -- parity unpinned for that branch. */
static inline quat rotation_between(v3 from, v3 to)
{
    quat q;
    float dot = v3_dot(from, to);
    if (dot == -1.f) {
        q.w = -4.37113883e-8f;
        q.v = v3_make(0.f, 1.f, 0.f);
        return q;
    }
    float w = 1.f + dot;
    v3 c = v3_cross(from, to);
    float nrm = sqrtf(((c.x * c.x + c.y * c.y) + c.z * c.z) + w * w);
    q.w = w / nrm;
    q.v = v3_div(c, nrm);
    return q;
}

/* [3P nalgebra] UnitQuaternion * Vector3: t = 2*(q.v x x); (t*q.w + q.v x t) + x */
static inline v3 quat_rotate(quat q, v3 x)
{
    v3 t = v3_scale(v3_cross(q.v, x), 2.f);
    v3 c = v3_cross(q.v, t);
    return v3_add(v3_add(v3_scale(t, q.w), c), x);
}

/* common.rs:64-78 with brdf.rs:10-17,36-44; the two uniforms are passed in. */
void orc_sample_diffuse_ray(const float dir_out[3], const float normal[3], float u0, float u1,
                            float dir_in[3], float weight[3])
{
    v3 n = v3_load(normal), wo = v3_load(dir_out);
    quat q = rotation_between(n, v3_make(0.f, 0.f, 1.f));
    v3 wo_l = quat_rotate(q, wo);
    float a = sqrtf(u0);
    float sn, cs;
    orc_sincos_2pi(u1, &sn, &cs);
    v3 wi_l = v3_make(a * cs, a * sn, sqrtf(1.f - u0));
    float pdf = wi_l.z / ORC_PI;
    float f = 0.8f / ORC_PI;
    wi_l.z = copysignf(wi_l.z, wo_l.z);
    quat qi = q;
    qi.v = v3_neg(q.v);
    v3 wi = quat_rotate(qi, wi_l);
    float ndotl = fabsf(v3_dot(n, wi));
    float w = f * ndotl / pdf;
    v3_store(dir_in, wi);
    weight[0] = weight[1] = weight[2] = w;
}

/* ------------------------------------------------------------------------ */
/* Camera, src/camera.rs                                                     */

typedef struct {
    v3 eye, forward, right, up;
    v3 step_x, step_y, top_left;
    float pixel_size;
} cam_frame;

static cam_frame cam_setup(const orc_camera *c, uint32_t w, uint32_t h)
{
    cam_frame f;
    f.eye = v3_load(c->eye);
    f.forward = v3_normalize(v3_sub(v3_load(c->target), f.eye)); /* camera.rs:26-28 */
    f.right = v3_normalize(v3_cross(f.forward, v3_load(c->up)));
    f.up = v3_cross(f.right, f.forward);
    float half_w = tanf(c->fov_deg * (ORC_PI / 180.f) / 2.f); /* camera.rs:63-70 */
    float half_h = half_w * (((float)h - 1.f) / ((float)w - 1.f));
    f.step_x = v3_scale(f.right, 2.f * half_w / ((float)w - 1.f));
    f.step_y = v3_scale(v3_neg(f.up), 2.f * half_h / ((float)h - 1.f));
    f.pixel_size = half_w * 2.f / ((float)w - 1.f);
    f.top_left = v3_add(v3_sub(f.forward, v3_scale(f.right, half_w)), v3_scale(f.up, half_h));
    return f;
}

/* camera.rs:72-75, 98-113: row start recomputed, x advanced by repeated += */
static v3 cam_pixel_ray(const cam_frame *f, uint32_t tile_size, uint32_t x, uint32_t y)
{
    uint32_t rx = x / tile_size * tile_size, ry = y / tile_size * tile_size;
    v3 region_tl = v3_add(v3_add(f->top_left, v3_scale(f->step_x, (float)rx)),
                          v3_scale(f->step_y, (float)ry));
    uint32_t cy = y - ry, cx = x - rx;
    v3 cur = region_tl;
    if (cy > 0) cur = v3_add(region_tl, v3_scale(f->step_y, (float)cy));
    for (uint32_t i = 0; i < cx; i++) cur = v3_add(cur, f->step_x);
    return cur;
}

/* camera.rs:115-120 */
static v3 cam_sample(const cam_frame *f, v3 pixel_ray, float u0, float u1)
{
    v3 off = v3_scale(v3_add(v3_scale(f->right, u0 - 0.5f), v3_scale(f->up, u1 - 0.5f)),
                      f->pixel_size);
    return v3_normalize(v3_add(pixel_ray, off));
}

void orc_camera_sample_dir(const orc_camera *cam, uint32_t width, uint32_t height,
                           uint32_t tile_size, uint32_t x, uint32_t y, float u0, float u1,
                           float dir[3])
{
    cam_frame f = cam_setup(cam, width, height);
    v3_store(dir, cam_sample(&f, cam_pixel_ray(&f, tile_size, x, y), u0, u1));
}

void orc_generate_primary(const orc_camera *cam, const orc_render_params *p, orc_ray *rays)
{
    cam_frame f = cam_setup(cam, p->width, p->height);
    size_t k = 0;
    for (uint32_t y = 0; y < p->height; y++)
        for (uint32_t x = 0; x < p->width; x++) {
            v3 pr = cam_pixel_ray(&f, p->tile_size, x, y);
            uint32_t pix = y * p->width + x;
            for (uint32_t sidx = 0; sidx < p->samples; sidx++, k++) {
                uint32_t gs = p->sample_offset + sidx;
                float u0 = orc_rng_f32(p->seed, pix, gs, 0);
                float u1 = orc_rng_f32(p->seed, pix, gs, 1);
                v3 d = cam_sample(&f, pr, u0, u1);
                float o3[3], d3[3];
                v3_store(o3, f.eye);
                v3_store(d3, d);
                orc_ray_new(&rays[k], o3, d3, 0.f, INFINITY);
            }
        }
}

/* ------------------------------------------------------------------------ */
/* Tile integrators, src/integrators/integrate_stream.rs, tile_integrator.rs */

typedef struct {
    v3 weight;
    uint32_t dest;   /* pixel index inside the tile (common.rs:17) */
    uint32_t pixel;  /* global pixel index, RNG key */
    uint32_t sample; /* global sample index, RNG key */
    uint32_t bounces;
} path_info;

typedef struct {
    const orc_scene *s;
    const orc_light *light;
    const orc_render_params *p;
    cam_frame cam;
    float *accum;
    uint32_t ntx, nty;
    atomic_uint next_tile;
    orc_stats stats; /* merged under lock */
    pthread_mutex_t lock;
} render_job;

static inline v3 hit_pos_of(const orc_ray *r) /* ray.rs:40-42 */
{
    return v3_add(v3_load(r->origin_far), v3_scale(v3_load(r->direction), r->origin_far[3]));
}

static void make_ray(orc_ray *r, v3 o, v3 d, float near_t, float far_t)
{
    float o3[3], d3[3];
    v3_store(o3, o);
    v3_store(d3, d);
    orc_ray_new(r, o3, d3, near_t, far_t);
}

static const orc_trace_fns scalar_trace_fns = {intersect_stream_impl, occluded_stream_impl, orc_intersect1,
                                               orc_occluded1};

static void render_tile(render_job *job, uint32_t tx, uint32_t ty, orc_stats *st,
                        stream_scratch *sc)
{
    const orc_render_params *p = job->p;
    const orc_scene *s = job->s;
    /* the same integrator over either restatement of the traversal */
    const orc_trace_fns *fn = (p->mode & ORC_MODE_SIMD) ? &orc_simd_trace_fns : &scalar_trace_fns;
    const uint32_t mode = p->mode & ~(uint32_t)ORC_MODE_SIMD;
    uint32_t x0 = tx * p->tile_size, y0 = ty * p->tile_size;
    uint32_t tw = p->tile_size < p->width - x0 ? p->tile_size : p->width - x0;   /* main.rs:76-77 */
    uint32_t th = p->tile_size < p->height - y0 ? p->tile_size : p->height - y0;
    size_t max_rays = (size_t)tw * th * p->samples;
    float *tile_px = (float *)calloc((size_t)tw * th * 4, sizeof(float));

    if (mode == ORC_MODE_SINGLE_RAY) {
        /* tile_integrator.rs:68-99 + 18-66 */
        for (uint32_t py = 0; py < th; py++)
            for (uint32_t px = 0; px < tw; px++) {
                uint32_t gx = x0 + px, gy = y0 + py, pix = gy * p->width + gx;
                v3 pr = cam_pixel_ray(&job->cam, p->tile_size, gx, gy);
                float *dst = tile_px + 4 * ((size_t)py * tw + px);
                dst[3] += (float)p->samples;
                for (uint32_t si = 0; si < p->samples; si++) {
                    uint32_t gs = p->sample_offset + si;
                    v3 d = cam_sample(&job->cam, pr, orc_rng_f32(p->seed, pix, gs, 0),
                                      orc_rng_f32(p->seed, pix, gs, 1));
                    orc_ray ray;
                    make_ray(&ray, job->cam.eye, d, 0.f, INFINITY);
                    uint32_t bounces = 0;
                    v3 weight = v3_make(1.f, 1.f, 1.f);
                    uint32_t hit;
                    while ((hit = fn->intersect1(s, &ray, st)) != ORC_NONE) {
                        v3 hp = hit_pos_of(&ray);
                        v3 wo = v3_neg(v3_load(ray.direction));
                        const float *nrm = s->geom_normals + 3 * (size_t)hit;
                        float wo3[3], hp3[3], wi3[3], dist, sw[3];
                        v3_store(wo3, wo);
                        v3_store(hp3, hp);
                        if (orc_sample_light(wo3, nrm, hp3, job->light, wi3, &dist, sw)) {
                            orc_ray sh;
                            make_ray(&sh, hp, v3_load(wi3), SHADOW_RAY_NEAR, dist);
                            if (!fn->occluded1(s, &sh, st)) {
                                v3 c = v3_mul(weight, v3_load(sw));
                                dst[0] += c.x; dst[1] += c.y; dst[2] += c.z;
                            }
                        }
                        if (bounces >= p->max_bounces) break;
                        float din[3], dw[3];
                        orc_sample_diffuse_ray(wo3, nrm, orc_rng_f32(p->seed, pix, gs, 2 + 2 * bounces),
                                               orc_rng_f32(p->seed, pix, gs, 3 + 2 * bounces), din, dw);
                        make_ray(&ray, hp, v3_load(din), SHADOW_RAY_NEAR, INFINITY);
                        bounces += 1;
                        weight = v3_mul(weight, v3_load(dw));
                    }
                }
            }
    } else {
        /* integrate_stream.rs:56-164 (stream) and 166-242 (shadow immediate) */
        orc_ray *rays = (orc_ray *)malloc(sizeof(orc_ray) * (max_rays ? max_rays : 1));
        path_info *infos = (path_info *)malloc(sizeof(path_info) * (max_rays ? max_rays : 1));
        uint32_t *hits = (uint32_t *)malloc(sizeof(uint32_t) * (max_rays ? max_rays : 1));
        orc_ray *sh_rays = NULL;
        path_info *sh_infos = NULL;
        uint8_t *sh_occ = NULL;
        if (mode == ORC_MODE_STREAM) {
            sh_rays = (orc_ray *)malloc(sizeof(orc_ray) * (max_rays ? max_rays : 1));
            sh_infos = (path_info *)malloc(sizeof(path_info) * (max_rays ? max_rays : 1));
            sh_occ = (uint8_t *)malloc(max_rays ? max_rays : 1);
        }
        size_t n = 0;
        /* generate_rays, integrate_stream.rs:14-54 */
        for (uint32_t py = 0; py < th; py++)
            for (uint32_t px = 0; px < tw; px++) {
                uint32_t gx = x0 + px, gy = y0 + py, pix = gy * p->width + gx;
                v3 pr = cam_pixel_ray(&job->cam, p->tile_size, gx, gy);
                for (uint32_t si = 0; si < p->samples; si++, n++) {
                    uint32_t gs = p->sample_offset + si;
                    v3 d = cam_sample(&job->cam, pr, orc_rng_f32(p->seed, pix, gs, 0),
                                      orc_rng_f32(p->seed, pix, gs, 1));
                    make_ray(&rays[n], job->cam.eye, d, 0.f, INFINITY);
                    infos[n].weight = v3_make(1.f, 1.f, 1.f);
                    infos[n].dest = py * tw + px;
                    infos[n].pixel = pix;
                    infos[n].sample = gs;
                    infos[n].bounces = 0;
                }
            }
        for (size_t i = 0; i < (size_t)tw * th; i++) tile_px[4 * i + 3] += (float)p->samples;

        while (n > 0) {
            for (size_t i = 0; i < n; i++) hits[i] = ORC_NONE;
            fn->intersect_stream(s, rays, hits, (uint32_t)n, st, sc);
            size_t new_n = 0, n_sh = 0;
            for (size_t i = 0; i < n; i++) {
                if (hits[i] == ORC_NONE) continue;
                const orc_ray *ray = &rays[i];
                path_info info = infos[i];
                v3 hp = hit_pos_of(ray);
                v3 wo = v3_neg(v3_load(ray->direction));
                const float *nrm = s->geom_normals + 3 * (size_t)hits[i];
                float wo3[3], hp3[3], wi3[3], dist, sw[3];
                v3_store(wo3, wo);
                v3_store(hp3, hp);
                if (orc_sample_light(wo3, nrm, hp3, job->light, wi3, &dist, sw)) {
                    if (mode == ORC_MODE_STREAM) {
                        make_ray(&sh_rays[n_sh], hp, v3_load(wi3), SHADOW_RAY_NEAR, dist);
                        sh_infos[n_sh] = info;
                        sh_infos[n_sh].weight = v3_mul(info.weight, v3_load(sw));
                        n_sh++;
                    } else {
                        orc_ray sh;
                        make_ray(&sh, hp, v3_load(wi3), SHADOW_RAY_NEAR, dist);
                        if (!fn->occluded1(s, &sh, st)) {
                            v3 c = v3_mul(info.weight, v3_load(sw));
                            float *dst = tile_px + 4 * (size_t)info.dest;
                            dst[0] += c.x; dst[1] += c.y; dst[2] += c.z;
                        }
                    }
                }
                if (info.bounces < p->max_bounces) {
                    float din[3], dw[3];
                    orc_sample_diffuse_ray(wo3, nrm,
                                           orc_rng_f32(p->seed, info.pixel, info.sample, 2 + 2 * info.bounces),
                                           orc_rng_f32(p->seed, info.pixel, info.sample, 3 + 2 * info.bounces),
                                           din, dw);
                    make_ray(&rays[new_n], hp, v3_load(din), SHADOW_RAY_NEAR, INFINITY);
                    infos[new_n] = info;
                    infos[new_n].weight = v3_mul(info.weight, v3_load(dw)); /* common.rs:23-29 */
                    infos[new_n].bounces = info.bounces + 1;
                    new_n++;
                }
            }
            n = new_n;
            if (mode == ORC_MODE_STREAM && n_sh > 0) {
                memset(sh_occ, 0, n_sh);
                fn->occluded_stream(s, sh_rays, sh_occ, (uint32_t)n_sh, st, sc);
                for (size_t i = 0; i < n_sh; i++)
                    if (!sh_occ[i]) {
                        float *dst = tile_px + 4 * (size_t)sh_infos[i].dest;
                        dst[0] += sh_infos[i].weight.x;
                        dst[1] += sh_infos[i].weight.y;
                        dst[2] += sh_infos[i].weight.z;
                    }
            }
        }
        free(rays); free(infos); free(hits);
        free(sh_rays); free(sh_infos); free(sh_occ);
    }

    /* each pixel belongs to exactly one tile (main.rs:88-95): no locking */
    for (uint32_t py = 0; py < th; py++)
        for (uint32_t px = 0; px < tw; px++) {
            float *dst = job->accum + 4 * ((size_t)(y0 + py) * p->width + (x0 + px));
            const float *src = tile_px + 4 * ((size_t)py * tw + px);
            dst[0] += src[0]; dst[1] += src[1]; dst[2] += src[2]; dst[3] += src[3];
        }
    free(tile_px);
}

static void *render_worker(void *arg)
{
    render_job *job = (render_job *)arg;
    orc_stats st;
    memset(&st, 0, sizeof st);
    stream_scratch sc = {0, 0};
    uint32_t ntiles = job->ntx * job->nty;
    uint32_t sc_n = job->p->shard_count ? job->p->shard_count : 1;
    for (;;) {
        uint32_t i = atomic_fetch_add(&job->next_tile, 1);
        if (i >= ntiles) break;
        if (i % sc_n != job->p->shard_index) continue;
        render_tile(job, i % job->ntx, i / job->ntx, &st, &sc);
    }
    free(sc.buf);
    pthread_mutex_lock(&job->lock);
    job->stats.total_rays += st.total_rays;
    job->stats.inner_visit += st.inner_visit;
    job->stats.inner_rays += st.inner_rays;
    job->stats.leaf_visit += st.leaf_visit;
    job->stats.leaf_rays += st.leaf_rays;
    job->stats.leaf_objs += st.leaf_objs;
    job->stats.obj_intersect += st.obj_intersect;
    pthread_mutex_unlock(&job->lock);
    return NULL;
}

/* src/main.rs:61-98,133-168: row-major tiles, one worker per thread */
void orc_render(const orc_scene *s, const orc_camera *cam, const orc_light *light,
                const orc_render_params *p, float *accum, orc_stats *stats)
{
    render_job job;
    memset(&job, 0, sizeof job);
    job.s = s;
    job.light = light;
    job.p = p;
    job.cam = cam_setup(cam, p->width, p->height);
    job.accum = accum;
    job.ntx = (p->width + p->tile_size - 1) / p->tile_size;  /* main.rs:68-69 */
    job.nty = (p->height + p->tile_size - 1) / p->tile_size;
    atomic_init(&job.next_tile, 0);
    pthread_mutex_init(&job.lock, NULL);
    uint32_t nt = p->nthreads ? p->nthreads : 1;
    if (nt == 1) {
        render_worker(&job);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nt);
        for (uint32_t i = 0; i < nt; i++) pthread_create(&th[i], NULL, render_worker, &job);
        for (uint32_t i = 0; i < nt; i++) pthread_join(th[i], NULL);
        free(th);
    }
    pthread_mutex_destroy(&job.lock);
    if (stats) {
        stats->total_rays += job.stats.total_rays;
        stats->inner_visit += job.stats.inner_visit;
        stats->inner_rays += job.stats.inner_rays;
        stats->leaf_visit += job.stats.leaf_visit;
        stats->leaf_rays += job.stats.leaf_rays;
        stats->leaf_objs += job.stats.leaf_objs;
        stats->obj_intersect += job.stats.obj_intersect;
    }
}

/* src/main.rs:34-41: sqrt gamma, *255, Rust saturating `as u8` (NaN -> 0) */
void orc_resolve(const float *accum, uint32_t width, uint32_t height, uint8_t *rgb)
{
    size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; i++) {
        float w = accum[4 * i + 3];
        for (int c = 0; c < 3; c++) {
            float v = sqrtf(accum[4 * i + c] / w) * 255.f;
            uint8_t b;
            if (!(v > 0.f)) b = 0;
            else if (v >= 255.f) b = 255;
            else b = (uint8_t)v;
            rgb[3 * i + c] = b;
        }
    }
}
