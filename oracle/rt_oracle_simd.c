/*
 * rt_oracle_simd.c -- TEST INFRASTRUCTURE ONLY (see rt_oracle.h).
 *
 * The reference's four traversal routines written with the x86 intrinsics the reference itself
 * uses.  The Rust code calls safe_arch 1.0.0, whose functions are thin wrappers of one intrinsic
 * each ([3P], Cargo.lock): shuffle_abi_f32_all_m128 = _mm_shuffle_ps, blend_varying_m128 =
 * _mm_blendv_ps, fused_mul_sub_m128 = _mm_fmsub_ps, max_m128 / min_m128 = _mm_max_ps / _mm_min_ps,
 * set_m128_m256(hi, lo) = _mm256_set_m128, load_m128_splat_m256 = _mm256_broadcast_ps,
 * shuffle_m256 = _mm256_shuffle_ps, cmp_op_mask_m256<LessEqualOrdered> = _mm256_cmp_ps(_CMP_LE_OQ),
 * move_mask_m256 = _mm256_movemask_ps, `^` = xor.  Every statement below stands for the statement
 * of the reference at the cited line, operands in the reference's order, so that what the scalar
 * restatement (rt_oracle.c) has to emulate -- the second-operand-on-NaN rule of max/min, the sign-bit
 * lanes of the blends, the single rounding of the fused multiply-subtract -- comes from the
 * hardware here.  PARITY UNPINNED all the same: this is a second restatement, not the reference.
 *
 * Build: -mavx2 -mfma (x86-64-v3) and -ffp-contract=off.  The latter matters: with contraction on,
 * gcc fuses _mm_add_ps(_mm_mul_ps(..)) in the triangle test, which Rust never does.
 */
#include "rt_oracle_internal.h"

#include <immintrin.h>

/* ------------------------------------------------------------------------ */
/* src/bvh/intersect1.rs                                                     */

typedef struct {
    __m128 dir_recip_x, dir_recip_y, dir_recip_z;
    __m128 origin_dir_recip_x, origin_dir_recip_y, origin_dir_recip_z;
    __m128 ray_near_far;
} ray1;

/* intersect1.rs:14-30 (occluded1) and 117-133 (intersect1): identical text */
static inline ray1 ray1_setup(const orc_ray *ray)
{
    ray1 r;
    const __m128 origin_xyz_far = _mm_loadu_ps(ray->origin_far);
    const __m128 origin_x = _mm_shuffle_ps(origin_xyz_far, origin_xyz_far, 0x00);
    const __m128 origin_y = _mm_shuffle_ps(origin_xyz_far, origin_xyz_far, 0x55);
    const __m128 origin_z = _mm_shuffle_ps(origin_xyz_far, origin_xyz_far, 0xAA);
    const __m128 dir_recip_xyz_near = _mm_loadu_ps(ray->dir_recip_near);
    r.dir_recip_x = _mm_shuffle_ps(dir_recip_xyz_near, dir_recip_xyz_near, 0x00);
    r.dir_recip_y = _mm_shuffle_ps(dir_recip_xyz_near, dir_recip_xyz_near, 0x55);
    r.dir_recip_z = _mm_shuffle_ps(dir_recip_xyz_near, dir_recip_xyz_near, 0xAA);
    /* lanes {near, near, far, far}: loaded ONCE, so node culling sees the far the ray arrived with */
    r.ray_near_far = _mm_shuffle_ps(dir_recip_xyz_near, origin_xyz_far, 0xFF);
    r.origin_dir_recip_x = _mm_mul_ps(origin_x, r.dir_recip_x);
    r.origin_dir_recip_y = _mm_mul_ps(origin_y, r.dir_recip_y);
    r.origin_dir_recip_z = _mm_mul_ps(origin_z, r.dir_recip_z);
    return r;
}

/* intersect1.rs:41-76 / 144-179 */
static inline void node1(const node *n, const ray1 *r, int *hit_l, int *hit_r)
{
    const __m128 bb_min_max_x = _mm_loadu_ps(n->bx);
    const __m128 bb_min_max_y = _mm_loadu_ps(n->by);
    const __m128 bb_min_max_z = _mm_loadu_ps(n->bz);

    const __m128 bb_max_min_x = _mm_shuffle_ps(bb_min_max_x, bb_min_max_x, 0x4E); /* 0b01_00_11_10 */
    const __m128 bb_max_min_y = _mm_shuffle_ps(bb_min_max_y, bb_min_max_y, 0x4E);
    const __m128 bb_max_min_z = _mm_shuffle_ps(bb_min_max_z, bb_min_max_z, 0x4E);

    const __m128 bb_near_far_x = _mm_blendv_ps(bb_min_max_x, bb_max_min_x, r->dir_recip_x);
    const __m128 bb_near_far_y = _mm_blendv_ps(bb_min_max_y, bb_max_min_y, r->dir_recip_y);
    const __m128 bb_near_far_z = _mm_blendv_ps(bb_min_max_z, bb_max_min_z, r->dir_recip_z);

    const __m128 tnear_far_x = _mm_fmsub_ps(bb_near_far_x, r->dir_recip_x, r->origin_dir_recip_x);
    const __m128 tnear_far_y = _mm_fmsub_ps(bb_near_far_y, r->dir_recip_y, r->origin_dir_recip_y);
    const __m128 tnear_far_z = _mm_fmsub_ps(bb_near_far_z, r->dir_recip_z, r->origin_dir_recip_z);

    const __m128 tnear = _mm_max_ps(_mm_max_ps(tnear_far_x, tnear_far_y), _mm_max_ps(tnear_far_z, r->ray_near_far));
    const __m128 tfar = _mm_min_ps(_mm_min_ps(tnear_far_x, tnear_far_y), _mm_min_ps(tnear_far_z, r->ray_near_far));

    float tnear_far[4];
    _mm_storeu_ps(tnear_far, _mm_shuffle_ps(tnear, tfar, 0xE4)); /* 0b11_10_01_00 */
    *hit_l = tnear_far[0] <= tnear_far[2];
    *hit_r = tnear_far[1] <= tnear_far[3];
}

#define SIMD_STACK1 (32 * 4) /* ArrayVec<_, 32> in the reference (it panics beyond); head room for deep test meshes */

/* intersect1.rs:109-210 */
uint32_t orc_simd_intersect1(const orc_scene *s, orc_ray *ray, orc_stats *st)
{
    orc_stats dummy;
    if (!st) { memset(&dummy, 0, sizeof dummy); st = &dummy; }
    st->total_rays += 1;
    uint32_t hit_obj = ORC_NONE;
    const node *stack[SIMD_STACK1];
    int sp = 0;
    const ray1 r = ray1_setup(ray);
    const v3 o = v3_load(ray->origin_far), d = v3_load(ray->direction);
    const node *cur = s->root;
    while (cur) {
        if (!cur->is_leaf) {
            stats_inner(st, 1);
            int hl, hr;
            node1(cur, &r, &hl, &hr);
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
                const float hit = tri_intersect(t->v0, t->v0v1, t->v0v2, o, d);
                if (ray->dir_recip_near[3] <= hit && hit < ray->origin_far[3]) { /* (near..far).contains(&hit) */
                    ray->origin_far[3] = hit;
                    hit_obj = s->object_indices[i];
                }
            }
            cur = sp ? stack[--sp] : NULL;
        }
    }
    return hit_obj;
}

/* intersect1.rs:7-106 */
int orc_simd_occluded1(const orc_scene *s, const orc_ray *ray, orc_stats *st)
{
    orc_stats dummy;
    if (!st) { memset(&dummy, 0, sizeof dummy); st = &dummy; }
    st->total_rays += 1;
    const node *stack[SIMD_STACK1];
    int sp = 0;
    const ray1 r = ray1_setup(ray);
    const v3 o = v3_load(ray->origin_far), d = v3_load(ray->direction);
    const node *cur = s->root;
    while (cur) {
        if (!cur->is_leaf) {
            stats_inner(st, 1);
            int hl, hr;
            node1(cur, &r, &hl, &hr);
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
                const float hit = tri_intersect(t->v0, t->v0v1, t->v0v2, o, d);
                if (ray->dir_recip_near[3] <= hit && hit < ray->origin_far[3]) return 1;
            }
            cur = sp ? stack[--sp] : NULL;
        }
    }
    return 0;
}

void orc_simd_intersect1_n(const orc_scene *s, orc_ray *rays, uint32_t *hits, uint32_t n, orc_stats *st)
{
    for (uint32_t i = 0; i < n; i++) hits[i] = orc_simd_intersect1(s, &rays[i], st);
}

void orc_simd_occluded1_n(const orc_scene *s, const orc_ray *rays, uint8_t *occluded, uint32_t n, orc_stats *st)
{
    for (uint32_t i = 0; i < n; i++) occluded[i] = (uint8_t)orc_simd_occluded1(s, &rays[i], st);
}

/* ------------------------------------------------------------------------ */
/* src/bvh/intersect_stream.rs                                               */

/* The node's boxes as the stream code prepares them once per node visit
 * (intersect_stream.rs:50-64 and 239-253): both 128-bit halves hold the same four lanes, the two
 * upper lanes of each half negated, {minL, minR, -maxL, -maxR} and {maxL, maxR, -minL, -minR}. */
typedef struct {
    __m256 bb_min_max_x, bb_min_max_y, bb_min_max_z;
    __m256 bb_max_min_x, bb_max_min_y, bb_max_min_z;
} node2;

static inline node2 node2_setup(const node *n)
{
    node2 b;
    const __m256 pos_neg_mask = _mm256_setr_ps(0.f, 0.f, -0.f, -0.f, 0.f, 0.f, -0.f, -0.f);
    const __m256 x = _mm256_broadcast_ps((const __m128 *)n->bx);
    const __m256 y = _mm256_broadcast_ps((const __m128 *)n->by);
    const __m256 z = _mm256_broadcast_ps((const __m128 *)n->bz);
    b.bb_max_min_x = _mm256_xor_ps(_mm256_shuffle_ps(x, x, 0x4E), pos_neg_mask);
    b.bb_max_min_y = _mm256_xor_ps(_mm256_shuffle_ps(y, y, 0x4E), pos_neg_mask);
    b.bb_max_min_z = _mm256_xor_ps(_mm256_shuffle_ps(z, z, 0x4E), pos_neg_mask);
    b.bb_min_max_x = _mm256_xor_ps(x, pos_neg_mask);
    b.bb_min_max_y = _mm256_xor_ps(y, pos_neg_mask);
    b.bb_min_max_z = _mm256_xor_ps(z, pos_neg_mask);
    return b;
}

/* Rays a (low half) and b (high half) against both children: intersect_stream.rs:75-142 and
 * 269-336.  Bits of the result per ray (a: 0..3, b: 4..7): 0 = left child hit, 1 = right child
 * hit, 2 = left first (nearL <= nearR). */
static inline int node2_mask(const node2 *b, const orc_ray *ray_a, const orc_ray *ray_b)
{
    const __m256 neg_mask = _mm256_set1_ps(-0.f);
    const __m256 origin_xyz_far_ab = _mm256_set_m128(_mm_loadu_ps(ray_b->origin_far), _mm_loadu_ps(ray_a->origin_far));
    const __m256 origin_xyz_far_ab_neg = _mm256_xor_ps(origin_xyz_far_ab, neg_mask);
    const __m256 dir_recip_xyz_near_ab =
        _mm256_set_m128(_mm_loadu_ps(ray_b->dir_recip_near), _mm_loadu_ps(ray_a->dir_recip_near));

    const __m256 origin_x = _mm256_shuffle_ps(origin_xyz_far_ab, origin_xyz_far_ab_neg, 0x00);
    const __m256 origin_y = _mm256_shuffle_ps(origin_xyz_far_ab, origin_xyz_far_ab_neg, 0x55);
    const __m256 origin_z = _mm256_shuffle_ps(origin_xyz_far_ab, origin_xyz_far_ab_neg, 0xAA);

    const __m256 dir_recip_x = _mm256_shuffle_ps(dir_recip_xyz_near_ab, dir_recip_xyz_near_ab, 0x00);
    const __m256 dir_recip_y = _mm256_shuffle_ps(dir_recip_xyz_near_ab, dir_recip_xyz_near_ab, 0x55);
    const __m256 dir_recip_z = _mm256_shuffle_ps(dir_recip_xyz_near_ab, dir_recip_xyz_near_ab, 0xAA);

    const __m256 ray_near_far = _mm256_shuffle_ps(dir_recip_xyz_near_ab, origin_xyz_far_ab_neg, 0xFF);

    const __m256 origin_dir_recip_x = _mm256_mul_ps(origin_x, dir_recip_x);
    const __m256 origin_dir_recip_y = _mm256_mul_ps(origin_y, dir_recip_y);
    const __m256 origin_dir_recip_z = _mm256_mul_ps(origin_z, dir_recip_z);

    const __m256 bb_near_far_x = _mm256_blendv_ps(b->bb_min_max_x, b->bb_max_min_x, dir_recip_x);
    const __m256 bb_near_far_y = _mm256_blendv_ps(b->bb_min_max_y, b->bb_max_min_y, dir_recip_y);
    const __m256 bb_near_far_z = _mm256_blendv_ps(b->bb_min_max_z, b->bb_max_min_z, dir_recip_z);

    const __m256 tnear_far_x = _mm256_fmsub_ps(bb_near_far_x, dir_recip_x, origin_dir_recip_x);
    const __m256 tnear_far_y = _mm256_fmsub_ps(bb_near_far_y, dir_recip_y, origin_dir_recip_y);
    const __m256 tnear_far_z = _mm256_fmsub_ps(bb_near_far_z, dir_recip_z, origin_dir_recip_z);

    /* near 0, near 1, -far 0, -far 1 */
    const __m256 tnear_far =
        _mm256_max_ps(_mm256_max_ps(_mm256_max_ps(tnear_far_x, tnear_far_y), tnear_far_z), ray_near_far);

    /* near 0, near 1, near 0, near 0  <=  far 0, far 1, near 1, near 1 */
    return _mm256_movemask_ps(_mm256_cmp_ps(_mm256_shuffle_ps(tnear_far, tnear_far, 0x04), /* 0b00_00_01_00 */
                                            _mm256_shuffle_ps(_mm256_xor_ps(tnear_far, neg_mask), tnear_far, 0x5E), /* 0b01_01_11_10 */
                                            _CMP_LE_OQ));
}

/* TriangleOpt::intersect_simd, src/triangle_opt.rs:46-69, on four rays.  [3P nalgebra 0.34.1 / simba
 * 0.9.1 / wide 0.7.33]: cross and dot are the scalar formulas lane by lane, un-fused. */
typedef struct { __m128 x, y, z; } v3x4;

static inline v3x4 v3x4_cross(v3x4 a, v3x4 b)
{
    v3x4 r;
    r.x = _mm_sub_ps(_mm_mul_ps(a.y, b.z), _mm_mul_ps(a.z, b.y));
    r.y = _mm_sub_ps(_mm_mul_ps(a.z, b.x), _mm_mul_ps(a.x, b.z));
    r.z = _mm_sub_ps(_mm_mul_ps(a.x, b.y), _mm_mul_ps(a.y, b.x));
    return r;
}
static inline __m128 v3x4_dot(v3x4 a, v3x4 b)
{
    return _mm_add_ps(_mm_add_ps(_mm_mul_ps(a.x, b.x), _mm_mul_ps(a.y, b.y)), _mm_mul_ps(a.z, b.z));
}
static inline v3x4 v3x4_splat(v3 a)
{
    v3x4 r = {_mm_set1_ps(a.x), _mm_set1_ps(a.y), _mm_set1_ps(a.z)};
    return r;
}

static inline __m128 tri_intersect_x4(const tri_opt *t, v3x4 ray_origin, v3x4 ray_direction)
{
    const v3x4 v0v1 = v3x4_splat(t->v0v1);
    const v3x4 v0v2 = v3x4_splat(t->v0v2);
    const v3x4 pvec = v3x4_cross(ray_direction, v0v1);
    const __m128 det = v3x4_dot(v0v2, pvec);

    const __m128 epsilon = _mm_set1_ps(1.1920929e-7f);
    const __m128 det_valid = _mm_cmpge_ps(_mm_andnot_ps(_mm_set1_ps(-0.f), det), epsilon);

    const __m128 inv_det = _mm_div_ps(_mm_set1_ps(1.f), det);

    const v3x4 v0 = v3x4_splat(t->v0);
    const v3x4 tvec = {_mm_sub_ps(ray_origin.x, v0.x), _mm_sub_ps(ray_origin.y, v0.y), _mm_sub_ps(ray_origin.z, v0.z)};
    const __m128 u = _mm_mul_ps(v3x4_dot(tvec, pvec), inv_det);
    const __m128 u_valid = _mm_and_ps(_mm_cmpge_ps(u, _mm_setzero_ps()), _mm_cmple_ps(u, _mm_set1_ps(1.f)));

    const v3x4 qvec = v3x4_cross(tvec, v0v2);
    const __m128 v = _mm_mul_ps(v3x4_dot(ray_direction, qvec), inv_det);
    const __m128 v_valid =
        _mm_and_ps(_mm_cmpge_ps(v, _mm_setzero_ps()), _mm_cmple_ps(_mm_add_ps(u, v), _mm_set1_ps(1.f)));

    const __m128 tt = _mm_mul_ps(v3x4_dot(v0v1, qvec), inv_det);

    const __m128 t_valid = _mm_and_ps(_mm_and_ps(det_valid, u_valid), v_valid);
    return _mm_blendv_ps(_mm_set1_ps(INFINITY), tt, t_valid);
}

/* intersect_objs / occluded_objs, intersect_stream.rs:384-426 and 428-469: chunks of four rays (a
 * short chunk is padded with its first ray), every triangle of the leaf against the chunk, then the
 * rays of the chunk one by one against their own, shrinking far. */
static inline void leaf_objs(const orc_scene *s, const node *nd, const uint32_t *ray_indices, size_t count,
                             orc_ray *rays, uint32_t *hit_objects, uint8_t *occluded)
{
    for (size_t c = 0; c < count; c += 4) {
        const size_t len = count - c < 4 ? count - c : 4;
        const uint32_t *chunk = ray_indices + c;
        uint32_t padded[4];
        for (size_t k = 0; k < 4; k++) padded[k] = k < len ? chunk[k] : chunk[0];
        v3x4 ray_origins, ray_directions;
        ray_origins.x = _mm_setr_ps(rays[padded[0]].origin_far[0], rays[padded[1]].origin_far[0],
                                    rays[padded[2]].origin_far[0], rays[padded[3]].origin_far[0]);
        ray_origins.y = _mm_setr_ps(rays[padded[0]].origin_far[1], rays[padded[1]].origin_far[1],
                                    rays[padded[2]].origin_far[1], rays[padded[3]].origin_far[1]);
        ray_origins.z = _mm_setr_ps(rays[padded[0]].origin_far[2], rays[padded[1]].origin_far[2],
                                    rays[padded[2]].origin_far[2], rays[padded[3]].origin_far[2]);
        ray_directions.x = _mm_setr_ps(rays[padded[0]].direction[0], rays[padded[1]].direction[0],
                                       rays[padded[2]].direction[0], rays[padded[3]].direction[0]);
        ray_directions.y = _mm_setr_ps(rays[padded[0]].direction[1], rays[padded[1]].direction[1],
                                       rays[padded[2]].direction[1], rays[padded[3]].direction[1]);
        ray_directions.z = _mm_setr_ps(rays[padded[0]].direction[2], rays[padded[1]].direction[2],
                                       rays[padded[2]].direction[2], rays[padded[3]].direction[2]);
        for (uint32_t tri_idx = nd->first; tri_idx < nd->first + nd->count; tri_idx++) {
            float hit[4];
            _mm_storeu_ps(hit, tri_intersect_x4(&s->triangles[tri_idx], ray_origins, ray_directions));
            for (size_t k = 0; k < len; k++) {
                orc_ray *ray = &rays[chunk[k]];
                if (ray->dir_recip_near[3] <= hit[k] && hit[k] < ray->origin_far[3]) {
                    ray->origin_far[3] = hit[k];
                    if (hit_objects) hit_objects[chunk[k]] = s->object_indices[tri_idx];
                    if (occluded) occluded[chunk[k]] = 1;
                }
            }
        }
    }
}

/* intersect_stream.rs:10-198 */
static void simd_intersect_stream_impl(const orc_scene *s, orc_ray *rays, uint32_t *hits, uint32_t n,
                                       orc_stats *st, stream_scratch *sc)
{
    st->total_rays += n;
    if (n == 0) return;
    const size_t ray_list_len = (size_t)n * 32;
    uint32_t *base = scratch_get(sc, ray_list_len * 3);
    uint32_t *ray_lists[3] = {base, base + ray_list_len, base + 2 * ray_list_len};
    size_t ray_list_sizes[3] = {0, 0, 0};
    for (uint32_t i = 0; i < n; i++) ray_lists[0][i] = i;
    ray_list_sizes[0] = n;
    stream_item stack[64 * 4];
    int sp = 0;
    stack[sp].n = s->root; stack[sp].list = 0; stack[sp].start = 0; sp++;
    while (sp) {
        const stream_item it = stack[--sp];
        size_t active_ray_idx = it.start;
        const size_t last_active_ray_idx = ray_list_sizes[it.list];
        ray_list_sizes[it.list] = it.start;
        const node *nd = it.n;
        if (!nd->is_leaf) {
            stats_inner(st, last_active_ray_idx - active_ray_idx);
            const node2 bb = node2_setup(nd);
            const size_t orig[3] = {ray_list_sizes[0], ray_list_sizes[1], ray_list_sizes[2]};
            while (active_ray_idx < last_active_ray_idx) {
                const uint32_t ray_idx_a = ray_lists[it.list][active_ray_idx];
                const uint32_t ray_idx_b =
                    active_ray_idx + 1 < last_active_ray_idx ? ray_lists[it.list][active_ray_idx + 1] : ray_idx_a;
                const unsigned mask = (unsigned)node2_mask(&bb, &rays[ray_idx_a], &rays[ray_idx_b]);

                const unsigned left_hit_a = mask & 1u, right_hit_a = (mask >> 1) & 1u, left_first_a = (mask >> 2) & 1u;
                ray_lists[0][ray_list_sizes[0]] = ray_idx_a;
                ray_lists[1][ray_list_sizes[1]] = ray_idx_a;
                ray_lists[2][ray_list_sizes[2]] = ray_idx_a;
                ray_list_sizes[0] += left_hit_a & left_first_a;
                ray_list_sizes[1] += right_hit_a;
                ray_list_sizes[2] += left_hit_a & (left_first_a ^ 1u);

                if (ray_idx_a != ray_idx_b) {
                    const unsigned left_hit_b = (mask >> 4) & 1u, right_hit_b = (mask >> 5) & 1u,
                                   left_first_b = (mask >> 6) & 1u;
                    ray_lists[0][ray_list_sizes[0]] = ray_idx_b;
                    ray_lists[1][ray_list_sizes[1]] = ray_idx_b;
                    ray_lists[2][ray_list_sizes[2]] = ray_idx_b;
                    ray_list_sizes[0] += left_hit_b & left_first_b;
                    ray_list_sizes[1] += right_hit_b;
                    ray_list_sizes[2] += left_hit_b & (left_first_b ^ 1u);
                }
                active_ray_idx += 2;
            }
            if (ray_list_sizes[2] - orig[2] > 0) { stack[sp].n = nd->child[0]; stack[sp].list = 2; stack[sp].start = orig[2]; sp++; }
            if (ray_list_sizes[1] - orig[1] > 0) { stack[sp].n = nd->child[1]; stack[sp].list = 1; stack[sp].start = orig[1]; sp++; }
            if (ray_list_sizes[0] - orig[0] > 0) { stack[sp].n = nd->child[0]; stack[sp].list = 0; stack[sp].start = orig[0]; sp++; }
        } else {
            stats_leaf(st, last_active_ray_idx - active_ray_idx, nd->count);
            leaf_objs(s, nd, ray_lists[it.list] + active_ray_idx, last_active_ray_idx - active_ray_idx, rays, hits, NULL);
        }
    }
}

/* intersect_stream.rs:200-382 */
static void simd_occluded_stream_impl(const orc_scene *s, orc_ray *rays, uint8_t *occluded, uint32_t n,
                                      orc_stats *st, stream_scratch *sc)
{
    st->total_rays += n;
    if (n == 0) return;
    const size_t ray_list_len = (size_t)n * 32;
    uint32_t *base = scratch_get(sc, ray_list_len * 2);
    uint32_t *ray_lists[2] = {base, base + ray_list_len};
    size_t ray_list_sizes[2] = {0, 0};
    for (uint32_t i = 0; i < n; i++) ray_lists[0][i] = i;
    ray_list_sizes[0] = n;
    stream_item stack[64 * 4];
    int sp = 0;
    stack[sp].n = s->root; stack[sp].list = 0; stack[sp].start = 0; sp++;
    while (sp) {
        const stream_item it = stack[--sp];
        size_t active_ray_idx = it.start;
        const size_t last_active_ray_idx = ray_list_sizes[it.list];
        ray_list_sizes[it.list] = it.start;
        const node *nd = it.n;
        if (!nd->is_leaf) {
            stats_inner(st, last_active_ray_idx - active_ray_idx);
            const node2 bb = node2_setup(nd);
            const size_t orig[2] = {ray_list_sizes[0], ray_list_sizes[1]};
            while (active_ray_idx < last_active_ray_idx) {
                const uint32_t ray_idx_a = ray_lists[it.list][active_ray_idx];
                if (occluded[ray_idx_a]) { /* only the first ray of a pair is skipped, 258-261 */
                    active_ray_idx += 1;
                    continue;
                }
                const uint32_t ray_idx_b =
                    active_ray_idx + 1 < last_active_ray_idx ? ray_lists[it.list][active_ray_idx + 1] : ray_idx_a;
                const unsigned mask = (unsigned)node2_mask(&bb, &rays[ray_idx_a], &rays[ray_idx_b]);

                ray_lists[0][ray_list_sizes[0]] = ray_idx_a;
                ray_lists[1][ray_list_sizes[1]] = ray_idx_a;
                ray_list_sizes[0] += mask & 1u;
                ray_list_sizes[1] += (mask >> 1) & 1u;
                if (ray_idx_a != ray_idx_b) {
                    ray_lists[0][ray_list_sizes[0]] = ray_idx_b;
                    ray_lists[1][ray_list_sizes[1]] = ray_idx_b;
                    ray_list_sizes[0] += (mask >> 4) & 1u;
                    ray_list_sizes[1] += (mask >> 5) & 1u;
                }
                active_ray_idx += 2;
            }
            if (ray_list_sizes[1] - orig[1] > 0) { stack[sp].n = nd->child[1]; stack[sp].list = 1; stack[sp].start = orig[1]; sp++; }
            if (ray_list_sizes[0] - orig[0] > 0) { stack[sp].n = nd->child[0]; stack[sp].list = 0; stack[sp].start = orig[0]; sp++; }
        } else {
            stats_leaf(st, last_active_ray_idx - active_ray_idx, nd->count);
            leaf_objs(s, nd, ray_lists[it.list] + active_ray_idx, last_active_ray_idx - active_ray_idx, rays, NULL, occluded);
        }
    }
}

/* chunks of 2^16 rays keep the 32-entries-per-ray lists bounded, as in the scalar restatement */
void orc_simd_intersect_stream(const orc_scene *s, orc_ray *rays, uint32_t *hits, uint32_t n, orc_stats *st)
{
    orc_stats dummy;
    if (!st) { memset(&dummy, 0, sizeof dummy); st = &dummy; }
    stream_scratch sc = {0, 0};
    for (uint32_t i = 0; i < n; i++) hits[i] = ORC_NONE;
    const uint32_t chunk = 1u << 16;
    for (uint32_t off = 0; off < n; off += chunk) {
        const uint32_t m = n - off < chunk ? n - off : chunk;
        simd_intersect_stream_impl(s, rays + off, hits + off, m, st, &sc);
    }
    free(sc.buf);
}

void orc_simd_occluded_stream(const orc_scene *s, orc_ray *rays, uint8_t *occluded, uint32_t n, orc_stats *st)
{
    orc_stats dummy;
    if (!st) { memset(&dummy, 0, sizeof dummy); st = &dummy; }
    stream_scratch sc = {0, 0};
    const uint32_t chunk = 1u << 16;
    for (uint32_t off = 0; off < n; off += chunk) {
        const uint32_t m = n - off < chunk ? n - off : chunk;
        simd_occluded_stream_impl(s, rays + off, occluded + off, m, st, &sc);
    }
    free(sc.buf);
}

const orc_trace_fns orc_simd_trace_fns = {simd_intersect_stream_impl, simd_occluded_stream_impl, orc_simd_intersect1,
                                          orc_simd_occluded1};
