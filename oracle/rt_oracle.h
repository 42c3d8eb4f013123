/*
 * rt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement, in plain C, of the hot path of sati290/rust-raytracer
 * (BVH build, BVH traversal, ray/triangle test, diffuse + point-light shading,
 * tile integrator).  It exists to CHECK the HIP path; nothing in the product
 * library links, loads or calls it.  Only tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py may use it.
 *
 * PARITY UNPINNED: the reference ships no tests, no golden vectors and no
 * buildable binary in this environment (no Rust toolchain, scene assets are
 * Git-LFS stubs), so this restatement is pinned only by its own known-answer
 * tests and by a brute-force all-triangles cross-check.  Arithmetic that lives
 * in third-party crates absent from the reference tree (nalgebra 0.34.1,
 * simba 0.9.1, wide 0.7.33, rand 0.8.5) is restated from their published
 * algorithms and marked [3P] where it appears.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference repository root).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/ray.rs:6-24  StreamRay, repr(C, align(16)), 48 bytes.
 * origin_far = {ox,oy,oz,far}; dir_recip_near = {1/dx,1/dy,1/dz,near};
 * direction = {dx,dy,dz,0}. */
typedef struct orc_ray {
    float origin_far[4];
    float dir_recip_near[4];
    float direction[4];
} orc_ray;

/* src/trace_stats.rs:4-12 */
typedef struct orc_stats {
    uint64_t total_rays;
    uint64_t inner_visit;
    uint64_t inner_rays;
    uint64_t leaf_visit;
    uint64_t leaf_rays;
    uint64_t leaf_objs;
    uint64_t obj_intersect;
} orc_stats;

/* src/bvh/build.rs:5-11 */
typedef struct orc_bvh_stats {
    uint32_t num_leaves;
    uint32_t num_objs;
    uint32_t max_depth;
    uint32_t max_objs;
} orc_bvh_stats;

/* Flattened tree, pre-order (left child directly follows its parent).
 * Child reference: bit 31 clear -> index of an inner node; bit 31 set ->
 * leaf, low 31 bits = first slot in the leaf-ordered triangle array, and the
 * matching count field gives the number of triangles.
 * bx/by/bz lanes are {minL, minR, maxL, maxR} (src/bvh.rs:19-23). */
typedef struct orc_flat_node {
    float bx[4];
    float by[4];
    float bz[4];
    uint32_t left;
    uint32_t right;
    uint32_t left_count;
    uint32_t right_count;
} orc_flat_node;

/* leaf-ordered triangle, src/triangle_opt.rs:10-14 plus the mesh triangle id
 * that src/bvh.rs:32 keeps in object_indices. */
typedef struct orc_flat_tri {
    float v0[3];
    float v0v1[3];
    float v0v2[3];
    uint32_t id;
    uint32_t pad[2];
} orc_flat_tri;

/* src/camera.rs:25-37 inputs */
typedef struct orc_camera {
    float eye[3];
    float target[3];
    float up[3];
    float fov_deg;
} orc_camera;

/* src/light.rs:4-16 (intensity = color * power) */
typedef struct orc_light {
    float pos[3];
    float intensity[3];
} orc_light;

enum {
    ORC_MODE_STREAM = 0,                  /* integrate_stream.rs:56-164  */
    ORC_MODE_STREAM_SHADOW_IMMEDIATE = 1, /* integrate_stream.rs:166-242 */
    ORC_MODE_SINGLE_RAY = 3,              /* tile_integrator.rs:18-99    */
    /* or-ed into a mode: the traversal calls go to the SSE/AVX2 restatement (rt_oracle_simd.c) */
    ORC_MODE_SIMD = 0x100
};

typedef struct orc_render_params {
    uint32_t width, height;   /* src/main.rs:103-104 (hard-coded there)     */
    uint32_t samples;         /* src/args.rs:24                             */
    uint32_t sample_offset;   /* first global sample index of this call     */
    uint32_t max_bounces;     /* src/args.rs:27                             */
    uint32_t tile_size;       /* src/args.rs:30                             */
    uint32_t shard_index;     /* tiles i with i % shard_count == shard_index */
    uint32_t shard_count;
    uint32_t mode;            /* ORC_MODE_*                                 */
    uint32_t nthreads;        /* worker threads (tiles are the work items)  */
    uint64_t seed;            /* src/args.rs:36                             */
} orc_render_params;

typedef struct orc_scene orc_scene;

/* scene: src/bvh/build.rs:45-63 on an indexed mesh (src/mesh.rs:9-15). */
orc_scene *orc_scene_create(const float *positions, uint32_t num_vertices,
                            const uint32_t *indices, uint32_t num_triangles,
                            const float *geom_normals);
void orc_scene_destroy(orc_scene *s);
void orc_scene_bvh_stats(const orc_scene *s, orc_bvh_stats *out);
uint32_t orc_scene_num_inner(const orc_scene *s);
uint32_t orc_scene_num_triangles(const orc_scene *s);
/* root reference in the flat encoding (a leaf reference when the whole mesh
 * fits one leaf) and its triangle count (0 for an inner root). */
void orc_scene_root(const orc_scene *s, uint32_t *root_ref, uint32_t *root_count);
void orc_scene_flatten(const orc_scene *s, orc_flat_node *nodes, orc_flat_tri *tris);

/* ray helpers */
void orc_ray_new(orc_ray *r, const float origin[3], const float direction[3],
                 float near_t, float far_t);
float orc_tri_intersect(const float v0[3], const float v0v1[3], const float v0v2[3],
                        const float origin[3], const float direction[3]);

/* traversal.  hit ids use 0xFFFFFFFF for None. */
uint32_t orc_intersect1(const orc_scene *s, orc_ray *ray, orc_stats *stats);
int orc_occluded1(const orc_scene *s, const orc_ray *ray, orc_stats *stats);
void orc_intersect1_n(const orc_scene *s, orc_ray *rays, uint32_t *hits, uint32_t n,
                      orc_stats *stats);
void orc_occluded1_n(const orc_scene *s, const orc_ray *rays, uint8_t *occluded, uint32_t n,
                     orc_stats *stats);
void orc_intersect_stream(const orc_scene *s, orc_ray *rays, uint32_t *hits, uint32_t n,
                          orc_stats *stats);
void orc_occluded_stream(const orc_scene *s, orc_ray *rays, uint8_t *occluded, uint32_t n,
                         orc_stats *stats);
/* all-triangles loop in mesh order with the same triangle arithmetic. */
void orc_intersect_brute(const orc_scene *s, orc_ray *rays, uint32_t *hits, uint32_t n);

/* The same four traversals written with the reference's own intrinsics, instruction for instruction
 * (rt_oracle_simd.c: SSE + FMA single-ray code of src/bvh/intersect1.rs, AVX2 two-rays-by-two-children
 * node filter and four-ray leaf loop of src/bvh/intersect_stream.rs).  Second witness for the scalar
 * restatement above: NaN handling of min/max and the blend lanes come from the hardware here. */
uint32_t orc_simd_intersect1(const orc_scene *s, orc_ray *ray, orc_stats *stats);
int orc_simd_occluded1(const orc_scene *s, const orc_ray *ray, orc_stats *stats);
void orc_simd_intersect1_n(const orc_scene *s, orc_ray *rays, uint32_t *hits, uint32_t n, orc_stats *stats);
void orc_simd_occluded1_n(const orc_scene *s, const orc_ray *rays, uint8_t *occluded, uint32_t n,
                          orc_stats *stats);
void orc_simd_intersect_stream(const orc_scene *s, orc_ray *rays, uint32_t *hits, uint32_t n,
                               orc_stats *stats);
void orc_simd_occluded_stream(const orc_scene *s, orc_ray *rays, uint8_t *occluded, uint32_t n,
                              orc_stats *stats);

/* shading pieces (exposed for known-answer tests) */
int orc_sample_light(const float dir_out[3], const float normal[3], const float hit_pos[3],
                     const orc_light *light, float dir_in[3], float *dist, float weight[3]);
void orc_sample_diffuse_ray(const float dir_out[3], const float normal[3], float u0, float u1,
                            float dir_in[3], float weight[3]);
void orc_sincos_2pi(float u, float *s, float *c);
uint32_t orc_rng_u32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim);
float orc_rng_f32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim);
/* direction of sample (u0,u1) of pixel (x,y); tile origin fixes the rounding of
 * the incremental walk (src/camera.rs:98-120). */
void orc_camera_sample_dir(const orc_camera *cam, uint32_t width, uint32_t height,
                           uint32_t tile_size, uint32_t x, uint32_t y, float u0, float u1,
                           float dir[3]);
/* primary rays of a whole frame pass in (pixel-major, sample-minor) order:
 * rays[(y*width+x)*samples + s]. */
void orc_generate_primary(const orc_camera *cam, const orc_render_params *p, orc_ray *rays);

/* frame render: accum is width*height float4 (sum r, sum g, sum b, sum samples),
 * src/main.rs:108; contributions are ADDED to it. */
void orc_render(const orc_scene *s, const orc_camera *cam, const orc_light *light,
                const orc_render_params *p, float *accum, orc_stats *stats);
/* src/main.rs:34-41 */
void orc_resolve(const float *accum, uint32_t width, uint32_t height, uint8_t *rgb);

#ifdef __cplusplus
}
#endif
#endif
