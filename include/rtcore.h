/*
 * rtcore.h -- C ABI of the MI355X ray-tracing core (librtcore.so).
 *
 * The reference (sati290/rust-raytracer) is a single Rust binary with no FFI
 * or plugin layer.  Its de-facto trace boundary is four methods on `Bvh` and
 * the per-tile integrator function pointer chosen in `main`; each entry point
 * below names the reference item it stands in for (paths relative to the
 * reference repository root).  Signatures use plain pointers and sizes only.
 *
 * Conventions
 *   - every function returns RT_OK (0) or an RT_ERR_* code and never throws or
 *     aborts across the ABI (the reference panics instead: src/scene.rs:31,48,
 *     src/bvh/intersect1.rs:115 ArrayVec overflow); rt_last_error() returns a
 *     thread-local message for the last failing call.
 *   - "host" entry points take caller-owned host buffers and copy them; the
 *     "_device" variants take device pointers of the context's GPU and run on
 *     the given HIP stream (hipStream_t passed as void*, NULL = default
 *     stream) without synchronising it.
 *   - one in-flight call per rt_ctx; an rt_scene is immutable after creation.
 */
#ifndef RTCORE_H
#define RTCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID 1 /* bad argument                                   */
#define RT_ERR_HIP 2     /* HIP runtime failure (message has the HIP text) */
#define RT_ERR_NOMEM 3
#define RT_ERR_LIMIT 4   /* size beyond a documented limit                 */

#define RT_NO_HIT 0xFFFFFFFFu /* Option<u32>::None of src/bvh/intersect1.rs:109 */

typedef struct rt_ctx rt_ctx;     /* one per GPU                         */
typedef struct rt_scene rt_scene; /* device-resident BVH + triangles     */

/* src/ray.rs:6-24 StreamRay, repr(C, align(16)), 48 bytes.  origin_far.w is
 * the far distance and is shrunk to the hit distance by closest-hit calls;
 * dir_recip_near.xyz = 1/direction (unclamped), .w = near. */
typedef struct rt_stream_ray {
    float origin_far[4];
    float dir_recip_near[4];
    float direction[4];
} rt_stream_ray;

/* src/trace_stats.rs:4-12.  The GPU traverses one ray per lane, so
 * inner_visit == inner_rays and leaf_visit == leaf_rays here (as in the
 * reference's single-ray traversal); total_rays, inner_rays, leaf_rays,
 * leaf_objs and obj_intersect are the sums the reference counts per ray. */
typedef struct rt_trace_stats {
    uint64_t total_rays;
    uint64_t inner_visit;
    uint64_t inner_rays;
    uint64_t leaf_visit;
    uint64_t leaf_rays;
    uint64_t leaf_objs;
    uint64_t obj_intersect;
} rt_trace_stats;

/* src/bvh/build.rs:5-11 BvhStats */
typedef struct rt_bvh_stats {
    uint32_t num_leaves;
    uint32_t num_objs;
    uint32_t max_depth;
    uint32_t max_objs;
} rt_bvh_stats;

typedef struct rt_scene_info {
    uint32_t num_triangles;
    uint32_t num_inner_nodes;
    uint32_t root_ref;   /* canonical flat encoding, see rt_flat_node */
    uint32_t root_count;
    uint64_t device_bytes;
} rt_scene_info;

/* Canonical flattened tree (pre-order, left child follows its parent) for
 * inspection and builder parity checks.  bx/by/bz lanes are
 * {minL,minR,maxL,maxR} (src/bvh.rs:19-23).  Child reference: bit 31 clear =
 * inner node index; bit 31 set = leaf, low 31 bits = first slot in the
 * leaf-ordered triangle array, *_count = number of triangles. */
typedef struct rt_flat_node {
    float bx[4];
    float by[4];
    float bz[4];
    uint32_t left;
    uint32_t right;
    uint32_t left_count;
    uint32_t right_count;
} rt_flat_node;

/* src/triangle_opt.rs:10-14 in leaf order plus the mesh triangle id that
 * src/bvh.rs:32 keeps in object_indices. */
typedef struct rt_flat_tri {
    float v0[3];
    float v0v1[3];
    float v0v2[3];
    uint32_t id;
    uint32_t pad[2];
} rt_flat_tri;

/* Camera::new inputs, src/camera.rs:25 */
typedef struct rt_camera {
    float eye[3];
    float target[3];
    float up[3];
    float fov_deg;
} rt_camera;

/* src/light.rs:4-16, intensity = color * power */
typedef struct rt_light {
    float pos[3];
    float intensity[3];
} rt_light;

/* Arguments of the reference's integrator call (src/main.rs:143-149) plus the
 * image size it hard-codes (src/main.rs:103-104) and the sharding this build
 * adds.  Tiles are the reference's row-major tile_size x tile_size grid
 * (src/main.rs:61-98); a call renders the tiles i with
 * i % shard_count == shard_index, samples [sample_offset, sample_offset+samples)
 * of each of their pixels.  Random numbers are keyed by (seed, pixel, sample,
 * dimension), so the image does not depend on how a frame is split. */
#define RT_RENDER_VISIT_STATS 1u /* fill the per-visit counters of rt_trace_stats, not only total_rays */

typedef struct rt_render_params {
    uint32_t width, height;
    uint32_t samples;          /* src/args.rs:24, default 128           */
    uint32_t sample_offset;
    uint32_t max_bounces;      /* src/args.rs:27, default 2             */
    uint32_t tile_size;        /* src/args.rs:30, default 16            */
    uint32_t shard_index;
    uint32_t shard_count;      /* 0 is read as 1                        */
    uint32_t samples_per_pass; /* 0 = library default                   */
    uint32_t flags;            /* RT_RENDER_*                           */
    uint64_t seed;             /* src/args.rs:36, default 1235468       */
} rt_render_params;

const char *rt_last_error(void);
const char *rt_version(void);

/* No reference counterpart (the reference has no device).  device_id is a HIP
 * device ordinal. */
int rt_ctx_create(int device_id, rt_ctx **out);
void rt_ctx_destroy(rt_ctx *ctx);
/* Tuning knobs (no reference counterpart).  Names and ranges:
 *   "coop_steps"        loop iterations after which a ray is finished by its whole
 *                       wave (0 = never; default 1024).  Rays handled that way keep their exact
 *                       hit distance; their visit counters and the id on an
 *                       exact-distance tie may differ from the single-lane walk.  Sets
 *                       "coop_steps_binary" as well.
 *   "coop_steps_binary" the same for the forms on the binary tree alone (0, 1, 4; default 512: their
 *                       rays take about twice the iterations, and the unoccluded shadow rays of the
 *                       left-first any-hit walk are the longest of all)
 *   "min_active"        1..64, a wave fetches new rays when fewer lanes are busy (default 40);
 *   "min_active_depth0" the same for the depth-0 launches of rt_render (default 20: their rays
 *                       are coherent and a refill costs more than a few idle lanes)
 *   "traversal_depth0", "traversal_deep", "traversal_stream"   traversal form of the
 *                       depth-0 launches of rt_render, of its deeper launches, and of the
 *                       stream entry points: 0 = while-while loop on the binary tree, 1 = one
 *                       item (node or triangle) per iteration on the binary tree, 2 = one
 *                       item per iteration on the quantized 4-wide tree, 3 = while-while loop
 *                       on the quantized 4-wide tree.  "traversal_stream" also takes 4 = form 1 in
 *                       the visit order of the reference's single-ray closest hit
 *                       (src/bvh/intersect1.rs:109-210: left child first, boxes culled against
 *                       the far distance the ray arrived with).  Defaults 3, 2, 1.  Forms 0 and 1 walk
 *                       the reference's tree in the reference's order (exact distances, ids
 *                       and visit counters); 2 and 3 walk finer, conservatively quantized
 *                       boxes over the same triangles in another order: same distances except
 *                       where two surfaces meet within rounding, ids may differ on exact
 *                       ties, no visit counters, and 1/d is bounded as in the reference's
 *                       packet rays (src/simd_ray.rs:46-47) so that axis-parallel rays are
 *                       culled properly.  Calls that ask for visit counters always run form
 *                       0 or 1.  "traversal_deep" and "traversal_stream" also take four
 *                       experimental variants of form 2 with the same results (kept for
 *                       measurements, none is faster): 5 = two rays per lane, the loads of
 *                       one issued behind the arithmetic of the other; 6 = a lane walks
 *                       nodes and tests the triangles of the last leaf it reached side by
 *                       side, one node and one triangle per iteration; 7 = lanes only walk
 *                       nodes and the triangles of the leaves they reach go through a job
 *                       queue of the wave, tested 64 at a time by whichever lanes are free
 *                       (of triangles at exactly the same distance the lowest id wins); 8 = an
 *                       iteration at a leaf loads its two triangles together with the item that
 *                       follows the leaf (same visits in the same order as form 2: same ids too).
 *                       rt_intersect_stream_noalloc runs form 2 where 5 or 7 is selected.
 *   "traversal_deep_connect" form of the deeper any-hit launches of rt_render alone
 *                       (same values as "traversal_deep"; -1, the default, follows it)
 *   "fuse_primary"      0/1 (default 1): depth 0 of rt_render without ray and PathInfo queues;
 *                       the extend launch makes the camera rays itself.  Same image bit for
 *                       bit; not used by calls that ask for visit counters.
 *   "fast_tree_buckets" 0 or 2..64 (default 32), read by rt_scene_create: the quantized 4-wide
 *                       tree of forms 2 and 3 is collapsed from a tree of its own, built like the
 *                       reference's (binned SAH) but with that many buckets, every axis tried and
 *                       leaves of at most "fast_tree_max_leaf" triangles; 0 collapses it from the
 *                       reference's tree.  Forms 0 and 1 always walk the reference's tree.
 *   "fast_tree_device_build" 0 = the tree of forms 2 and 3 is built on the host, 1 (default) = on the
 *                       GPU for meshes of 2^18 triangles and more, 2 = always on the GPU, together with the
 *                       4-wide collapse, quantization and packing of the traversal items; 3 = the binary
 *                       tree on the GPU, the items on the host.  The tree and the items are the same either
 *                       way (rt_bvh_build_search_device, rt_scene_get_wide_items).
 *   "fast_tree_max_leaf" 1..3 (default 2), read by rt_scene_create: most triangles in a leaf of
 *                       the tree of forms 2 and 3
 *   "fast_tree_align"   0, 1, 3 or 4 (default 4), read by rt_scene_create: placement of the 48-byte items of
 *                       that tree against 128-byte cache lines.  0 = back to back; 1 = nodes on 64-byte
 *                       boundaries; 3 = no item lies across a line (gaps only where one would); 4 = as 3, and a
 *                       leaf of one or two triangles lies within one line (6 % more memory, 18 % fewer L1
 *                       fills on incoherent rays).  Same tree, same results.
 *   "sort_extend_from", "sort_connect_from" first depth whose queue is Morton-ordered
 *                       (defaults: never = 0xFFFFFFFF for the bounce rays, 1 for the shadow rays)
 *   "sort_bits"         3..30 key bits that take part in the order
 *   "sort_octant"       0/1 (default 0): bounce rays carry the octant of their direction in the three
 *                       lowest ordered key bits
 *   "timing"            0/1 per-launch HIP events
 *   "overlap_connect"   0/1 (default 0): rt_render runs the any-hit launch of depth d on a stream of
 *                       the context's own beside the closest-hit launch of depth d + 1, so that the tail
 *                       of one persistent launch can be filled by the start of the other.  Same image bit
 *                       for bit.  Per-launch event times then overlap and no longer add up to the frame.
 *                       Measured 0.7-1 % slower on one GPU, hence off.
 *   "workspace_limit_mb" test hook: a render workspace beyond this many MiB counts as an allocation
 *                       failure (0 = off), which exercises the smaller-pass fallback of rt_render */
int rt_ctx_set_option(rt_ctx *ctx, const char *name, int64_t value);

/* Upper bound of paths per render pass, 4096..2^30 (default 2^28: a 1920x1080 frame at 128 samples
 * per pixel in one pass; the workspace, about 180 bytes per path, is sized lazily from what a
 * call needs, so only frames that large take the 48 GB that go with it). */
int rt_ctx_set_max_paths(rt_ctx *ctx, uint64_t max_paths);

/* Bvh::build, src/bvh/build.rs:45-63 (binned SAH, 8 buckets, leaf <= 8) on the
 * indexed mesh of src/mesh.rs:9-15; flattens and uploads once.  positions:
 * 3*num_vertices floats; indices: 3*num_triangles; geom_normals:
 * 3*num_triangles (src/mesh.rs:180-188). */
int rt_scene_create(rt_ctx *ctx, const float *positions, uint32_t num_vertices,
                    const uint32_t *indices, uint32_t num_triangles, const float *geom_normals,
                    rt_scene **out);
void rt_scene_destroy(rt_scene *scene);
/* BvhStats printed at src/bvh/build.rs:51 */
int rt_scene_stats(const rt_scene *scene, rt_bvh_stats *out);
int rt_scene_get_info(const rt_scene *scene, rt_scene_info *out);
/* nodes: num_inner_nodes entries, tris: num_triangles entries (either may be NULL) */
int rt_scene_get_flat(const rt_scene *scene, rt_flat_node *nodes, rt_flat_tri *tris);

/* Bvh::build alone, src/bvh/build.rs:45-63, on the host: the same builder
 * rt_scene_create runs, without touching a GPU.  nodes must hold
 * num_triangles entries (a binary tree over n leaves has fewer inner nodes),
 * order num_triangles entries (leaf slot -> mesh triangle id,
 * object_indices of src/bvh.rs:32).  info->device_bytes is 0.  threads = 0
 * picks a default. */
int rt_bvh_build_host(const float *positions, uint32_t num_vertices, const uint32_t *indices,
                      uint32_t num_triangles, uint32_t threads, rt_flat_node *nodes, uint32_t *order,
                      rt_scene_info *info, rt_bvh_stats *stats);

/* The builder of the tree the fast traversal forms walk (no counterpart in the
 * reference): binned SAH like src/bvh/build.rs:45-207, but with `buckets`
 * buckets (2..64), every axis that has a centroid extent tried instead of the
 * largest one only, candidates with an empty side skipped, and leaves of at
 * most max_leaf triangles (1..8; more only where all centroids coincide).
 * Same output conventions as rt_bvh_build_host.  rt_scene_create runs it with
 * (fast_tree_buckets, fast_tree_max_leaf) unless the former is 0. */
int rt_bvh_build_search_host(const float *positions, uint32_t num_vertices, const uint32_t *indices,
                             uint32_t num_triangles, uint32_t threads, uint32_t buckets, uint32_t max_leaf,
                             rt_flat_node *nodes, uint32_t *order, rt_scene_info *info, rt_bvh_stats *stats);

/* The same builder run on the GPU of `ctx`, breadth first: one pass per tree level over all triangles
 * instead of one recursion per node (no counterpart in the reference, whose builder recurses on the CPU
 * with rayon::join, src/bvh/build.rs:174-193).  The same nodes, boxes, leaf ranges and BvhStats as
 * rt_bvh_build_search_host; nodes are numbered breadth first instead of in pre-order and the
 * triangles inside a leaf may come in another order.  rt_scene_create uses it for meshes of 2^18
 * triangles and more (option "fast_tree_device_build") while the host threads build the reference's
 * tree, and falls back to the host builder when the GPU build cannot run (RT_ERR_HIP here). */
int rt_bvh_build_search_device(rt_ctx *ctx, const float *positions, uint32_t num_vertices, const uint32_t *indices,
                               uint32_t num_triangles, uint32_t buckets, uint32_t max_leaf, rt_flat_node *nodes,
                               uint32_t *order, rt_scene_info *info, rt_bvh_stats *stats);
/* Box of all vertices of the scene's mesh (no counterpart in the reference; the library scales the
 * Morton keys of its ray-queue sorts with it). */
int rt_scene_bounds(const rt_scene *scene, float lo[3], float hi[3]);
/* *out = 0 when the scene's fast-form tree came from the host builder, 1 when from the device-side builder,
 * 2 when the traversal items (4-wide collapse, quantization, packing) were made on the device as well */
int rt_scene_fast_tree_on_device(const rt_scene *scene, int *out);
/* The traversal items of the fast forms as they lie in device memory: *num_units 16-byte units (a 4-wide
 * quantized node or a triangle is three of them).  units may be NULL to ask for the count.  No counterpart
 * in the reference; tests compare the device-made items with the host-made ones through it. */
int rt_scene_get_wide_items(const rt_scene *scene, void *units, uint64_t capacity_units, uint64_t *num_units);

/* 4-wide traversal form of the tree (no counterpart in the reference, whose tree
 * is binary, src/bvh.rs:19-34).
 *
 * rt_wide_node is the exact form: lo/hi are [axis][child], the boxes of up to
 * four descendants of one binary node, copied bit for bit from the binary
 * nodes.  Child references: bit 31 clear = wide node index; bit 31 set = leaf,
 * with bit 30 clear: bits 3..29 = first slot in the leaf-ordered triangle
 * array and bits 0..2 = count-1; bit 30 set: low 30 bits number the leaves of
 * more than 8 triangles (src/bvh/build.rs:93-101) in node order.  An unused
 * slot has lo = +inf, hi = -inf and repeats child[0].  bin = index of the
 * canonical binary node the wide node was collapsed from.
 *
 * rt_qwide_node (48 bytes) is the form the device traverses: the same node with its child boxes
 * on an 8-bit grid, plane = origin[axis] + byte * step[axis], where step is the f32 whose bits
 * 21..30 are the ten-bit field `axis` of exps (an exponent and two mantissa bits), child i
 * in byte i of lo[axis] / hi[axis], every box rounded outward (by at least 2^-19 of the node's
 * largest coordinate or extent), so a ray that meets an exact child box meets the quantized one.
 * Unused slots have lo byte 255 and hi byte 0.  Nodes and triangles of the wide tree share one
 * array of 16-byte units on the device (three units each); the children of a node -- a node for an
 * inner child, its 1..3 triangles (48 bytes each, rt_flat_tri) for a leaf child -- are stored back
 * to back from unit base4 >> 2, and byte i of meta is
 *     (units from there to child i) << 2 | triangles of child i (0 = inner node),
 * which makes  base4 + byte i  the traversal reference (unit << 2 | triangles) of child i. */
typedef struct rt_wide_node {
    float lo[3][4];
    float hi[3][4];
    uint32_t child[4];
    uint32_t bin;
    uint32_t pad[3];
} rt_wide_node;

typedef struct rt_qwide_node {
    float origin[3];
    uint32_t exps;
    uint32_t base4;
    uint32_t meta;
    uint32_t lo[3];
    uint32_t hi[3];
} rt_qwide_node;

/* Derives the 4-wide forms from a canonical flat tree on the host: children of
 * inner children are pulled up, largest surface area first, until a node has
 * four; the children of a wide node are then ordered largest box first (the
 * order in which the any-hit walk visits hit children).  out and qout may be NULL and otherwise
 * hold num_nodes entries each (the wide tree never has more nodes than the binary one); nodes are
 * written in depth-first pre-order.  wide_root_ref is 0, or the leaf reference when the
 * root is a leaf.  The device form (qout, and where its items go: node_unit[i] = first unit of
 * wide node i, slot_unit[s] = first unit of the triangle in leaf slot s, *num_units = units in
 * all; each may be NULL) needs every leaf to hold at most 3 triangles of [0, num_triangles):
 * pass the output of rt_bvh_refine_host (max_tris <= 3) or of rt_bvh_build_search_host
 * (max_leaf <= 3), otherwise RT_ERR_INVALID.  rt_scene_create runs the same code. */
int rt_bvh_wide_host(const rt_flat_node *nodes, uint32_t num_nodes, uint32_t root_ref, uint32_t root_count,
                     uint32_t num_triangles, rt_wide_node *out, rt_qwide_node *qout, uint32_t *node_unit,
                     uint32_t *slot_unit, uint32_t *num_wide, uint32_t *wide_root_ref, uint32_t *wide_depth,
                     uint32_t *num_units);

/* Finer leaves for the wide tree (no reference counterpart; the reference stops
 * at 8 triangles per leaf, src/bvh/build.rs:80-81).  Takes the canonical flat
 * tree and leaf order of rt_bvh_build_host and returns a copy in which every
 * leaf holding more than max_tris triangles is replaced by a small
 * binary subtree over its own triangles (ordered by centroid on the leaf's
 * largest centroid-extent axis, cut where area(A)*|A| + area(B)*|B| is least;
 * leaves of more than 8 triangles or without a centroid extent, the work of
 * src/bvh/build.rs:93-101, are cut in halves by slot; child boxes are the
 * exact unions of the triangles' vertex bounds).  The
 * original nodes keep their indices, new nodes are appended; xorder is a
 * permutation of order that only moves triangles inside their reference leaf.
 * xnodes must hold num_nodes + num_triangles entries, xorder num_triangles.
 * rt_scene_create builds the wide tree over exactly this (max_tris = 3) when
 * the option "fast_tree_buckets" is 0, and over rt_bvh_build_search_host's
 * tree otherwise. */
int rt_bvh_refine_host(const float *positions, uint32_t num_vertices, const uint32_t *indices,
                       uint32_t num_triangles, const rt_flat_node *nodes, uint32_t num_nodes, uint32_t root_ref,
                       uint32_t root_count, const uint32_t *order, uint32_t max_tris, rt_flat_node *xnodes,
                       uint32_t *xorder, uint32_t *num_xnodes, uint32_t *xroot_ref, uint32_t *xroot_count,
                       uint32_t *extra_depth);

/* Bvh::intersect_stream, src/bvh/intersect_stream.rs:10-15: closest hit for n
 * rays.  hits[i] = mesh triangle id or RT_NO_HIT; rays[i].origin_far[3] is
 * shrunk to the hit distance.  Accept rule near <= t < far
 * (src/bvh/intersect_stream.rs:418).  stats may be NULL.  The host variant
 * uploads and traces in chunks of 2^20 rays (upload of the next chunk overlaps
 * the trace of the current one) and brings back hit id and far distance only;
 * every other byte of rays[] is left as the caller wrote it. */
int rt_intersect_stream(rt_scene *scene, rt_stream_ray *rays, uint32_t *hits, uint32_t n,
                        rt_trace_stats *stats);
int rt_intersect_stream_device(rt_scene *scene, rt_stream_ray *d_rays, uint32_t *d_hits,
                               uint32_t n, rt_trace_stats *stats, void *hip_stream);

/* rt_intersect_stream_device for callers that cannot allocate: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference (which
 * has no device).  Same inputs, same outputs and the same traversal form (option
 * "traversal_stream") as rt_intersect_stream_device without statistics, but everything the
 * kernels need beyond d_rays and d_hits -- the ray-fetch counters and the part of the per-lane
 * traversal stacks that does not fit in LDS -- lives in one fixed device array of
 * RT_NOALLOC_SCRATCH_BYTES that is part of the library's code object (every GPU the library runs
 * on holds its own copy until the process ends); the call makes no hipMalloc, hipFree, event
 * creation or other allocating call.  The price: a grid of 512 blocks instead of every resident
 * one, no straggler hand-off (option "coop_steps" is not used), and two limits, beyond which the
 * call returns RT_ERR_LIMIT: n <= RT_NOALLOC_MAX_RAYS, and a scene whose trees can defer at most
 * RT_NOALLOC_MAX_STACK nodes per ray of the selected form (tree depth + 1 for the binary forms, three
 * times the depth of the 4-wide tree + 1 for the others; the stand-in scenes need 24 to 46).  All calls on one GPU share the array: calls that overlap there (two streams, two
 * contexts, two threads) race; calls on one stream, or otherwise ordered, are fine. */
#define RT_NOALLOC_MAX_RAYS 268435456u /* 2^28 */
#define RT_NOALLOC_MAX_STACK 56
#define RT_NOALLOC_SCRATCH_BYTES 17825792u /* 17 MiB */
int rt_intersect_stream_noalloc(rt_scene *scene, rt_stream_ray *d_rays, uint32_t *d_hits, uint32_t n,
                                void *hip_stream);

/* Bvh::occluded_stream / Bvh::occluded1, src/bvh/intersect_stream.rs:200-205,
 * src/bvh/intersect1.rs:7: any hit in [near, far).  occluded[i] = 0/1.  Rays
 * are not modified. */
int rt_occluded_stream(rt_scene *scene, const rt_stream_ray *rays, uint8_t *occluded, uint32_t n,
                       rt_trace_stats *stats);
int rt_occluded_stream_device(rt_scene *scene, const rt_stream_ray *d_rays, uint8_t *d_occluded,
                              uint32_t n, rt_trace_stats *stats, void *hip_stream);

/* Bvh::intersect1, src/bvh/intersect1.rs:109-210, and Bvh::occluded1, src/bvh/intersect1.rs:7-106:
 * one ray.  rt_intersect1 walks in that function's own order -- left child first, boxes culled
 * against the far distance the ray arrives with, triangles accepted against the shrinking far
 * -- so its visit counters are that function's, and on an exact-distance tie it returns
 * the triangle that function returns; *hit = mesh triangle id or RT_NO_HIT and
 * ray->origin_far[3] is shrunk to the hit distance.  One ray per call keeps one lane of one wave
 * busy: these exist so that a host written against the single-ray methods links and
 * gives the same answers; arrays of rays go to the stream entry points (option
 * "traversal_stream" = 4 gives them this visit order). */
int rt_intersect1(rt_scene *scene, rt_stream_ray *ray, uint32_t *hit, rt_trace_stats *stats);
int rt_occluded1(rt_scene *scene, const rt_stream_ray *ray, uint8_t *occluded, rt_trace_stats *stats);

/* The integrator call of src/main.rs:133-161 for a whole frame (all tiles of
 * the shard) instead of one tile: integrate_tile_stream,
 * src/integrators/integrate_stream.rs:56-164, as generate / extend / shade /
 * connect kernels over compacted ray queues.  accum is width*height float4
 * (sum r, sum g, sum b, sum of sample counts; src/main.rs:108); contributions
 * are ADDED to it.  stats may be NULL. */
int rt_render(rt_scene *scene, const rt_camera *camera, const rt_light *light,
              const rt_render_params *params, float *accum, rt_trace_stats *stats);
int rt_render_device(rt_scene *scene, const rt_camera *camera, const rt_light *light,
                     const rt_render_params *params, float *d_accum, rt_trace_stats *stats,
                     void *hip_stream);

/* One-node multi-GPU rendering (no reference counterpart: the reference is one
 * process; its tiles are independent, src/main.rs:157-161, which is what makes
 * the split exact).  One process per GPU, full scene replica each.
 *
 * rt_shard_plan_make: the part of a frame of `samples` samples per pixel that
 * rank `rank` of `world` renders.  mode RT_SHARD_SAMPLE: a contiguous range of
 * the samples of every pixel (ranks below samples % world take one more);
 * RT_SHARD_TILE: all samples of the tiles i with i % world == rank.  Pure host
 * code. */
#define RT_SHARD_SAMPLE 0
#define RT_SHARD_TILE 1
typedef struct rt_shard_plan {
    uint32_t samples;
    uint32_t sample_offset;
    uint32_t shard_index;
    uint32_t shard_count;
} rt_shard_plan;
int rt_shard_plan_make(uint32_t rank, uint32_t world, int mode, uint32_t samples, uint32_t sample_offset,
                       rt_shard_plan *out);

/* Renders this rank's part of the frame described by `frame` (its samples /
 * sample_offset are the whole frame's; shard_index / shard_count are ignored)
 * into d_accum (device, width*height*4 floats, added to) and, when nccl_comm is
 * not NULL, sum-reduces d_accum to rank 0 of that communicator on hip_stream
 * (in place; channel w sums the sample counts, src/main.rs:39).  nccl_comm is an
 * ncclComm_t the host created over the node's GPUs (RCCL); librccl.so is
 * opened on first use, so hosts that never call this do not need it.  stats
 * counts this rank's rays only. */
int rt_node_render(rt_scene *scene, const rt_camera *camera, const rt_light *light,
                   const rt_render_params *frame, uint32_t rank, uint32_t world, int mode, void *nccl_comm,
                   float *d_accum, rt_trace_stats *stats, void *hip_stream);

/* Primary rays of a frame pass as StreamRays, generate_rays of
 * src/integrators/integrate_stream.rs:14-54 for every pixel of the frame:
 * rays[(y*width + x)*samples + s].  Used for per-ray parity checks. */
int rt_generate_primary(rt_ctx *ctx, const rt_camera *camera, const rt_render_params *params,
                        rt_stream_ray *rays);

/* color_vec_to_rgb_norm_gamma, src/main.rs:34-41: rgb = sqrt(sum/w)*255 with
 * Rust's saturating float->u8 cast.  Host buffers. */
int rt_resolve(rt_ctx *ctx, const float *accum, uint32_t width, uint32_t height, uint8_t *rgb);

/* Duration in milliseconds of the traversal kernels of the most recent trace
 * or render call on this context, measured with HIP events on the stream the
 * kernels ran on: [0] closest-hit (extend), [1] any-hit (connect), [2] all
 * other kernels; launches[k] = number of launches summed into ms[k].  The
 * events are synchronised by this call. */
int rt_ctx_last_kernel_ms(rt_ctx *ctx, float ms[3], uint32_t launches[3]);

/* Per-launch form of rt_ctx_last_kernel_ms: the first min(cap, count) launches
 * of the last call in issue order; kind[i] = 0 closest-hit, 1 any-hit,
 * 2 other.  *count receives the total number of timed launches. */
int rt_ctx_last_launch_ms(rt_ctx *ctx, float *ms, uint32_t *kind, uint32_t cap, uint32_t *count);

/* Diagnostic: the ordering step the renderer applies to its ray queues, on
 * host buffers.  perm receives the indices 0..n-1 ordered by bits
 * [32-bits, 32) of keys (stable).  No reference counterpart: the reference
 * never reorders rays. */
int rt_debug_sort_keys(rt_ctx *ctx, const uint32_t *keys, uint32_t n, uint32_t bits, uint32_t *perm);

/* Counters of the most recent trace or render call that asked for statistics,
 * split by traversal kind: out[0] closest-hit (extend) launches, out[1]
 * any-hit (connect) launches.  Their sum is what that call added to its
 * rt_trace_stats argument (src/trace_stats.rs keeps one total only). */
int rt_ctx_last_stats_split(rt_ctx *ctx, rt_trace_stats out[2]);

/* Counters of the most recent rt_render* call that asked for statistics by path depth (0 = camera
 * rays and their shadow rays), out[0] closest-hit launches, out[1] any-hit launches: rays[d] always;
 * inner_rays[d] / obj_intersect[d] with RT_RENDER_VISIT_STATS.  Depths from RT_STAT_DEPTHS-1 on
 * share the last slot.  No reference counterpart (src/trace_stats.rs keeps totals); this is what
 * a per-kernel roofline needs, since each depth is a launch of its own. */
#define RT_STAT_DEPTHS 8
typedef struct rt_depth_stats {
    uint64_t rays[RT_STAT_DEPTHS];
    uint64_t inner_rays[RT_STAT_DEPTHS];
    uint64_t obj_intersect[RT_STAT_DEPTHS];
} rt_depth_stats;
int rt_ctx_last_depth_stats(rt_ctx *ctx, rt_depth_stats out[2]);

/* Longest single ray (node visits + triangle tests) of the most recent call
 * that collected per-visit statistics: out[0] closest-hit, out[1] any-hit.  A
 * kernel cannot end before its longest ray does; this is the tail diagnostic. */
int rt_ctx_last_max_ray_steps(rt_ctx *ctx, uint64_t out[2]);

/* Most deferred nodes any single ray held on its traversal stack at once, same call and split as
 * rt_ctx_last_max_ray_steps.  The reference's stacks are ArrayVecs of 32 (single ray,
 * src/bvh/intersect1.rs:12,115) and 64 (stream, src/bvh/intersect_stream.rs:21) entries and panic
 * beyond that; here the first RT_LDS_STACK_ENTRIES entries of a lane are kept in LDS and deeper
 * ones in a global overflow area sized from the scene's tree depth, so any depth up to 1024 works. */
#define RT_LDS_STACK_ENTRIES 24
int rt_ctx_last_max_stack_entries(rt_ctx *ctx, uint64_t out[2]);

/* Path slots per pass the last rt_render / rt_render_device call planned with (no counterpart in
 * the reference).  At most the context's max_paths; smaller when the workspace for a pass of that
 * size could not be allocated: the call then releases what it had grown, halves the pass and plans
 * the frame again (down to one tile or 2^16 paths) before it gives up with RT_ERR_NOMEM. */
int rt_ctx_last_pass_paths(rt_ctx *ctx, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif
