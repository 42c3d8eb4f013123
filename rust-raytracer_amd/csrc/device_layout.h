// Device-side data layout of the ray-tracing core (gfx950).
//
// Everything here is uploaded once per scene and then only read.
#pragma once

#include <stdint.h>

namespace rtc {

// Child reference as stored in device nodes and on traversal stacks.
//   bit 31 clear : index of an inner node
//   bit 31 set   : leaf.  bit 30 clear: bits 3..29 = first slot in the
//                  leaf-ordered triangle array, bits 0..2 = count-1 (count 1..8,
//                  the reference's leaf rule, src/bvh/build.rs:80-81).
//                  bit 30 set: low 30 bits index `big_leaves` {first,count}
//                  (leaves made by the zero-centroid-extent rule,
//                  src/bvh/build.rs:93-101, can hold any number of triangles).
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kBigLeafBit = 0x40000000u;
constexpr uint32_t kMaxPackedFirst = 1u << 27;
constexpr uint32_t kDone = 0xFFFFFFFFu; // traversal sentinel (never a valid ref)

// 64-byte inner node: both children's boxes in the reference's lane order
// {minL, minR, maxL, maxR} per axis (src/bvh.rs:19-23) + two child refs.
struct alignas(64) DevNode {
    float bx[4];
    float by[4];
    float bz[4];
    uint32_t left;
    uint32_t right;
    uint32_t pad0;
    uint32_t pad1;
};
static_assert(sizeof(DevNode) == 64, "DevNode must be 64 bytes");

// 4-wide form: the binary tree with every other level removed.  A wide node
// holds up to four descendants of one binary node.
//
// DevWideNode (128 bytes) is the exact form the host builds first: every child
// box is a child box of some binary node, bit for bit; `lo`/`hi` are
// [axis][child].  An unused slot has the empty box (+inf, -inf) and repeats the
// reference of slot 0 (always a leaf in such a node).
struct alignas(128) DevWideNode {
    float lo[3][4];
    float hi[3][4];
    uint32_t child[4]; // same reference encoding as DevNode; inner = wide node index
    uint32_t bin;      // binary node this one was collapsed from
    uint32_t pad[3];
};
static_assert(sizeof(DevWideNode) == 128, "DevWideNode must be 128 bytes");

// DevQNode (48 bytes) is what the traversal reads: the four child boxes on an 8-bit grid local to
// the node, plane = origin[axis] + byte * step[axis], each box rounded OUTWARD by at
// least the margin the builder documents.  A ray that meets an exact child box therefore meets its
// quantized box, so the leaves reached are a superset of those of the exact tree and the triangle
// tests decide the hit as before.  Byte i of every lo/hi word belongs to child i.
//
// Nodes and triangles of the wide tree live in ONE array of 16-byte units (a node and a triangle
// are three units each, so every traversal item is three 16-byte loads).  The children of a node
// are stored back to back in one block -- a 48-byte node for an inner child, its 1..3 triangles for
// a leaf child -- and the node holds the block's first unit and one byte per child:
//     meta byte i = (units from the block start to child i) << 2 | triangles of child i (0 = inner)
// so that the traversal reference of child i is  base4 + byte i  =  unit << 2 | triangles.
// An unused slot has lo = 255, hi = 0 on every axis (the grid step is > 0, so near > far) and
// repeats the meta byte of slot 0, so that even a ray whose slab test degenerates to NaN on every
// axis only re-tests triangles it has already seen.
struct alignas(16) DevQNode {
    float origin[3];
    uint32_t exps;  // grid steps, ten bits per axis (x | y << 10 | z << 20): the top ten bits of a positive
                    // f32 below its sign, i.e. step = 2^(e - 127) * (1 + m / 4) for field e << 2 | m
    uint32_t base4; // first unit of the children block << 2
    uint32_t meta;
    uint32_t lo[3]; // [axis], child i in byte i
    uint32_t hi[3];
};
static_assert(sizeof(DevQNode) == 48, "DevQNode must be 48 bytes");
struct alignas(16) Unit {
    uint32_t w[4];
};
constexpr uint32_t kWideLeafTris = 3;  // most triangles a leaf of the wide tree holds (two bits of a reference)
constexpr uint32_t kUnitsPerItem = 3;  // 16-byte units of a node or a triangle

// 48-byte triangle in leaf order: TriangleOpt {v0v1, v0v2, v0}
// (src/triangle_opt.rs:10-14) read as three float4, with the mesh triangle id
// (object_indices, src/bvh.rs:32) in the spare lanes.
struct alignas(16) DevTri {
    float v0[3];
    float e1[3]; // v0v1
    float e2[3]; // v0v2
    uint32_t id;
    uint32_t pad[2];
};
static_assert(sizeof(DevTri) == 48, "DevTri must be 48 bytes");

struct BigLeaf {
    uint32_t first;
    uint32_t count;
};

// Internal queue ray (32 bytes): the reciprocal direction of StreamRay
// (src/ray.rs:17-24) is recomputed in the kernel by the same IEEE division.
struct alignas(16) QRay {
    float o[3];
    float near_t;
    float d[3];
    float far_t;
};
static_assert(sizeof(QRay) == 32, "QRay must be 32 bytes");

// PathInfo of src/integrators/common.rs:14-19: the path's weight and the slot that owns its
// radiance.  The reference also carries the pixel; here pixel and sample index (the keys of the
// counter-based random numbers) follow from the slot and the pass, and the bounce count is the
// depth of the launch, so the queue entry is 16 bytes.
struct alignas(16) PathInfo {
    float w[3];
    uint32_t slot; // index into the pass's radiance arrays: sample * pixel slots + pixel slot
};
static_assert(sizeof(PathInfo) == 16, "PathInfo must be 16 bytes");

struct alignas(16) ShadowInfo {
    float w[3]; // path weight * light weight
    uint32_t slot;
};

struct SceneDev {
    const DevNode *nodes;
    const DevTri *tris;
    const BigLeaf *big_leaves;
    const float *normals; // float4 per mesh triangle (xyz, 0)
    uint32_t root_ref;
    uint32_t max_depth;
    // 4-wide quantized tree over the same triangles (the fast traversal forms), null when absent:
    // DevQNode and DevTri items in one array of 16-byte units, see DevQNode.  `nodes`/`tris`
    // above stay the reference's tree for the exact forms.
    const Unit *witems;
    uint32_t wroot_ref;   // unit << 2 | triangles (a leaf when the mesh has at most kWideLeafTris triangles)
    uint32_t wmax_depth;  // inner levels of the wide tree
};

// Counters of src/trace_stats.rs:4-12 as kept on the device.
struct StatsDev {
    unsigned long long total_rays;
    unsigned long long inner_rays;
    unsigned long long leaf_rays;
    unsigned long long obj_intersect;
    unsigned long long max_ray_steps; // most node visits + triangle tests any single ray needed
    unsigned long long max_stack_entries; // most deferred nodes any single ray held at once
    // the same sums by path depth of the launch (render calls; depths >= kStatDepths-1 share the last slot)
    unsigned long long depth_rays[8];
    unsigned long long depth_inner[8];
    unsigned long long depth_tris[8];
};
constexpr unsigned kStatDepths = 8;

} // namespace rtc
