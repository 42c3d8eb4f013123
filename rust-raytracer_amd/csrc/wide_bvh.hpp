// 4-wide collapse of the flattened binary tree (host side).
//
// The binary tree stays the canonical form (it is what the reference builds,
// src/bvh/build.rs:45-207); the wide form is derived from it by pulling the
// children of inner children up into their grandparent until a node has four
// children, always opening the child box with the largest surface area.  No box
// is recomputed: every child box of a wide node is a child box of some binary
// node, bit for bit.
#pragma once

#include <cstdint>
#include <vector>

#include "device_layout.h"

namespace rtc {

// bn: device-form binary nodes (packed leaf references); root_ref: reference of
// the root (may be a leaf).  Returns the root reference of the wide tree (the
// leaf itself when the root is a leaf, else 0) and appends the nodes to `out`
// in depth-first pre-order.  max_wide_depth (optional) receives the depth of the
// wide tree in inner nodes.
uint32_t collapse_wide4(const DevNode *bn, size_t num_nodes, uint32_t root_ref, std::vector<DevWideNode> &out,
                        uint32_t *max_wide_depth);

// Finer leaves for the wide tree.  The reference stops splitting at 8 triangles
// (src/bvh/build.rs:80-81), a leaf size chosen for 8-wide CPU SIMD; a GPU lane
// tests triangles one at a time and every test costs three 16-byte gathers.
// Every leaf of more than `max_tris` triangles is replaced by a small
// binary subtree over its own triangles: triangles are ordered by centroid
// along the leaf's largest centroid-extent axis and cut where
// area(A)*|A| + area(B)*|B| is smallest, recursively; child boxes are the exact
// unions of the triangles' vertex bounds.  Leaves whose centroids all coincide
// (the zero-centroid-extent rule, src/bvh/build.rs:93-101, which also makes the
// leaves of more than 8 triangles listed in `big`) are cut in halves by slot.
// The result is a copy of the binary tree (`xn`: the original nodes at their
// original indices, new nodes appended) without a leaf above max_tris, and of the
// leaf order (`xorder`: a permutation of `order` that only moves triangles
// inside their reference leaf).  Returns the number of levels the deepest new
// subtree adds.
uint32_t refine_leaves(const std::vector<DevNode> &bn, uint32_t root_ref, const float *positions,
                       const uint32_t *indices, const std::vector<uint32_t> &order, unsigned max_tris,
                       const std::vector<BigLeaf> &big, std::vector<DevNode> &xn, uint32_t &xroot,
                       std::vector<uint32_t> &xorder);

// Where the items of the wide tree go in the unified array of 16-byte units (DevQNode in
// device_layout.h): the root node at unit 0, then, depth first, the children block of every node.
struct WideLayout {
    std::vector<uint32_t> node_unit; // per wide node
    std::vector<uint32_t> slot_unit; // per slot of the leaf order the wide tree was built over
    uint32_t num_units = 0;
    uint32_t root_ref = 0;           // unit << 2 | triangles
};
// wroot: what collapse_wide4 returned.  False when a leaf holds more than kWideLeafTris triangles.
bool layout_wide4(const std::vector<DevWideNode> &wide, uint32_t wroot, uint32_t num_triangles, WideLayout &out,
                  unsigned align_mode = 0);
// The same from the child references alone: four per wide node, unused slots hold kNoWideChild.
constexpr uint32_t kNoWideChild = 0x7FFFFFFFu; // neither a leaf reference of a tree that fits nor a node index
bool layout_wide4(const uint32_t *children, size_t num_wide, uint32_t wroot, uint32_t num_triangles, WideLayout &out,
                  unsigned align_mode = 0);

// One node on the 8-bit grid.  Per axis: origin is placed `margin` below the lowest child plane
// and the grid step is the smallest value of the form (1 + m / 4) * 2^e, m in 0..3 (what the ten-bit step
// code of DevQNode holds), with which byte 255 reaches `margin` above the highest, where
//     margin = 2^-18 * max(|lowest|, |highest|, extent)
// (several hundred times the rounding error of the traversal's
// fma(byte, step/d, (origin - o)/d) against the exact plane); lower planes are
// rounded down and upper planes up so that the decoded box, evaluated in double
// precision, contains the exact one enlarged by half the margin.
DevQNode quantize_node(const DevWideNode &src, const WideLayout &layout);

// The unified array: every node quantized, every triangle of `xtris` (leaf order of the wide
// tree) copied to its units; one spare unit at the end (a wave-uniform item is read as 64 bytes).
// the same into a caller's buffer of layout.num_units + 1 units (gaps_cleared: it is zero-filled already)
void pack_wide4(const std::vector<DevWideNode> &wide, const WideLayout &layout, const DevTri *xtris,
                uint32_t num_triangles, Unit *items, bool gaps_cleared);
void pack_wide4(const std::vector<DevWideNode> &wide, const WideLayout &layout, const DevTri *xtris,
                uint32_t num_triangles, std::vector<Unit> &items);

} // namespace rtc
