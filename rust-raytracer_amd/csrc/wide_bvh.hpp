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

// Quantizes the child boxes of `wide` onto the 8-bit grid of DevQWideNode.
// Per node and axis: origin is placed `margin` below the lowest child plane and
// scale makes byte 255 reach `margin` above the highest, with
//     margin = 2^-18 * max(|lowest|, |highest|, extent)
// (several hundred times the rounding error of the traversal's
// fma(byte, scale/d, (origin - o)/d) against the exact plane); lower planes are
// rounded down and upper planes up so that the decoded box, evaluated in double
// precision, contains the exact one enlarged by half the margin.
void quantize_wide4(const std::vector<DevWideNode> &wide, std::vector<DevQWideNode> &out,
                    std::vector<uint32_t> &bin);

} // namespace rtc
