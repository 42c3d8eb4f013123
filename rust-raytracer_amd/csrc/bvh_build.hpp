// Host-side BVH construction for the MI355X ray-tracing core.
//
// Produces the same tree as the reference's Bvh::build
// (src/bvh/build.rs:45-207): top-down binned SAH, 8 buckets on the largest
// centroid-extent axis, leaves of 1..8 triangles, Hoare partition.  The tree
// is emitted directly in flattened pre-order form.
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/rtcore.h"

namespace rtc {

struct HostBvh {
    std::vector<rt_flat_node> nodes; // pre-order, left child = parent + 1 when inner
    std::vector<uint32_t> order;     // leaf slot -> mesh triangle id
    uint32_t root_ref = 0;
    uint32_t root_count = 0;
    rt_bvh_stats stats{};
};

// The search builder (no counterpart in the reference): the same binned SAH
// with more buckets, every axis with a centroid extent tried, and smaller
// leaves.  Used for the tree the fast traversal forms walk; the reference's
// tree stays what the exact forms walk.
struct BuildTuning {
    uint32_t buckets;  // 2..64
    uint32_t max_leaf; // 1..8 triangles
};

// positions: 3*nv floats; indices: 3*nt.  `threads` > 1 builds disjoint
// subtrees concurrently (the result does not depend on it).  tuning == nullptr:
// the reference's builder.
void build_bvh(const float *positions, uint32_t nv, const uint32_t *indices, uint32_t nt,
               unsigned threads, HostBvh &out, const BuildTuning *tuning = nullptr);

} // namespace rtc
