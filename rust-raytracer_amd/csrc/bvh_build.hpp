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

// positions: 3*nv floats; indices: 3*nt.  `threads` > 1 builds disjoint
// subtrees concurrently (the result does not depend on it).
void build_bvh(const float *positions, uint32_t nv, const uint32_t *indices, uint32_t nt,
               unsigned threads, HostBvh &out);

} // namespace rtc
