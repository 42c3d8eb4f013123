// Device-side build of the search tree (the tree the fast traversal forms walk): the builder of
// bvh_build.cpp with a BuildTuning, run breadth first on the current HIP device.  No counterpart in
// the reference, whose builder is recursive on the CPU (src/bvh/build.rs:45-207, rayon::join).
#pragma once

#include <functional>
#include <string>

#include "bvh_build.hpp"
#include "device_layout.h"

namespace rtc {

// Builds the tree over the mesh on the current device and brings it back as a HostBvh: the same
// nodes, boxes, leaf ranges and BvhStats as build_bvh(..., &tune); nodes are numbered breadth first
// instead of in pre-order, and the triangles inside a leaf may come in another order.  False (with
// the reason in `why`, and `out` unspecified) when the GPU cannot do it -- no memory for the build, a
// mesh the binning cannot split, a HIP error: the caller then runs the host builder.
//
// With a DeviceWideRequest the build goes on to what the fast traversal forms read -- 4-wide collapse,
// quantization and packing on the device (wide_node_ops.hpp: the host's arithmetic, so the items are those
// of collapse_wide4 + layout_wide4 + pack_wide4 over the same tree, bit for bit); only the placement of the
// items runs on the host.  Then `done` is set, the items are in the buffer `alloc` returned, and `out` holds the
// tree's root and BvhStats but no nodes and no order.  When the tree has a leaf of more than `leaf_tris`
// triangles (it needs refine_leaves() on the host) or the device cannot do it (`why` of the request), the
// function returns the tree as without a request and the caller runs the host pipeline.
struct DeviceWideRequest {
    unsigned align_mode = 0;                  // layout_wide4()
    uint32_t leaf_tris = 3;                   // most triangles a leaf may hold for the items to be made here
    std::function<Unit *(size_t)> alloc;      // device buffer of that many units, owned by the caller
    bool done = false;
    uint32_t num_units = 0, root_ref = 0, wide_depth = 0, num_wide_nodes = 0;
    std::string why;
};
bool build_bvh_search_device(const float *positions, uint32_t nv, const uint32_t *indices, uint32_t nt,
                             BuildTuning tune, HostBvh &out, std::string &why, DeviceWideRequest *wide_request = nullptr);

} // namespace rtc
