// Device-side build of the search tree (the tree the fast traversal forms walk): the builder of
// bvh_build.cpp with a BuildTuning, run breadth first on the current HIP device.  No counterpart in
// the reference, whose builder is recursive on the CPU (src/bvh/build.rs:45-207, rayon::join).
#pragma once

#include <string>

#include "bvh_build.hpp"

namespace rtc {

// Builds the tree over the mesh on the current device and brings it back as a HostBvh: the same
// nodes, boxes, leaf ranges and BvhStats as build_bvh(..., &tune); nodes are numbered breadth first
// instead of in pre-order, and the triangles inside a leaf may come in another order.  False (with
// the reason in `why`, and `out` unspecified) when the GPU cannot do it -- no memory for the build, a
// mesh the binning cannot split, a HIP error: the caller then runs the host builder.
bool build_bvh_search_device(const float *positions, uint32_t nv, const uint32_t *indices, uint32_t nt,
                             BuildTuning tune, HostBvh &out, std::string &why);

} // namespace rtc
