// Shared by the two hierarchy builders of the device tree build (rt_build.hip, rt_build_sah.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rta {

struct Box6 {
  double lo[3], hi[3];
};

// The hierarchy both builders hand to the common back end (boxes fitted bottom-up, breadth-first numbering, nodes
// written in the traversal layout): n - 1 inner nodes with ids below n - 1 and n leaves, leaf k being the sphere
// order[k].  left/right[id] name a child: an inner node's id, or 0x80000000 | k for leaf k.  parent_inner[id] and
// parent_leaf[k] name the parent (0xffffffff for the root).

// BVH::new (src/bvh.rs:92-266), the binned surface-area build, for spheres that all end up in leaves of their own:
// the same boxes, buckets, costs and stable partitions as the host builder (bvh_builder.cpp), so the same tree.  Where
// the reference would stop and make a leaf of several spheres (no split cheaper than the leaf: spheres on top of each
// other), the segment is cut in the middle instead.  prim_box and centroid: Sphere::aabb and its centroid per sphere.
// axis[id]: the axis the node was split along (child 0 holds the lower buckets); node_box[id]: the node's box (the union
// of its spheres' boxes: what fitting the boxes bottom-up would give).  *root_id: the root's id.
hipError_t sah_hierarchy(const Box6* prim_box, const double* centroid, uint32_t n, uint32_t* left, uint32_t* right, uint32_t* parent_inner,
                         uint32_t* parent_leaf, uint8_t* axis, Box6* node_box, uint32_t* order, uint32_t* root_id, hipStream_t stream);

}  // namespace rta
