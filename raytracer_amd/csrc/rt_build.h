// BVH construction on the device (rt_build.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "rt.h"

namespace rta {

// Device arrays of a finished tree, in the layout of SceneDev (rt_launch.h).  The arrays are
// allocated with hipMalloc by the builder and owned by the caller, also when the build fails
// half-way (free whatever is not null).
struct DeviceTree {
  void* d_nodes = nullptr;    // n_nodes x 64 B, breadth-first
  void* d_spheres = nullptr;  // n x 32 B in leaf order
  void* d_aux = nullptr;      // n x 16 B in leaf order
  uint32_t n_nodes = 0;
  uint32_t depth = 0;  // levels of the tree, leaves included
  uint32_t root = 0;
  float bound = 0.0f;
};

// How the hierarchy is made: by clustering the spheres along their Morton order from the bottom up, or by the
// reference's own binned surface-area splits from the top down (the host builder's tree, rt_build_sah.hip).
enum class DeviceBuilder { kClustering, kBinnedSah };

// The tree over n >= 2 spheres (host array), one sphere per leaf.  material_type[m] is the kind
// (0..2) of material m; it is packed next to the material index in the shading data.
// Returns when the tree is complete.
hipError_t device_build(const rt_sphere* spheres, size_t n, const uint8_t* material_type, size_t n_materials, DeviceBuilder builder,
                        DeviceTree* out, hipStream_t stream);

// The hierarchy the device's surface-area build makes, brought back to the host so that it can be compared with the
// reference's tree: n - 1 nodes and n leaves.  left/right[id]: a child -- a node's id, or 0x80000000 | k for leaf k, which
// holds sphere order[k]; boxes: 6 doubles per node (low xyz, high xyz); axis[id]: 0..2, or 3 where the reference's builder
// stops and makes one leaf of the node's spheres (src/bvh.rs:124) while the device build goes on cutting in the middle.
struct HostHierarchy {
  std::vector<uint32_t> left, right, order;
  std::vector<uint8_t> axis;
  std::vector<double> boxes;
  uint32_t root = 0;
};
hipError_t device_build_hierarchy(const rt_sphere* spheres, size_t n, HostHierarchy* out, hipStream_t stream);

// The four-wide tree (WideNode records, bvh_builder.h) collapsed on the device from n_two_wide two-wide nodes in
// device memory (root = node 0), by the algorithm of collapse_wide on the host, breadth-first.  *d_wide_out is
// allocated with hipMalloc and owned by the caller.  levels: levels of wide nodes.
hipError_t device_collapse_wide(const void* d_two_wide_nodes, uint32_t n_two_wide, void** d_wide_out, uint32_t* n_wide_out, uint32_t* levels_out,
                                hipStream_t stream);

}  // namespace rta
