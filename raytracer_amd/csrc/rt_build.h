// BVH construction on the device (rt_build.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

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

// The four-wide tree (WideNode records, bvh_builder.h) collapsed on the device from n_two_wide two-wide nodes in
// device memory (root = node 0), by the algorithm of collapse_wide on the host, breadth-first.  *d_wide_out is
// allocated with hipMalloc and owned by the caller.  levels: levels of wide nodes.
hipError_t device_collapse_wide(const void* d_two_wide_nodes, uint32_t n_two_wide, void** d_wide_out, uint32_t* n_wide_out, uint32_t* levels_out,
                                hipStream_t stream);

}  // namespace rta
