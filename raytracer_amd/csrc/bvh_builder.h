// Host BVH construction and flattening for the gfx950 kernels.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "rt.h"

namespace rta {

struct Box {
  double min[3], max[3];
};

// Node of the host tree in the reference's order (src/bvh.rs:17-27): pre-order,
// left child at i+1, right child at i+skip.
struct HostNode {
  Box box;
  int32_t kind;    // 0..2: split on that axis, 3: leaf
  uint64_t skip;   // split: distance to the right child
  uint64_t first;  // leaf: offset into leaf_indices
  uint64_t count;  // leaf: number of primitives
};

struct HostBvh {
  std::vector<HostNode> nodes;
  std::vector<uint64_t> leaf_indices;  // sphere indices, leaves concatenated in node order
  uint32_t depth = 0;
};

// Binned SAH over sphere bounding boxes with the reference's constants and
// tie-breaks (src/bvh.rs:87-90,105-138,140-156,196-228,250-266), object splits
// only.  O(n) work per node instead of the reference's O(31 n) set rebuilds.
HostBvh build_bvh(const rt_sphere* spheres, size_t n);

// 32-byte device node: two 16-byte words.
//   lo.xyz / hi.xyz : f32 box, rounded outwards from the f64 box
//   a : split -> index of the left child (the right child is a+1)
//       leaf  -> first sphere slot
//   b : split -> axis (0..2);  leaf -> 0x80000000 | sphere count
struct DevNode {
  float lo[3];
  uint32_t a;
  float hi[3];
  uint32_t b;
};
static_assert(sizeof(DevNode) == 32, "DevNode must be 32 bytes");

// Sphere record in leaf order ("slot" order): 32 bytes, f64.
struct DevSphere {
  double cx, cy, cz, r;
};
// Shading data of a slot, read once per hit.
struct DevSphereAux {
  double inv_r;
  uint32_t material;  // index into the material table
  uint32_t prim;      // index of the sphere in the caller's array
};
struct DevMaterial {
  double albedo[3];
  double param;
  uint32_t type;
  uint32_t pad[3];
};
static_assert(sizeof(DevMaterial) == 48, "DevMaterial must be 48 bytes");

struct FlatBvh {
  std::vector<DevNode> nodes;  // breadth-first: any prefix is the top of the tree, siblings adjacent
  std::vector<DevSphere> spheres;
  std::vector<DevSphereAux> aux;
};

FlatBvh flatten_bvh(const HostBvh& bvh, const rt_sphere* spheres, size_t n);

}  // namespace rta
