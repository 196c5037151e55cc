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

// 64-byte device node: one per inner node of the host tree, holding the f32
// boxes of both children (rounded outwards from the f64 boxes) as four
// 16-byte words:
//   x = (lo0.x, hi0.x, lo1.x, hi1.x), y and z likewise,
//   ref0, ref1 : child references, axis : split axis (0..2)
// A reference is the index of an inner node (bit 31 clear) or a leaf:
// bit 31 set, bits 30..24 sphere count (1..127), bits 23..0 first sphere slot.
struct DevNode {
  float x[4], y[4], z[4];
  uint32_t ref0, ref1, axis, pad;
};
static_assert(sizeof(DevNode) == 64, "DevNode must be 64 bytes");
constexpr uint32_t kRefLeaf = 0x80000000u;
constexpr uint32_t kRefNone = 0xffffffffu;
constexpr uint32_t kMaxLeafCount = 127;
constexpr uint32_t kMaxSlots = 0x00fffff0u;  // slots stay below this so that no leaf ref collides with the kernels' reserved values (0xfffffffd and up)

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
struct FlatBvh {
  std::vector<DevNode> nodes;  // inner nodes, breadth-first: any prefix is the top of the tree
  std::vector<DevSphere> spheres;
  std::vector<DevSphereAux> aux;
  uint32_t root = kRefNone;  // reference of the root (a leaf reference when the tree is a single leaf)
  uint32_t depth = 0;        // pushes a traversal can have outstanding, at most
  float bound = 0.0f;        // largest |coordinate| of any box plane
};

FlatBvh flatten_bvh(const HostBvh& bvh, const rt_sphere* spheres, size_t n);

// 128-byte node of the four-wide tree that scenes too large for LDS are walked through: the f32 boxes
// (rounded outwards) of up to four children, so that one fetch decides four children and a ray makes half
// as many dependent fetches as through the two-wide nodes.  Per axis the four low planes and the four high
// planes sit in separate 16-byte words (x low, x high, y low, y high, z low, z high): a lane reads the word it
// meets first and the word it meets last by address, with no per-plane min/max.  ref: as in DevNode; a child
// that does not exist has ref kRefNone and an inverted, infinite box.
struct WideNode {
  float planes[6][4];
  uint32_t ref[4];
  uint32_t pad[4];
};
static_assert(sizeof(WideNode) == 128, "WideNode must be 128 bytes");
struct WideBvh {
  std::vector<WideNode> nodes;  // breadth-first: any prefix is the top of the tree; node 0 is the root
  uint32_t depth = 0;           // levels of wide nodes
  uint32_t stack_entries = 0;   // entries a traversal can have outstanding, at most
  float bound = 0.0f;           // largest |coordinate| of any reconstructed plane
};
// Collapses the two-wide device tree below inner node `root` into the four-wide form: a node takes its two
// children and then, twice, replaces the inner child of largest surface area by that child's own children.
WideBvh collapse_wide(const DevNode* nodes, size_t n_nodes, uint32_t root = 0);

}  // namespace rta
