// Shared by the two hierarchy builders of the device tree build (rt_build.hip, rt_build_sah.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rta {

struct Box6 {
  double lo[3], hi[3];
};

// Scratch memory of one build: one allocation that the arrays are carved out of (a build makes some forty of them, and forty
// allocations and frees cost more than the kernels of a small build).  An array that does not fit gets an allocation of its own.
struct Arena {
  char* base = nullptr;
  size_t cap = 0, used = 0;
  ~Arena() {
    if (base) (void)hipFree(base);
  }
  void reserve(size_t bytes) {  // (no memory for it: the arrays are allocated one by one)
    if (hipMalloc(reinterpret_cast<void**>(&base), bytes) == hipSuccess)
      cap = bytes;
    else
      base = nullptr, (void)hipGetLastError();
  }
  void* take(size_t bytes) {
    const size_t at = (used + 255) & ~size_t(255);
    if (!base || at + bytes > cap) return nullptr;
    used = at + bytes;
    return base + at;
  }
};
struct Scratch {
  void* p = nullptr;
  bool own = false;
  ~Scratch() {
    if (p && own) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes, Arena* arena) {
    if (!bytes) bytes = 16;
    if (arena && (p = arena->take(bytes))) return hipSuccess;
    own = true;
    return hipMalloc(&p, bytes);
  }
  template <class T>
  T* as() const {
    return static_cast<T*>(p);
  }
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
                         uint32_t* parent_leaf, uint8_t* axis, Box6* node_box, uint32_t* order, uint32_t* root_id, Arena* arena, hipStream_t stream);
// Scratch bytes per sphere that device builds reserve in their arena (the larger of the two builders' needs, with room to spare).
constexpr size_t kBuildScratchPerSphere = 352;

}  // namespace rta
