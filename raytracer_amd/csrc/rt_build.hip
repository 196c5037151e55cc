// BVH construction on the device.  A hierarchy over the spheres comes from one of two builders -- the reference's own
// binned surface-area splits (BVH::new, src/bvh.rs:92-266; rt_build_sah.hip), which make the reference's tree, or, here,
// the spheres in Morton order clustered bottom-up (parallel locally-ordered clustering), which make a different tree over
// the same spheres: the closest hits do not depend on the tree -- and the back end in this file fits the boxes bottom-up
// where the builder has not supplied them, numbers the nodes breadth-first and writes them straight into the layout the
// traversal kernels read.  It stands in for BVH::new (src/bvh.rs:44-48) where building on the host would take longer.
#include <cstring>
#include <utility>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "rt_build.h"
#include "rt_build_internal.h"

namespace rta {
namespace {

constexpr int kThreads = 256;
constexpr uint32_t kLeafBit = 0x80000000u;

__device__ __forceinline__ float round_down(double v) {
  float f = (float)v;
  return ((double)f > v) ? nextafterf(f, -__builtin_inff()) : f;
}
__device__ __forceinline__ float round_up(double v) {
  float f = (float)v;
  return ((double)f < v) ? nextafterf(f, __builtin_inff()) : f;
}

// Sphere::aabb (src/sphere.rs:50-52) and the box's centroid (src/aabb.rs:23-25) per sphere;
// per block, the bounds of the boxes (all spheres: `flags` null) or of the centroids of the spheres that are not flagged.
__global__ __launch_bounds__(kThreads) void boxes_kernel(const rt_sphere* __restrict__ spheres, uint32_t n, Box6* __restrict__ prim_box,
                                                          double* __restrict__ centroid, const uint8_t* __restrict__ flags,
                                                          double* __restrict__ block_bounds) {
  __shared__ double red[6][kThreads];
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  double c[3], lo[3], hi[3];
  for (int a = 0; a < 3; ++a) {
    lo[a] = __builtin_inf();
    hi[a] = -__builtin_inf();
  }
  if (i < n) {
    const rt_sphere s = spheres[i];
    Box6 b;
    for (int a = 0; a < 3; ++a) {
      const double p0 = s.center[a] - s.radius, p1 = s.center[a] + s.radius;
      b.lo[a] = fmin(p0, p1);
      b.hi[a] = fmax(p0, p1);
      c[a] = b.lo[a] + 0.5 * (b.hi[a] - b.lo[a]);
      if (!flags) {
        centroid[3 * (size_t)i + a] = c[a];
        lo[a] = b.lo[a];
        hi[a] = b.hi[a];
      } else if (!flags[i]) {
        lo[a] = hi[a] = c[a];
      }
    }
    if (!flags) prim_box[i] = b;
  }
  for (int a = 0; a < 3; ++a) {
    red[a][threadIdx.x] = lo[a];
    red[3 + a][threadIdx.x] = hi[a];
  }
  __syncthreads();
  for (int off = kThreads / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
      for (int a = 0; a < 3; ++a) {
        red[a][threadIdx.x] = fmin(red[a][threadIdx.x], red[a][threadIdx.x + off]);
        red[3 + a][threadIdx.x] = fmax(red[3 + a][threadIdx.x], red[3 + a][threadIdx.x + off]);
      }
    __syncthreads();
  }
  if (threadIdx.x < 6) block_bounds[6 * (size_t)blockIdx.x + threadIdx.x] = red[threadIdx.x][0];
}

__global__ __launch_bounds__(kThreads) void bounds_kernel(const double* __restrict__ block_bounds, uint32_t n_blocks, double* __restrict__ bounds) {
  __shared__ double red[6][kThreads];
  double lo[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()}, hi[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
  for (uint32_t b = threadIdx.x; b < n_blocks; b += kThreads)
    for (int a = 0; a < 3; ++a) {
      lo[a] = fmin(lo[a], block_bounds[6 * (size_t)b + a]);
      hi[a] = fmax(hi[a], block_bounds[6 * (size_t)b + 3 + a]);
    }
  for (int a = 0; a < 3; ++a) {
    red[a][threadIdx.x] = lo[a];
    red[3 + a][threadIdx.x] = hi[a];
  }
  __syncthreads();
  for (int off = kThreads / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
      for (int a = 0; a < 3; ++a) {
        red[a][threadIdx.x] = fmin(red[a][threadIdx.x], red[a][threadIdx.x + off]);
        red[3 + a][threadIdx.x] = fmax(red[3 + a][threadIdx.x], red[3 + a][threadIdx.x + off]);
      }
    __syncthreads();
  }
  if (threadIdx.x < 6) bounds[threadIdx.x] = red[threadIdx.x][0];
}

__device__ __forceinline__ uint64_t spread21(uint64_t v) {  // bit k of v moves to bit 3k
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

// Spheres as large as the scene -- a radius above a quarter of the largest extent of all boxes: a ground sphere -- stand
// apart: they take no part in the bounds the Morton codes are scaled by (so the codes keep their resolution for everything
// else) and sort behind all others, which puts them in a subtree of their own directly under the root instead of somewhere
// inside the Morton order, where such a sphere's box would blow up every ancestor's.
__global__ __launch_bounds__(kThreads) void flag_kernel(const rt_sphere* __restrict__ spheres, uint32_t n, const double* __restrict__ box_bounds,
                                                         uint8_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const double extent = fmax(box_bounds[3] - box_bounds[0], fmax(box_bounds[4] - box_bounds[1], box_bounds[5] - box_bounds[2]));
  flags[i] = fabs(spheres[i].radius) > 0.25 * extent ? 1 : 0;
}

// 63-bit Morton code of each centroid inside the bounds of the centroids (those of the spheres that do not stand apart);
// bit 63 for the spheres that do.
__global__ __launch_bounds__(kThreads) void morton_kernel(const double* __restrict__ centroid, const double* __restrict__ bounds,
                                                           const uint8_t* __restrict__ flags, uint32_t n, uint64_t* __restrict__ keys,
                                                           uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  // One scale for the three axes (the largest extent): cells of the Morton grid are cubes, so a scene that is flat along one axis
  // is not cut along that axis until the cells are as small as it is thick.  (Scaled per axis, the codes of a field of spheres on a
  // large ground sphere were cut into height bands -- rings -- at every level: 20 % more node steps per ray.)
  const double ext = fmax(bounds[3] - bounds[0], fmax(bounds[4] - bounds[1], bounds[5] - bounds[2]));
  uint64_t q[3];
  for (int a = 0; a < 3; ++a) {
    double f = ext > 0.0 ? (centroid[3 * (size_t)i + a] - bounds[a]) / ext : 0.0;  // (no sphere inside the bounds: they are inverted, every code is 0)
    f = fmin(fmax(f * 2097152.0, 0.0), 2097151.0);
    q[a] = (uint64_t)f;
  }
  keys[i] = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2) | (flags[i] ? 1ull << 63 : 0ull);
  vals[i] = i;
}

// ------------------------------------------------------------- clustering ----
// The hierarchy: parallel locally-ordered clustering (Meister and Bittner 2018) over the Morton order.  The clusters --
// at first the spheres in Morton order -- sit in an array; in every round each cluster looks kRadius places to either side
// for the neighbour whose union with it has the least surface area, pairs that chose each other merge into a new node that
// takes the place of the first of the two, and the array is packed.  The order of the array is kept, so the result is the
// same in every run.  Rounds go on until one cluster is left: the root.  Against splitting at the highest differing bit of
// the codes (Karras 2012) this builds what the surface-area heuristic wants from the bottom up: renders come within 4 - 8 % of
// those through the host's binned-SAH tree (13 - 18 % before).  Only the topology is decided here, on f32 copies of the boxes;
// the exact boxes are fitted afterwards (fit_kernel).  (Measured on fields of 10^4 .. 10^6 spheres: a radius of 3 or 4 gives the
// best trees, 6 and 12 are 1 - 5 % slower to render, 25 and 50 up to 15 %: with a long reach small clusters pair up with
// clusters of their own size far along the curve instead of growing compact.)
#ifndef RTA_PLOC_RADIUS
#define RTA_PLOC_RADIUS 4
#endif
constexpr int kRadius = RTA_PLOC_RADIUS;
struct ClusterBox {
  float lo[3], hi[3];
};
__device__ __forceinline__ float half_area_of_union(const ClusterBox& a, const ClusterBox& b) {
  const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]), dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]),
              dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
  return dx * dy + dy * dz + dz * dx;
}
__global__ __launch_bounds__(kThreads) void ploc_init_kernel(const Box6* __restrict__ prim_box, const uint32_t* __restrict__ vals, uint32_t n,
                                                              uint32_t* __restrict__ ref, ClusterBox* __restrict__ box) {
  const uint32_t k = blockIdx.x * kThreads + threadIdx.x;
  if (k >= n) return;
  const Box6 b = prim_box[vals[k]];
  ClusterBox c;
  for (int a = 0; a < 3; ++a) c.lo[a] = (float)b.lo[a], c.hi[a] = (float)b.hi[a];
  ref[k] = kLeafBit | k;
  box[k] = c;
}
// nearest[i]: the neighbour within kRadius places whose union with cluster i has the least area (of equals the closer, then the earlier one)
__global__ __launch_bounds__(kThreads) void ploc_nearest_kernel(const ClusterBox* __restrict__ box, int m, uint32_t* __restrict__ nearest) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= m) return;
  const ClusterBox mine = box[i];
  float best = __builtin_inff();
  int best_j = i > 0 ? i - 1 : 1;  // (what stays if no area compares below infinity)
  for (int d = 1; d <= kRadius; ++d) {
    for (int s = -1; s <= 1; s += 2) {
      const int j = i + s * d;
      if (j < 0 || j >= m) continue;
      const float a = half_area_of_union(mine, box[j]);
      if (a < best) {
        best = a;
        best_j = j;
      }
    }
  }
  nearest[i] = (uint32_t)best_j;  // (m >= 2: there is a neighbour)
}
// marks[i]: low word 1 if cluster i stays in the array (it is not the second of a merging pair), high word 1 if it is the first of one
__global__ __launch_bounds__(kThreads) void ploc_mark_kernel(const uint32_t* __restrict__ nearest, int m, unsigned long long* __restrict__ marks) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= m) return;
  const int j = (int)nearest[i];
  const bool mutual = (int)nearest[j] == i;
  marks[i] = (mutual && j < i ? 0ull : 1ull) | (mutual && i < j ? 1ull << 32 : 0ull);
}
// `scan` = exclusive prefix sums of marks.  Writes the packed array and the new nodes (ids from next_id on, in array order).
__global__ __launch_bounds__(kThreads) void ploc_merge_kernel(const uint32_t* __restrict__ ref, const ClusterBox* __restrict__ box,
                                                               const uint32_t* __restrict__ nearest, const unsigned long long* __restrict__ marks,
                                                               const unsigned long long* __restrict__ scan, int m, uint32_t next_id,
                                                               uint32_t* __restrict__ ref_out, ClusterBox* __restrict__ box_out,
                                                               uint32_t* __restrict__ left, uint32_t* __restrict__ right,
                                                               uint32_t* __restrict__ parent_inner, uint32_t* __restrict__ parent_leaf,
                                                               unsigned long long* __restrict__ totals) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= m) return;
  const unsigned long long mark = marks[i], before = scan[i];
  if (i == m - 1) *totals = before + mark;  // clusters left (low word) and nodes made (high word) after this round
  if (!(mark & 1ull)) return;
  const uint32_t pos = (uint32_t)before;
  if (mark >> 32) {
    const int j = (int)nearest[i];
    const uint32_t id = next_id + (uint32_t)(before >> 32);
    const uint32_t a = ref[i], b = ref[j];
    left[id] = a;
    right[id] = b;
    if (a & kLeafBit) parent_leaf[a & ~kLeafBit] = id; else parent_inner[a] = id;
    if (b & kLeafBit) parent_leaf[b & ~kLeafBit] = id; else parent_inner[b] = id;
    parent_inner[id] = 0xffffffffu;  // until a later round gives it a parent: the last node made is the root
    const ClusterBox x = box[i], y = box[j];
    ClusterBox u;
    for (int k = 0; k < 3; ++k) u.lo[k] = fminf(x.lo[k], y.lo[k]), u.hi[k] = fmaxf(x.hi[k], y.hi[k]);
    ref_out[pos] = id;
    box_out[pos] = u;
  } else {
    ref_out[pos] = ref[i];
    box_out[pos] = box[i];
  }
}

__device__ __forceinline__ Box6 child_box(uint32_t ref, const Box6* __restrict__ prim_box, const uint32_t* __restrict__ vals,
                                           const Box6* node_box) {
  if (ref & kLeafBit) return prim_box[vals[ref & ~kLeafBit]];
  // written by another thread of this launch: read past the caches of this CU
  Box6 b;
  const double* p = reinterpret_cast<const double*>(node_box + ref);
  for (int a = 0; a < 3; ++a) {
    b.lo[a] = __builtin_nontemporal_load(p + a);
    b.hi[a] = __builtin_nontemporal_load(p + 3 + a);
  }
  return b;
}

// One thread per leaf climbs towards the root; the second thread to reach a node fits its box.
__global__ __launch_bounds__(kThreads) void fit_kernel(int n, const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                                        const uint32_t* __restrict__ parent_inner, const uint32_t* __restrict__ parent_leaf,
                                                        const Box6* __restrict__ prim_box, const uint32_t* __restrict__ vals, Box6* node_box,
                                                        uint32_t* arrivals) {
  const int k = blockIdx.x * kThreads + threadIdx.x;
  if (k >= n) return;
  uint32_t node = parent_leaf[k];
  while (node != 0xffffffffu) {
    __threadfence();
    if (atomicAdd(&arrivals[node], 1u) == 0u) return;  // the sibling subtree is not finished yet
    __threadfence();
    const Box6 a = child_box(left[node], prim_box, vals, node_box), b = child_box(right[node], prim_box, vals, node_box);
    Box6 u;
    for (int x = 0; x < 3; ++x) {
      u.lo[x] = fmin(a.lo[x], b.lo[x]);
      u.hi[x] = fmax(a.hi[x], b.hi[x]);
    }
    node_box[node] = u;
    node = parent_inner[node];
  }
}

// Level of each inner node (root = 1); the deepest level seen.
__global__ __launch_bounds__(kThreads) void level_kernel(int n_inner, const uint32_t* __restrict__ parent_inner, uint32_t* __restrict__ level,
                                                          uint32_t* __restrict__ ids, uint32_t* max_level) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n_inner) return;
  uint32_t l = 1, p = parent_inner[i];
  while (p != 0xffffffffu) {
    ++l;
    p = parent_inner[p];
  }
  level[i] = l;
  ids[i] = (uint32_t)i;
  atomicMax(max_level, l);
}

__global__ __launch_bounds__(kThreads) void rank_kernel(int n_inner, const uint32_t* __restrict__ order, uint32_t* __restrict__ new_of) {
  const int k = blockIdx.x * kThreads + threadIdx.x;
  if (k < n_inner) new_of[order[k]] = (uint32_t)k;
}

// The 64-byte traversal node of each inner node, at its breadth-first rank.
__global__ __launch_bounds__(kThreads) void emit_nodes_kernel(int n_inner, const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                                               const uint32_t* __restrict__ new_of, const Box6* __restrict__ prim_box,
                                                               const uint32_t* __restrict__ vals, const Box6* __restrict__ node_box,
                                                               float4* __restrict__ nodes, float* __restrict__ bound, uint32_t root,
                                                               const uint8_t* __restrict__ split_axis) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n_inner) return;
  uint32_t r0 = left[i], r1 = right[i];
  Box6 b0 = (r0 & kLeafBit) ? prim_box[vals[r0 & ~kLeafBit]] : node_box[r0];
  Box6 b1 = (r1 & kLeafBit) ? prim_box[vals[r1 & ~kLeafBit]] : node_box[r1];
  // split axis: the builder's, with its order of the children (child 0 holds the lower buckets); where it has none to
  // give, the axis along which the children's centres are furthest apart, child 0 the lower one, as in the host tree
  int axis = 0;
  double best = -1.0, diff[3];
  for (int a = 0; a < 3; ++a) {
    diff[a] = (b1.lo[a] + b1.hi[a]) - (b0.lo[a] + b0.hi[a]);
    if (fabs(diff[a]) > best) {
      best = fabs(diff[a]);
      axis = a;
    }
  }
  const bool given = split_axis && split_axis[i] < 3;
  if (given) axis = split_axis[i];
  if (!given && diff[axis] < 0.0) {
    const Box6 tb = b0;
    b0 = b1;
    b1 = tb;
    const uint32_t tr = r0;
    r0 = r1;
    r1 = tr;
  }
  auto ref = [&](uint32_t r) { return (r & kLeafBit) ? (kLeafBit | (1u << 24) | (r & ~kLeafBit)) : new_of[r]; };
  float4 q0 = make_float4(round_down(b0.lo[0]), round_up(b0.hi[0]), round_down(b1.lo[0]), round_up(b1.hi[0]));
  float4 q1 = make_float4(round_down(b0.lo[1]), round_up(b0.hi[1]), round_down(b1.lo[1]), round_up(b1.hi[1]));
  float4 q2 = make_float4(round_down(b0.lo[2]), round_up(b0.hi[2]), round_down(b1.lo[2]), round_up(b1.hi[2]));
  float4 q3 = make_float4(__uint_as_float(ref(r0)), __uint_as_float(ref(r1)), __uint_as_float((uint32_t)axis), 0.0f);
  float4* dst = nodes + 4 * (size_t)new_of[i];
  dst[0] = q0;
  dst[1] = q1;
  dst[2] = q2;
  dst[3] = q3;
  if ((uint32_t)i == root) {  // the root's box bounds every plane of the tree
    const Box6 rb = node_box[root];
    double m = 0.0;
    for (int a = 0; a < 3; ++a) m = fmax(m, fmax(fabs(rb.lo[a]), fabs(rb.hi[a])));
    *bound = round_up(m);
  }
}

// Sphere records and shading data in Morton order (slot k = sorted position k).
__global__ __launch_bounds__(kThreads) void emit_spheres_kernel(const rt_sphere* __restrict__ spheres, const uint32_t* __restrict__ vals, uint32_t n,
                                                                 const uint8_t* __restrict__ material_type, double* __restrict__ out_spheres,
                                                                 double2* __restrict__ out_aux) {
  const uint32_t k = blockIdx.x * kThreads + threadIdx.x;
  if (k >= n) return;
  const uint32_t i = vals[k];
  const rt_sphere s = spheres[i];
  double* o = out_spheres + 4 * (size_t)k;
  o[0] = s.center[0];
  o[1] = s.center[1];
  o[2] = s.center[2];
  o[3] = s.radius;
  const uint32_t mat = s.material | ((uint32_t)material_type[s.material] << 30);
  const unsigned long long bits = (unsigned long long)mat | ((unsigned long long)i << 32);
  out_aux[k] = make_double2(1.0 / s.radius, __longlong_as_double((long long)bits));
}

#define RTB(call)                   \
  do {                              \
    hipError_t e_ = (call);         \
    if (e_ != hipSuccess) return e_; \
  } while (0)

}  // namespace

// ------------------------------------------------- four-wide collapse ----
// The four-wide tree of bvh_builder.h (WideNode) from two-wide device nodes, level by level: a wide node takes
// the two children of its two-wide root and then, twice, replaces the inner child of largest surface area by that
// child's own children.  Per level: count each wide node's inner children, exclusive scan, emit -- so the
// numbering is breadth-first and the same in every run, children in child order (what the host collapse makes).
struct WideChild {
  float lo[3], hi[3];
  uint32_t ref;
};
__device__ __forceinline__ void two_wide_children(const float4* __restrict__ nodes, uint32_t b, WideChild out[2]) {
  const float4 x = nodes[4 * (size_t)b], y = nodes[4 * (size_t)b + 1], z = nodes[4 * (size_t)b + 2], r = nodes[4 * (size_t)b + 3];
  out[0] = WideChild{{x.x, y.x, z.x}, {x.y, y.y, z.y}, __float_as_uint(r.x)};
  out[1] = WideChild{{x.z, y.z, z.z}, {x.w, y.w, z.w}, __float_as_uint(r.y)};
}
__device__ __forceinline__ double half_area(const WideChild& c) {
  const double dx = (double)c.hi[0] - c.lo[0], dy = (double)c.hi[1] - c.lo[1], dz = (double)c.hi[2] - c.lo[2];
  return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ int gather_wide_children(const float4* __restrict__ nodes, uint32_t root, WideChild ch[4]) {
  int n = 2;
  two_wide_children(nodes, root, ch);
  while (n < 4) {
    int pick = -1;
    double best = -1.0;
    for (int k = 0; k < n; ++k)
      if (!(ch[k].ref & 0x80000000u) && half_area(ch[k]) > best) {
        best = half_area(ch[k]);
        pick = k;
      }
    if (pick < 0) break;
    WideChild two[2];
    two_wide_children(nodes, ch[pick].ref, two);
    ch[pick] = two[0];
    ch[n++] = two[1];
  }
  return n;
}
__global__ __launch_bounds__(kThreads) void wide_count_kernel(const float4* __restrict__ nodes, const uint32_t* __restrict__ roots, uint32_t count,
                                                            uint32_t* __restrict__ n_inner_children) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= count) return;
  WideChild ch[4];
  const int n = gather_wide_children(nodes, roots[i], ch);
  uint32_t inner = 0;
  for (int k = 0; k < n; ++k) inner += (ch[k].ref & 0x80000000u) ? 0u : 1u;
  n_inner_children[i] = inner;
}
__global__ __launch_bounds__(kThreads) void wide_emit_kernel(const float4* __restrict__ nodes, const uint32_t* __restrict__ roots, uint32_t count,
                                                           const uint32_t* __restrict__ child_offset, uint32_t level_base, uint32_t next_base,
                                                           uint32_t* __restrict__ next_roots, float4* __restrict__ wide) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= count) return;
  WideChild ch[4];
  const int n = gather_wide_children(nodes, roots[i], ch);
  const float inf = __builtin_inff();
  float4 q[8];
  for (int a = 0; a < 3; ++a) {
    float lo[4], hi[4];
    for (int k = 0; k < 4; ++k) {
      lo[k] = k < n ? ch[k].lo[a] : inf;   // a child that does not exist: inverted, infinite box
      hi[k] = k < n ? ch[k].hi[a] : -inf;
    }
    q[2 * a] = make_float4(lo[0], lo[1], lo[2], lo[3]);
    q[2 * a + 1] = make_float4(hi[0], hi[1], hi[2], hi[3]);
  }
  uint32_t refs[4], at = child_offset[i];
  for (int k = 0; k < 4; ++k) {
    if (k >= n) {
      refs[k] = 0xffffffffu;
    } else if (ch[k].ref & 0x80000000u) {
      refs[k] = ch[k].ref;
    } else {
      next_roots[at] = ch[k].ref;
      refs[k] = next_base + at;
      ++at;
    }
  }
  q[6] = make_float4(__uint_as_float(refs[0]), __uint_as_float(refs[1]), __uint_as_float(refs[2]), __uint_as_float(refs[3]));
  q[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float4* dst = wide + 8 * (size_t)(level_base + i);
  for (int k = 0; k < 8; ++k) dst[k] = q[k];
}

hipError_t device_collapse_wide(const void* d_two_wide_nodes, uint32_t n_two_wide, void** d_wide_out, uint32_t* n_wide_out, uint32_t* levels_out,
                                hipStream_t stream) {
  *d_wide_out = nullptr;
  const float4* nodes = static_cast<const float4*>(d_two_wide_nodes);
  Arena arena_, *const arena = &arena_;
  arena_.reserve((size_t)n_two_wide * 160 + (4u << 20));
  Scratch roots_a, roots_b, counts, offsets, scan_tmp, all;
  RTB(roots_a.alloc((size_t)n_two_wide * 4, arena));
  RTB(roots_b.alloc((size_t)n_two_wide * 4, arena));
  RTB(counts.alloc((size_t)n_two_wide * 4 + 4, arena));
  RTB(offsets.alloc((size_t)n_two_wide * 4 + 4, arena));
  RTB(all.alloc((size_t)n_two_wide * 128, arena));  // at most one wide node per two-wide node; trimmed below
  size_t tmp_bytes = 0;
  RTB(rocprim::exclusive_scan(nullptr, tmp_bytes, counts.as<uint32_t>(), offsets.as<uint32_t>(), 0u, (size_t)n_two_wide + 1, rocprim::plus<uint32_t>(), stream));
  RTB(scan_tmp.alloc(tmp_bytes, arena));
  const uint32_t zero = 0;
  RTB(hipMemcpyAsync(roots_a.p, &zero, 4, hipMemcpyHostToDevice, stream));  // the root of the two-wide tree is node 0
  uint32_t count = 1, base = 0, levels = 0;
  uint32_t* cur = roots_a.as<uint32_t>();
  uint32_t* next = roots_b.as<uint32_t>();
  while (count) {
    const unsigned blocks = (count + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(wide_count_kernel, dim3(blocks), dim3(kThreads), 0, stream, nodes, cur, count, counts.as<uint32_t>());
    RTB(hipMemsetAsync(counts.as<uint32_t>() + count, 0, 4, stream));  // one entry past the end: its offset is the level's total
    RTB(rocprim::exclusive_scan(scan_tmp.p, tmp_bytes, counts.as<uint32_t>(), offsets.as<uint32_t>(), 0u, (size_t)count + 1, rocprim::plus<uint32_t>(), stream));
    hipLaunchKernelGGL(wide_emit_kernel, dim3(blocks), dim3(kThreads), 0, stream, nodes, cur, count, offsets.as<uint32_t>(), base, base + count, next,
                       all.as<float4>());
    RTB(hipGetLastError());
    uint32_t next_count = 0;
    RTB(hipMemcpyAsync(&next_count, offsets.as<uint32_t>() + count, 4, hipMemcpyDeviceToHost, stream));
    RTB(hipStreamSynchronize(stream));
    base += count;
    count = next_count;
    std::swap(cur, next);
    ++levels;
    if (base + (size_t)count > n_two_wide) return hipErrorUnknown;  // cannot happen for a tree; never write past the buffers
  }
  RTB(hipMalloc(d_wide_out, (size_t)base * 128));
  RTB(hipMemcpyAsync(*d_wide_out, all.p, (size_t)base * 128, hipMemcpyDeviceToDevice, stream));
  RTB(hipStreamSynchronize(stream));
  *n_wide_out = base;
  *levels_out = levels;
  return hipSuccess;
}

hipError_t device_build_hierarchy(const rt_sphere* h_spheres, size_t n_, HostHierarchy* out, hipStream_t stream) {
  const uint32_t n = (uint32_t)n_;
  if (n < 2) return hipErrorInvalidValue;
  const uint32_t n_inner = n - 1;
  const unsigned blocks = (n + kThreads - 1) / kThreads;
  Arena arena_, *const arena = &arena_;
  arena_.reserve((size_t)n * kBuildScratchPerSphere + (4u << 20));
  Scratch d_in, prim_box, centroid, block_bounds, left, right, parent_inner, parent_leaf, node_box, axis, order, root;
  RTB(d_in.alloc((size_t)n * sizeof(rt_sphere), arena));
  RTB(hipMemcpyAsync(d_in.p, h_spheres, (size_t)n * sizeof(rt_sphere), hipMemcpyHostToDevice, stream));
  RTB(prim_box.alloc((size_t)n * sizeof(Box6), arena));
  RTB(centroid.alloc((size_t)n * 24, arena));
  RTB(block_bounds.alloc((size_t)blocks * 48, arena));
  RTB(left.alloc((size_t)n_inner * 4, arena));
  RTB(right.alloc((size_t)n_inner * 4, arena));
  RTB(parent_inner.alloc((size_t)n_inner * 4, arena));
  RTB(parent_leaf.alloc((size_t)n * 4, arena));
  RTB(node_box.alloc((size_t)n_inner * sizeof(Box6), arena));
  RTB(axis.alloc(n_inner, arena));
  RTB(order.alloc((size_t)n * 4, arena));
  RTB(root.alloc(4, arena));
  hipLaunchKernelGGL(boxes_kernel, dim3(blocks), dim3(kThreads), 0, stream, d_in.as<rt_sphere>(), n, prim_box.as<Box6>(), centroid.as<double>(),
                     (const uint8_t*)nullptr, block_bounds.as<double>());
  RTB(hipGetLastError());
  RTB(sah_hierarchy(prim_box.as<Box6>(), centroid.as<double>(), n, left.as<uint32_t>(), right.as<uint32_t>(), parent_inner.as<uint32_t>(),
                    parent_leaf.as<uint32_t>(), axis.as<uint8_t>(), node_box.as<Box6>(), order.as<uint32_t>(), root.as<uint32_t>(), arena, stream));
  out->left.resize(n_inner);
  out->right.resize(n_inner);
  out->axis.resize(n_inner);
  out->boxes.resize((size_t)n_inner * 6);
  out->order.resize(n);
  static_assert(sizeof(Box6) == 48, "six doubles");
  RTB(hipMemcpyAsync(out->left.data(), left.p, (size_t)n_inner * 4, hipMemcpyDeviceToHost, stream));
  RTB(hipMemcpyAsync(out->right.data(), right.p, (size_t)n_inner * 4, hipMemcpyDeviceToHost, stream));
  RTB(hipMemcpyAsync(out->axis.data(), axis.p, n_inner, hipMemcpyDeviceToHost, stream));
  RTB(hipMemcpyAsync(out->boxes.data(), node_box.p, (size_t)n_inner * 48, hipMemcpyDeviceToHost, stream));
  RTB(hipMemcpyAsync(out->order.data(), order.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
  RTB(hipMemcpyAsync(&out->root, root.p, 4, hipMemcpyDeviceToHost, stream));
  RTB(hipStreamSynchronize(stream));
  return out->root < n_inner ? hipSuccess : hipErrorUnknown;
}

hipError_t device_build(const rt_sphere* h_spheres, size_t n_, const uint8_t* h_material_type, size_t n_materials, DeviceBuilder builder,
                        DeviceTree* out, hipStream_t stream) {
  const uint32_t n = (uint32_t)n_;
  const uint32_t n_inner = n - 1;  // the caller guarantees n >= 2
  const unsigned blocks = (n + kThreads - 1) / kThreads, blocks_inner = (n_inner + kThreads - 1) / kThreads;

  Arena arena_, *const arena = &arena_;  // (declared before the arrays: it outlives them)
  arena_.reserve((size_t)n * kBuildScratchPerSphere + n_materials + (4u << 20));
  Scratch d_in, d_mtype, prim_box, centroid, block_bounds, bounds, keys, keys2, vals, vals2, left, right, parent_inner, parent_leaf, node_box,
      arrivals, level, level2, ids, order, new_of, misc, sort_tmp, flags;
  RTB(d_in.alloc((size_t)n * sizeof(rt_sphere), arena));
  RTB(d_mtype.alloc(n_materials, arena));
  RTB(hipMemcpyAsync(d_in.p, h_spheres, (size_t)n * sizeof(rt_sphere), hipMemcpyHostToDevice, stream));
  RTB(hipMemcpyAsync(d_mtype.p, h_material_type, n_materials, hipMemcpyHostToDevice, stream));
  RTB(prim_box.alloc((size_t)n * sizeof(Box6), arena));
  RTB(centroid.alloc((size_t)n * 24, arena));
  RTB(block_bounds.alloc((size_t)blocks * 48, arena));
  RTB(bounds.alloc(48, arena));
  RTB(flags.alloc(n, arena));
  RTB(keys.alloc((size_t)n * 8, arena));
  RTB(keys2.alloc((size_t)n * 8, arena));
  RTB(vals.alloc((size_t)n * 4, arena));
  RTB(vals2.alloc((size_t)n * 4, arena));
  RTB(left.alloc((size_t)n_inner * 4, arena));
  RTB(right.alloc((size_t)n_inner * 4, arena));
  RTB(parent_inner.alloc((size_t)n_inner * 4, arena));
  RTB(parent_leaf.alloc((size_t)n * 4, arena));
  RTB(node_box.alloc((size_t)n_inner * sizeof(Box6), arena));
  RTB(arrivals.alloc((size_t)n_inner * 4, arena));
  RTB(level.alloc((size_t)n_inner * 4, arena));
  RTB(level2.alloc((size_t)n_inner * 4, arena));
  RTB(ids.alloc((size_t)n_inner * 4, arena));
  RTB(order.alloc((size_t)n_inner * 4, arena));
  RTB(new_of.alloc((size_t)n_inner * 4, arena));
  RTB(misc.alloc(16, arena));  // [0] deepest level (u32), [1] bound (f32), [2] the root's id (surface-area build)
  RTB(hipMemsetAsync(arrivals.p, 0, (size_t)n_inner * 4, stream));
  RTB(hipMemsetAsync(misc.p, 0, 16, stream));

  size_t tmp_bytes = 0, tmp_bytes2 = 0;
  RTB(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys.as<uint64_t>(), keys2.as<uint64_t>(), vals.as<uint32_t>(), vals2.as<uint32_t>(), n, 0, 64,
                                stream));
  RTB(rocprim::radix_sort_pairs(nullptr, tmp_bytes2, level.as<uint32_t>(), level2.as<uint32_t>(), ids.as<uint32_t>(), order.as<uint32_t>(), n_inner,
                                0, 8, stream));
  RTB(sort_tmp.alloc(tmp_bytes > tmp_bytes2 ? tmp_bytes : tmp_bytes2, arena));

  Scratch split_axis;
  uint32_t root_id = 0;
  const bool sah = builder == DeviceBuilder::kBinnedSah;
  if (sah) {
    // boxes and centroids, then the hierarchy of the reference's own algorithm (rt_build_sah.hip); leaf k is sphere vals2[k]
    RTB(split_axis.alloc(n_inner, arena));
    hipLaunchKernelGGL(boxes_kernel, dim3(blocks), dim3(kThreads), 0, stream, d_in.as<rt_sphere>(), n, prim_box.as<Box6>(), centroid.as<double>(),
                       (const uint8_t*)nullptr, block_bounds.as<double>());
    RTB(hipGetLastError());
    RTB(sah_hierarchy(prim_box.as<Box6>(), centroid.as<double>(), n, left.as<uint32_t>(), right.as<uint32_t>(), parent_inner.as<uint32_t>(),
                      parent_leaf.as<uint32_t>(), split_axis.as<uint8_t>(), node_box.as<Box6>(), vals2.as<uint32_t>(), misc.as<uint32_t>() + 2, arena, stream));
    RTB(hipMemcpyAsync(&root_id, misc.as<uint32_t>() + 2, 4, hipMemcpyDeviceToHost, stream));
    RTB(hipStreamSynchronize(stream));
    if (root_id >= n_inner) return hipErrorUnknown;
  } else {
    // boxes and their bounds; which spheres stand apart; the centroid bounds of the others; Morton codes
    hipLaunchKernelGGL(boxes_kernel, dim3(blocks), dim3(kThreads), 0, stream, d_in.as<rt_sphere>(), n, prim_box.as<Box6>(), centroid.as<double>(),
                       (const uint8_t*)nullptr, block_bounds.as<double>());
    hipLaunchKernelGGL(bounds_kernel, dim3(1), dim3(kThreads), 0, stream, block_bounds.as<double>(), blocks, bounds.as<double>());
    hipLaunchKernelGGL(flag_kernel, dim3(blocks), dim3(kThreads), 0, stream, d_in.as<rt_sphere>(), n, bounds.as<double>(), flags.as<uint8_t>());
    hipLaunchKernelGGL(boxes_kernel, dim3(blocks), dim3(kThreads), 0, stream, d_in.as<rt_sphere>(), n, prim_box.as<Box6>(), centroid.as<double>(),
                       flags.as<uint8_t>(), block_bounds.as<double>());
    hipLaunchKernelGGL(bounds_kernel, dim3(1), dim3(kThreads), 0, stream, block_bounds.as<double>(), blocks, bounds.as<double>());
    hipLaunchKernelGGL(morton_kernel, dim3(blocks), dim3(kThreads), 0, stream, centroid.as<double>(), bounds.as<double>(), flags.as<uint8_t>(), n,
                       keys.as<uint64_t>(), vals.as<uint32_t>());
    RTB(hipGetLastError());

    RTB(rocprim::radix_sort_pairs(sort_tmp.p, tmp_bytes, keys.as<uint64_t>(), keys2.as<uint64_t>(), vals.as<uint32_t>(), vals2.as<uint32_t>(), n, 0, 64,
                                  stream));

    // clustering rounds over the Morton order (ping-pong arrays; the host reads the two counts back after every round)
    Scratch ref_a, ref_b, box_a, box_b, nearest, marks, scan, totals;
    RTB(ref_a.alloc((size_t)n * 4, arena));
    RTB(ref_b.alloc((size_t)n * 4, arena));
    RTB(box_a.alloc((size_t)n * sizeof(ClusterBox), arena));
    RTB(box_b.alloc((size_t)n * sizeof(ClusterBox), arena));
    RTB(nearest.alloc((size_t)n * 4, arena));
    RTB(marks.alloc((size_t)n * 8, arena));
    RTB(scan.alloc((size_t)n * 8, arena));
    RTB(totals.alloc(8, arena));
    size_t scan_bytes = 0;
    RTB(rocprim::exclusive_scan(nullptr, scan_bytes, marks.as<unsigned long long>(), scan.as<unsigned long long>(), 0ull, n,
                                rocprim::plus<unsigned long long>(), stream));
    Scratch scan_tmp;
    RTB(scan_tmp.alloc(scan_bytes, arena));
    hipLaunchKernelGGL(ploc_init_kernel, dim3(blocks), dim3(kThreads), 0, stream, prim_box.as<Box6>(), vals2.as<uint32_t>(), n, ref_a.as<uint32_t>(),
                       box_a.as<ClusterBox>());
    uint32_t m = n, next_id = 0;
    Scratch *ref_in = &ref_a, *ref_out = &ref_b, *box_in = &box_a, *box_out = &box_b;
    while (m > 1) {
      const unsigned bm = (m + kThreads - 1) / kThreads;
      hipLaunchKernelGGL(ploc_nearest_kernel, dim3(bm), dim3(kThreads), 0, stream, box_in->as<ClusterBox>(), (int)m, nearest.as<uint32_t>());
      hipLaunchKernelGGL(ploc_mark_kernel, dim3(bm), dim3(kThreads), 0, stream, nearest.as<uint32_t>(), (int)m, marks.as<unsigned long long>());
      size_t bytes = scan_bytes;
      RTB(rocprim::exclusive_scan(scan_tmp.p, bytes, marks.as<unsigned long long>(), scan.as<unsigned long long>(), 0ull, m,
                                  rocprim::plus<unsigned long long>(), stream));
      hipLaunchKernelGGL(ploc_merge_kernel, dim3(bm), dim3(kThreads), 0, stream, ref_in->as<uint32_t>(), box_in->as<ClusterBox>(),
                         nearest.as<uint32_t>(), marks.as<unsigned long long>(), scan.as<unsigned long long>(), (int)m, next_id,
                         ref_out->as<uint32_t>(), box_out->as<ClusterBox>(), left.as<uint32_t>(), right.as<uint32_t>(), parent_inner.as<uint32_t>(),
                         parent_leaf.as<uint32_t>(), totals.as<unsigned long long>());
      RTB(hipGetLastError());
      unsigned long long t = 0;
      RTB(hipMemcpyAsync(&t, totals.p, 8, hipMemcpyDeviceToHost, stream));
      RTB(hipStreamSynchronize(stream));
      const uint32_t left_over = (uint32_t)t, made = (uint32_t)(t >> 32);
      if (made == 0 || left_over + made != m) return hipErrorUnknown;  // (every round merges at least the closest pair)
      next_id += made;
      m = left_over;
      std::swap(ref_in, ref_out);
      std::swap(box_in, box_out);
    }
    if (next_id != n_inner) return hipErrorUnknown;
    root_id = n_inner - 1;  // the last node made
  }
  if (!sah)  // (the surface-area build knows every node's box already)
    hipLaunchKernelGGL(fit_kernel, dim3(blocks), dim3(kThreads), 0, stream, (int)n, left.as<uint32_t>(), right.as<uint32_t>(),
                       parent_inner.as<uint32_t>(), parent_leaf.as<uint32_t>(), prim_box.as<Box6>(), vals2.as<uint32_t>(), node_box.as<Box6>(),
                       arrivals.as<uint32_t>());
  hipLaunchKernelGGL(level_kernel, dim3(blocks_inner), dim3(kThreads), 0, stream, (int)n_inner, parent_inner.as<uint32_t>(), level.as<uint32_t>(),
                     ids.as<uint32_t>(), misc.as<uint32_t>());
  RTB(hipGetLastError());
  // Breadth-first numbering: a stable sort of the node ids by level (levels stay below 2^8: 63 key bits and 32 index bits).
  RTB(rocprim::radix_sort_pairs(sort_tmp.p, tmp_bytes2, level.as<uint32_t>(), level2.as<uint32_t>(), ids.as<uint32_t>(), order.as<uint32_t>(), n_inner,
                                0, 8, stream));
  hipLaunchKernelGGL(rank_kernel, dim3(blocks_inner), dim3(kThreads), 0, stream, (int)n_inner, order.as<uint32_t>(), new_of.as<uint32_t>());

  RTB(hipMalloc(&out->d_nodes, (size_t)n_inner * 64));
  RTB(hipMalloc(&out->d_spheres, (size_t)n * 32));
  RTB(hipMalloc(&out->d_aux, (size_t)n * 16));
  hipLaunchKernelGGL(emit_nodes_kernel, dim3(blocks_inner), dim3(kThreads), 0, stream, (int)n_inner, left.as<uint32_t>(), right.as<uint32_t>(),
                     new_of.as<uint32_t>(), prim_box.as<Box6>(), vals2.as<uint32_t>(), node_box.as<Box6>(), static_cast<float4*>(out->d_nodes),
                     misc.as<float>() + 1, root_id, sah ? split_axis.as<uint8_t>() : (const uint8_t*)nullptr);
  hipLaunchKernelGGL(emit_spheres_kernel, dim3(blocks), dim3(kThreads), 0, stream, d_in.as<rt_sphere>(), vals2.as<uint32_t>(), n,
                     d_mtype.as<uint8_t>(), static_cast<double*>(out->d_spheres), static_cast<double2*>(out->d_aux));
  RTB(hipGetLastError());

  uint32_t host_misc[4] = {0, 0, 0, 0};
  RTB(hipMemcpyAsync(host_misc, misc.p, 16, hipMemcpyDeviceToHost, stream));
  RTB(hipStreamSynchronize(stream));
  out->n_nodes = n_inner;
  out->depth = host_misc[0] + 1;  // inner levels plus the leaves
  std::memcpy(&out->bound, &host_misc[1], 4);
  out->root = 0;  // the root is alone on level 1, so it is first in breadth-first order
  return hipSuccess;
}

}  // namespace rta
