// BVH::new (src/bvh.rs:92-266) on the device: the binned surface-area build of the host builder (bvh_builder.cpp),
// restated so that it makes the same tree.  Every number that decides a split is computed as the host computes it:
// boxes and centroids in f64, box unions and counts (exact, whatever the order), the bucket of a centroid and the
// cost of a split by the same expressions in the same order (this file is compiled with floating-point contraction
// off, like the host's), the cheapest split by f64::total_cmp with the first of equals, partitions that keep the
// spheres of each side in their order.
//
// Segments (the spheres of one node, contiguous in an index array) of more than kSmall spheres are worked on level by
// level, many blocks to a segment: bounds and bucket histograms are gathered with atomic minima and maxima on
// order-preserving integer images of the doubles, one wave per segment picks the split, and the index array is
// partitioned into its twin.  Segments of at most kSmall spheres are finished by one wave each, which walks down the
// whole subtree with the histogram in LDS.  A node's id is the position of its split (the gap between two neighbouring
// leaves: each gap belongs to exactly one node), so ids need no allocation and come out the same in every run.
#include <hip/hip_runtime.h>

#include <utility>

#include "rt_build_internal.h"

namespace rta {
namespace {

constexpr int kBuckets = 32;              // src/bvh.rs:89
constexpr double kLeafCost = 1.0;         // src/bvh.rs:87
constexpr double kSplitCost = 1.0 / 8.0;  // src/bvh.rs:88
constexpr uint32_t kSmall = 128;          // segments up to this size are finished by one wave
constexpr uint32_t kChunk = 2048;         // positions of a large segment that one block works on
constexpr int kBlock = 256;
constexpr uint32_t kNone = 0xffffffffu, kLeafBit = 0x80000000u;
// Below this level the host builder stops splitting (bvh_builder.cpp: inputs such as spheres in geometric progression would
// otherwise cost a level per sphere); here the segments are cut in the middle from there on, so the tree stays below 64 + log2(n) levels.
constexpr uint32_t kMaxBuildDepth = 64;

struct Seg {
  uint32_t lo, hi;  // positions [lo, hi) of the index array
  uint32_t parent;  // id of the parent node, kNone for the root
  uint32_t side;    // bit 0: 0 = the parent's child 0, 1 = its child 1; bit 1 (segments handed to the small kernel): which index array holds the segment; bits 8 and up: the node's level (root: 1)
};
__device__ __forceinline__ uint32_t level_of(const Seg& s) { return s.side >> 8; }

// Doubles as integers that compare the same way (finite values, infinities and zeros of either sign).
__device__ __forceinline__ unsigned long long okey(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return b ^ ((b >> 63) ? ~0ull : 0x8000000000000000ull);
}
__device__ __forceinline__ double oval(unsigned long long k) {
  const unsigned long long b = k ^ ((k >> 63) ? 0x8000000000000000ull : ~0ull);
  return __longlong_as_double((long long)b);
}
constexpr unsigned long long kKeyMinInit = ~0ull, kKeyMaxInit = 0ull;  // what atomicMin / atomicMax start from

// src/bvh.rs:250-258 as in bvh_builder.cpp (the cast saturates and maps NaN to 0 as Rust's `as usize` does)
__device__ __forceinline__ int choose_bucket(double min_t, double max_t, double t) {
  const double f = (t - min_t) / (max_t - min_t);
  const double scaled = (double)kBuckets * f;
  if (!(scaled > 0.0)) return 0;
  if (scaled >= (double)kBuckets) return kBuckets - 1;
  return (int)scaled;
}
__device__ __forceinline__ double surface_area(const double lo[3], const double hi[3]) {  // src/aabb.rs:31-34
  const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
  return 2.0 * (x * y + x * z + z * y);
}
__device__ __forceinline__ long long total_order_key(double v) {  // f64::total_cmp (min_by at src/bvh.rs:227)
  long long b = __double_as_longlong(v);
  b ^= (long long)((unsigned long long)(b >> 63) >> 1);
  return b;
}

// Per axis and bucket: the union of the boxes of the spheres whose centroid falls into the bucket, and their number.
struct Hist {
  unsigned long long lo[3][kBuckets][3];  // okey of the low planes: minima
  unsigned long long hi[3][kBuckets][3];  // okey of the high planes: maxima
  uint32_t cnt[3][kBuckets];
};
// What the blocks of a large segment gather.
struct BigAcc {
  unsigned long long mn[6];  // okey: box low xyz, centroid minimum xyz
  unsigned long long mx[6];  // okey: box high xyz, centroid maximum xyz
  Hist h;
};
struct Bounds {
  double lo[3], hi[3];      // the node's box
  double cmin[3], cmax[3];  // bounds of the centroids
};
struct Decision {
  int axis, split;  // spheres whose bucket along `axis` is below `split` go to child 0 ...
  uint32_t nl;      // ... nl of them
  bool sah;         // false: no split is cheaper than a leaf (or none exists): the segment is cut in the middle as it stands
};

// LDS that one wave needs to pick a split.
struct Sweep {
  double pa[3][kBuckets], sa[3][kBuckets];    // surface area of the union of buckets [0, i) and of buckets [i, kBuckets)
  uint32_t pn[3][kBuckets], sn[3][kBuckets];  // spheres in them
};

__device__ __forceinline__ void lane_sync() { __syncthreads(); }  // (blocks of one wave)

// The split of a segment of n spheres, by the 64 lanes of a wave together (every lane returns the same answer).
// The sweep of src/bvh.rs:196-227: per axis whose centroids differ, every boundary between buckets, cost
// kSplitCost + kLeafCost * (n_left * area_left + n_right * area_right) / area_node, the lowest by total order, the first of equals.
__device__ __forceinline__ Decision pick_split(const Hist* h, const Bounds& b, uint32_t n, bool may_split, Sweep* sw, uint32_t lane) {
  bool valid[3];
  for (int a = 0; a < 3; ++a) valid[a] = (b.cmax[a] - b.cmin[a]) > 0.0;
  if (lane < 6) {
    const int a = (int)(lane >> 1);
    if (valid[a]) {
      double lo[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()}, hi[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
      uint32_t count = 0;
      double area = surface_area(lo, hi);  // (of nothing: never used, bucket 0 and bucket kBuckets - 1 are never empty)
      // (most buckets of a small segment are empty: the union and its area only change at the others)
      if ((lane & 1u) == 0u) {
        for (int k = 0; k < kBuckets - 1; ++k) {  // unions of buckets [0, k]
          const uint32_t c = h->cnt[a][k];
          if (c) {
            for (int x = 0; x < 3; ++x) lo[x] = fmin(lo[x], oval(h->lo[a][k][x])), hi[x] = fmax(hi[x], oval(h->hi[a][k][x]));
            count += c;
            area = surface_area(lo, hi);
          }
          sw->pa[a][k + 1] = area;
          sw->pn[a][k + 1] = count;
        }
      } else {
        for (int k = kBuckets - 1; k >= 1; --k) {  // unions of buckets [k, kBuckets)
          const uint32_t c = h->cnt[a][k];
          if (c) {
            for (int x = 0; x < 3; ++x) lo[x] = fmin(lo[x], oval(h->lo[a][k][x])), hi[x] = fmax(hi[x], oval(h->hi[a][k][x]));
            count += c;
            area = surface_area(lo, hi);
          }
          sw->sa[a][k] = area;
          sw->sn[a][k] = count;
        }
      }
    }
  }
  lane_sync();
  const double node_area = surface_area(b.lo, b.hi);
  constexpr int kCandidates = 3 * (kBuckets - 1);
  auto cost_of = [&](int c) {
    const int a = c / (kBuckets - 1), i = c % (kBuckets - 1) + 1;
    return kSplitCost + kLeafCost * ((double)sw->pn[a][i] * sw->pa[a][i] + (double)sw->sn[a][i] * sw->sa[a][i]) / node_area;  // src/bvh.rs:218-226
  };
  long long best_key = 0x7fffffffffffffffll;
  int best_c = 1 << 30;
  for (int c = (int)lane; c < kCandidates; c += 64) {
    if (!valid[c / (kBuckets - 1)]) continue;
    const long long key = total_order_key(cost_of(c));
    if (best_c == (1 << 30) || key < best_key) {  // (a lane's candidates come in rising order: the first of equals stays)
      best_key = key;
      best_c = c;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const long long k2 = __shfl_xor(best_key, off);
    const int c2 = __shfl_xor(best_c, off);
    if (c2 != (1 << 30) && (best_c == (1 << 30) || k2 < best_key || (k2 == best_key && c2 < best_c))) {
      best_key = k2;
      best_c = c2;
    }
  }
  Decision d;
  d.sah = false;
  d.axis = 0;
  d.split = 0;
  d.nl = n / 2;
  if (best_c != (1 << 30)) {
    const double best_cost = cost_of(best_c);
    if (best_cost < kLeafCost * (double)n && may_split) {  // src/bvh.rs:124
      d.sah = true;
      d.axis = best_c / (kBuckets - 1);
      d.split = best_c % (kBuckets - 1) + 1;
      d.nl = sw->pn[d.axis][d.split];
    }
  }
  lane_sync();  // the sweep arrays may be written again after this
  if (d.sah && (d.nl == 0u || d.nl >= n)) {  // cannot happen (buckets 0 and kBuckets - 1 are never empty); never trust a split that leaves a side empty
    d.sah = false;
    d.nl = n / 2;
  }
  return d;
}

__device__ __forceinline__ void hist_clear(Hist* h, uint32_t tid, uint32_t nthreads) {
  unsigned long long* lo = &h->lo[0][0][0];
  unsigned long long* hi = &h->hi[0][0][0];
  uint32_t* cnt = &h->cnt[0][0];
  for (uint32_t w = tid; w < 3u * kBuckets * 3u; w += nthreads) lo[w] = kKeyMinInit, hi[w] = kKeyMaxInit;
  for (uint32_t w = tid; w < 3u * kBuckets; w += nthreads) cnt[w] = 0u;
}
__device__ __forceinline__ void hist_add(Hist* h, int a, int bucket, const Box6& box) {
  for (int x = 0; x < 3; ++x) {
    atomicMin(&h->lo[a][bucket][x], okey(box.lo[x]));
    atomicMax(&h->hi[a][bucket][x], okey(box.hi[x]));
  }
  atomicAdd(&h->cnt[a][bucket], 1u);
}

// Where the hierarchy is written.  make_node: the node of a segment once its split is known (its box, its split axis and its
// links to the parent); make_leaf: a child that is a single sphere.
struct Links {
  uint32_t *left, *right, *parent_inner, *parent_leaf, *order, *root_id;
  uint8_t* axis;
  Box6* node_box;
};
__device__ __forceinline__ uint32_t make_node(const Links& L, const Seg& s, const Decision& d, const double lo[3], const double hi[3]) {
  const uint32_t id = s.lo + d.nl - 1u;
  Box6 box;
  for (int a = 0; a < 3; ++a) box.lo[a] = lo[a], box.hi[a] = hi[a];
  L.node_box[id] = box;
  if (s.parent != kNone)
    ((s.side & 1u) ? L.right : L.left)[s.parent] = id;
  else
    *L.root_id = id;
  L.parent_inner[id] = s.parent;
  L.axis[id] = (uint8_t)(d.sah ? d.axis : 3);
  return id;
}
// (the leaf's sphere, order[pos], is written by whoever moved it there)
__device__ __forceinline__ void make_leaf(const Links& L, uint32_t id, uint32_t side, uint32_t pos) {
  (side ? L.right : L.left)[id] = kLeafBit | pos;
  L.parent_leaf[pos] = id;
}

// ------------------------------------------------------------ small segments ----
__global__ __launch_bounds__(64) void small_kernel(const Seg* __restrict__ roots, uint32_t n_roots, const Box6* __restrict__ prim_box,
                                                   const double* __restrict__ centroid, uint32_t* idx0, uint32_t* idx1, Links L) {
  __shared__ Hist h;
  __shared__ Sweep sw;
  __shared__ Seg stack[kSmall];
  const uint32_t lane = threadIdx.x;
  if (blockIdx.x >= n_roots) return;
  if (lane == 0) stack[0] = roots[blockIdx.x];
  int sp = 1;
  lane_sync();
  while (sp > 0) {
    const Seg s = stack[--sp];
    lane_sync();  // everyone has read the entry before a push can overwrite it
    const uint32_t n = s.hi - s.lo, buf = (s.side >> 1) & 1u;
    const uint32_t* in = buf ? idx1 : idx0;
    uint32_t* out = buf ? idx0 : idx1;
    if (n == 2u) {
      // Two spheres (a third of all nodes): along any axis on which their centroids differ one lands in the first bucket
      // and one in the last, every boundary between buckets separates them, and the cost is the same whatever the axis; so
      // the sweep's answer is the first such axis with the lower centroid in child 0.  One lane does it.
      if (lane == 0) {
        const uint32_t p0 = in[s.lo], p1 = in[s.lo + 1u];
        const Box6 b0 = prim_box[p0], b1 = prim_box[p1];
        Decision d;
        d.sah = false;
        d.axis = 0;
        d.split = 1;
        d.nl = 1u;
        bool swap = false;
        for (int a = 2; a >= 0; --a) {
          const double c0 = centroid[3 * (size_t)p0 + a], c1 = centroid[3 * (size_t)p1 + a];
          if ((fmax(c0, c1) - fmin(c0, c1)) > 0.0) {
            d.axis = a;
            d.sah = true;
            swap = c1 < c0;
          }
        }
        double lo[3], hi[3];
        for (int a = 0; a < 3; ++a) lo[a] = fmin(b0.lo[a], b1.lo[a]), hi[a] = fmax(b0.hi[a], b1.hi[a]);
        if (d.sah) {
          const Box6 &first = swap ? b1 : b0, &second = swap ? b0 : b1;
          const double cost = kSplitCost + kLeafCost * ((double)1u * surface_area(first.lo, first.hi) + (double)1u * surface_area(second.lo, second.hi)) / surface_area(lo, hi);
          if (!(cost < kLeafCost * (double)2u) || level_of(s) >= kMaxBuildDepth) d.sah = swap = false;  // src/bvh.rs:124: the reference keeps the two in one leaf; here they stay as they are
        }
        const uint32_t id = make_node(L, s, d, lo, hi);
        L.left[id] = kLeafBit | s.lo;
        L.right[id] = kLeafBit | (s.lo + 1u);
        L.parent_leaf[s.lo] = L.parent_leaf[s.lo + 1u] = id;
        L.order[s.lo] = swap ? p1 : p0;
        L.order[s.lo + 1u] = swap ? p0 : p1;
      }
      lane_sync();
      continue;
    }
    // The first 64 spheres of the segment stay in registers through the three passes (bounds, histogram, partition):
    // most segments have no more.  (The index arrays are written by some lanes and read by others a round later: the wave's
    // own L1 sees its stores, and the fence below keeps the order; the accesses are marked as streaming.)
    const uint32_t p0 = s.lo + lane;
    const bool have0 = p0 < s.hi;
    const uint32_t prim0 = have0 ? __builtin_nontemporal_load(in + p0) : 0u;
    Box6 box0;
    double c0[3];
    if (have0) {
      box0 = prim_box[prim0];
      for (int a = 0; a < 3; ++a) c0[a] = centroid[3 * (size_t)prim0 + a];
    } else {
      for (int a = 0; a < 3; ++a) box0.lo[a] = c0[a] = __builtin_inf(), box0.hi[a] = -__builtin_inf();
    }
    // bounds of the boxes and of the centroids (AABB::bounding_box, src/aabb.rs:40-46)
    Bounds b;
    for (int a = 0; a < 3; ++a) b.lo[a] = box0.lo[a], b.hi[a] = box0.hi[a], b.cmin[a] = have0 ? c0[a] : __builtin_inf(), b.cmax[a] = have0 ? c0[a] : -__builtin_inf();
    for (uint32_t p = p0 + 64u; p < s.hi; p += 64u) {
      const uint32_t prim = __builtin_nontemporal_load(in + p);
      const Box6 box = prim_box[prim];
      for (int a = 0; a < 3; ++a) {
        const double c = centroid[3 * (size_t)prim + a];
        b.lo[a] = fmin(b.lo[a], box.lo[a]);
        b.hi[a] = fmax(b.hi[a], box.hi[a]);
        b.cmin[a] = fmin(b.cmin[a], c);
        b.cmax[a] = fmax(b.cmax[a], c);
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      b.lo[0] = fmin(b.lo[0], __shfl_xor(b.lo[0], off)), b.lo[1] = fmin(b.lo[1], __shfl_xor(b.lo[1], off)), b.lo[2] = fmin(b.lo[2], __shfl_xor(b.lo[2], off));
      b.hi[0] = fmax(b.hi[0], __shfl_xor(b.hi[0], off)), b.hi[1] = fmax(b.hi[1], __shfl_xor(b.hi[1], off)), b.hi[2] = fmax(b.hi[2], __shfl_xor(b.hi[2], off));
      b.cmin[0] = fmin(b.cmin[0], __shfl_xor(b.cmin[0], off)), b.cmin[1] = fmin(b.cmin[1], __shfl_xor(b.cmin[1], off)), b.cmin[2] = fmin(b.cmin[2], __shfl_xor(b.cmin[2], off));
      b.cmax[0] = fmax(b.cmax[0], __shfl_xor(b.cmax[0], off)), b.cmax[1] = fmax(b.cmax[1], __shfl_xor(b.cmax[1], off)), b.cmax[2] = fmax(b.cmax[2], __shfl_xor(b.cmax[2], off));
    }
    hist_clear(&h, lane, 64u);
    lane_sync();
    if (have0) {
      if ((b.cmax[0] - b.cmin[0]) > 0.0) hist_add(&h, 0, choose_bucket(b.cmin[0], b.cmax[0], c0[0]), box0);
      if ((b.cmax[1] - b.cmin[1]) > 0.0) hist_add(&h, 1, choose_bucket(b.cmin[1], b.cmax[1], c0[1]), box0);
      if ((b.cmax[2] - b.cmin[2]) > 0.0) hist_add(&h, 2, choose_bucket(b.cmin[2], b.cmax[2], c0[2]), box0);
    }
    for (uint32_t p = p0 + 64u; p < s.hi; p += 64u) {
      const uint32_t prim = __builtin_nontemporal_load(in + p);
      const Box6 box = prim_box[prim];
      for (int a = 0; a < 3; ++a)
        if ((b.cmax[a] - b.cmin[a]) > 0.0) hist_add(&h, a, choose_bucket(b.cmin[a], b.cmax[a], centroid[3 * (size_t)prim + a]), box);
    }
    lane_sync();
    const Decision d = pick_split(&h, b, n, level_of(s) < kMaxBuildDepth, &sw, lane);
    const uint32_t mid = s.lo + d.nl;
    const bool leaf_left = d.nl == 1u, leaf_right = s.hi - mid == 1u;
    uint32_t newbuf = buf;
    if (d.sah) {  // stable partition into the other array; a sphere that ends up alone is a leaf: its place in the final order is written here
      uint32_t n_left = 0, n_right = 0;
      const double cmin = d.axis == 0 ? b.cmin[0] : d.axis == 1 ? b.cmin[1] : b.cmin[2], cmax = d.axis == 0 ? b.cmax[0] : d.axis == 1 ? b.cmax[1] : b.cmax[2];
      for (uint32_t base = s.lo; base < s.hi; base += 64u) {
        const uint32_t p = base + lane;
        const bool act = p < s.hi, first = base == s.lo;
        const uint32_t prim = first ? prim0 : (act ? __builtin_nontemporal_load(in + p) : 0u);
        const double c = first ? (d.axis == 0 ? c0[0] : d.axis == 1 ? c0[1] : c0[2]) : (act ? centroid[3 * (size_t)prim + d.axis] : 0.0);
        const bool to_left = act && choose_bucket(cmin, cmax, c) < d.split;
        const unsigned long long m_left = __ballot(to_left), m_right = __ballot(act && !to_left);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (act) {
          const uint32_t dest = to_left ? s.lo + n_left + (uint32_t)__popcll(m_left & below) : mid + n_right + (uint32_t)__popcll(m_right & below);
          __builtin_nontemporal_store(prim, out + dest);
          if (to_left ? leaf_left : leaf_right) L.order[dest] = prim;
        }
        n_left += (uint32_t)__popcll(m_left);
        n_right += (uint32_t)__popcll(m_right);
      }
      newbuf = buf ^ 1u;
    } else {  // cut in the middle, nothing moves
      if (leaf_left && lane == 0) L.order[s.lo] = prim0;
      if (leaf_right) {
        const uint32_t last = s.hi - 1u;
        if (((last - s.lo) & 63u) == lane) L.order[last] = last - s.lo < 64u ? prim0 : __builtin_nontemporal_load(in + last);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the stores above before the loads of the next round (one wave: program order does the rest)
    lane_sync();
    if (lane == 0) {
      const uint32_t id = make_node(L, s, d, b.lo, b.hi);
      // child 1 first, so that child 0 is taken off the stack first
      if (leaf_right)
        make_leaf(L, id, 1u, mid);
      else
        stack[sp] = Seg{mid, s.hi, id, 1u | (newbuf << 1) | ((level_of(s) + 1u) << 8)};
      if (leaf_left)
        make_leaf(L, id, 0u, s.lo);
      else
        stack[sp + (leaf_right ? 0 : 1)] = Seg{s.lo, mid, id, 0u | (newbuf << 1) | ((level_of(s) + 1u) << 8)};
    }
    sp += (leaf_right ? 0 : 1) + (leaf_left ? 0 : 1);
    lane_sync();
  }
}

// ------------------------------------------------------------ large segments ----
// Block `item` works on chunk (item - chunk_off[k]) of segment k, where chunk_off[k] <= item < chunk_off[k + 1].
__device__ __forceinline__ uint32_t segment_of_item(const uint32_t* __restrict__ chunk_off, uint32_t n_segs, uint32_t item) {
  uint32_t lo = 0, hi = n_segs;  // chunk_off[lo] <= item < chunk_off[hi]
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (chunk_off[mid] <= item)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(kBlock) void big_clear_kernel(BigAcc* __restrict__ acc, uint32_t n_segs) {
  const uint32_t k = blockIdx.x;
  if (k >= n_segs) return;
  if (threadIdx.x < 6) acc[k].mn[threadIdx.x] = kKeyMinInit, acc[k].mx[threadIdx.x] = kKeyMaxInit;
  hist_clear(&acc[k].h, threadIdx.x, kBlock);
}

__global__ __launch_bounds__(kBlock) void big_bounds_kernel(const Seg* __restrict__ segs, uint32_t n_segs, const uint32_t* __restrict__ chunk_off,
                                                            const Box6* __restrict__ prim_box, const double* __restrict__ centroid,
                                                            const uint32_t* __restrict__ in, BigAcc* __restrict__ acc) {
  __shared__ double red[12][kBlock / 64];
  const uint32_t k = segment_of_item(chunk_off, n_segs, blockIdx.x);
  const Seg s = segs[k];
  const uint32_t begin = s.lo + (blockIdx.x - chunk_off[k]) * kChunk, end = min(begin + kChunk, s.hi);
  double v[12];  // minima 0..5 (box low, centroid), maxima 6..11 (box high, centroid)
  for (int a = 0; a < 6; ++a) v[a] = __builtin_inf(), v[6 + a] = -__builtin_inf();
  for (uint32_t p = begin + threadIdx.x; p < end; p += kBlock) {
    const uint32_t prim = in[p];
    const Box6 box = prim_box[prim];
    for (int a = 0; a < 3; ++a) {
      const double c = centroid[3 * (size_t)prim + a];
      v[a] = fmin(v[a], box.lo[a]);
      v[3 + a] = fmin(v[3 + a], c);
      v[6 + a] = fmax(v[6 + a], box.hi[a]);
      v[9 + a] = fmax(v[9 + a], c);
    }
  }
  for (int off = 32; off > 0; off >>= 1)
    for (int a = 0; a < 6; ++a) {
      v[a] = fmin(v[a], __shfl_xor(v[a], off));
      v[6 + a] = fmax(v[6 + a], __shfl_xor(v[6 + a], off));
    }
  if ((threadIdx.x & 63u) == 0u)
    for (int a = 0; a < 12; ++a) red[a][threadIdx.x >> 6] = v[a];
  __syncthreads();
  if (threadIdx.x < 12) {
    double r = red[threadIdx.x][0];
    for (int w = 1; w < kBlock / 64; ++w) r = threadIdx.x < 6 ? fmin(r, red[threadIdx.x][w]) : fmax(r, red[threadIdx.x][w]);
    if (threadIdx.x < 6)
      atomicMin(&acc[k].mn[threadIdx.x], okey(r));
    else
      atomicMax(&acc[k].mx[threadIdx.x - 6], okey(r));
  }
}

__device__ __forceinline__ Bounds bounds_of(const BigAcc& a) {
  Bounds b;
  for (int x = 0; x < 3; ++x) {
    b.lo[x] = oval(a.mn[x]);
    b.cmin[x] = oval(a.mn[3 + x]);
    b.hi[x] = oval(a.mx[x]);
    b.cmax[x] = oval(a.mx[3 + x]);
  }
  return b;
}

__global__ __launch_bounds__(kBlock) void big_hist_kernel(const Seg* __restrict__ segs, uint32_t n_segs, const uint32_t* __restrict__ chunk_off,
                                                          const Box6* __restrict__ prim_box, const double* __restrict__ centroid,
                                                          const uint32_t* __restrict__ in, BigAcc* acc) {
  __shared__ Hist h;
  const uint32_t k = segment_of_item(chunk_off, n_segs, blockIdx.x);
  const Seg s = segs[k];
  const uint32_t begin = s.lo + (blockIdx.x - chunk_off[k]) * kChunk, end = min(begin + kChunk, s.hi);
  double cmin[3], cmax[3];
  for (int a = 0; a < 3; ++a) cmin[a] = oval(acc[k].mn[3 + a]), cmax[a] = oval(acc[k].mx[3 + a]);
  hist_clear(&h, threadIdx.x, kBlock);
  __syncthreads();
  for (uint32_t p = begin + threadIdx.x; p < end; p += kBlock) {
    const uint32_t prim = in[p];
    const Box6 box = prim_box[prim];
    for (int a = 0; a < 3; ++a)
      if ((cmax[a] - cmin[a]) > 0.0) hist_add(&h, a, choose_bucket(cmin[a], cmax[a], centroid[3 * (size_t)prim + a]), box);
  }
  __syncthreads();
  if (threadIdx.x < 3 * kBuckets) {
    const int a = threadIdx.x / kBuckets, bk = threadIdx.x % kBuckets;
    const uint32_t c = h.cnt[a][bk];
    if (c) {
      for (int x = 0; x < 3; ++x) {
        atomicMin(&acc[k].h.lo[a][bk][x], h.lo[a][bk][x]);
        atomicMax(&acc[k].h.hi[a][bk][x], h.hi[a][bk][x]);
      }
      atomicAdd(&acc[k].h.cnt[a][bk], c);
    }
  }
}

struct BigDecision {
  Decision d;
  double cmin, cmax;  // of the split's axis
  uint32_t id;
};

__global__ __launch_bounds__(64) void big_decide_kernel(const Seg* __restrict__ segs, uint32_t n_segs, const BigAcc* __restrict__ acc,
                                                        BigDecision* __restrict__ dec, Links L) {
  __shared__ Sweep sw;
  const uint32_t k = blockIdx.x;
  if (k >= n_segs) return;
  const Seg s = segs[k];
  const Bounds b = bounds_of(acc[k]);
  const Decision d = pick_split(&acc[k].h, b, s.hi - s.lo, level_of(s) < kMaxBuildDepth, &sw, threadIdx.x);
  if (threadIdx.x == 0) {
    BigDecision out;
    out.d = d;
    out.cmin = b.cmin[d.axis];
    out.cmax = b.cmax[d.axis];
    out.id = make_node(L, s, d, b.lo, b.hi);
    dec[k] = out;
  }
}

__device__ __forceinline__ bool goes_left(const BigDecision& bd, const Seg& s, uint32_t p, uint32_t prim, const double* __restrict__ centroid) {
  if (!bd.d.sah) return p - s.lo < bd.d.nl;
  return choose_bucket(bd.cmin, bd.cmax, centroid[3 * (size_t)prim + bd.d.axis]) < bd.d.split;
}

// Spheres of each chunk that go to child 0.
__global__ __launch_bounds__(kBlock) void big_count_kernel(const Seg* __restrict__ segs, uint32_t n_segs, const uint32_t* __restrict__ chunk_off,
                                                           const BigDecision* __restrict__ dec, const double* __restrict__ centroid,
                                                           const uint32_t* __restrict__ in, uint32_t* __restrict__ chunk_left) {
  __shared__ uint32_t total;
  const uint32_t k = segment_of_item(chunk_off, n_segs, blockIdx.x);
  const Seg s = segs[k];
  const BigDecision bd = dec[k];
  const uint32_t begin = s.lo + (blockIdx.x - chunk_off[k]) * kChunk, end = min(begin + kChunk, s.hi);
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  uint32_t mine = 0;
  for (uint32_t p = begin + threadIdx.x; p < end; p += kBlock) mine += goes_left(bd, s, p, in[p], centroid) ? 1u : 0u;
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
  if ((threadIdx.x & 63u) == 0u) atomicAdd(&total, mine);
  __syncthreads();
  if (threadIdx.x == 0) chunk_left[blockIdx.x] = total;
}

// chunk_left -> the number of spheres that go to child 0 in the chunks of the same segment before each chunk (a thread per
// segment: few segments have more than a chunk or two).
__global__ __launch_bounds__(kBlock) void big_offsets_kernel(uint32_t n_segs, const uint32_t* __restrict__ chunk_off, uint32_t* __restrict__ chunk_left) {
  const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= n_segs) return;
  uint32_t run = 0;
  for (uint32_t c = chunk_off[k]; c < chunk_off[k + 1]; ++c) {
    const uint32_t v = chunk_left[c];
    chunk_left[c] = run;
    run += v;
  }
}

// The stable partition: thread t of a block takes kChunk / kBlock consecutive positions of the chunk.
__global__ __launch_bounds__(kBlock) void big_scatter_kernel(const Seg* __restrict__ segs, uint32_t n_segs, const uint32_t* __restrict__ chunk_off,
                                                             const BigDecision* __restrict__ dec, const double* __restrict__ centroid,
                                                             const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                             const uint32_t* __restrict__ chunk_left) {
  constexpr uint32_t kPer = kChunk / kBlock;
  __shared__ uint32_t sums[kBlock];
  const uint32_t k = segment_of_item(chunk_off, n_segs, blockIdx.x);
  const Seg s = segs[k];
  const BigDecision bd = dec[k];
  const uint32_t chunk = blockIdx.x - chunk_off[k];
  const uint32_t begin = s.lo + chunk * kChunk, end = min(begin + kChunk, s.hi);
  const uint32_t first = begin + threadIdx.x * kPer;
  uint32_t prim[kPer];
  bool to_left[kPer];
  uint32_t mine = 0;
  for (uint32_t j = 0; j < kPer; ++j) {
    const uint32_t p = first + j;
    prim[j] = p < end ? in[p] : 0u;
    to_left[j] = p < end && goes_left(bd, s, p, prim[j], centroid);
    mine += to_left[j] ? 1u : 0u;
  }
  sums[threadIdx.x] = mine;
  __syncthreads();
  for (int off = 1; off < kBlock; off <<= 1) {  // inclusive scan of the threads' counts
    const uint32_t v = (int)threadIdx.x >= off ? sums[threadIdx.x - off] : 0u;
    __syncthreads();
    sums[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t left_before = chunk_left[blockIdx.x] + sums[threadIdx.x] - mine;  // in the segment, before this thread's positions
  uint32_t right_before = (first - s.lo) - left_before;
  for (uint32_t j = 0; j < kPer; ++j) {
    const uint32_t p = first + j;
    if (p >= end) break;
    if (to_left[j])
      out[s.lo + left_before++] = prim[j];
    else
      out[s.lo + bd.d.nl + right_before++] = prim[j];
  }
}

// The children of the level's segments: single spheres become leaves, segments of at most kSmall spheres go on the list
// for the small kernel (`buf`: the index array they are in now), larger ones on the next level's list.
__global__ __launch_bounds__(kBlock) void big_children_kernel(const Seg* __restrict__ segs, uint32_t n_segs, const BigDecision* __restrict__ dec,
                                                              const uint32_t* __restrict__ arr, uint32_t buf, Seg* __restrict__ next,
                                                              Seg* __restrict__ small, uint32_t* __restrict__ counts, Links L) {
  const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= n_segs) return;
  const Seg s = segs[k];
  const BigDecision bd = dec[k];
  const uint32_t mid = s.lo + bd.d.nl;
  for (uint32_t side = 0; side < 2u; ++side) {
    const uint32_t lo = side ? mid : s.lo, hi = side ? s.hi : mid;
    if (hi - lo == 1u) {
      make_leaf(L, bd.id, side, lo);
      L.order[lo] = arr[lo];
    }
    else if (hi - lo <= kSmall)
      small[atomicAdd(&counts[1], 1u)] = Seg{lo, hi, bd.id, side | (buf << 1) | ((level_of(s) + 1u) << 8)};
    else
      next[atomicAdd(&counts[0], 1u)] = Seg{lo, hi, bd.id, side | ((level_of(s) + 1u) << 8)};
  }
}

// chunk_off for the next level's segments; counts[2] = the number of chunks.  One block: thread t sums the chunks of a run of
// consecutive segments, the runs' totals are scanned in LDS, and each thread writes its segments' offsets.
constexpr int kPlanThreads = 1024;
__global__ __launch_bounds__(kPlanThreads) void big_plan_kernel(const Seg* __restrict__ segs, uint32_t* __restrict__ chunk_off, uint32_t* counts) {
  __shared__ uint32_t sums[kPlanThreads];
  const uint32_t n_segs = counts[0], per = (n_segs + kPlanThreads - 1u) / kPlanThreads;
  const uint32_t first = threadIdx.x * per, last = min(first + per, n_segs);
  uint32_t mine = 0;
  for (uint32_t k = first; k < last; ++k) mine += (segs[k].hi - segs[k].lo + kChunk - 1u) / kChunk;
  sums[threadIdx.x] = mine;
  __syncthreads();
  for (int off = 1; off < kPlanThreads; off <<= 1) {
    const uint32_t v = (int)threadIdx.x >= off ? sums[threadIdx.x - off] : 0u;
    __syncthreads();
    sums[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = sums[threadIdx.x] - mine;
  for (uint32_t k = first; k < last; ++k) {
    chunk_off[k] = run;
    run += (segs[k].hi - segs[k].lo + kChunk - 1u) / kChunk;
  }
  if (threadIdx.x == kPlanThreads - 1) {
    chunk_off[n_segs] = sums[kPlanThreads - 1];
    counts[2] = sums[kPlanThreads - 1];
  }
}

__global__ __launch_bounds__(kBlock) void iota_kernel(uint32_t* __restrict__ idx, uint32_t n) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) idx[i] = i;
}

#define RTB(call)                    \
  do {                               \
    hipError_t e_ = (call);          \
    if (e_ != hipSuccess) return e_; \
  } while (0)

}  // namespace

hipError_t sah_hierarchy(const Box6* prim_box, const double* centroid, uint32_t n, uint32_t* left, uint32_t* right, uint32_t* parent_inner,
                         uint32_t* parent_leaf, uint8_t* axis, Box6* node_box, uint32_t* order, uint32_t* root_id, Arena* arena, hipStream_t stream) {
  if (n < 2) return hipErrorInvalidValue;
  const uint32_t max_big = n / kSmall + 2, max_small = n / 2 + 2, max_chunks = n / kChunk + max_big + 2;
  Scratch idx0, idx1, segs_a, segs_b, small, acc, dec, chunk_off, chunk_left, counts;
  RTB(idx0.alloc((size_t)n * 4, arena));
  RTB(idx1.alloc((size_t)n * 4, arena));
  RTB(segs_a.alloc((size_t)max_big * sizeof(Seg), arena));
  RTB(segs_b.alloc((size_t)max_big * sizeof(Seg), arena));
  RTB(small.alloc((size_t)max_small * sizeof(Seg), arena));
  RTB(acc.alloc((size_t)max_big * sizeof(BigAcc), arena));
  RTB(dec.alloc((size_t)max_big * sizeof(BigDecision), arena));
  RTB(chunk_off.alloc((size_t)(max_big + 1) * 4, arena));
  RTB(chunk_left.alloc((size_t)max_chunks * 4, arena));
  RTB(counts.alloc(16, arena));  // [0] segments of the next level, [1] segments for the small kernel, [2] chunks of the next level
  RTB(hipMemsetAsync(counts.p, 0, 16, stream));
  hipLaunchKernelGGL(iota_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, idx0.as<uint32_t>(), n);  // ascending sphere order, as on the host
  const Links L{left, right, parent_inner, parent_leaf, order, root_id, axis, node_box};

  const Seg root{0u, n, kNone, 1u << 8};
  uint32_t n_segs = 0, n_chunks = 0, n_small = 0;
  Scratch *cur = &segs_a, *nxt = &segs_b;
  if (n <= kSmall) {
    RTB(hipMemcpyAsync(small.p, &root, sizeof(Seg), hipMemcpyHostToDevice, stream));
    n_small = 1;
  } else {
    RTB(hipMemcpyAsync(cur->p, &root, sizeof(Seg), hipMemcpyHostToDevice, stream));
    const uint32_t off[2] = {0u, (n + kChunk - 1u) / kChunk};
    RTB(hipMemcpyAsync(chunk_off.p, off, 8, hipMemcpyHostToDevice, stream));
    n_segs = 1;
    n_chunks = off[1];
  }
  uint32_t buf = 0;  // the index array the level's segments are in
  for (uint32_t level = 0; n_segs > 0; ++level) {
    if (level > kMaxBuildDepth + 40 || n_segs > max_big || n_chunks > max_chunks) return hipErrorUnknown;  // cannot happen; never run past the buffers
    const uint32_t* in = buf ? idx1.as<uint32_t>() : idx0.as<uint32_t>();
    uint32_t* out = buf ? idx0.as<uint32_t>() : idx1.as<uint32_t>();
    hipLaunchKernelGGL(big_clear_kernel, dim3(n_segs), dim3(kBlock), 0, stream, acc.as<BigAcc>(), n_segs);
    hipLaunchKernelGGL(big_bounds_kernel, dim3(n_chunks), dim3(kBlock), 0, stream, cur->as<Seg>(), n_segs, chunk_off.as<uint32_t>(), prim_box, centroid,
                       in, acc.as<BigAcc>());
    hipLaunchKernelGGL(big_hist_kernel, dim3(n_chunks), dim3(kBlock), 0, stream, cur->as<Seg>(), n_segs, chunk_off.as<uint32_t>(), prim_box, centroid, in,
                       acc.as<BigAcc>());
    hipLaunchKernelGGL(big_decide_kernel, dim3(n_segs), dim3(64), 0, stream, cur->as<Seg>(), n_segs, acc.as<BigAcc>(), dec.as<BigDecision>(), L);
    hipLaunchKernelGGL(big_count_kernel, dim3(n_chunks), dim3(kBlock), 0, stream, cur->as<Seg>(), n_segs, chunk_off.as<uint32_t>(),
                       dec.as<BigDecision>(), centroid, in, chunk_left.as<uint32_t>());
    hipLaunchKernelGGL(big_offsets_kernel, dim3((n_segs + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, n_segs, chunk_off.as<uint32_t>(), chunk_left.as<uint32_t>());
    hipLaunchKernelGGL(big_scatter_kernel, dim3(n_chunks), dim3(kBlock), 0, stream, cur->as<Seg>(), n_segs, chunk_off.as<uint32_t>(),
                       dec.as<BigDecision>(), centroid, in, out, chunk_left.as<uint32_t>());
    RTB(hipMemsetAsync(counts.p, 0, 4, stream));  // the next level's list starts empty; the small list keeps growing
    hipLaunchKernelGGL(big_children_kernel, dim3((n_segs + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, cur->as<Seg>(), n_segs, dec.as<BigDecision>(),
                       out, buf ^ 1u, nxt->as<Seg>(), small.as<Seg>(), counts.as<uint32_t>(), L);
    hipLaunchKernelGGL(big_plan_kernel, dim3(1), dim3(kPlanThreads), 0, stream, nxt->as<Seg>(), chunk_off.as<uint32_t>(), counts.as<uint32_t>());
    RTB(hipGetLastError());
    uint32_t host_counts[4] = {0, 0, 0, 0};
    RTB(hipMemcpyAsync(host_counts, counts.p, 16, hipMemcpyDeviceToHost, stream));
    RTB(hipStreamSynchronize(stream));
    n_segs = host_counts[0];
    n_small = host_counts[1];
    n_chunks = host_counts[2];
    if (n_small > max_small) return hipErrorUnknown;
    std::swap(cur, nxt);
    buf ^= 1u;
  }
  if (n_small > 0) {
    hipLaunchKernelGGL(small_kernel, dim3(n_small), dim3(64), 0, stream, small.as<Seg>(), n_small, prim_box, centroid, idx0.as<uint32_t>(),
                       idx1.as<uint32_t>(), L);
    RTB(hipGetLastError());
  }
  RTB(hipStreamSynchronize(stream));  // the scratch arrays go out of scope
  return hipSuccess;
}

}  // namespace rta
