#include "bvh_builder.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <exception>
#include <system_error>
#include <cstdlib>
#include <thread>

namespace rta {
namespace {

const double kInf = std::numeric_limits<double>::infinity();
const double kLeafCost = 1.0;         // src/bvh.rs:87
const double kSplitCost = 1.0 / 8.0;  // src/bvh.rs:88
const int kBuckets = 32;              // src/bvh.rs:89
// Splitting stops at this level (the rest becomes one leaf): inputs such as spheres in geometric
// progression would otherwise recurse once per sphere.  No tree of a realistic scene comes near it.
const uint32_t kMaxBuildDepth = 64;

inline Box empty_box() {  // AABB::new, src/aabb.rs:11-16
  return Box{{kInf, kInf, kInf}, {-kInf, -kInf, -kInf}};
}
inline void grow(Box& a, const Box& b) {  // AABB::update / union, src/aabb.rs:36-38,48-53
  for (int k = 0; k < 3; ++k) {
    a.min[k] = std::fmin(a.min[k], b.min[k]);
    a.max[k] = std::fmax(a.max[k], b.max[k]);
  }
}
inline double surface_area(const Box& b) {  // src/aabb.rs:31-34
  double x = b.max[0] - b.min[0], y = b.max[1] - b.min[1], z = b.max[2] - b.min[2];
  return 2.0 * (x * y + x * z + z * y);
}
// Key that orders doubles as f64::total_cmp does (min_by at src/bvh.rs:227).
inline int64_t total_order_key(double v) {
  int64_t b;
  std::memcpy(&b, &v, 8);
  b ^= (int64_t)((uint64_t)(b >> 63) >> 1);
  return b;
}
// src/bvh.rs:250-258; the cast saturates and maps NaN to 0 as Rust's `as usize` does.
inline int choose_bucket(double min_t, double max_t, double t) {
  double f = (t - min_t) / (max_t - min_t);
  double scaled = (double)kBuckets * f;
  if (!(scaled > 0.0)) return 0;
  if (scaled >= (double)kBuckets) return kBuckets - 1;  // f == 1 for the largest centroid
  return (int)scaled;
}

// Work below this many spheres stays on the calling thread.
const size_t kParallelGrain = 16384;

// Threads a build may use: the machine's, 16 at most; RT_HOST_THREADS in the environment sets another limit (1 .. 64).
inline unsigned host_threads() {
  if (const char* env = std::getenv("RT_HOST_THREADS")) {
    const long v = std::strtol(env, nullptr, 10);
    if (v >= 1 && v <= 64) return (unsigned)v;
  }
  unsigned hw = std::thread::hardware_concurrency();
  return std::max(1u, std::min(hw ? hw : 1u, 16u));
}

// Runs f(part, begin, end) over [lo, hi) cut into `parts` contiguous pieces, one thread each; a piece whose
// thread cannot be created runs on the calling thread.  f must not throw.
template <class F>
void for_parts(size_t lo, size_t hi, unsigned parts, F f) {
  if (parts <= 1) {
    f(0u, lo, hi);
    return;
  }
  std::vector<std::thread> th;
  th.reserve(parts - 1);
  const size_t n = hi - lo;
  unsigned started = 1;  // pieces [1, started) have a thread
  for (; started < parts; ++started) {
    try {
      th.emplace_back(f, started, lo + n * started / parts, lo + n * (started + 1) / parts);
    } catch (const std::system_error&) {
      break;
    }
  }
  f(0u, lo, lo + n / parts);
  for (unsigned p = started; p < parts; ++p) f(p, lo + n * p / parts, lo + n * (p + 1) / parts);
  for (auto& t : th) t.join();
}

// A subtree in the reference's pre-order layout, possibly with holes: where a large node's
// children were built by two threads, their subtrees hang off the hole that follows the node and
// are spliced in once, at the end.  `total` counts the nodes including those in holes.
struct Subtree {
  struct Hole {
    size_t pos;  // the pieces go before nodes[pos]
    std::unique_ptr<Subtree> left, right;
  };
  std::vector<HostNode> nodes;
  std::vector<Hole> holes;
  size_t total = 0;
  uint32_t depth = 0;
  void flatten_into(std::vector<HostNode>& dst) const {
    size_t h = 0;
    for (size_t i = 0; i <= nodes.size(); ++i) {
      while (h < holes.size() && holes[h].pos == i) {
        holes[h].left->flatten_into(dst);
        holes[h].right->flatten_into(dst);
        ++h;
      }
      if (i < nodes.size()) dst.push_back(nodes[i]);
    }
  }
};

struct BucketSet {
  Box box[kBuckets];
  uint64_t count[kBuckets];
  void clear() {
    for (int b = 0; b < kBuckets; ++b) {
      box[b] = empty_box();
      count[b] = 0;
    }
  }
};

// The binned-SAH object-split builder of src/bvh.rs:92-266.  Large nodes are binned by several
// threads (box unions and counts are exact and order-independent, so the tree is the same as a
// serial build's) and large left subtrees are built by their own thread.
struct Builder {
  std::vector<Box> prim_box;            // Sphere::aabb, src/sphere.rs:50-52
  std::vector<double> centroid[3];      // AABB::centroid, src/aabb.rs:23-25
  std::vector<uint64_t> idx, tmp;
  size_t grain = kParallelGrain;
  unsigned max_threads = 1;

  void build(size_t lo, size_t hi, uint32_t depth, Subtree& out) {
    const size_t n = hi - lo;
    out.depth = std::max(out.depth, depth);
    const unsigned parts = n >= 2 * kParallelGrain ? (unsigned)std::min<size_t>(max_threads, n / kParallelGrain) : 1u;

    // Node box and bounds of the centroids (AABB::bounding_box, src/aabb.rs:40-46).
    struct Bounds {
      Box box;
      double cmin[3], cmax[3];
    };
    Bounds pb1;
    std::vector<Bounds> pbv(parts > 1 ? parts : 0);
    Bounds* pb = parts > 1 ? pbv.data() : &pb1;
    for_parts(lo, hi, parts, [&](unsigned p, size_t b0, size_t b1) {
      Bounds r{empty_box(), {kInf, kInf, kInf}, {-kInf, -kInf, -kInf}};
      for (size_t k = b0; k < b1; ++k) {
        const uint64_t i = idx[k];
        grow(r.box, prim_box[i]);
        for (int a = 0; a < 3; ++a) {
          const double c = centroid[a][i];
          r.cmin[a] = std::fmin(r.cmin[a], c);
          r.cmax[a] = std::fmax(r.cmax[a], c);
        }
      }
      pb[p] = r;
    });
    Box box = pb[0].box;
    double cmin[3], cmax[3];
    for (int a = 0; a < 3; ++a) cmin[a] = pb[0].cmin[a], cmax[a] = pb[0].cmax[a];
    for (unsigned p = 1; p < parts; ++p) {
      grow(box, pb[p].box);
      for (int a = 0; a < 3; ++a) {
        cmin[a] = std::fmin(cmin[a], pb[p].cmin[a]);
        cmax[a] = std::fmax(cmax[a], pb[p].cmax[a]);
      }
    }

    bool have = false;
    int best_axis = 0, best_split = 0;
    double best_cost = 0.0;
    const double node_area = surface_area(box);
    BucketSet sets1;
    std::vector<BucketSet> setsv(parts > 1 ? parts : 0);
    BucketSet* sets = parts > 1 ? setsv.data() : &sets1;
    for (int axis = 0; axis < 3; ++axis) {  // iter_bounds_axes, src/bvh.rs:196-201
      if (!((cmax[axis] - cmin[axis]) > 0.0)) continue;
      for_parts(lo, hi, parts, [&](unsigned p, size_t b0, size_t b1) {
        BucketSet& s = sets[p];
        s.clear();
        for (size_t k = b0; k < b1; ++k) {
          const uint64_t i = idx[k];
          int b = choose_bucket(cmin[axis], cmax[axis], centroid[axis][i]);
          s.count[b]++;
          grow(s.box[b], prim_box[i]);
        }
      });
      BucketSet& all = sets[0];
      for (unsigned p = 1; p < parts; ++p)
        for (int b = 0; b < kBuckets; ++b) {
          all.count[b] += sets[p].count[b];
          grow(all.box[b], sets[p].box[b]);
        }
      // Suffix unions, then one forward sweep: the same boxes and counts that
      // combine_buckets (src/bvh.rs:260-266) yields for every split point,
      // because min/max are exact and associative.
      Box suffix[kBuckets + 1];
      uint64_t suffix_n[kBuckets + 1];
      suffix[kBuckets] = empty_box();
      suffix_n[kBuckets] = 0;
      for (int b = kBuckets - 1; b >= 0; --b) {
        suffix[b] = all.box[b];
        grow(suffix[b], suffix[b + 1]);
        suffix_n[b] = all.count[b] + suffix_n[b + 1];
      }
      Box prefix = empty_box();
      uint64_t prefix_n = 0;
      for (int i = 1; i < kBuckets; ++i) {
        grow(prefix, all.box[i - 1]);
        prefix_n += all.count[i - 1];
        double cost = kSplitCost + kLeafCost * ((double)prefix_n * surface_area(prefix) + (double)suffix_n[i] * surface_area(suffix[i])) /
                                       node_area;  // src/bvh.rs:218-226
        if (!have || total_order_key(cost) < total_order_key(best_cost)) {
          have = true;
          best_axis = axis;
          best_split = i;
          best_cost = cost;
        }
      }
    }

    if (have && best_cost < kLeafCost * (double)n && depth < kMaxBuildDepth) {  // src/bvh.rs:124
      // Stable partition keeps both sides in ascending sphere order; tmp[lo, hi) is this node's scratch.
      size_t nl = 0, nr = 0;
      for (size_t k = lo; k < hi; ++k) {
        int b = choose_bucket(cmin[best_axis], cmax[best_axis], centroid[best_axis][idx[k]]);
        if (b < best_split)
          idx[lo + nl++] = idx[k];
        else
          tmp[lo + nr++] = idx[k];
      }
      std::copy(tmp.begin() + lo, tmp.begin() + lo + nr, idx.begin() + lo + nl);
      const size_t me = out.nodes.size(), me_total = out.total;
      out.nodes.push_back(HostNode{box, best_axis, 0, 0, 0});
      out.total++;
      if (nl >= grain && nr >= grain / 4) {
        Subtree::Hole hole{out.nodes.size(), std::make_unique<Subtree>(), std::make_unique<Subtree>()};
        Subtree &left = *hole.left, &right = *hole.right;
        // The left subtree on a thread of its own where one can be had.  An exception in either half (out of
        // memory) is carried to this thread and rethrown once both halves have stopped.
        std::exception_ptr left_error, right_error;
        auto build_left = [&]() {
          try {
            build(lo, lo + nl, depth + 1, left);
          } catch (...) {
            left_error = std::current_exception();
          }
        };
        std::thread worker;
        try {
          worker = std::thread(build_left);
        } catch (const std::system_error&) {
        }
        try {
          build(lo + nl, hi, depth + 1, right);
        } catch (...) {
          right_error = std::current_exception();
        }
        if (worker.joinable())
          worker.join();
        else if (!right_error)
          build_left();
        if (left_error) std::rethrow_exception(left_error);
        if (right_error) std::rethrow_exception(right_error);
        out.nodes[me].skip = left.total + 1;  // src/bvh.rs:126-130
        out.total += left.total + right.total;
        out.depth = std::max({out.depth, left.depth, right.depth});
        out.holes.push_back(std::move(hole));
      } else {
        build(lo, lo + nl, depth + 1, out);
        out.nodes[me].skip = out.total - me_total;
        build(lo + nl, hi, depth + 1, out);
      }
      return;
    }
    // Leaf (src/bvh.rs:108-110,137).  The leaves partition idx in pre-order, so the concatenated
    // leaf index list is idx itself and a leaf's offset into it is lo.
    out.nodes.push_back(HostNode{box, 3, 0, lo, n});
    out.total++;
  }
};

inline float round_down(double v) {
  float f = (float)v;
  if ((double)f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
  return f;
}
inline float round_up(double v) {
  float f = (float)v;
  if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
  return f;
}

}  // namespace

HostBvh build_bvh(const rt_sphere* spheres, size_t n) {
  Builder b;
  b.prim_box.resize(n);
  for (int a = 0; a < 3; ++a) b.centroid[a].resize(n);
  b.idx.resize(n);
  b.tmp.resize(n);
  for (size_t i = 0; i < n; ++i) {
    Box& bx = b.prim_box[i];
    for (int a = 0; a < 3; ++a) {
      double p0 = spheres[i].center[a] - spheres[i].radius, p1 = spheres[i].center[a] + spheres[i].radius;
      bx.min[a] = std::fmin(p0, p1);
      bx.max[a] = std::fmax(p0, p1);
      b.centroid[a][i] = bx.min[a] + 0.5 * (bx.max[a] - bx.min[a]);
    }
    b.idx[i] = i;
  }
  b.max_threads = host_threads();
  b.grain = std::max(kParallelGrain, n / 64);
  Subtree t;
  b.build(0, n, 1, t);
  HostBvh out;
  out.nodes.reserve(t.total);
  t.flatten_into(out.nodes);
  out.leaf_indices = std::move(b.idx);
  out.depth = t.depth;
  return out;
}

namespace {

// The host tree with leaves of more than kMaxLeafCount spheres cut into chains
// of smaller leaves under extra splits that keep the leaf's box.  (Such leaves
// only arise when many spheres share one centre.)  Spheres keep their order
// for rays with a non-negative x direction.
struct Work {
  struct N {
    Box box;
    int32_t kind;        // 0..2 split, 3 leaf
    int64_t left, right; // split: child indices in `nodes`
    uint64_t first, count;  // leaf: slot range
  };
  std::vector<N> nodes;
  int64_t add_leaf_range(const Box& box, uint64_t first, uint64_t count) {
    if (count <= kMaxLeafCount) {
      nodes.push_back(N{box, 3, -1, -1, first, count});
      return (int64_t)nodes.size() - 1;
    }
    uint64_t half = count / 2;
    int64_t me = (int64_t)nodes.size();
    nodes.push_back(N{box, 0, -1, -1, 0, 0});
    int64_t l = add_leaf_range(box, first, half);
    int64_t r = add_leaf_range(box, first + half, count - half);
    nodes[me].left = l;
    nodes[me].right = r;
    return me;
  }
};

}  // namespace

FlatBvh flatten_bvh(const HostBvh& bvh, const rt_sphere* spheres, size_t n) {
  FlatBvh f;
  const size_t nn = bvh.nodes.size();
  f.spheres.resize(n);
  f.aux.resize(n);

  // Slots follow the leaves in pre-order, so neighbouring leaves are neighbours in memory.
  Work w;
  std::vector<int64_t> work_of_host(nn, -1);
  {
    uint64_t slot = 0;
    // Pre-order array: children of a split at i are i+1 and i+skip; build bottom-up by walking backwards.
    for (size_t i = 0; i < nn; ++i) {
      const HostNode& h = bvh.nodes[i];
      if (h.kind != 3) continue;
      for (uint64_t k = 0; k < h.count; ++k) {
        uint64_t pi = bvh.leaf_indices[h.first + k];
        const rt_sphere& s = spheres[pi];
        f.spheres[slot + k] = DevSphere{s.center[0], s.center[1], s.center[2], s.radius};
        f.aux[slot + k] = DevSphereAux{1.0 / s.radius, s.material, (uint32_t)pi};
      }
      work_of_host[i] = w.add_leaf_range(h.box, slot, h.count);
      slot += h.count;
    }
    for (size_t i = nn; i-- > 0;) {
      const HostNode& h = bvh.nodes[i];
      if (h.kind == 3) continue;
      w.nodes.push_back(Work::N{h.box, h.kind, work_of_host[i + 1], work_of_host[i + h.skip], 0, 0});
      work_of_host[i] = (int64_t)w.nodes.size() - 1;
    }
  }
  if (nn == 0 || (bvh.nodes[0].kind == 3 && bvh.nodes[0].count == 0)) return f;  // empty scene: root stays kRefNone

  auto leaf_ref = [](const Work::N& x) { return kRefLeaf | ((uint32_t)x.count << 24) | (uint32_t)x.first; };
  const int64_t root = work_of_host[0];
  {
    const Box& b = w.nodes[root].box;
    double m = 0.0;
    for (int a = 0; a < 3; ++a) m = std::fmax(m, std::fmax(std::fabs(b.min[a]), std::fabs(b.max[a])));
    f.bound = round_up(m);
  }
  if (w.nodes[root].kind == 3) {
    f.root = leaf_ref(w.nodes[root]);
    f.depth = 1;
    return f;
  }

  // Breadth-first numbering of the inner nodes.
  std::vector<int64_t> order{root};
  std::vector<uint32_t> level{1};
  std::vector<uint32_t> new_of(w.nodes.size(), 0);
  for (size_t head = 0; head < order.size(); ++head) {
    const Work::N& x = w.nodes[order[head]];
    new_of[order[head]] = (uint32_t)head;
    f.depth = std::max(f.depth, level[head]);
    for (int64_t c : {x.left, x.right})
      if (w.nodes[c].kind != 3) {
        order.push_back(c);
        level.push_back(level[head] + 1);
      }
  }
  // new_of for nodes later in `order` is assigned when they are reached; fill all first.
  for (size_t k = 0; k < order.size(); ++k) new_of[order[k]] = (uint32_t)k;
  f.nodes.resize(order.size());
  for (size_t k = 0; k < order.size(); ++k) {
    const Work::N& x = w.nodes[order[k]];
    DevNode d{};
    const Work::N* ch[2] = {&w.nodes[x.left], &w.nodes[x.right]};
    for (int c = 0; c < 2; ++c) {
      d.x[2 * c] = round_down(ch[c]->box.min[0]);
      d.x[2 * c + 1] = round_up(ch[c]->box.max[0]);
      d.y[2 * c] = round_down(ch[c]->box.min[1]);
      d.y[2 * c + 1] = round_up(ch[c]->box.max[1]);
      d.z[2 * c] = round_down(ch[c]->box.min[2]);
      d.z[2 * c + 1] = round_up(ch[c]->box.max[2]);
    }
    d.ref0 = ch[0]->kind == 3 ? leaf_ref(*ch[0]) : new_of[x.left];
    d.ref1 = ch[1]->kind == 3 ? leaf_ref(*ch[1]) : new_of[x.right];
    d.axis = (uint32_t)x.kind;
    f.nodes[k] = d;
  }
  f.root = 0;
  return f;
}

WideBvh collapse_wide(const DevNode* nodes, size_t n_nodes, uint32_t root) {
  WideBvh w;
  if (n_nodes == 0 || root >= n_nodes) return w;
  struct Child {
    float lo[3], hi[3];
    uint32_t ref;
  };
  auto children_of = [&](uint32_t b, Child out[2]) {
    const DevNode& d = nodes[b];
    for (int c = 0; c < 2; ++c) {
      out[c].lo[0] = d.x[2 * c];
      out[c].hi[0] = d.x[2 * c + 1];
      out[c].lo[1] = d.y[2 * c];
      out[c].hi[1] = d.y[2 * c + 1];
      out[c].lo[2] = d.z[2 * c];
      out[c].hi[2] = d.z[2 * c + 1];
      out[c].ref = c == 0 ? d.ref0 : d.ref1;
    }
  };
  auto area = [](const Child& c) {
    const double dx = (double)c.hi[0] - c.lo[0], dy = (double)c.hi[1] - c.lo[1], dz = (double)c.hi[2] - c.lo[2];
    return dx * dy + dy * dz + dz * dx;
  };
  // breadth-first over the wide nodes; each is rooted at a two-wide inner node
  std::vector<uint32_t> root_of{root};
  std::vector<uint32_t> level{1};
  w.nodes.reserve(n_nodes / 2 + 1);
  double bound = 0.0;
  for (size_t head = 0; head < root_of.size(); ++head) {
    Child ch[4];
    int n = 2;
    children_of(root_of[head], ch);
    while (n < 4) {
      int pick = -1;
      double best = -1.0;
      for (int k = 0; k < n; ++k)
        if (!(ch[k].ref & kRefLeaf) && area(ch[k]) > best) {
          best = area(ch[k]);
          pick = k;
        }
      if (pick < 0) break;
      Child two[2];
      children_of(ch[pick].ref, two);
      ch[pick] = two[0];
      ch[n++] = two[1];
    }
    WideNode node{};
    for (int a = 0; a < 3; ++a)
      for (int k = 0; k < 4; ++k) {
        node.planes[2 * a][k] = k < n ? ch[k].lo[a] : std::numeric_limits<float>::infinity();
        node.planes[2 * a + 1][k] = k < n ? ch[k].hi[a] : -std::numeric_limits<float>::infinity();
        if (k < n) bound = std::fmax(bound, std::fmax(std::fabs((double)ch[k].lo[a]), std::fabs((double)ch[k].hi[a])));
      }
    for (int k = 0; k < 4; ++k) {
      if (k >= n) {
        node.ref[k] = kRefNone;
      } else if (ch[k].ref & kRefLeaf) {
        node.ref[k] = ch[k].ref;
      } else {
        node.ref[k] = (uint32_t)root_of.size();
        root_of.push_back(ch[k].ref);
        level.push_back(level[head] + 1);
      }
    }
    w.depth = std::max(w.depth, level[head]);
    w.nodes.push_back(node);
  }
  w.stack_entries = 3 * w.depth + 1;  // at most three children set aside per level
  w.bound = round_up(bound);
  return w;
}

}  // namespace rta
