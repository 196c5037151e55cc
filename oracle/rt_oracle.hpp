// CPU oracle: an f64 restatement of the render hot path of rfsmi/raytracer.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library may include,
// link or call this file; it exists so that tests, the smoke check and the
// CPU-baseline leg of the benchmark have an independent statement of what
// the reference computes.  Every function cites the reference source it
// follows (paths are relative to the reference checkout, `src/...`).
//
// Parity status: the slab test, the AABB algebra and Sphere::aabb /
// clipped_aabb are pinned by the reference's own unit tests
// (src/aabb.rs:89-194, src/sphere.rs:83-117).  The camera, get_ray, the sky
// branch and the recursion of ray_colour, Sphere::hit (on the ground sphere and
// on the unit spheres), HitRecord::new with its front-face rule, the Lambertian
// scatter, the Metal scatter without fuzz (with vector::reflect), the refracting
// branch of the Dielectric scatter (with vector::refract), the lens sampler and
// write_colour are pinned by the example render the reference keeps in its
// repository (image.png): the sky and the sky mirrored in the large metal sphere
// byte for byte; the far ground and the sky seen through the large glass sphere
// by their mean colours (the latter tells an index of refraction of 1.5 from 1.45);
// the lens by the profile of the smeared horizon (tests/test_reference_image.py).
// BVH::new and BVH::hit beyond the four fixed spheres, the fuzz of the metal
// scatter and Schlick's reflectance (which weighs a few per cent in the glass
// region) have no test, fixture or output in the reference that reaches them,
// and the reference cannot be built here (no Rust toolchain), so for those
// parity is UNPINNED: the restatement is reviewable line against line and is
// exercised by analytic known-answer tests only.
//
// Arithmetic rules taken from the reference's dependencies (glam 0.24.2,
// rand 0.8.5; versions from Cargo.lock): DVec3 is three scalar f64; dot is
// x*x'+y*y'+z*z' evaluated left to right; normalize multiplies by 1/length;
// min/max are IEEE minNum/maxNum (fmin/fmax); rustc never contracts a*b+c,
// so this file must be compiled with -ffp-contract=off.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace orc {

// ---------------------------------------------------------------- DVec3 ----
struct V3 {
  double x, y, z;
};
static inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
static inline V3 splat(double s) { return V3{s, s, s}; }
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
static inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
static inline V3 operator/(double s, V3 a) { return {s / a.x, s / a.y, s / a.z}; }
static inline V3 operator+(V3 a, double s) { return {a.x + s, a.y + s, a.z + s}; }
static inline V3 operator-(V3 a, double s) { return {a.x - s, a.y - s, a.z - s}; }
static inline V3 operator-(double s, V3 a) { return {s - a.x, s - a.y, s - a.z}; }
static inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline double length_squared(V3 a) { return dot(a, a); }
static inline double length(V3 a) { return std::sqrt(dot(a, a)); }
static inline V3 normalize(V3 a) { return a * (1.0 / length(a)); }
static inline V3 cross(V3 a, V3 b) {
  return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
static inline V3 vmin(V3 a, V3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
static inline V3 vmax(V3 a, V3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }
static inline double min_element(V3 a) { return std::fmin(a.x, std::fmin(a.y, a.z)); }
static inline double max_element(V3 a) { return std::fmax(a.x, std::fmax(a.y, a.z)); }
static inline bool operator==(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
static inline double get_axis(V3 v, int axis) { return axis == 0 ? v.x : axis == 1 ? v.y : v.z; }  // src/bvh.rs:182-194
static inline V3 axis_vec(int axis) { return {axis == 0 ? 1.0 : 0.0, axis == 1 ? 1.0 : 0.0, axis == 2 ? 1.0 : 0.0}; }

static const double INF = std::numeric_limits<double>::infinity();

// ------------------------------------------------------- Ray / Interval ----
// src/ray.rs:9-13,30-41
struct Interval {
  double min, max;
  bool surrounds(double x) const { return min < x && x < max; }              // src/ray.rs:35-37 (strict)
  double clamp(double x) const { return std::fmin(std::fmax(x, min), max); }  // src/ray.rs:39-41
};

// src/ray.rs:3-7,16-27.  Ray::new normalises the direction and stores 1/direction.
struct Ray {
  V3 origin, direction, inv_direction;
  V3 at(double t) const { return origin + t * direction; }  // src/ray.rs:25-27
};
static inline Ray ray_new(V3 origin, V3 direction) {
  V3 d = normalize(direction);
  return Ray{origin, d, 1.0 / d};
}

// ----------------------------------------------------------------- AABB ----
// src/aabb.rs:5-83
struct AABB {
  V3 min, max;
};
static inline AABB aabb_new() { return {splat(INF), splat(-INF)}; }                      // src/aabb.rs:11-16
static inline V3 aabb_size(const AABB& b) { return b.max - b.min; }                       // src/aabb.rs:27-29
static inline bool aabb_is_empty(const AABB& b) { return min_element(aabb_size(b)) < 0.0; }  // src/aabb.rs:19-21
static inline V3 aabb_centroid(const AABB& b) { return b.min + 0.5 * (b.max - b.min); }   // src/aabb.rs:23-25
static inline double aabb_surface_area(const AABB& b) {                                   // src/aabb.rs:31-34
  V3 s = aabb_size(b);
  return 2.0 * (s.x * s.y + s.x * s.z + s.z * s.y);
}
static inline AABB aabb_union2(const AABB& a, const AABB& b) {  // fold step of src/aabb.rs:48-53
  return {vmin(a.min, b.min), vmax(a.max, b.max)};
}
static inline AABB aabb_union(const AABB* boxes, size_t n) {  // src/aabb.rs:48-53
  AABB acc = aabb_new();
  for (size_t i = 0; i < n; ++i) acc = aabb_union2(acc, boxes[i]);
  return acc;
}
static inline AABB aabb_bounding_box(const V3* pts, size_t n) {  // src/aabb.rs:40-46
  if (n == 0) return aabb_new();
  V3 lo = pts[0], hi = pts[0];
  for (size_t i = 1; i < n; ++i) {
    lo = vmin(lo, pts[i]);
    hi = vmax(hi, pts[i]);
  }
  return {lo, hi};
}
static inline AABB aabb_intersection(const AABB* boxes, size_t n) {  // src/aabb.rs:56-67
  if (n == 0) return AABB{splat(-INF), splat(INF)};
  AABB acc = boxes[0];
  for (size_t i = 1; i < n; ++i) acc = AABB{vmax(acc.min, boxes[i].min), vmin(acc.max, boxes[i].max)};
  return acc;
}
// src/aabb.rs:69-82.  Returns whether the reference returns Some(..); the
// payload is never read by BVH::hit (src/bvh.rs:58), so only the flag is kept.
static inline bool aabb_ray_intersection(const AABB& b, const Ray& r, Interval ray_t) {
  if (aabb_is_empty(b)) return false;
  V3 t1 = (b.min - r.origin) * r.inv_direction;
  V3 t2 = (b.max - r.origin) * r.inv_direction;
  V3 lo = vmin(t1, t2), hi = vmax(t2, t1);
  double tmin = std::fmax(std::fmax(lo.x, lo.y), lo.z);
  double tmax = std::fmin(std::fmin(hi.x, hi.y), hi.z);
  if (tmax < tmin || tmax < ray_t.min || ray_t.max < tmin) return false;
  return true;
}

// ------------------------------------------------------------ Materials ----
enum MatType : uint32_t { LAMBERTIAN = 0, METAL = 1, DIELECTRIC = 2 };
// Flat form of src/material.rs:10-19 (one record per Arc<dyn Material>).
struct Material {
  uint32_t type;
  uint32_t pad;
  double albedo[3];
  double param;  // Metal: fuzz, Dielectric: ir, Lambertian: unused
};
// Flat form of src/sphere.rs:12-16.
struct SphereIn {
  double c[3];
  double r;
  uint32_t mat;
  uint32_t pad;
};

// src/hit.rs:9-27
struct HitRecord {
  bool front_face;
  uint32_t material;
  int32_t prim;
  V3 normal, p;
  double t;
};
static inline HitRecord hit_record_new(const Ray& r, V3 p, double t, V3 out_normal, uint32_t material, int32_t prim) {
  bool front_face = dot(r.direction, out_normal) < 0.0;  // src/hit.rs:19
  return HitRecord{front_face, material, prim, (front_face ? 1.0 : -1.0) * out_normal, p, t};
}

// --------------------------------------------------------------- Sphere ----
struct Sphere {
  V3 center;
  double radius;
  uint32_t material;
};
// src/sphere.rs:29-48
static inline bool sphere_hit(const Sphere& s, int32_t prim, const Ray& r, Interval ray_t, HitRecord& out) {
  V3 oc = r.origin - s.center;
  double a = length_squared(r.direction);
  double half_b = dot(oc, r.direction);
  double c = length_squared(oc) - s.radius * s.radius;
  double discriminant = half_b * half_b - a * c;
  if (discriminant < 0.0) return false;
  const double f[2] = {-1.0, 1.0};
  for (int k = 0; k < 2; ++k) {
    double t = (-half_b + f[k] * std::sqrt(discriminant)) / a;
    if (ray_t.surrounds(t)) {
      V3 p = r.at(t);
      V3 outward_normal = (p - s.center) / s.radius;
      out = hit_record_new(r, p, t, outward_normal, s.material, prim);
      return true;
    }
  }
  return false;
}
// src/sphere.rs:50-52
static inline AABB sphere_aabb(const Sphere& s) {
  V3 pts[2] = {s.center - s.radius, s.center + s.radius};
  return aabb_bounding_box(pts, 2);
}
// src/sphere.rs:54-74 (only reachable from the unfinished spatial-split
// builder; kept because the reference's own test pins it).
static inline AABB sphere_clipped_aabb(const Sphere& s, int axis, double t1, double t2) {
  V3 ax = axis_vec(axis);
  V3 d1 = (t1 - s.center) * ax;
  V3 d2 = (t2 - s.center) * ax;
  double h = std::sqrt(s.radius * s.radius - length_squared(vmin(d1, d2)));
  if (std::isnan(h)) h = -s.radius;
  V3 other_axis = splat(1.0) - ax;
  AABB box{s.center + d1 - other_axis * h, s.center + d2 + other_axis * h};
  if (min_element(d1) < 0.0 && 0.0 < max_element(d2)) {
    V3 pts[2] = {s.center - other_axis * s.radius, s.center + other_axis * s.radius};
    box = aabb_union2(box, aabb_bounding_box(pts, 2));
  }
  AABB both[2] = {sphere_aabb(s), box};
  return aabb_intersection(both, 2);
}

// ------------------------------------------------------------------ BVH ----
// src/bvh.rs:17-27.  Pre-order array; a Split at i has its left child at i+1
// and its right child at i+skip.
struct Node {
  AABB aabb;
  bool leaf;
  int axis;      // Split only: 0,1,2 for DVec3::X/Y/Z
  size_t skip;   // Split only
  std::vector<size_t> indices;  // Leaf only
};

static const double LEAF_COST = 1.0;        // src/bvh.rs:87
static const double SPLIT_COST = 1.0 / 8.0;  // src/bvh.rs:88
static const size_t N_BUCKETS = 32;          // src/bvh.rs:89

// f64::total_cmp as a signed-integer key (used by min_by at src/bvh.rs:227).
static inline int64_t total_order_key(double v) {
  int64_t b;
  std::memcpy(&b, &v, 8);
  b ^= (int64_t)((uint64_t)(b >> 63) >> 1);
  return b;
}

struct Bucket {  // src/bvh.rs:230-248; the HashSet is replaced by an index list kept ascending
  std::vector<size_t> primitives;
  AABB aabb = aabb_new();
};

// src/bvh.rs:250-258
static inline size_t choose_bucket(size_t n_buckets, double min_t, double max_t, double t) {
  double f = (t - min_t) / (max_t - min_t);
  double scaled = (double)n_buckets * f;
  // Rust `as usize` saturates and maps NaN to 0.
  size_t i = std::isnan(scaled) ? 0 : scaled <= 0.0 ? 0 : scaled >= 1.8e19 ? SIZE_MAX : (size_t)scaled;
  return i == n_buckets ? i - 1 : i;
}

struct BVH {
  std::vector<Node> nodes;
  std::vector<Sphere> primitives;
};

struct Builder {
  const std::vector<Sphere>& prims;

  AABB prim_aabb(size_t i) const { return sphere_aabb(prims[i]); }

  // src/bvh.rs:105-138 with the spatial-split attempt (:112-120) left out.
  // In the reference a spatial-split candidate's right side always holds every
  // primitive of the node (src/bvh.rs:169-175 adds each primitive to all
  // buckets from its first to the last), so choosing one would recurse on an
  // identical node for ever: whenever the reference terminates it has used
  // object splits only, which is what this builder does.
  void build(const std::vector<size_t>& indices, std::vector<Node>& out) const {
    AABB aabb = aabb_new();
    for (size_t i : indices) aabb = aabb_union2(aabb, prim_aabb(i));

    // object_buckets, src/bvh.rs:140-156
    std::vector<V3> cents(indices.size());
    for (size_t k = 0; k < indices.size(); ++k) cents[k] = aabb_centroid(prim_aabb(indices[k]));
    AABB bounds = aabb_bounding_box(cents.data(), cents.size());

    bool have = false;
    int best_axis = 0;
    size_t best_i = 0;
    double best_cost = 0.0;
    std::vector<Bucket> best_buckets;
    for (int axis = 0; axis < 3; ++axis) {  // iter_bounds_axes, src/bvh.rs:196-201
      double min_t = get_axis(bounds.min, axis), max_t = get_axis(bounds.max, axis);
      if (!((max_t - min_t) > 0.0)) continue;
      std::vector<Bucket> buckets(N_BUCKETS);
      for (size_t k = 0; k < indices.size(); ++k) {
        double t = get_axis(cents[k], axis);
        size_t bi = choose_bucket(N_BUCKETS, min_t, max_t, t);
        buckets[bi].primitives.push_back(indices[k]);
        buckets[bi].aabb = aabb_union2(buckets[bi].aabb, prim_aabb(indices[k]));
      }
      // best_split, src/bvh.rs:203-228: every split point, first minimum wins.
      for (size_t i = 1; i < N_BUCKETS; ++i) {
        size_t na = 0, nb = 0;
        AABB a = aabb_new(), b = aabb_new();
        for (size_t k = 0; k < i; ++k) {  // combine_buckets, src/bvh.rs:260-266
          na += buckets[k].primitives.size();
          a = aabb_union2(a, buckets[k].aabb);
        }
        for (size_t k = i; k < N_BUCKETS; ++k) {
          nb += buckets[k].primitives.size();
          b = aabb_union2(b, buckets[k].aabb);
        }
        double cost = SPLIT_COST +
                      LEAF_COST * ((double)na * aabb_surface_area(a) + (double)nb * aabb_surface_area(b)) /
                          aabb_surface_area(aabb);
        if (!have || total_order_key(cost) < total_order_key(best_cost)) {
          have = true;
          best_axis = axis;
          best_i = i;
          best_cost = cost;
          best_buckets = buckets;
        }
      }
    }
    if (!have) {  // src/bvh.rs:108-110
      out.push_back(Node{aabb, true, 0, 0, indices});
      return;
    }
    if (best_cost < LEAF_COST * (double)indices.size()) {  // src/bvh.rs:124
      std::vector<size_t> lhs, rhs;
      for (size_t k = 0; k < best_i; ++k) lhs.insert(lhs.end(), best_buckets[k].primitives.begin(), best_buckets[k].primitives.end());
      for (size_t k = best_i; k < N_BUCKETS; ++k) rhs.insert(rhs.end(), best_buckets[k].primitives.begin(), best_buckets[k].primitives.end());
      // The reference's sets are unordered; ascending order is this oracle's choice.
      std::sort(lhs.begin(), lhs.end());
      std::sort(rhs.begin(), rhs.end());
      size_t parent = out.size();
      out.push_back(Node{aabb, false, best_axis, 0, {}});
      build(lhs, out);
      out[parent].skip = out.size() - parent;  // = len(left subtree) + 1, src/bvh.rs:126-130
      build(rhs, out);
      return;
    }
    out.push_back(Node{aabb, true, 0, 0, indices});  // src/bvh.rs:137
  }
};

// src/bvh.rs:44-48
static inline BVH bvh_new(const std::vector<Sphere>& prims) {
  BVH b;
  b.primitives = prims;
  std::vector<size_t> all(prims.size());
  for (size_t i = 0; i < prims.size(); ++i) all[i] = i;
  Builder{b.primitives}.build(all, b.nodes);
  return b;
}

struct TraceStats {
  uint64_t rays = 0, node_visits = 0, sphere_tests = 0;
};

// src/bvh.rs:50-84
static inline bool bvh_hit(const BVH& bvh, const Ray& r, Interval ray_t, HitRecord& best, TraceStats* st = nullptr) {
  if (st) st->rays++;
  if (bvh.nodes.empty()) return false;
  // The reference returns None for an empty primitive list because the root
  // leaf's box is AABB::new(), which is_empty(); the same happens here.
  size_t stack[256];
  int sp = 0;
  stack[sp++] = 0;
  bool found = false;
  while (sp > 0) {
    size_t i = stack[--sp];
    const Node& node = bvh.nodes[i];
    if (st) st->node_visits++;
    if (!aabb_ray_intersection(node.aabb, r, ray_t)) continue;
    if (!node.leaf) {
      if (get_axis(r.direction, node.axis) < 0.0) {
        stack[sp++] = i + 1;
        stack[sp++] = i + node.skip;
      } else {
        stack[sp++] = i + node.skip;
        stack[sp++] = i + 1;
      }
    } else {
      for (size_t pi : node.indices) {
        if (st) st->sphere_tests++;
        HitRecord hr{};
        if (sphere_hit(bvh.primitives[pi], (int32_t)pi, r, ray_t, hr)) {
          ray_t.max = hr.t;
          best = hr;
          found = true;
        }
      }
    }
  }
  return found;
}

// --------------------------------------------------------------- Philox ----
// Philox4x32 (Salmon et al., SC'11).  The reference draws from an
// unseeded thread-local ChaCha12 (src/vector.rs:10-12, src/camera.rs:115-116,
// src/material.rs:58) and is not reproducible; a counter-based generator
// with the same output distributions takes its place.  Ten rounds are the
// published default (scene generation, the sequential stream of the
// rejection samplers); the draws of the render loop in direct mode use seven,
// the fewest the authors found to pass BigCrush, as the GPU path does.
static inline void philox4x32(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4], int rounds) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < rounds; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline void philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) { philox4x32(key, ctr, out, 10); }
static const int RENDER_ROUNDS = 7;
// rand 0.8.5 Standard for f64: 53 random bits scaled by 2^-53, in [0,1).
static inline double u64_to_unit(uint32_t lo, uint32_t hi) {
  uint64_t v = ((uint64_t)hi << 32) | lo;
  return (double)(v >> 11) * (1.0 / 9007199254740992.0);
}

// Counter layout: (pixel index, sample index, slot, domain).
enum Domain : uint32_t { DOM_CAMERA = 0, DOM_SCATTER = 1, DOM_STREAM = 2, DOM_SCENE = 3 };
static inline void philox_pair(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t slot, uint32_t domain, double u[2]) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t ctr[4] = {pixel, sample, slot, domain};
  uint32_t o[4];
  philox4x32_10(key, ctr, o);
  u[0] = u64_to_unit(o[0], o[1]);
  u[1] = u64_to_unit(o[2], o[3]);
}
// The render loop's draws in direct mode: two uniforms of 53 bits per scatter event ...
static inline void render_pair(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t slot, uint32_t domain, double u[2]) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t ctr[4] = {pixel, sample, slot, domain};
  uint32_t o[4];
  philox4x32(key, ctr, o, RENDER_ROUNDS);
  u[0] = u64_to_unit(o[0], o[1]);
  u[1] = u64_to_unit(o[2], o[3]);
}
// ... and four of 32 bits (word * 2^-32) per camera ray: the position in the pixel and the point on the lens.
static inline void render_quad(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t slot, uint32_t domain, double u[4]) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t ctr[4] = {pixel, sample, slot, domain};
  uint32_t o[4];
  philox4x32(key, ctr, o, RENDER_ROUNDS);
  for (int k = 0; k < 4; ++k) u[k] = (double)o[k] * (1.0 / 4294967296.0);
}

// ------------------------------------------------ vector.rs helpers --------
static inline bool near_zero(V3 v) {  // src/vector.rs:52-54 (abs_diff_eq with 1e-6)
  return std::fabs(v.x) <= 1e-6 && std::fabs(v.y) <= 1e-6 && std::fabs(v.z) <= 1e-6;
}
static inline V3 reflect(V3 v, V3 n) { return v - 2.0 * dot(v, n) * n; }  // src/vector.rs:56-58
static inline V3 refract(V3 v, V3 n, double etai_over_etat) {             // src/vector.rs:60-65
  double cos_theta = std::fmin(dot(-v, n), 1.0);
  V3 r_out_perp = etai_over_etat * (v + cos_theta * n);
  V3 r_out_parallel = -std::sqrt(std::fabs(1.0 - length_squared(r_out_perp))) * n;
  return r_out_perp + r_out_parallel;
}

// Two sources of randomness with the same distributions.
//
// SAMPLER_DIRECT: fixed draw budget per event; the unit vector and the disk
// point are direct maps of two uniforms (uniform on the sphere: z = 1-2u1,
// phi = 2 pi u2; uniform in the disk: r = sqrt(u1), phi = 2 pi u2).  This is
// what the GPU path evaluates, so it is the mode used for parity.
//
// SAMPLER_REJECTION: the reference's own rejection loops (src/vector.rs:16-40)
// fed from a sequential Philox stream per (pixel, sample); used to show that
// the direct maps leave the image statistics unchanged.
enum SamplerMode : int { SAMPLER_DIRECT = 0, SAMPLER_REJECTION = 1 };

struct PathRng {
  uint64_t seed;
  uint32_t pixel, sample;
  int mode;
  uint32_t stream_slot = 0;  // rejection mode: next block of the sequential stream
  double buf[2] = {0.0, 0.0};
  int buf_left = 0;

  PathRng(uint64_t seed_, uint32_t pixel_, uint32_t sample_, int mode_) : seed(seed_), pixel(pixel_), sample(sample_), mode(mode_) {}

  double stream_next() {  // sequential stream, domain DOM_STREAM
    if (buf_left == 0) {
      philox_pair(seed, pixel, sample, stream_slot++, DOM_STREAM, buf);
      buf_left = 2;
    }
    return buf[2 - buf_left--];
  }
};

static inline V3 unit_from_uniforms(double u1, double u2) {
  double z = 1.0 - 2.0 * u1;
  double rxy = std::sqrt(std::fmax(0.0, 1.0 - z * z));
  double phi = 2.0 * M_PI * u2;
  return v3(rxy * std::cos(phi), rxy * std::sin(phi), z);
}
static inline V3 disk_from_uniforms(double u1, double u2) {
  double r = std::sqrt(u1);
  double phi = 2.0 * M_PI * u2;
  return v3(r * std::cos(phi), r * std::sin(phi), 0.0);
}
// src/vector.rs:20-27 / :16-18
static inline V3 random_unit_rejection(PathRng& g) {
  for (;;) {
    double x = -1.0 + 2.0 * g.stream_next();
    double y = -1.0 + 2.0 * g.stream_next();
    double z = -1.0 + 2.0 * g.stream_next();
    V3 p = v3(x, y, z);
    if (length_squared(p) < 1.0) return normalize(p);
  }
}
// src/vector.rs:29-40
static inline V3 random_disk_rejection(PathRng& g) {
  for (;;) {
    double x = -1.0 + 2.0 * g.stream_next();
    double y = -1.0 + 2.0 * g.stream_next();
    V3 p = v3(x, y, 0.0);
    if (length_squared(p) < 1.0) return p;
  }
}

// What one scatter event may consume.
struct ScatterDraw {
  V3 unit;      // value of vector::random_unit() for this event
  double coin;  // value of rand::random::<f64>() for this event
};

// src/material.rs:21-65.  Returns false where the reference returns None.
static inline bool scatter(const Material& m, const Ray& r, const HitRecord& hr, const ScatterDraw& d, V3& attenuation,
                           Ray& scattered) {
  switch (m.type) {
    case LAMBERTIAN: {  // src/material.rs:21-30
      V3 dir = hr.normal + d.unit;
      if (near_zero(dir)) dir = hr.normal;
      attenuation = v3(m.albedo[0], m.albedo[1], m.albedo[2]);
      scattered = ray_new(hr.p, dir);  // HitRecord::ray, src/hit.rs:29-31
      return true;
    }
    case METAL: {  // src/material.rs:32-41
      V3 reflected = reflect(normalize(r.direction), hr.normal) + m.param * d.unit;
      if (dot(reflected, hr.normal) <= 0.0) return false;
      attenuation = v3(m.albedo[0], m.albedo[1], m.albedo[2]);
      scattered = ray_new(hr.p, reflected);
      return true;
    }
    default: {  // Dielectric, src/material.rs:43-65
      double ir = m.param;
      double refraction_ratio = hr.front_face ? 1.0 / ir : ir;  // ir.powi(-1) / ir.powi(1)
      V3 unit_direction = normalize(r.direction);
      double cos_theta = std::fmin(dot(-unit_direction, hr.normal), 1.0);
      double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
      bool cannot_refract = refraction_ratio * sin_theta > 1.0;
      double r0 = (1.0 - refraction_ratio) / (1.0 + refraction_ratio);
      r0 = r0 * r0;
      double one_minus = 1.0 - cos_theta;
      // powi(5): LLVM expands it as x^4 * x with x^4 = (x^2)^2.
      double x2 = one_minus * one_minus;
      double p5 = x2 * x2 * one_minus;
      double reflectance = r0 + (1.0 - r0) * p5;
      V3 direction = (cannot_refract || reflectance > d.coin) ? reflect(unit_direction, hr.normal)
                                                              : refract(unit_direction, hr.normal, refraction_ratio);
      attenuation = splat(1.0);
      scattered = ray_new(hr.p, direction);
      return true;
    }
  }
}

// --------------------------------------------------------------- Camera ----
// src/camera.rs:15-26 (defaults are applied by the caller)
struct Config {
  double aspect_ratio;
  uint32_t image_width;
  uint32_t pad;
  uint64_t samples_per_pixel;
  uint64_t max_depth;
  double vfov;
  double lookfrom[3], lookat[3], vup[3];
  double defocus_angle;
  double focus_dist;
};
// src/camera.rs:28-37
struct Camera {
  Config config;
  uint32_t image_height;
  V3 center, pixel00_loc, pixel_delta_u, pixel_delta_v, defocus_disk_u, defocus_disk_v;
};
static inline double to_radians(double deg) { return deg * (M_PI / 180.0); }  // f64::to_radians
// src/camera.rs:40-82
static inline Camera config_camera(const Config& c) {
  double hf = std::fmax((double)c.image_width / c.aspect_ratio, 1.0);
  uint32_t image_height = hf >= 4294967295.0 ? 4294967295u : (uint32_t)hf;
  V3 lookfrom = v3(c.lookfrom[0], c.lookfrom[1], c.lookfrom[2]);
  V3 lookat = v3(c.lookat[0], c.lookat[1], c.lookat[2]);
  V3 vup = v3(c.vup[0], c.vup[1], c.vup[2]);
  V3 center = lookfrom;
  double theta = to_radians(c.vfov);
  double h = std::tan(theta / 2.0);
  double viewport_height = 2.0 * h * c.focus_dist;
  double viewport_width = viewport_height * (double)c.image_width / (double)image_height;
  V3 w = normalize(lookfrom - lookat);
  V3 u = normalize(cross(vup, w));
  V3 v = cross(w, u);
  V3 viewport_u = viewport_width * u;
  V3 viewport_v = viewport_height * (-v);
  V3 pixel_delta_u = viewport_u / (double)c.image_width;
  V3 pixel_delta_v = viewport_v / (double)image_height;
  V3 viewport_upper_left = center - c.focus_dist * w - (viewport_u + viewport_v) / 2.0;
  V3 pixel00_loc = viewport_upper_left + (pixel_delta_u + pixel_delta_v) / 2.0;
  double defocus_radius = c.focus_dist * std::tan(to_radians(c.defocus_angle / 2.0));
  V3 defocus_disk_u = u * defocus_radius;
  V3 defocus_disk_v = v * defocus_radius;
  return Camera{c, image_height, center, pixel00_loc, pixel_delta_u, pixel_delta_v, defocus_disk_u, defocus_disk_v};
}

// src/camera.rs:114-136
static inline Ray get_ray(const Camera& cam, uint32_t i, uint32_t j, PathRng& g) {
  V3 pixel_center = cam.pixel00_loc + ((double)i * cam.pixel_delta_u) + ((double)j * cam.pixel_delta_v);
  double u[4] = {0.0, 0.0, 0.0, 0.0};
  if (g.mode == SAMPLER_DIRECT) {
    render_quad(g.seed, g.pixel, g.sample, 0, DOM_CAMERA, u);  // one block per camera ray: pixel position, lens point
  } else {
    u[0] = g.stream_next();
    u[1] = g.stream_next();
  }
  double px = -0.5 + u[0];
  double py = -0.5 + u[1];
  V3 pixel_sample = pixel_center + ((px * cam.pixel_delta_u) + (py * cam.pixel_delta_v));
  V3 ray_origin;
  if (cam.config.defocus_angle <= 0.0) {
    ray_origin = cam.center;
  } else {
    V3 p = g.mode == SAMPLER_DIRECT ? disk_from_uniforms(u[2], u[3]) : random_disk_rejection(g);
    ray_origin = cam.center + p.x * cam.defocus_disk_u + p.y * cam.defocus_disk_v;
  }
  V3 ray_direction = pixel_sample - ray_origin;
  return ray_new(ray_origin, ray_direction);
}

struct Scene {
  BVH bvh;
  std::vector<Material> materials;
};

static inline ScatterDraw draw_for_bounce(PathRng& g, const Material& m, const HitRecord& hr, const Ray& r, uint32_t bounce) {
  ScatterDraw d{v3(0, 0, 0), 0.0};
  if (g.mode == SAMPLER_DIRECT) {
    double u[2];
    render_pair(g.seed, g.pixel, g.sample, bounce, DOM_SCATTER, u);
    d.unit = unit_from_uniforms(u[0], u[1]);
    d.coin = u[0];
  } else {
    // Draw exactly what the reference would draw for this material.
    if (m.type == DIELECTRIC) {
      // `cannot_refract || reflectance > random()` short-circuits (src/material.rs:58)
      double ir = m.param;
      double ratio = hr.front_face ? 1.0 / ir : ir;
      double cos_theta = std::fmin(dot(-normalize(r.direction), hr.normal), 1.0);
      double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
      if (!(ratio * sin_theta > 1.0)) d.coin = g.stream_next();
    } else {
      d.unit = random_unit_rejection(g);
    }
  }
  return d;
}

// src/camera.rs:98-112, recursion kept so that the product of attenuations
// is formed in the reference's order.
static inline V3 ray_colour(const Scene& sc, const Ray& r, uint64_t depth, uint32_t bounce, PathRng& g, TraceStats* st) {
  if (depth == 0) return v3(0, 0, 0);
  HitRecord hr{};
  if (!bvh_hit(sc.bvh, r, Interval{1e-3, INF}, hr, st)) {
    V3 unit_direction = normalize(r.direction);
    double a = (unit_direction.y + 1.0) / 2.0;
    return (1.0 - a) * splat(1.0) + a * v3(0.5, 0.7, 1.0);
  }
  const Material& m = sc.materials[hr.material];
  ScatterDraw d = draw_for_bounce(g, m, hr, r, bounce);
  V3 attenuation;
  Ray scattered;
  if (!scatter(m, r, hr, d, attenuation, scattered)) return v3(0, 0, 0);
  return attenuation * ray_colour(sc, scattered, depth - 1, bounce + 1, g, st);
}

struct Rect {
  uint32_t x0, y0, x1, y1;  // half-open pixel rectangle
};

// Per-pixel body of src/camera.rs:144-153; samples are summed in index order
// (the reference's Rayon reduce has no fixed order).
static inline V3 pixel_sum(const Scene& sc, const Camera& cam, uint64_t seed, int mode, uint32_t i, uint32_t j, uint32_t s0,
                           uint32_t s1, TraceStats* st) {
  V3 sum = v3(0, 0, 0);
  uint32_t pixel = j * cam.config.image_width + i;
  for (uint32_t s = s0; s < s1; ++s) {
    PathRng g{seed, pixel, s, mode};
    Ray r = get_ray(cam, i, j, g);
    sum = sum + ray_colour(sc, r, cam.config.max_depth, 0, g, st);
  }
  return sum;
}

// src/camera.rs:138-156 without the text output: adds the samples
// [s0,s1) of every pixel of `tile` into rgb_sum (W*H*3 doubles, row major).
static inline TraceStats render(const Scene& sc, const Config& cfg, uint64_t seed, uint32_t s0, uint32_t s1, Rect tile, int mode,
                                int nthreads, double* rgb_sum) {
  Camera cam = config_camera(cfg);
  if (nthreads < 1) nthreads = 1;
  std::atomic<uint32_t> next_row{tile.y0};
  std::vector<TraceStats> stats(nthreads);
  auto work = [&](int tid) {
    TraceStats st;
    for (;;) {
      uint32_t j = next_row.fetch_add(1);
      if (j >= tile.y1) break;
      for (uint32_t i = tile.x0; i < tile.x1; ++i) {
        V3 c = pixel_sum(sc, cam, seed, mode, i, j, s0, s1, &st);
        double* px = rgb_sum + 3 * ((size_t)j * cfg.image_width + i);
        px[0] += c.x;
        px[1] += c.y;
        px[2] += c.z;
      }
    }
    stats[tid] = st;
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nthreads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto& th : pool) th.join();
  TraceStats total;
  for (auto& s : stats) {
    total.rays += s.rays;
    total.node_visits += s.node_visits;
    total.sphere_tests += s.sphere_tests;
  }
  return total;
}

// src/camera.rs:85-95: divide by the sample count, gamma 2, clamp, truncate.
static inline void quantise(const double rgb[3], uint64_t samples_per_pixel, uint8_t out[3]) {
  Interval intensity{0.0, 1.0};
  for (int k = 0; k < 3; ++k) {
    double c = rgb[k] / (double)samples_per_pixel;
    double g = std::sqrt(c);
    double v = 255.0 * intensity.clamp(g);
    // Rust `as u8`: truncate toward zero, saturate, NaN -> 0.
    out[k] = std::isnan(v) ? 0 : v <= 0.0 ? 0 : v >= 255.0 ? 255 : (uint8_t)v;
  }
}

}  // namespace orc
