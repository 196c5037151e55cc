// Entry lists for camera rays (see rt_launch.h): one thread per 8x8 pixel tile walks the tree with the beam that holds
// every camera ray of the tile and writes down the nodes the render kernel starts those rays at.
//
// The beam.  A camera ray (Camera::get_ray, src/camera.rs:120-136) leaves a point o of the lens, |o - c| <= rho0 around
// the lens centre c, towards a point s of the focus plane, |s - m| <= h around the tile's middle m.  Its points are
// r(l) = o + l (s - o), l > 0, and those of the central ray are rc(l) = c + l (m - c), so that
//   |r(l) - rc(l)| = |(1 - l)(o - c) + l (s - m)| <= |1 - l| rho0 + l h =: w(l).
// A ray of the tile that touches a box B at parameter l therefore has rc(l) inside B grown by w(l) on every side.
// The test is made in two passes: first with the largest w any point of the tree can see (W1, from a bound on l over the
// tree's extent), which gives an interval of l in which the central ray can be inside the grown box at all; then with the
// largest w over that interval (w is convex: the larger of its two ends).  All of it in f64 on the f32 planes of the device
// tree (which hold the true boxes), with a pad far above any rounding of these few operations.
#include <hip/hip_runtime.h>

#include "bvh_builder.h"
#include "rt_launch.h"

namespace rta {
namespace {

struct V3 {
  double x, y, z;
};
__device__ __forceinline__ double comp(const V3& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : v.z; }

struct Beam {
  V3 c, inv;       // origin of the central ray and the reciprocal of its unit direction (+-inf where it does not move)
  double inv_len;  // 1 / |m - c|: l = t * inv_len for the time t along the unit direction
  double rho0, h;
  double t_far;    // no point of the tree's extent is seen at a later time
  double w1;       // w over [0, t_far]
  double pad_rel, pad_abs;
};

__device__ __forceinline__ double beam_width(const Beam& b, double t) {
  const double l = t * b.inv_len;
  return (b.rho0 * fabs(1.0 - l) + b.h * l) * b.pad_rel + b.pad_abs;
}

// Can the central ray be inside the box grown by w on every side, and from when to when at most?
__device__ __forceinline__ bool grown_box(const Beam& b, const float lo[3], const float hi[3], double w, double& tn, double& tf) {
  tn = 0.0;
  tf = b.t_far;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double inv = comp(b.inv, k), o = comp(b.c, k);
    const bool down = __builtin_signbit(inv);  // the high plane is met first
    const double near = down ? (double)hi[k] + w : (double)lo[k] - w, far = down ? (double)lo[k] - w : (double)hi[k] + w;
    // fmax/fmin drop a NaN (0 * inf: the ray runs inside one of the planes), which leaves that plane out: the conservative answer
    tn = fmax(tn, (near - o) * inv);
    tf = fmin(tf, (far - o) * inv);
  }
  return tn <= tf;
}

// A child that does not exist has an inverted, infinite box: its near plane lies at +inf for every direction.
__device__ __forceinline__ bool beam_box(const Beam& b, const float lo[3], const float hi[3], double& t_entry) {
  double tn, tf;
  if (!grown_box(b, lo, hi, b.w1, tn, tf)) return false;
  const double w2 = fmax(beam_width(b, tn), beam_width(b, tf));
  if (!grown_box(b, lo, hi, w2, tn, tf)) return false;
  t_entry = tn;
  return true;
}

// The children of inner node `node`: references and boxes.
template <bool kWide>
struct Children;
template <>
struct Children<false> {
  static constexpr int kCount = 2;
  uint32_t ref[2];
  float lo[2][3], hi[2][3];
  __device__ __forceinline__ void load(const SceneDev& sc, uint32_t node) {
    const DevNode d = reinterpret_cast<const DevNode*>(sc.nodes)[node];
    ref[0] = d.ref0, ref[1] = d.ref1;
    for (int c = 0; c < 2; ++c) {
      lo[c][0] = d.x[2 * c], hi[c][0] = d.x[2 * c + 1];
      lo[c][1] = d.y[2 * c], hi[c][1] = d.y[2 * c + 1];
      lo[c][2] = d.z[2 * c], hi[c][2] = d.z[2 * c + 1];
    }
  }
};
template <>
struct Children<true> {
  static constexpr int kCount = 4;
  uint32_t ref[4];
  float lo[4][3], hi[4][3];
  __device__ __forceinline__ void load(const SceneDev& sc, uint32_t node) {
    const WideNode d = reinterpret_cast<const WideNode*>(sc.nodes)[node];
    for (int c = 0; c < 4; ++c) {
      ref[c] = d.ref[c];
      for (int k = 0; k < 3; ++k) lo[c][k] = d.planes[2 * k][c], hi[c][k] = d.planes[2 * k + 1][c];
    }
  }
};

// f32 not above / not below a double
__device__ __forceinline__ float f32_down(double v) {
  float f = (float)v;
  return (double)f > v ? nextafterf(f, -__builtin_inff()) : f;
}
__device__ __forceinline__ float f32_up(double v) {
  float f = (float)v;
  return (double)f < v ? nextafterf(f, __builtin_inff()) : f;
}

template <bool kWide>
__global__ __launch_bounds__(64) void tile_cuts_kernel(SceneDev sc, CameraDev cam, double rho0, double h_pixel, uint32_t x0, uint32_t y0,
                                                       uint32_t tiles_x, uint32_t n_tiles, uint32_t row_first, uint32_t row_stride, uint32_t k_max,
                                                       uint32_t* __restrict__ cuts) {
  constexpr uint32_t kSide = 8u, kEntries = cut_entries(kWide), kWords = cut_entry_words(kWide);
  constexpr int kKids = Children<kWide>::kCount;
  const uint32_t t = blockIdx.x * 64u + threadIdx.x;
  if (t >= n_tiles) return;
  const double h_tile = h_pixel * (double)kSide;  // from the middle of the tile to its corners
  struct Entry {
    uint32_t ref;
    float lo[3], hi[3];
  };
  Entry out[kEntries];
  uint32_t n_out = 0;
  auto write = [&]() {
    uint32_t* dst = cuts + (size_t)t * (kEntries * kWords);
    if constexpr (kWide) {  // groups of four entries in the form of a staged four-wide node (rt_launch.h)
      const float inf = __builtin_inff();
      for (uint32_t k = 0; k < kEntries; ++k) {
        const bool used = k < n_out;
        uint32_t* g = dst + (k / 4u) * kWideCutGroupWords + (k & 3u);
        for (int a = 0; a < 3; ++a) {
          g[8 * a] = __float_as_uint(used ? out[k].lo[a] : inf);
          g[8 * a + 4] = __float_as_uint(used ? out[k].hi[a] : -inf);
        }
        g[24] = used ? out[k].ref : kRefNone;
      }
      for (uint32_t w = (kEntries / 4u) * kWideCutGroupWords; w < kEntries * kWords; ++w) dst[w] = 0u;
    } else {
      for (uint32_t k = 0; k < kEntries; ++k) dst[k] = k < n_out ? out[k].ref : kRefNone;
    }
  };
  const uint32_t root = sc.root;
  Entry whole;  // the root with a box that holds everything
  whole.ref = root;
  for (int a = 0; a < 3; ++a) whole.lo[a] = -3.0e38f, whole.hi[a] = 3.0e38f;
  if (root == kRefNone || (root & kRefLeaf) || k_max < 1u) {  // no tree to speak of: the walk starts where it always did
    if (root != kRefNone) out[n_out++] = whole;
    write();
    return;
  }
  // the tree's extent, from the boxes of the root's children
  Children<kWide> ch;
  ch.load(sc, root);
  double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
  for (int c = 0; c < kKids; ++c)
    if (ch.ref[c] != kRefNone)
      for (int k = 0; k < 3; ++k) blo[k] = fmin(blo[k], (double)ch.lo[c][k]), bhi[k] = fmax(bhi[k], (double)ch.hi[c][k]);
  const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
  const double half = 0.5 * (double)(kSide - 1u);
  const double fx = (double)(x0 + tx * 8u) + half, fy = (double)(y0 + (ty * row_stride + row_first) * 8u) + half;
  Beam b;
  b.c = V3{cam.center[0], cam.center[1], cam.center[2]};
  const V3 m = {cam.pixel00[0] + fx * cam.du[0] + fy * cam.dv[0], cam.pixel00[1] + fx * cam.du[1] + fy * cam.dv[1],
                cam.pixel00[2] + fx * cam.du[2] + fy * cam.dv[2]};
  const V3 D = {m.x - b.c.x, m.y - b.c.y, m.z - b.c.z};
  const double len = sqrt(D.x * D.x + D.y * D.y + D.z * D.z);
  b.inv_len = 1.0 / len;
  b.inv = V3{1.0 / (D.x * b.inv_len), 1.0 / (D.y * b.inv_len), 1.0 / (D.z * b.inv_len)};
  b.rho0 = rho0;
  b.h = h_tile;
  b.pad_rel = 1.0 + 1e-6;
  b.pad_abs = 1e-7 * (1.0 + (double)sc.bound);
  // A point p of the tree seen at time t along the central ray: t = |rc - c| <= |p - c| + w(t) <= reach + rho0 + t g.
  const double g = (rho0 + h_tile) * b.inv_len;
  const double ex = 0.5 * (bhi[0] - blo[0]), ey = 0.5 * (bhi[1] - blo[1]), ez = 0.5 * (bhi[2] - blo[2]);
  const double mx = 0.5 * (bhi[0] + blo[0]) - b.c.x, my = 0.5 * (bhi[1] + blo[1]) - b.c.y, mz = 0.5 * (bhi[2] + blo[2]) - b.c.z;
  const double reach = sqrt(mx * mx + my * my + mz * mz) + sqrt(ex * ex + ey * ey + ez * ez);
  if (!(len > 0.0) || !(g < 0.5) || !(reach < 1e300)) {  // a lens or tile as wide as the focus distance: no beam worth the name
    out[n_out++] = whole;
    write();
    return;
  }
  b.t_far = (reach + rho0) / (1.0 - g) * (1.0 + 1e-9);
  b.w1 = fmax(beam_width(b, 0.0), beam_width(b, b.t_far));

  // Depth-first, nearest child first.  A node whose children the beam touches is opened if all of them are inner nodes and
  // the list has room for them; else it goes on the list itself.  (Entries are inner nodes: a ray tests a leaf's box at
  // the leaf's parent before it tests the sphere.)  Entries listed and nodes waiting never exceed K together.
  const uint32_t K = k_max < kEntries ? k_max : kEntries;
  Entry stack[kEntries];
  uint32_t n_st = 0;
  for (int a = 0; a < 3; ++a) whole.lo[a] = f32_down(blo[a]), whole.hi[a] = f32_up(bhi[a]);
  stack[n_st++] = whole;
  while (n_st > 0) {
    const Entry node = stack[--n_st];
    if (node.ref != root) ch.load(sc, node.ref);
    bool hit[kKids];
    double tn[kKids];
    uint32_t kids = 0;
    bool any_leaf = false;
    for (int c = 0; c < kKids; ++c) {
      tn[c] = 0.0;
      hit[c] = ch.ref[c] != kRefNone && beam_box(b, ch.lo[c], ch.hi[c], tn[c]);
      if (hit[c]) {
        ++kids;
        any_leaf = any_leaf || (ch.ref[c] & kRefLeaf);
      }
    }
    if (kids == 0) continue;
    if (any_leaf || n_out + n_st + kids > K) {
      out[n_out++] = node;
      continue;
    }
    for (uint32_t n = 0; n < kids; ++n) {  // the farthest first, so that the nearest is taken off first
      int far = -1;
      for (int c = 0; c < kKids; ++c)
        if (hit[c] && (far < 0 || tn[c] > tn[far])) far = c;
      hit[far] = false;
      Entry e;
      e.ref = ch.ref[far];
      for (int a = 0; a < 3; ++a) e.lo[a] = ch.lo[far][a], e.hi[a] = ch.hi[far][a];
      stack[n_st++] = e;
    }
  }
  write();
}

}  // namespace

hipError_t launch_tile_cuts(const SceneDev& sc, const CameraDev& cam, double rho0, double h_tile, uint32_t x0, uint32_t y0, uint32_t tiles_x,
                            uint32_t n_tiles, uint32_t row_first, uint32_t row_stride, uint32_t k, uint32_t* cuts, hipStream_t stream) {
  if (n_tiles == 0) return hipSuccess;
  const unsigned blocks = (n_tiles + 63u) / 64u;
  if (sc.wide)
    hipLaunchKernelGGL(tile_cuts_kernel<true>, dim3(blocks), dim3(64), 0, stream, sc, cam, rho0, h_tile, x0, y0, tiles_x, n_tiles, row_first,
                       row_stride, k, cuts);
  else
    hipLaunchKernelGGL(tile_cuts_kernel<false>, dim3(blocks), dim3(64), 0, stream, sc, cam, rho0, h_tile, x0, y0, tiles_x, n_tiles, row_first,
                       row_stride, k, cuts);
  return hipGetLastError();
}

}  // namespace rta
