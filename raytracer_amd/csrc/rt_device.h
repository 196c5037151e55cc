// Device-side building blocks of the gfx950 path tracer: Philox, the camera,
// the f32 node test, the f64 sphere test, BVH traversal and the three
// scatter functions.  Included only by rt_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_launch.h"

namespace rta {

// --------------------------------------------------------------- Philox ----
// Philox4x32-10.  Counter = (pixel, sample, slot, domain), key = seed.
enum : uint32_t { DOM_CAMERA = 0, DOM_SCATTER = 1 };

__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0;
    c1 = lo1;
    c2 = n2;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}
// 53 random bits scaled by 2^-53: uniform in [0,1) like rand's Standard f64.
__device__ __forceinline__ double unit_from_bits(uint32_t lo, uint32_t hi) {
  unsigned long long v = ((unsigned long long)hi << 32) | lo;
  return (double)(v >> 11) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ void philox_pair(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t slot, uint32_t domain,
                                            double& u0, double& u1) {
  uint32_t o[4];
  philox4x32_10(k0, k1, pixel, sample, slot, domain, o);
  u0 = unit_from_bits(o[0], o[1]);
  u1 = unit_from_bits(o[2], o[3]);
}

// ---------------------------------------------------------------- maths ----
struct D3 {
  double x, y, z;
};
__device__ __forceinline__ D3 d3(double x, double y, double z) { return D3{x, y, z}; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator*(double s, D3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ D3 operator*(D3 a, D3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 normalize(D3 a) { return rsqrt(dot(a, a)) * a; }

// Uniform point on the unit sphere from two uniforms: z = 1-2u0, phi = 2 pi u1.
// Same distribution as normalising a point drawn uniformly in the unit ball
// (vector::random_unit, src/vector.rs:16-27).
__device__ __forceinline__ D3 unit_from_uniforms(double u0, double u1) {
  double z = 1.0 - 2.0 * u0;
  double rxy = sqrt(fmax(0.0, 1.0 - z * z));
  double s, c;
  sincospi(2.0 * u1, &s, &c);
  return d3(rxy * c, rxy * s, z);
}
// Uniform point in the unit disk: r = sqrt(u0), phi = 2 pi u1
// (vector::random_within_unit_disk, src/vector.rs:29-40).
__device__ __forceinline__ void disk_from_uniforms(double u0, double u1, double& px, double& py) {
  double r = sqrt(u0);
  double s, c;
  sincospi(2.0 * u1, &s, &c);
  px = r * c;
  py = r * s;
}

// --------------------------------------------------------------- camera ----
// Camera::get_ray (src/camera.rs:114-136) for pixel (i,j), sample s.
__device__ __forceinline__ void camera_ray(const CameraDev& cam, uint32_t k0, uint32_t k1, uint32_t i, uint32_t j, uint32_t s, D3& org,
                                           D3& dir) {
  const uint32_t pixel = j * cam.width + i;
  const D3 du = d3(cam.du[0], cam.du[1], cam.du[2]), dv = d3(cam.dv[0], cam.dv[1], cam.dv[2]);
  D3 pixel_center = d3(cam.pixel00[0], cam.pixel00[1], cam.pixel00[2]) + ((double)i * du) + ((double)j * dv);
  double u0, u1;
  philox_pair(k0, k1, pixel, s, 0, DOM_CAMERA, u0, u1);
  double px = -0.5 + u0, py = -0.5 + u1;
  D3 pixel_sample = pixel_center + ((px * du) + (py * dv));
  org = d3(cam.center[0], cam.center[1], cam.center[2]);
  if (cam.defocus) {
    philox_pair(k0, k1, pixel, s, 1, DOM_CAMERA, u0, u1);
    double dx, dy;
    disk_from_uniforms(u0, u1, dx, dy);
    org = org + dx * d3(cam.disk_u[0], cam.disk_u[1], cam.disk_u[2]) + dy * d3(cam.disk_v[0], cam.disk_v[1], cam.disk_v[2]);
  }
  dir = normalize(pixel_sample - org);
}

// ------------------------------------------------------------- node test ----
// The reference tests a node with f64 slabs (AABB::ray_intersection,
// src/aabb.rs:69-82) and only the yes/no answer is used (src/bvh.rs:58).
// Here the test runs in f32 on boxes rounded outwards.  Every rounding error
// is covered by padding so that "miss" is returned only where the f64 test
// also returns a miss; the closest hit found by traversal is therefore the
// one the f64 traversal finds.
//
// Per axis, with r = 1/d:  t = b*r - o*r, evaluated as fma(b, rf, c) where
// c = -(o*r) -/+ e folds in an absolute pad e = 12 ulp_half * |o*r| (covers the
// rounding of o, r and o*r), and the entry/exit distances get a relative pad
// of 6 half-ulps.  An axis whose reciprocal is not a usable f32 number
// (direction component ~ 0) constrains nothing, which is conservative.
struct RayF32 {
  float rf[3], cn[3], cf[3];
};
#define RTA_U24 5.9604644775390625e-08f /* 2^-24 */

__device__ __forceinline__ void ray_f32_axis(double o, double d, float& rf, float& cn, float& cf) {
  float df = (float)d;
  float r = __builtin_amdgcn_rcpf(df);
  float of = (float)o;
  float orf = of * r;
  float e = fabsf(orf) * (12.0f * RTA_U24);
  bool ok = fabsf(r) <= 1e18f;  // false for inf and NaN too
  rf = ok ? r : 0.0f;
  cn = ok ? (-orf - e) : -__builtin_inff();
  cf = ok ? (-orf + e) : __builtin_inff();
}
__device__ __forceinline__ RayF32 ray_f32(D3 o, D3 d) {
  RayF32 q;
  ray_f32_axis(o.x, d.x, q.rf[0], q.cn[0], q.cf[0]);
  ray_f32_axis(o.y, d.y, q.rf[1], q.cn[1], q.cf[1]);
  ray_f32_axis(o.z, d.z, q.rf[2], q.cn[2], q.cf[2]);
  return q;
}
__device__ __forceinline__ float f32_below(double v) {  // largest f32 <= v
  float f = (float)v;
  return ((double)f > v) ? nextafterf(f, -__builtin_inff()) : f;
}
__device__ __forceinline__ float f32_above(double v) {  // smallest f32 >= v
  float f = (float)v;
  return ((double)f < v) ? nextafterf(f, __builtin_inff()) : f;
}
// negx/negy/negz: direction component < 0 (selects which plane is entered first).
__device__ __forceinline__ bool node_test(const RayF32& q, bool negx, bool negy, bool negz, float4 n0, float4 n1, float tminf,
                                          float tmaxf) {
  float bnx = negx ? n1.x : n0.x, bfx = negx ? n0.x : n1.x;
  float bny = negy ? n1.y : n0.y, bfy = negy ? n0.y : n1.y;
  float bnz = negz ? n1.z : n0.z, bfz = negz ? n0.z : n1.z;
  float tnx = fmaf(bnx, q.rf[0], q.cn[0]), tfx = fmaf(bfx, q.rf[0], q.cf[0]);
  float tny = fmaf(bny, q.rf[1], q.cn[1]), tfy = fmaf(bfy, q.rf[1], q.cf[1]);
  float tnz = fmaf(bnz, q.rf[2], q.cn[2]), tfz = fmaf(bfz, q.rf[2], q.cf[2]);
  float tn = fmaxf(fmaxf(tnx, tny), tnz);
  float tf = fminf(fminf(tfx, tfy), tfz);
  tn = fmaf(-fabsf(tn), 6.0f * RTA_U24, tn);
  tf = fmaf(fabsf(tf), 6.0f * RTA_U24, tf);
  return !(tf < tn || tf < tminf || tmaxf < tn);
}

// ------------------------------------------------------------- traversal ----
struct TreeView {
  const float4* gnodes;
  const float4* lnodes;  // LDS copy of nodes [0, n_lds)
  const double2* gsph;
  const double2* lsph;  // LDS copy of the sphere slots, or nullptr
  uint32_t n_nodes, n_lds, n_sph_lds;
};

__device__ __forceinline__ void fetch_node(const TreeView& tv, uint32_t i, float4& n0, float4& n1) {
  if (i < tv.n_lds) {
    n0 = tv.lnodes[2 * i];
    n1 = tv.lnodes[2 * i + 1];
  } else {
    n0 = tv.gnodes[2 * i];
    n1 = tv.gnodes[2 * i + 1];
  }
}
__device__ __forceinline__ void fetch_sphere(const TreeView& tv, uint32_t slot, double2& s0, double2& s1) {
  if (slot < tv.n_sph_lds) {
    s0 = tv.lsph[2 * slot];
    s1 = tv.lsph[2 * slot + 1];
  } else {
    s0 = tv.gsph[2 * slot];
    s1 = tv.gsph[2 * slot + 1];
  }
}

// BVH::hit (src/bvh.rs:50-84): closest hit in (t_min, t_max), the nearer child
// first (by the sign of the direction on the split axis), spheres of a leaf in
// stored order, a later hit must be strictly closer (Interval::surrounds,
// src/ray.rs:35-37).  The reference pushes both children and pops the nearer;
// here the nearer child is entered directly and the other is pushed, which
// visits nodes in the same order.  `stack` is this lane's column of the LDS
// stack, entries `stride` words apart; its capacity is the depth of the tree.
// Returns the slot of the hit sphere (or -1) and its distance in t_best.
__device__ __forceinline__ int traverse(const TreeView& tv, uint32_t* stack, uint32_t stride, D3 o, D3 d, double t_min, double t_max,
                                        double& t_best, uint32_t& n_visits) {
  int best = -1;
  t_best = t_max;
  if (tv.n_nodes == 0) return best;
  const RayF32 q = ray_f32(o, d);
  const bool negx = d.x < 0.0, negy = d.y < 0.0, negz = d.z < 0.0;
  const float tminf = f32_below(t_min);
  float tmaxf = (t_max < 3.0e38) ? f32_above(t_max) : __builtin_inff();
  // Sphere::hit (src/sphere.rs:29-48) terms that depend on the ray only.
  const double a = dot(d, d);
  const double inv_a = 1.0 / a;
  uint32_t cur = 0, sp = 0;
  for (;;) {
    float4 n0, n1;
    fetch_node(tv, cur, n0, n1);
    ++n_visits;
    if (node_test(q, negx, negy, negz, n0, n1, tminf, tmaxf)) {
      const uint32_t na = __float_as_uint(n0.w), nb = __float_as_uint(n1.w);
      if (!(nb & 0x80000000u)) {
        const bool neg = nb == 0 ? negx : nb == 1 ? negy : negz;
        // direction < 0 on the split axis: right child first (src/bvh.rs:64-68)
        stack[sp * stride] = na + (neg ? 0u : 1u);
        ++sp;
        cur = na + (neg ? 1u : 0u);
        continue;
      }
      const uint32_t count = nb & 0x7fffffffu;
      for (uint32_t k = 0; k < count; ++k) {
        double2 s0, s1;
        fetch_sphere(tv, na + k, s0, s1);
        D3 oc = d3(o.x - s0.x, o.y - s0.y, o.z - s1.x);
        double half_b = dot(oc, d);
        double c = dot(oc, oc) - s1.y * s1.y;
        double disc = half_b * half_b - a * c;
        if (disc >= 0.0) {
          double sq = sqrt(disc);
          double t = (-half_b - sq) * inv_a;
          if (!(t_min < t && t < t_best)) t = (-half_b + sq) * inv_a;
          if (t_min < t && t < t_best) {
            t_best = t;
            best = (int)(na + k);
            tmaxf = f32_above(t);
          }
        }
      }
    }
    if (sp == 0) break;
    --sp;
    cur = stack[sp * stride];
  }
  return best;
}

// ---------------------------------------------------------------- shading ----
struct Hit {
  D3 p, normal;
  bool front;
  uint32_t material, prim;
};
// HitRecord::new (src/hit.rs:18-27) for the sphere in `slot` at distance t.
__device__ __forceinline__ Hit make_hit(const TreeView& tv, const double2* aux, int slot, D3 o, D3 d, double t) {
  double2 s0, s1;
  fetch_sphere(tv, (uint32_t)slot, s0, s1);
  double2 ax = aux[slot];
  Hit h;
  h.p = d3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);
  D3 n = ax.x * d3(h.p.x - s0.x, h.p.y - s0.y, h.p.z - s1.x);
  h.front = dot(d, n) < 0.0;
  h.normal = h.front ? n : d3(-n.x, -n.y, -n.z);
  unsigned long long bits = (unsigned long long)__double_as_longlong(ax.y);
  h.material = (uint32_t)bits;
  h.prim = (uint32_t)(bits >> 32);
  return h;
}

// Material::scatter (src/material.rs:21-65).  `d` is the unit direction of the
// incoming ray.  Returns false where the reference returns None.
__device__ __forceinline__ bool scatter(const double2* materials, uint32_t mat, D3 d, const Hit& h, double u0, double u1, D3& attenuation,
                                        D3& out_dir) {
  const double2 m0 = materials[4 * mat], m1 = materials[4 * mat + 1], m2 = materials[4 * mat + 2], m3 = materials[4 * mat + 3];
  const uint32_t type = (uint32_t)__double_as_longlong(m3.x);
  attenuation = d3(m0.x, m0.y, m1.x);
  D3 dir;
  if (type == 2u) {  // Dielectric
    attenuation = d3(1.0, 1.0, 1.0);
    double ratio = h.front ? m2.x : m1.y;  // ir^-1 entering, ir leaving
    double cos_theta = fmin(-dot(d, h.normal), 1.0);
    double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
    bool cannot_refract = ratio * sin_theta > 1.0;
    double x = 1.0 - cos_theta, x2 = x * x;
    double reflectance = m2.y + (1.0 - m2.y) * (x2 * x2 * x);  // Schlick, src/material.rs:45-50
    if (cannot_refract || reflectance > u0) {
      dir = d - (2.0 * dot(d, h.normal)) * h.normal;  // vector::reflect, src/vector.rs:56-58
    } else {  // vector::refract, src/vector.rs:60-65
      D3 perp = ratio * (d + cos_theta * h.normal);
      double par = -sqrt(fabs(1.0 - dot(perp, perp)));
      dir = perp + par * h.normal;
    }
  } else {
    D3 unit = unit_from_uniforms(u0, u1);
    if (type == 0u) {  // Lambertian
      dir = h.normal + unit;
      if (fabs(dir.x) <= 1e-6 && fabs(dir.y) <= 1e-6 && fabs(dir.z) <= 1e-6) dir = h.normal;  // near_zero, src/vector.rs:52-54
    } else {  // Metal
      dir = (d - (2.0 * dot(d, h.normal)) * h.normal) + m1.y * unit;
      if (dot(dir, h.normal) <= 0.0) return false;
    }
  }
  out_dir = normalize(dir);  // Ray::new, src/ray.rs:16-23
  return true;
}

}  // namespace rta
