// Device-side building blocks of the gfx950 path tracer: Philox, the camera,
// the f32 node test, the f64 sphere test, BVH traversal, the three
// scatter functions and the staging of a scene in LDS.  Included by the kernel sources only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_launch.h"

namespace rta {

// --------------------------------------------------------------- Philox ----
// Philox4x32 (Salmon et al., SC'11).  Counter = (pixel, sample, slot, domain), key = seed.  The draws of the
// render loop use seven rounds, the fewest its authors found to pass BigCrush; ten rounds are the published
// default, which the host keeps for the scene generator.
enum : uint32_t { DOM_CAMERA = 0, DOM_SCATTER = 1 };
constexpr int kRenderRounds = 7;

__device__ __forceinline__ void philox4x32_render(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                  uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < kRenderRounds; ++r) {
    // one 32x32->64 multiply per word pair (v_mad_u64_u32) instead of separate high and low products
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c0 = n0;
    c1 = (uint32_t)p1;
    c2 = n2;
    c3 = (uint32_t)p0;
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
  // (the top 53 of the 64 bits) * 2^-53 = hi * 2^-32 + (lo without its low 11 bits) * 2^-64: every step is exact
  return fma((double)hi, 1.0 / 4294967296.0, (double)(lo & 0xfffff800u) * (1.0 / 18446744073709551616.0));
}
// Two uniforms of 53 bits from one block: what a scatter event draws.
__device__ __forceinline__ void philox_pair(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t slot, uint32_t domain,
                                            double& u0, double& u1) {
  uint32_t o[4];
  philox4x32_render(k0, k1, pixel, sample, slot, domain, o);
  u0 = unit_from_bits(o[0], o[1]);
  u1 = unit_from_bits(o[2], o[3]);
}
// Four uniforms of 32 bits (word * 2^-32, in [0,1)) from one block: what a camera ray draws -- the position in
// the pixel and the point on the lens, neither of which can tell 2^-32 from 2^-53.
__device__ __forceinline__ void philox_quad(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t slot, uint32_t domain,
                                            double u[4]) {
  uint32_t o[4];
  philox4x32_render(k0, k1, pixel, sample, slot, domain, o);
#pragma unroll
  for (int k = 0; k < 4; ++k) u[k] = (double)o[k] * (1.0 / 4294967296.0);
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
// v with its sign bit flipped where `mask` (0 or 0x80000000) has it set
__device__ __forceinline__ double flip_sign(double v, uint32_t mask) {
  return __hiloint2double(__double2hiint(v) ^ (int)mask, __double2loint(v));
}

// A double constant that is materialised where it is used (two scalar moves).
// Without this the compiler hoists the polynomial coefficients below out of
// the kernel's main loop into vector registers, runs out of registers, and
// reloads them from scratch memory one by one on every evaluation.
__device__ __forceinline__ double local_const(double c) {
  asm volatile("" : "+s"(c));
  return c;
}

// Square root of a finite x >= 0 that is not huge or tiny enough to need rescaling (0, or |x| within
// 1e-280 .. 1e280): the reciprocal-square-root instruction refined by the same Goldschmidt
// steps and residual corrections the compiler's sqrt uses, without its range checks.  A zero is lifted to 2^-1000
// instead of being singled out afterwards (the instruction's result for 0 is infinite): its root comes out as
// 2^-500 = 3e-151, which no sum with a number of the scene's scale can see.
__device__ __forceinline__ double sqrt_plain(double x) {
  double xx;
  asm("v_max_f64 %0, %1, %2" : "=v"(xx) : "v"(x), "v"(0x1p-1000));
  const double y = __builtin_amdgcn_rsq(xx);
  double g = xx * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  double d = fma(-g, g, xx);
  g = fma(d, h, g);
  d = fma(-g, g, xx);
  return fma(d, h, g);
}

// 1/sqrt(x) for a finite x > 0 that is not tiny or huge: the reciprocal-square-root instruction and
// the refinement step the compiler's rsqrt uses, without its check for zero and infinite results.
__device__ __forceinline__ double rsqrt_plain(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-(y * x), y, 1.0);
  return fma(y * e, fma(e, 0.375, 0.5), y);
}
// A direction of the path: never the zero vector (scatter replaces a degenerate one, the camera looks somewhere).
__device__ __forceinline__ D3 normalize(D3 a) { return rsqrt_plain(dot(a, a)) * a; }

// The sphere test's discriminant (x >= 0): an exactly tangent ray gets a root 1e-150 off.
__device__ __forceinline__ double sqrt_disc(double x) { return sqrt_plain(x); }

// sin and cos of 2*pi*u for u in [0,1): the argument is reduced exactly to a
// quarter turn q and a remainder |f| <= 1/8 of a turn, then the classic
// minimax polynomials for sin and cos on [-pi/4, pi/4] are evaluated
// (coefficients as in fdlibm's __kernel_sin/__kernel_cos; error about 1 ulp).
__device__ __forceinline__ void sincos_2pi(double u, double& s, double& c) {
  const double q = rint(4.0 * u);           // 0..4
  const double f = fma(q, -0.25, u);        // exact, |f| <= 0.125
  const double a = f * local_const(6.28318530717958647692528676655900577);
  const double z = a * a;
  double ps = local_const(1.58969099521155010221e-10);
  ps = fma(ps, z, local_const(-2.50507602534068634195e-08));
  ps = fma(ps, z, local_const(2.75573137070700676789e-06));
  ps = fma(ps, z, local_const(-1.98412698298579493134e-04));
  ps = fma(ps, z, local_const(8.33333333332248946124e-03));
  ps = fma(ps, z, local_const(-1.66666666666666324348e-01));
  const double sa = fma(a * z, ps, a);
  double pc = local_const(-1.13596475577881948265e-11);
  pc = fma(pc, z, local_const(2.08757232129817482790e-09));
  pc = fma(pc, z, local_const(-2.75573143513906633035e-07));
  pc = fma(pc, z, local_const(2.48015872894767294178e-05));
  pc = fma(pc, z, local_const(-1.38888888888741095749e-03));
  pc = fma(pc, z, local_const(4.16666666666666019037e-02));
  const double ca = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int k = (int)q;  // rotate by k quarter turns
  const double s1 = (k & 1) ? ca : sa, c1 = (k & 1) ? sa : ca;
  s = (k & 2) ? -s1 : s1;
  c = ((k + 1) & 2) ? -c1 : c1;
}

// The same through a table of sin and cos at the 512th parts of a turn (8 KB of LDS, filled by stage_scene with the
// function above): the remainder is at most 1/1024 of a turn (6.2e-3 rad), where two terms of the sine's series and three of
// the cosine's reach double precision (the first terms left out are 6e-20 and 8e-17), and the angle-sum formulas put the two
// together.  About 2 ulp; under half the instructions.
constexpr uint32_t kTrigEntries = 512;
struct TrigSeries {};  // no table: the polynomials over a quarter turn
struct TrigTable {
  __attribute__((address_space(3))) double* tab;  // kTrigEntries x (sin, cos)
};
__device__ __forceinline__ void sincos_2pi(double u, double& s, double& c, TrigSeries) { sincos_2pi(u, s, c); }
__device__ __forceinline__ void sincos_2pi(double u, double& s, double& c, TrigTable t) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) d2 lds_d2;
  const double q = rint((double)kTrigEntries * u);           // 0..kTrigEntries
  const double f = fma(q, -1.0 / kTrigEntries, u);           // exact, |f| <= 1/(2*kTrigEntries)
  const uint32_t k = (uint32_t)(int)q & (kTrigEntries - 1u);
  const d2 sc = *(lds_d2*)(t.tab + 2u * k);
  const double a = f * local_const(6.28318530717958647692528676655900577);
  const double z = a * a;
  double ps = local_const(1.0 / 120.0);
  ps = fma(ps, z, local_const(-1.0 / 6.0));
  const double sa = fma(a * z, ps, a);
  double pc = local_const(1.0 / 24.0);
  pc = fma(pc, z, -0.5);
  const double ca = fma(pc, z, 1.0);
  s = fma(sc.x, ca, sc.y * sa);
  c = fma(sc.y, ca, -(sc.x * sa));
}

// Uniform point on the unit sphere from two uniforms: z = 1-2u0, phi = 2 pi u1.
// Same distribution as normalising a point drawn uniformly in the unit ball
// (vector::random_unit, src/vector.rs:16-27).
template <class Trig = TrigSeries>
__device__ __forceinline__ D3 unit_from_uniforms(double u0, double u1, Trig trig = Trig{}) {
  double z = 1.0 - 2.0 * u0;
  double rxy = sqrt_plain(1.0 - z * z);  // (a rounding below zero is lifted like a zero)
  double s, c;
  sincos_2pi(u1, s, c, trig);
  return d3(rxy * c, rxy * s, z);
}
// Uniform point in the unit disk: r = sqrt(u0), phi = 2 pi u1
// (vector::random_within_unit_disk, src/vector.rs:29-40).
template <class Trig = TrigSeries>
__device__ __forceinline__ void disk_from_uniforms(double u0, double u1, double& px, double& py, Trig trig = Trig{}) {
  double r = sqrt_plain(u0);
  double s, c;
  sincos_2pi(u1, s, c, trig);
  px = r * c;
  py = r * s;
}

// --------------------------------------------------------------- camera ----
typedef __attribute__((address_space(3))) double lds_double;
// Where camera_ray reads the 18 camera constants from: the kernel argument
// block, or a copy in LDS (read at the point of use, so they do not occupy
// vector registers for the lifetime of the kernel).
enum { CAM_CENTER = 0, CAM_PIXEL00 = 3, CAM_DU = 6, CAM_DV = 9, CAM_DISK_U = 12, CAM_DISK_V = 15, CAM_DOUBLES = 18 };
struct CamArgs {
  const CameraDev& c;
  __device__ __forceinline__ D3 get(int k) const {
    const double* p = k == CAM_CENTER ? c.center : k == CAM_PIXEL00 ? c.pixel00 : k == CAM_DU ? c.du : k == CAM_DV ? c.dv : k == CAM_DISK_U ? c.disk_u : c.disk_v;
    return d3(p[0], p[1], p[2]);
  }
};
struct CamLds {
  lds_double* p;
  __device__ __forceinline__ D3 get(int k) const { return d3(p[k], p[k + 1], p[k + 2]); }
};
// Camera::get_ray (src/camera.rs:114-136) for pixel (i,j), sample s.
template <class Cam, class Trig = TrigSeries>
__device__ __forceinline__ void camera_ray(const Cam& cam, uint32_t width, bool defocus, uint32_t k0, uint32_t k1, uint32_t i, uint32_t j,
                                           uint32_t s, D3& org, D3& dir, Trig trig = Trig{}) {
  const uint32_t pixel = j * width + i;
  const D3 du = cam.get(CAM_DU), dv = cam.get(CAM_DV);
  D3 pixel_center = cam.get(CAM_PIXEL00) + ((double)i * du) + ((double)j * dv);
  double u[4];  // one block per camera ray: (0, 1) the position in the pixel, (2, 3) the point on the lens
  philox_quad(k0, k1, pixel, s, 0, DOM_CAMERA, u);
  double px = -0.5 + u[0], py = -0.5 + u[1];
  D3 pixel_sample = pixel_center + ((px * du) + (py * dv));
  org = cam.get(CAM_CENTER);
  if (defocus) {
    double dx, dy;
    disk_from_uniforms(u[2], u[3], dx, dy, trig);
    org = org + dx * cam.get(CAM_DISK_U) + dy * cam.get(CAM_DISK_V);
  }
  dir = normalize(pixel_sample - org);
}

// ------------------------------------------------------------- node test ----
// The reference tests a node with f64 slabs (AABB::ray_intersection,
// src/aabb.rs:69-82) and only the yes/no answer is used (src/bvh.rs:58).
// Here the test runs in f32 on boxes rounded outwards, and every rounding
// error is covered by a pad, so that "miss" is returned only where the f64
// test also returns a miss: traversal may enter a few more nodes than the
// reference, never fewer, and finds the same closest hit.
//
// Per axis, with r = 1/d, a plane b is crossed at t = b*r - o*r, evaluated
// as fma(b, rf, c).  The constant c = -(o*r) -/+ e carries the pad
//   e = 2^-24 * (6*B*|rf| + 12*|o*r|),  B = largest |plane coordinate| in the tree,
// which bounds the rounding of o, r, o*r, of the product b*rf and of the fma
// result (4*B*|r| + 7*|o*r| half-ulps are needed).  The pad is subtracted for
// the plane the ray enters through and added for the one it leaves through;
// which of lo/hi that is depends on the sign of r, so the two constants are
// stored as (c_lo, c_hi) and entry/exit are min/max of the two products.
// The reciprocal of a direction component ~ 0 is cut to +-1e18 (see ray_box_axis), which keeps the test conservative.
#define RTA_U24 5.9604644775390625e-08f /* 2^-24 */

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// The tests run in a shifted and scaled time, tau = (t - t0) * scale: t0 is the start of the query's interval
// (so that "not before the interval" is tau >= 0, which the clamp modifier of a multiply-add gives for free, see
// fma_clamp) and `scale` a power of two that brings the times of the scene below 1 (the clamp also cuts at 1:
// an entry time beyond that can only make a test answer "hit" more often).  Subtracting t0 rounds once more
// (covered by the pad: 2 * t0 more half-ulps); scaling by a power of two does not round.
struct TimeFrame {
  float t0, scale;
  float t0s;     // t0 * scale: the interval's start in scaled time
  float pad_t0;  // the share of the subtraction of t0s in the pad of a plane constant: 2^-24 * 2 * |t0s|
};
// A wave-uniform float kept in a scalar register (the scalar unit has no float arithmetic: left alone, the compiler
// works such values out per lane and holds them in vector registers).
__device__ __forceinline__ float uniform_f32(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ TimeFrame time_frame(double t_min, float scale) {
  float f = (float)t_min;
  if ((double)f > t_min) f = nextafterf(f, -__builtin_inff());  // the largest f32 <= t_min
  const float t0s = f * scale;
  return TimeFrame{f, scale, t0s, RTA_U24 * 2.0f * fabsf(t0s)};
}
// The same frame for every lane of the wave: its numbers in scalar registers.
__device__ __forceinline__ TimeFrame uniform_frame(TimeFrame tf) {
  return TimeFrame{uniform_f32(tf.t0), uniform_f32(tf.scale), uniform_f32(tf.t0s), uniform_f32(tf.pad_t0)};
}
// A power of two that brings the times a ray can spend in a scene of this extent (and 16 times more) below 1.
__device__ __forceinline__ float time_scale_for(float bound) {
  int e;
  (void)frexpf(fmaxf(bound, 1e-30f), &e);  // bound < 2^e
  return ldexpf(1.0f, -(e + 5));
}
// tau of a time t >= t0, rounded up (the far end of an interval)
__device__ __forceinline__ float tau_above(const TimeFrame& tf, double t) {
  if (!(t < 3.0e38)) return __builtin_inff();
  const double v = (t - (double)tf.t0) * (double)tf.scale;
  float f = (float)v;
  return ((double)f < v) ? nextafterf(f, __builtin_inff()) : f;
}

struct RayBox {
  float rx, ry, rz;  // reciprocal direction, times the time scale
  f32x2 cx, cy, cz;  // (c_lo, c_hi) per axis: constants for the box's low and high plane
  f32x2 nx, ny, nz;  // (c_near, c_far) per axis: the same two constants by the order the ray meets the planes
};

// `pad_bound` = 2^-24 * 6 * B (see above).  The reciprocal is cut to +-1e18: a ray that (almost) does not move along the axis
// then gets plane times of +-(distance to the plane) * 1e18 * scale, far outside the scene's times on the side the plane
// lies, unless the origin is within rounding of the plane, and that is what the pad (which grows with |r|) covers.  No
// infinity or NaN comes out of the constants this way, and nothing needs to be singled out.
__device__ __forceinline__ void ray_box_axis(double o, double d, float pad_bound, TimeFrame tf, float& rf, f32x2& c2, f32x2& s2) {
  const float r1 = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf((float)d), -1e18f, 1e18f);
  const float r = r1 * tf.scale;  // everything below is in scaled time (a power of two: no rounding)
  const float orf = (float)o * r;
  const float e = fmaf(pad_bound, fabsf(r), fmaf(RTA_U24 * 12.0f, fabsf(orf), tf.pad_t0));
  const float cn = -orf - (e + tf.t0s), cf = -orf + (e - tf.t0s);
  rf = r;
  c2 = r < 0.0f ? f32x2{cf, cn} : f32x2{cn, cf};
  s2 = f32x2{cn, cf};
}
__device__ __forceinline__ RayBox ray_box(D3 o, D3 d, float pad_bound, TimeFrame tf) {
  RayBox q;
  ray_box_axis(o.x, d.x, pad_bound, tf, q.rx, q.cx, q.nx);
  ray_box_axis(o.y, d.y, pad_bound, tf, q.ry, q.cy, q.ny);
  ray_box_axis(o.z, d.z, pad_bound, tf, q.rz, q.cz, q.nz);
  return q;
}
// 2^-24 * 6 * B for a tree whose plane coordinates are at most B in size
__device__ __forceinline__ float pad_for_bound(float bound) { return RTA_U24 * 6.0f * bound; }
__device__ __forceinline__ float f32_below(double v) {  // largest f32 <= v
  float f = (float)v;
  return ((double)f > v) ? nextafterf(f, -__builtin_inff()) : f;
}
__device__ __forceinline__ float f32_above(double v) {  // smallest f32 >= v
  float f = (float)v;
  return ((double)f < v) ? nextafterf(f, __builtin_inff()) : f;
}
// Some f32 >= v for a finite v > 0 well inside the f32 range: the nearest f32 moved one step up, without
// looking at which way the conversion rounded (at most two steps above v instead of at most one).
__device__ __forceinline__ float f32_beyond(double v) { return __uint_as_float(__float_as_uint((float)v) + 1u); }
// Some tau >= that of a time t inside the interval, in f32: t converted and moved one step up (>= t), scaled and
// shifted by one multiply-add (one rounding), and moved one step up again.
__device__ __forceinline__ float tau_beyond(const TimeFrame& tf, double t) {
  const float t_up = __uint_as_float(__float_as_uint((float)t) + 1u);
  return __uint_as_float(__float_as_uint(fmaf(t_up, tf.scale, -tf.t0s)) + 1u);
}
// One box given as (lo, hi) pairs per axis: does the ray cross it within [tminf, tmaxf] (in the ray's time frame)?
__device__ __forceinline__ bool box_test(const RayBox& q, f32x2 bx, f32x2 by, f32x2 bz, float tminf, float tmaxf, float& tn) {
  f32x2 tx = f32x2{fmaf(bx.x, q.rx, q.cx.x), fmaf(bx.y, q.rx, q.cx.y)};
  f32x2 ty = f32x2{fmaf(by.x, q.ry, q.cy.x), fmaf(by.y, q.ry, q.cy.y)};
  f32x2 tz = f32x2{fmaf(bz.x, q.rz, q.cz.x), fmaf(bz.y, q.rz, q.cz.y)};
  tn = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), tminf));
  float tf = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fminf(fmaxf(tz.x, tz.y), tmaxf));
  return tn <= tf;
}
// min/max as the bare instructions: fminf/fmaxf pass an operand that was not computed in sight of
// the compiler (the running far limit) through a quieting max(x, x) first.  A NaN operand (0 * inf on an axis the ray does not
// move along) is dropped in favour of the other operands, which is the conservative answer.
__device__ __forceinline__ float max_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float min_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// a * b + c cut to [0, 1] by the instruction's output modifier (a NaN comes out as 0)
__device__ __forceinline__ float fma_clamp(float a, float b, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// Both boxes of a node at once, from the direction-sorted planes (entry planes of box 0 and 1 in
// .xy, exit planes in .zw).  Scalar multiply-adds: the packed two-wide form measured slower here.
// The start of the interval is time 0 of the ray's frame: the clamp on one of the three entry times stands for it.
__device__ __forceinline__ void box_pair_sorted(const RayBox& q, f32x4 X, f32x4 Y, f32x4 Z, float tmaxf, float& tn0, float& tn1,
                                                bool& h0, bool& h1) {
  const f32x2 nx = {fmaf(X.x, q.rx, q.nx.x), fmaf(X.y, q.rx, q.nx.x)}, fx = {fmaf(X.z, q.rx, q.nx.y), fmaf(X.w, q.rx, q.nx.y)};
  const f32x2 ny = {fmaf(Y.x, q.ry, q.ny.x), fmaf(Y.y, q.ry, q.ny.x)}, fy = {fmaf(Y.z, q.ry, q.ny.y), fmaf(Y.w, q.ry, q.ny.y)};
  const f32x2 nz = {fma_clamp(Z.x, q.rz, q.nz.x), fma_clamp(Z.y, q.rz, q.nz.x)}, fz = {fmaf(Z.z, q.rz, q.nz.y), fmaf(Z.w, q.rz, q.nz.y)};
  tn0 = max3_raw(nx.x, ny.x, nz.x);
  tn1 = max3_raw(nx.y, ny.y, nz.y);
  const float tf0 = min3_raw(fx.x, fy.x, min_raw(fz.x, tmaxf));
  const float tf1 = min3_raw(fx.y, fy.y, min_raw(fz.y, tmaxf));
  h0 = tn0 <= tf0;
  h1 = tn1 <= tf1;
}

// ------------------------------------------------------------- traversal ----
// Device tree: inner nodes only, 64 bytes each, holding the boxes of both
// children (so a child that the ray misses is never fetched):
//   q0 = (lo0.x, hi0.x, lo1.x, hi1.x)   q1 = the same for y   q2 = for z
//   q3 = (ref0, ref1, split axis, 0) as integers
// A ref names an inner node (bit 31 clear) or a leaf: bit 31 set, bits 30..24
// the sphere count (1..127), bits 23..0 the first sphere slot.
constexpr uint32_t kRefLeaf = 0x80000000u;
constexpr uint32_t kRefDone = 0xffffffffu;  // traversal finished / empty tree
// Two more values the render kernel keeps in the same register to say what a lane wants next.
constexpr uint32_t kRefCamera = 0xfffffffeu;  // start the next sample
constexpr uint32_t kRefIdle = 0xfffffffdu;    // nothing left to do; every leaf ref is below this value
// A node staged in LDS is 7 quads (112 B): for each axis the four planes twice, ordered for rays
// that travel up the axis, (lo0, lo1, hi0, hi1), and for rays that travel down it,
// (hi0, hi1, lo0, lo1); then the references.  A lane reads the copy that matches its direction and
// gets the planes it meets first in .xy and those it meets last in .zw, with no min/max per axis.
constexpr uint32_t kLdsNodeQuads = 7;

typedef __attribute__((address_space(3))) f32x4 lds_float4;
typedef __attribute__((address_space(3))) f32x2 lds_float2;
typedef __attribute__((address_space(3))) f64x2 lds_double2;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

struct TreeView {
  const float4* gnodes;
  lds_float4* lnodes;  // LDS copy of nodes [0, n_lds), kLdsNodeQuads apart
  const double2* gsph;
  lds_double2* lsph;  // LDS copy of sphere slots [0, n_sph_lds)
  const double2* gaux;
  lds_double2* laux;  // LDS copy of the slots' shading data [0, n_aux_lds)
  const double2* gmat;
  lds_double2* lmat;  // LDS copy of materials [0, n_mat_lds)
  uint32_t n_lds, n_sph_lds, n_aux_lds, n_mat_lds;
  uint32_t root;
  uint32_t first_leaf;  // see SceneDev
  lds_double2* lfirst;  // LDS copy of that sphere's record
  float bound;
  float pad_bound;     // pad_for_bound(bound)
  TimeFrame time;      // the frame the node tests run in: the start of the queries' interval and the scene's time scale
  bool single_leaves;  // every leaf holds exactly one sphere
};

// Stack of the four-wide traversal: the first `k` entries of a lane's stack live in LDS, deeper ones (rare) in
// a column of global memory.
struct HybridStack {
  lds_u32* l;   // entry e < k at l[e * kWorkgroup]
  uint32_t* g;  // entry e >= k at g[(e - k) * gstride]
  uint32_t k, gstride;
};

// LDS address of staged node i as an integer (LDS pointers are 32 bits wide).
__device__ __forceinline__ uint32_t lds_node_address(lds_float4* lnodes, uint32_t i) {
  return (uint32_t)(uintptr_t)lnodes + i * (kLdsNodeQuads * 16u);
}
// Writes node i (global layout q0..q3) into its LDS slot.
// kScaled (trees that are in LDS as a whole): references to inner nodes are stored as the LDS
// address of the node's copy, so that following one needs neither a multiplication nor a base.
template <bool kScaled>
__device__ __forceinline__ void stage_node(lds_float4* lnodes, uint32_t i, f32x4 q0, f32x4 q1, f32x4 q2, f32x4 q3) {
  lds_float4* p = lnodes + kLdsNodeQuads * i;
  if (kScaled) {
    uint32_t r0 = __float_as_uint(q3.x), r1 = __float_as_uint(q3.y);
    if (!(r0 & kRefLeaf)) r0 = lds_node_address(lnodes, r0);
    if (!(r1 & kRefLeaf)) r1 = lds_node_address(lnodes, r1);
    q3.x = __uint_as_float(r0);
    q3.y = __uint_as_float(r1);
  }
  p[0] = f32x4{q0.x, q0.z, q0.y, q0.w};
  p[1] = f32x4{q0.y, q0.w, q0.x, q0.z};
  p[2] = f32x4{q1.x, q1.z, q1.y, q1.w};
  p[3] = f32x4{q1.y, q1.w, q1.x, q1.z};
  p[4] = f32x4{q2.x, q2.z, q2.y, q2.w};
  p[5] = f32x4{q2.y, q2.w, q2.x, q2.z};
  p[6] = q3;
}
// Node i from LDS for a ray that reads copy off / 16 of each axis (off = 0 or 16).
template <bool kScaled>
__device__ __forceinline__ void fetch_node_lds(const TreeView& tv, uint32_t i, uint32_t offx, uint32_t offy, uint32_t offz, f32x4& X, f32x4& Y,
                                               f32x4& Z, f32x2& refs) {
  typedef __attribute__((address_space(3))) char lds_char;
  // kScaled: i is the node's LDS address itself
  lds_char* node = kScaled ? (lds_char*)(uintptr_t)i : (lds_char*)tv.lnodes + __umul24(i, kLdsNodeQuads * 16u);
  X = *(lds_float4*)(node + offx);
  Y = *(lds_float4*)(node + 32u + offy);
  Z = *(lds_float4*)(node + 64u + offz);
  refs = *(lds_float2*)(node + 96u);  // only the two references: no dead destination registers
}
__device__ __forceinline__ void fetch_node_global(const TreeView& tv, uint32_t i, f32x4& q0, f32x4& q1, f32x4& q2, f32x2& refs) {
  const f32x4* p = reinterpret_cast<const f32x4*>(tv.gnodes) + 4 * (size_t)i;
  q0 = p[0];
  q1 = p[1];
  q2 = p[2];
  refs = *reinterpret_cast<const f32x2*>(p + 3);
}

template <bool kAllLds>
__device__ __forceinline__ void fetch_sphere(const TreeView& tv, uint32_t slot, f64x2& s0, f64x2& s1) {
  if (kAllLds || slot < tv.n_sph_lds) {
    s0 = tv.lsph[2 * slot];
    s1 = tv.lsph[2 * slot + 1];
  } else {
    const f64x2* p = reinterpret_cast<const f64x2*>(tv.gsph) + 2 * (size_t)slot;
    s0 = p[0];
    s1 = p[1];
  }
}

// State of one lane's closest-hit query, BVH::hit (src/bvh.rs:50-84).
//
// The reference pushes both children of a split and tests each node's own box
// when it is popped, taking first the child on the side the ray comes from
// (src/bvh.rs:64-68).  Here both child boxes are tested at the parent; of two
// children that are hit the one whose box the ray enters first becomes
// current and the other is pushed.  The closest hit does not depend on the order of visits: a later hit
// replaces the current one only if strictly closer (Interval::surrounds,
// src/ray.rs:35-37), spheres of a leaf are tested in stored order, and only two
// different spheres hit at exactly the same distance could be told apart by
// it.  `stack` is this lane's column of the stack (in LDS, or in global memory for
// trees too deep for LDS), entries `stride` words apart; the tree is never
// deeper than its capacity.
struct Trav {
  D3 o, d;
  RayBox q;
  double t_best;
  float tmaxf;
  uint32_t offx, offy, offz;  // per axis, which of the two copies in a node staged in LDS to read: byte offset 0 or 16 within the axis
  uint32_t cur, sp;  // sp: where the stack's free slot is (see stack_slot)
  uint32_t tos;  // the top of the stack lives here; `sp` entries below it are in LDS
  int best;
};

// `sp0`: the stack pointer of an empty stack (stack_origin of the lane's stack; 0 for the four-wide walk, which counts entries).
template <bool kWide = false>
__device__ __forceinline__ void trav_begin(Trav& tr, const TreeView& tv, D3 o, D3 d, double t_max, uint32_t sp0 = 0u) {
  tr.o = o;
  tr.d = d;
  tr.q = ray_box(o, d, tv.pad_bound, tv.time);
  tr.t_best = t_max;
  tr.tmaxf = tau_above(tv.time, t_max);
  // The copy must match the sign the node test sees, which is that of the f32 reciprocal (never a zero: it is the
  // reciprocal of a number, cut to +-1e18): 16 where the ray travels down the axis, from the sign bit.
  tr.offx = (__float_as_uint(tr.q.rx) >> 27) & 16u;
  tr.offy = (__float_as_uint(tr.q.ry) >> 27) & 16u;
  tr.offz = (__float_as_uint(tr.q.rz) >> 27) & 16u;
  tr.cur = tv.root;
  tr.sp = sp0;
  tr.tos = kRefDone;  // popping the bottom entry ends the traversal
  tr.best = -1;
}

// The next node is already in a register when it is needed; the entry below it is
// fetched from the stack behind the node test that follows.
// A stack in LDS is walked by address: `sp` is the LDS address of the free slot above the entries in use (entries of a
// lane are kWorkgroup words apart), so that a slot is reached without an add.  One row below the bottom is part of
// the stack's allocation (kStackSpareRows): the pop that ends a traversal moves `sp` there and reads a word nobody looks at.
// A stack in global memory counts `sp` in bytes from its bottom and stops there.
__device__ __forceinline__ uint32_t stack_origin(lds_u32* stack) { return (uint32_t)(uintptr_t)stack; }
__device__ __forceinline__ uint32_t stack_origin(uint32_t*) { return 0u; }
__device__ __forceinline__ lds_u32* stack_slot(lds_u32*, uint32_t sp) { return (lds_u32*)(uintptr_t)sp; }
__device__ __forceinline__ uint32_t* stack_slot(uint32_t* stack, uint32_t byte_offset) {
  return (uint32_t*)((char*)stack + byte_offset);
}
__device__ __forceinline__ uint32_t stack_down(lds_u32*, uint32_t sp, uint32_t bytes) { return sp - bytes; }
__device__ __forceinline__ uint32_t stack_down(uint32_t*, uint32_t sp, uint32_t bytes) { return __builtin_elementwise_sub_sat(sp, bytes); }
template <class StackPtr>
__device__ __forceinline__ void trav_pop(Trav& tr, StackPtr stack, uint32_t stride) {
  // With nothing below the top the top is kRefDone, the traversal ends with this pop and
  // what is read into the top is never looked at: no need to tell that case apart.
  tr.cur = tr.tos;
  tr.sp = stack_down(stack, tr.sp, 4u * stride);
  tr.tos = *stack_slot(stack, tr.sp);
}
template <class StackPtr>
__device__ __forceinline__ void trav_push(Trav& tr, StackPtr stack, uint32_t stride, uint32_t ref) {
  *stack_slot(stack, tr.sp) = tr.tos;
  tr.sp += 4u * stride;
  tr.tos = ref;
}

// One inner node: test both children, descend into the first one hit.
template <bool kAllLds, class StackPtr>
__device__ __forceinline__ void trav_inner_step(Trav& tr, const TreeView& tv, StackPtr stack, uint32_t stride) {
  f32x2 q3;
  bool h0, h1;
  float tn0 = 0.0f, tn1 = 0.0f;
  if (kAllLds || tr.cur < tv.n_lds) {
    f32x4 X, Y, Z;
    fetch_node_lds<kAllLds>(tv, tr.cur, tr.offx, tr.offy, tr.offz, X, Y, Z, q3);
    box_pair_sorted(tr.q, X, Y, Z, tr.tmaxf, tn0, tn1, h0, h1);
  } else {
    f32x4 q0, q1, q2;
    fetch_node_global(tv, tr.cur, q0, q1, q2, q3);
    h0 = box_test(tr.q, f32x2{q0.x, q0.y}, f32x2{q1.x, q1.y}, f32x2{q2.x, q2.y}, 0.0f, tr.tmaxf, tn0);
    h1 = box_test(tr.q, f32x2{q0.z, q0.w}, f32x2{q1.z, q1.w}, f32x2{q2.z, q2.w}, 0.0f, tr.tmaxf, tn1);
  }
  const uint32_t ref0 = __float_as_uint(q3.x), ref1 = __float_as_uint(q3.y);
  const bool swap = tn1 < tn0;  // of two children that are hit, the one entered first is visited first
  // Selects instead of a three-way branch: only the push and the pop are predicated.
  const uint32_t near = swap ? ref1 : ref0, far = swap ? ref0 : ref1;
  const bool both = h0 && h1;
  if (both) trav_push(tr, stack, stride, far);
  if (h0 || h1) {
    tr.cur = both ? near : (h0 ? ref0 : ref1);
  } else {
    trav_pop(tr, stack, stride);
  }
}

// Two inner-node steps with the second node's fetch issued before the first step's stack
// bookkeeping, so that the bookkeeping runs while the fetch is in flight.  Same effect on the
// traversal state as two calls of trav_inner_step (the second only if the cursor is still at
// an inner node).
struct NodeData {
  f32x4 a, b, c;
  f32x2 q3;  // the two child references
  bool sorted;  // a, b, c are the direction-sorted LDS copies (near planes in .xy, far in .zw) rather than q0, q1, q2
};
template <bool kAllLds>
__device__ __forceinline__ NodeData node_fetch(const Trav& tr, const TreeView& tv, uint32_t i) {
  NodeData n;
  if (kAllLds || i < tv.n_lds) {
    fetch_node_lds<kAllLds>(tv, i, tr.offx, tr.offy, tr.offz, n.a, n.b, n.c, n.q3);
    n.sorted = true;
  } else {
    fetch_node_global(tv, i, n.a, n.b, n.c, n.q3);
    n.sorted = false;
  }
  return n;
}
struct NodeChoice {
  uint32_t next, far;
  bool both, any;
};
template <bool kAllLds>
__device__ __forceinline__ NodeChoice node_choose(const Trav& tr, const NodeData& n) {
  bool h0, h1;
  float tn0 = 0.0f, tn1 = 0.0f;
  if (kAllLds || n.sorted) {
    box_pair_sorted(tr.q, n.a, n.b, n.c, tr.tmaxf, tn0, tn1, h0, h1);
  } else {
    h0 = box_test(tr.q, f32x2{n.a.x, n.a.y}, f32x2{n.b.x, n.b.y}, f32x2{n.c.x, n.c.y}, 0.0f, tr.tmaxf, tn0);
    h1 = box_test(tr.q, f32x2{n.a.z, n.a.w}, f32x2{n.b.z, n.b.w}, f32x2{n.c.z, n.c.w}, 0.0f, tr.tmaxf, tn1);
  }
  const uint32_t ref0 = __float_as_uint(n.q3.x), ref1 = __float_as_uint(n.q3.y);
  const bool swap = tn1 < tn0;
  NodeChoice c;
  c.both = h0 && h1;
  c.any = h0 || h1;
  c.far = swap ? ref0 : ref1;
  c.next = c.both ? (swap ? ref1 : ref0) : (c.any ? (h0 ? ref0 : ref1) : tr.tos);
  return c;
}
// The stack side of a step whose cursor move (to c.next) is done by the caller.
template <class StackPtr>
__device__ __forceinline__ void node_apply(Trav& tr, const NodeChoice& c, StackPtr stack, uint32_t stride) {
  // The store goes to the free slot above the stack's top in every lane; it only counts where
  // sp then moves up (both children hit).  The stack has a slot to spare for it.
  *stack_slot(stack, tr.sp) = tr.tos;
  if (!c.both && !c.any) {  // c.next was the top of the stack: bring up the entry below it (see trav_pop for sp == 0)
    tr.sp = stack_down(stack, tr.sp, 4u * stride);
    tr.tos = *stack_slot(stack, tr.sp);
  }
  tr.sp += c.both ? 4u * stride : 0u;
  tr.tos = c.both ? c.far : tr.tos;
}
// Up to kLeft node steps in a row for a lane that stays on inner nodes, the next node's fetch issued before the
// stack work of the current step.  Returns the number of steps made.
template <bool kAllLds, int kLeft, class StackPtr>
__device__ __forceinline__ uint32_t inner_chain(Trav& tr, const TreeView& tv, const NodeData& n, StackPtr stack, uint32_t stride) {
  const NodeChoice c = node_choose<kAllLds>(tr, n);
  if constexpr (kLeft > 1) {
    const bool go = !(c.next & kRefLeaf);
    NodeData m;
    if (go) m = node_fetch<kAllLds>(tr, tv, c.next);
    node_apply(tr, c, stack, stride);
    tr.cur = c.next;
    if (go) return 1u + inner_chain<kAllLds, kLeft - 1>(tr, tv, m, stack, stride);
    return 1u;
  } else {
    node_apply(tr, c, stack, stride);
    tr.cur = c.next;
    return 1u;
  }
}
// Node steps per round of the inner-node block (measured on the reference scene: 3 is 2 % faster than 2, 4 is 3 % slower, 5 is 1 % slower than 3).
constexpr int kInnerSteps = 3;
template <bool kAllLds, class StackPtr>
__device__ __forceinline__ uint32_t trav_inner_steps(Trav& tr, const TreeView& tv, StackPtr stack, uint32_t stride) {
  const NodeData n = node_fetch<kAllLds>(tr, tv, tr.cur);
  return inner_chain<kAllLds, kInnerSteps>(tr, tv, n, stack, stride);
}

// ------------------------------------------------------- four-wide nodes ----
// Scenes too large for LDS are walked through 128-byte nodes holding the f32 boxes of up to four children
// (WideNode in bvh_builder.h): per axis one 16-byte word of the four low planes and one of the four high planes,
// then the four references.  A lane reads, per axis, the word it meets first and the word it meets last (by
// address: tr.offx/y/z is 16 where the ray travels down the axis), so the test needs no per-plane min/max, and
// the planes go through the same padded multiply-add as those of the two-wide nodes.
constexpr uint32_t kWideLdsStride = 112;  // bytes between nodes staged in LDS (the padding is not staged): an odd multiple of 16 spreads them over the banks
constexpr uint32_t kWideGlobalQuads = 8, kWideStagedQuads = 7;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;

__device__ __forceinline__ lds_u32* hs_slot_l(const HybridStack& st, uint32_t e) { return st.l + e * kWorkgroup; }
__device__ __forceinline__ void hs_store(const HybridStack& st, uint32_t e, uint32_t v) {
  if (e < st.k)
    *hs_slot_l(st, e) = v;
  else
    st.g[(size_t)(e - st.k) * st.gstride] = v;
}
__device__ __forceinline__ uint32_t hs_load(const HybridStack& st, uint32_t e) {
  if (e < st.k) return *hs_slot_l(st, e);
  return st.g[(size_t)(e - st.k) * st.gstride];
}
// The stack of the four-wide walk counts `sp` in entries; as in the two-wide walk the top lives in a register.
__device__ __forceinline__ void trav_pop(Trav& tr, const HybridStack& st, uint32_t) {
  tr.cur = tr.tos;
  tr.sp = __builtin_elementwise_sub_sat(tr.sp, 1u);
  tr.tos = hs_load(st, tr.sp);
}
__device__ __forceinline__ void trav_push(Trav& tr, const HybridStack& st, uint32_t ref) {
  hs_store(st, tr.sp, tr.tos);
  tr.sp += 1u;
  tr.tos = ref;
}
// One four-wide node: test the four child boxes, continue with the one entered first, set the others aside.
struct WideData {
  f32x4 NX, FX, NY, FY, NZ, FZ;  // per axis the four children's planes met first and those met last
  u32x4 refs;
};
typedef __attribute__((address_space(3))) char lds_char;
// A node staged in LDS at p (also a group of a tile's entry list, which has the same form).
__device__ __forceinline__ WideData wide_fetch_lds(const Trav& tr, lds_char* p) {
  WideData w;
  w.NX = *(lds_float4*)(p + tr.offx);
  w.FX = *(lds_float4*)(p + (tr.offx ^ 16u));
  w.NY = *(lds_float4*)(p + 32u + tr.offy);
  w.FY = *(lds_float4*)(p + 32u + (tr.offy ^ 16u));
  w.NZ = *(lds_float4*)(p + 64u + tr.offz);
  w.FZ = *(lds_float4*)(p + 64u + (tr.offz ^ 16u));
  w.refs = *(lds_u32x4*)(p + 96);
  return w;
}
__device__ __forceinline__ WideData wide_fetch(const Trav& tr, const TreeView& tv, uint32_t node) {
  // tr.offx/y/z: 16 where the ray travels down the axis (its near planes are the high ones), else 0
  WideData w;
  if (node < tv.n_lds) {
    w = wide_fetch_lds(tr, (lds_char*)tv.lnodes + __umul24(node, kWideLdsStride));
  } else {
    const char* base = reinterpret_cast<const char*>(tv.gnodes);
    const uint32_t at = node * 128u;
    // The loads are written out so that the compiler does not know them for loads: both branches write the same
    // registers, in different lanes, and the compiler would have the LDS reads of the other lanes wait until these loads
    // have landed.  The wait below is for them.  (This relies on the compiler giving the two branches' results the
    // same registers, with no copy between a load and the wait -- it does, see the assembly; every render of a scene
    // beyond LDS in the tests would show stale node data at once.)
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(w.NX) : "v"(at + tr.offx), "s"(base) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(w.FX) : "v"(at + (tr.offx ^ 16u)), "s"(base) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=&v"(w.NY) : "v"(at + tr.offy), "s"(base) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=&v"(w.FY) : "v"(at + (tr.offy ^ 16u)), "s"(base) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=&v"(w.NZ) : "v"(at + tr.offz), "s"(base) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=&v"(w.FZ) : "v"(at + (tr.offz ^ 16u)), "s"(base) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:96" : "=&v"(w.refs) : "v"(at), "s"(base) : "memory");
  }
  // every lane: the loads above have landed when this returns (the LDS reads are the compiler's to wait for)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(w.NX), "+v"(w.FX), "+v"(w.NY), "+v"(w.FY), "+v"(w.NZ), "+v"(w.FZ), "+v"(w.refs) : : "memory");
  return w;
}
struct WideChoice {
  bool h0, h1, h2, h3, is0, f1, f2, f3;
  uint32_t next;
};
// The node test for the four boxes: where the ray enters each and whether it can hit it.
struct WideTimes {
  float tn0, tn1, tn2, tn3;
  bool h0, h1, h2, h3;
};
__device__ __forceinline__ WideTimes wide_times(const Trav& tr, const WideData& w) {
  const f32x4 &NX = w.NX, &FX = w.FX, &NY = w.NY, &FY = w.FY, &NZ = w.NZ, &FZ = w.FZ;
  const RayBox& q = tr.q;
  WideTimes t;
  t.tn0 = max3_raw(fmaf(NX.x, q.rx, q.nx.x), fmaf(NY.x, q.ry, q.ny.x), fma_clamp(NZ.x, q.rz, q.nz.x));
  t.tn1 = max3_raw(fmaf(NX.y, q.rx, q.nx.x), fmaf(NY.y, q.ry, q.ny.x), fma_clamp(NZ.y, q.rz, q.nz.x));
  t.tn2 = max3_raw(fmaf(NX.z, q.rx, q.nx.x), fmaf(NY.z, q.ry, q.ny.x), fma_clamp(NZ.z, q.rz, q.nz.x));
  t.tn3 = max3_raw(fmaf(NX.w, q.rx, q.nx.x), fmaf(NY.w, q.ry, q.ny.x), fma_clamp(NZ.w, q.rz, q.nz.x));
  const float tf0 = min3_raw(fmaf(FX.x, q.rx, q.nx.y), fmaf(FY.x, q.ry, q.ny.y), min_raw(fmaf(FZ.x, q.rz, q.nz.y), tr.tmaxf));
  const float tf1 = min3_raw(fmaf(FX.y, q.rx, q.nx.y), fmaf(FY.y, q.ry, q.ny.y), min_raw(fmaf(FZ.y, q.rz, q.nz.y), tr.tmaxf));
  const float tf2 = min3_raw(fmaf(FX.z, q.rx, q.nx.y), fmaf(FY.z, q.ry, q.ny.y), min_raw(fmaf(FZ.z, q.rz, q.nz.y), tr.tmaxf));
  const float tf3 = min3_raw(fmaf(FX.w, q.rx, q.nx.y), fmaf(FY.w, q.ry, q.ny.y), min_raw(fmaf(FZ.w, q.rz, q.nz.y), tr.tmaxf));
  // (a node has at least two children; an absent child's box is inverted and infinite, so it is never hit:
  // inf * r + c is +inf on its near side and -inf on its far side, or NaN on both, which the min/max drop, on an
  // axis the ray does not move along -- never on all three)
  t.h0 = t.tn0 <= tf0, t.h1 = t.tn1 <= tf1, t.h2 = t.tn2 <= tf2, t.h3 = t.tn3 <= tf3;
  return t;
}
__device__ __forceinline__ WideChoice wide_choose(const Trav& tr, const WideData& w) {
  const u32x4& refs = w.refs;
  const WideTimes x = wide_times(tr, w);
  WideChoice c;
  c.h0 = x.h0, c.h1 = x.h1, c.h2 = x.h2, c.h3 = x.h3;
  const float inf = __builtin_inff();
  const float t0 = c.h0 ? x.tn0 : inf, t1 = c.h1 ? x.tn1 : inf, t2 = c.h2 ? x.tn2 : inf, t3 = c.h3 ? x.tn3 : inf;
  const float m = min3_raw(t0, t1, min_raw(t2, t3));
  // the child entered first (the lowest of equals); selects, not branches
  c.is0 = t0 == m;
  const bool is1 = t1 == m, is2 = t2 == m;
  c.f1 = !c.is0 && is1;
  c.f2 = !c.is0 && !is1 && is2;
  c.f3 = !c.is0 && !is1 && !is2;
  uint32_t nx = refs.w;
  nx = c.f2 ? refs.z : nx;
  nx = c.f1 ? refs.y : nx;
  c.next = c.is0 ? refs.x : nx;
  return c;
}
__device__ __forceinline__ void wide_apply(Trav& tr, const HybridStack& st, const u32x4& refs, const WideChoice& c) {
  // The others that are hit wait on the stack, pushed in the order 3, 2, 1, 0 (last pushed, first visited: the
  // later children of a node come from the larger subtree that was opened up).  (Measured: storing to three slots
  // unconditionally instead of these branches, and one generic load sequence for LDS and global nodes, were both slower.)
  if (__builtin_amdgcn_ballot_w64(tr.sp + 3u > st.k) == 0) {
    // no lane of the wave can leave the part of its stack that is in LDS in this step: pushes and the pop without the test for it
    const uint32_t base = (uint32_t)(uintptr_t)st.l;  // (LDS pointers are 32 bits wide)
    uint32_t a = base + tr.sp * (4u * kWorkgroup);
    const auto push = [&](uint32_t ref) {
      *(lds_u32*)(uintptr_t)a = tr.tos;
      a += 4u * kWorkgroup;
      tr.tos = ref;
    };
    if (c.h3 && !c.f3) push(refs.w);
    if (c.h2 && !c.f2) push(refs.z);
    if (c.h1 && !c.f1) push(refs.y);
    if (c.h0 && !c.is0) push(refs.x);
    tr.sp = (a - base) / (4u * kWorkgroup);
    if (c.h0 || c.h1 || c.h2 || c.h3) {
      tr.cur = c.next;
    } else {
      tr.cur = tr.tos;
      tr.sp = __builtin_elementwise_sub_sat(tr.sp, 1u);
      tr.tos = *hs_slot_l(st, tr.sp);
    }
    return;
  }
  if (c.h3 && !c.f3) trav_push(tr, st, refs.w);
  if (c.h2 && !c.f2) trav_push(tr, st, refs.z);
  if (c.h1 && !c.f1) trav_push(tr, st, refs.y);
  if (c.h0 && !c.is0) trav_push(tr, st, refs.x);
  if (c.h0 || c.h1 || c.h2 || c.h3) {
    tr.cur = c.next;
  } else {
    trav_pop(tr, st, 0u);
  }
}
__device__ __forceinline__ void trav_wide_step(Trav& tr, const TreeView& tv, const HybridStack& st) {
  const WideData w = wide_fetch(tr, tv, tr.cur);
  const WideChoice c = wide_choose(tr, w);
  wide_apply(tr, st, w.refs, c);
}

// One leaf: Sphere::hit (src/sphere.rs:29-48) for each of its spheres.  The
// direction has unit length, so the reference's a = |d|^2 is taken as 1.
// Sphere::hit (src/sphere.rs:29-48) for the sphere in `slot`.  The direction has unit length, so
// the reference's a = |d|^2 is taken as 1.
// (s0, s1 = the sphere's record: centre x, y | centre z, radius)
__device__ __forceinline__ void sphere_test_record(Trav& tr, const TreeView& tv, uint32_t slot, f64x2 s0, f64x2 s1, double t_min) {
  D3 oc = d3(tr.o.x - s0.x, tr.o.y - s0.y, tr.o.z - s1.x);
  double half_b = dot(oc, tr.d);
  double c = dot(oc, oc) - s1.y * s1.y;
  double disc = half_b * half_b - c;
  if (disc >= 0.0) {
    double sq = sqrt_disc(disc);
    // The near root if it lies beyond t_min, else the far one (src/sphere.rs:38-44 takes the first of the two
    // inside the interval; a near root at or beyond t_best leaves nothing for the far root, which is later still).
    double t = -half_b - sq;
    if (!(t_min < t)) t = -half_b + sq;
    if (t_min < t && t < tr.t_best) {
      tr.t_best = t;
      tr.best = (int)slot;
      tr.tmaxf = tau_beyond(tv.time, t);
    }
  }
}
template <bool kAllLds>
__device__ __forceinline__ void sphere_test(Trav& tr, const TreeView& tv, uint32_t slot, double t_min) {
  f64x2 s0, s1;
  fetch_sphere<kAllLds>(tv, slot, s0, s1);
  sphere_test_record(tr, tv, slot, s0, s1, t_min);
}
// The sphere that stands outside the tree (TreeView::first_leaf), tested when a query starts.  In the reference's tree
// it is a leaf under the root whose box the query's origin lies in, so it is the first thing the walk would test anyway
// (src/bvh.rs:50-84); here that costs no node step and no round of the leaf block.  Returns the number of tests made.
template <bool kAllLds>
__device__ __forceinline__ uint32_t trav_first(Trav& tr, const TreeView& tv, double t_min) {
  if (tv.first_leaf == kRefDone) return 0u;
  sphere_test_record(tr, tv, tv.first_leaf, tv.lfirst[0], tv.lfirst[1], t_min);  // its record has a copy in LDS whatever else is there
  return 1u;
}
// One leaf: each of its spheres in stored order.  Trees whose leaves all hold one sphere (what
// both builders make of spheres that do not coincide) take the branch without the loop.
// kSingle: the caller knows at compile time that the tree is of that kind (the loop is then not even
// compiled in, which also spares the copies that merging the two forms costs).
template <bool kAllLds, bool kSingle = false, class StackPtr>
__device__ __forceinline__ void trav_leaf_step(Trav& tr, const TreeView& tv, StackPtr stack, uint32_t stride, double t_min,
                                               uint32_t& n_tests) {
  const uint32_t first = tr.cur & 0x00ffffffu;
  if (kSingle || tv.single_leaves) {
    ++n_tests;
    sphere_test<kAllLds>(tr, tv, first, t_min);
  } else {
    const uint32_t count = (tr.cur >> 24) & 0x7fu;
    for (uint32_t k = 0; k < count; ++k) {
      ++n_tests;
      sphere_test<kAllLds>(tr, tv, first + k, t_min);
    }
  }
  trav_pop(tr, stack, stride);
}

// Runs the query to completion (all lanes of the wave together).
template <bool kAllLds, class StackPtr>
__device__ __forceinline__ void trav_run(Trav& tr, const TreeView& tv, StackPtr stack, uint32_t stride, double t_min, uint32_t& n_steps,
                                         uint32_t& n_tests) {
  while (tr.cur != kRefDone) {
    while (!(tr.cur & kRefLeaf)) {
      trav_inner_step<kAllLds>(tr, tv, stack, stride);
      ++n_steps;
    }
    if (tr.cur != kRefDone) trav_leaf_step<kAllLds>(tr, tv, stack, stride, t_min, n_tests);
  }
}

// The same over the four-wide tree.
__device__ __forceinline__ void trav_run_wide(Trav& tr, const TreeView& tv, const HybridStack& st, double t_min, uint32_t& n_steps,
                                              uint32_t& n_tests) {
  while (tr.cur != kRefDone) {
    while (!(tr.cur & kRefLeaf)) {
      trav_wide_step(tr, tv, st);
      ++n_steps;
    }
    if (tr.cur != kRefDone) trav_leaf_step<false>(tr, tv, st, 0u, t_min, n_tests);
  }
}

// ---------------------------------------------------------------- shading ----
struct Hit {
  D3 p, normal;
  bool front;
  uint32_t material, prim;
};
// HitRecord::new (src/hit.rs:18-27) for the sphere in `slot` at distance t.
// h.material carries the material index (low 30 bits) and type (top 2 bits).
// What make_hit reads of the sphere in `slot`: asked for first, so that work that does not need it can go on meanwhile.
// (Centre x, y | centre z | 1/radius, material and index: the radius itself is not read.)
struct HitData {
  f64x2 s0;
  double cz;
  f64x2 ax;
};
template <bool kAllLds>
__device__ __forceinline__ HitData fetch_hit_data(const TreeView& tv, int slot) {
  HitData hd;
  if (kAllLds || (uint32_t)slot < tv.n_sph_lds) {
    hd.s0 = tv.lsph[2 * slot];
    hd.cz = ((const lds_double*)(tv.lsph + 2 * slot + 1))[0];
  } else {
    const f64x2* p = reinterpret_cast<const f64x2*>(tv.gsph) + 2 * (size_t)slot;
    hd.s0 = p[0];
    hd.cz = reinterpret_cast<const double*>(p + 1)[0];
  }
  if (kAllLds || (uint32_t)slot < tv.n_aux_lds)
    hd.ax = tv.laux[slot];
  else
    hd.ax = reinterpret_cast<const f64x2*>(tv.gaux)[slot];
  return hd;
}
__device__ __forceinline__ Hit make_hit_from(const HitData& hd, D3 o, D3 d, double t) {
  const f64x2 s0 = hd.s0, ax = hd.ax;
  Hit h;
  h.p = d3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);
  D3 n = ax.x * d3(h.p.x - s0.x, h.p.y - s0.y, h.p.z - hd.cz);
  h.front = dot(d, n) < 0.0;
  const uint32_t flip = h.front ? 0u : 0x80000000u;  // the normal against the ray: one select and three sign flips
  h.normal = d3(flip_sign(n.x, flip), flip_sign(n.y, flip), flip_sign(n.z, flip));
  unsigned long long bits = (unsigned long long)__double_as_longlong(ax.y);
  h.material = (uint32_t)bits;
  h.prim = (uint32_t)(bits >> 32);
  return h;
}
template <bool kAllLds>
__device__ __forceinline__ Hit make_hit(const TreeView& tv, int slot, D3 o, D3 d, double t) {
  return make_hit_from(fetch_hit_data<kAllLds>(tv, slot), o, d, t);
}

// Material::scatter (src/material.rs:21-65).  `d` is the unit direction of the
// incoming ray, `mat` the material index with its type in the top 2 bits.
// Material record (32 bytes): Lambertian/Metal (albedo.x, albedo.y)(albedo.z, fuzz);
// Dielectric (ir, 1/ir)(r0, -) with r0 = ((1-ir)/(1+ir))^2, which is the same
// number for ir and 1/ir.  Returns false where the reference returns None (out_dir is then a direction nobody follows:
// the lanes of a wave go through the normalisation together anyway).
template <bool kAllLds, class Trig = TrigSeries>
__device__ __forceinline__ bool scatter(const TreeView& tv, uint32_t mat, D3 d, const Hit& h, double u0, double u1, D3& attenuation,
                                        D3& out_dir, Trig trig = Trig{}) {
  const uint32_t type = mat >> 30, index = mat & 0x3fffffffu;
  f64x2 m0, m1;
  if (kAllLds || index < tv.n_mat_lds) {
    m0 = tv.lmat[2 * index];
    m1 = tv.lmat[2 * index + 1];
  } else {
    const f64x2* p = reinterpret_cast<const f64x2*>(tv.gmat) + 2 * (size_t)index;
    m0 = p[0];
    m1 = p[1];
  }
  D3 dir;
  bool scattered = true;
  if (type == 2u) {  // Dielectric
    attenuation = d3(1.0, 1.0, 1.0);
    double ratio = h.front ? m0.y : m0.x;  // ir^-1 entering, ir leaving (src/material.rs:52)
    double cos_theta = fmin(-dot(d, h.normal), 1.0);
    // ratio * sin_theta > 1 (src/material.rs:54-55) decided on the squares, which need no root; where those are too
    // close to 1 to tell (never, in practice: the wave skips it) the reference's own expression decides
    const double sin2 = fmax(0.0, 1.0 - cos_theta * cos_theta), k2 = (ratio * ratio) * sin2;
    bool cannot_refract = k2 > 1.0;
    if (fabs(k2 - 1.0) < 1e-12) cannot_refract = ratio * sqrt_plain(sin2) > 1.0;
    double x = 1.0 - cos_theta, x2 = x * x;
    double reflectance = m1.x + (1.0 - m1.x) * (x2 * x2 * x);  // Schlick, src/material.rs:45-50
    if (cannot_refract || reflectance > u0) {
      dir = d - (2.0 * dot(d, h.normal)) * h.normal;  // vector::reflect, src/vector.rs:56-58
    } else {  // vector::refract, src/vector.rs:60-65
      D3 perp = ratio * (d + cos_theta * h.normal);
      double par = -sqrt_plain(fabs(1.0 - dot(perp, perp)));
      dir = perp + par * h.normal;
    }
  } else {
    attenuation = d3(m0.x, m0.y, m1.x);
    D3 unit = unit_from_uniforms(u0, u1, trig);
    if (type == 0u) {  // Lambertian
      dir = h.normal + unit;
      const bool near_zero = fabs(dir.x) <= 1e-6 && fabs(dir.y) <= 1e-6 && fabs(dir.z) <= 1e-6;  // src/vector.rs:52-54
      if (__builtin_amdgcn_ballot_w64(near_zero)) {  // (a branch of the wave: practically never taken)
        if (near_zero) dir = h.normal;
      }
    } else {  // Metal
      dir = (d - (2.0 * dot(d, h.normal)) * h.normal) + m1.y * unit;
      scattered = !(dot(dir, h.normal) <= 0.0);
    }
  }
  out_dir = normalize(dir);  // Ray::new, src/ray.rs:16-23
  return scattered;
}

// ------------------------------------------------------------ LDS image ----
// LDS image of a workgroup:
//   [camera constants][sin/cos table][sphere outside the tree][entry lists][top-of-tree nodes][sphere slots][slot shading data][materials][per-lane stacks]
struct LdsImage {
  TreeView tv;
  CamLds cam;
  TrigTable trig;
  lds_u32* ring;  // per wave kCutRing entry lists of kCutEntries words (see rt_launch.h)
  lds_u32* stack_base;
};
// words of a wave's ring of entry lists, and of a workgroup's
constexpr uint32_t ring_words_per_wave(bool wide) { return kCutRing * cut_tile_words(wide); }
constexpr uint32_t ring_words(bool wide) { return (kWorkgroup / 64u) * ring_words_per_wave(wide); }

// `t_min`: where the intervals of the kernel's queries start; `time_scale`: see TimeFrame (0: from the scene's extent).
template <bool kAllLds, bool kWide = false>
__device__ __forceinline__ LdsImage stage_scene(const SceneDev& sc, const CameraDev* cam, float4* smem, uint32_t wg, double t_min,
                                                float time_scale = 0.0f) {
  lds_double* lcam = (lds_double*)smem;
  lds_double* ltrig = lcam + CAM_DOUBLES;
  lds_double2* lfirst = (lds_double2*)(ltrig + 2 * kTrigEntries);
  lds_u32* ring = (lds_u32*)(lfirst + 2);
  lds_float4* lnodes = (lds_float4*)(ring + ring_words(kWide));
  lds_double2* lsph = (lds_double2*)(lnodes + (kWide ? kWideLdsStride / 16 : kLdsNodeQuads) * sc.n_nodes_lds);
  lds_double2* laux = lsph + 2 * sc.n_spheres_lds;
  lds_double2* lmat = laux + sc.n_aux_lds;
  lds_u32* stack_base = (lds_u32*)(lmat + 2 * sc.n_materials_lds) + (kWide ? 0u : kStackSpareRows * wg);
  const f32x4* gn = reinterpret_cast<const f32x4*>(sc.nodes);
  const f64x2* gs = reinterpret_cast<const f64x2*>(sc.spheres);
  const f64x2* ga = reinterpret_cast<const f64x2*>(sc.aux);
  const f64x2* gm = reinterpret_cast<const f64x2*>(sc.materials);
  if (cam && threadIdx.x < CAM_DOUBLES) {
    const uint32_t k = threadIdx.x;
    const double* src = k < 3 ? cam->center : k < 6 ? cam->pixel00 : k < 9 ? cam->du : k < 12 ? cam->dv : k < 15 ? cam->disk_u : cam->disk_v;
    lcam[k] = src[k % 3];
  }
  if (threadIdx.x < 2 && sc.first_leaf != kRefDone) lfirst[threadIdx.x] = gs[2 * sc.first_leaf + threadIdx.x];
  for (uint32_t i = threadIdx.x; i < kTrigEntries; i += wg) {
    double sn, cs;
    sincos_2pi((double)i * (1.0 / kTrigEntries), sn, cs);
    ltrig[2 * i] = sn;
    ltrig[2 * i + 1] = cs;
  }
  for (uint32_t i = threadIdx.x; i < ring_words(kWide); i += wg) ring[i] = kRefDone;
  if (kWide) {  // four-wide nodes are staged as they are, 80 bytes apart
    for (uint32_t i = threadIdx.x; i < kWideStagedQuads * sc.n_nodes_lds; i += wg) {
      const uint32_t node = i / kWideStagedQuads, quad = i - node * kWideStagedQuads;
      lnodes[(kWideLdsStride / 16) * node + quad] = gn[kWideGlobalQuads * node + quad];
    }
  } else {
    for (uint32_t i = threadIdx.x; i < sc.n_nodes_lds; i += wg) stage_node<kAllLds>(lnodes, i, gn[4 * i], gn[4 * i + 1], gn[4 * i + 2], gn[4 * i + 3]);
  }
  for (uint32_t i = threadIdx.x; i < 2 * sc.n_spheres_lds; i += wg) lsph[i] = gs[i];
  for (uint32_t i = threadIdx.x; i < sc.n_aux_lds; i += wg) laux[i] = ga[i];
  for (uint32_t i = threadIdx.x; i < 2 * sc.n_materials_lds; i += wg) lmat[i] = gm[i];
  __syncthreads();
  LdsImage im;
  im.tv.gnodes = sc.nodes;
  im.tv.lnodes = lnodes;
  im.tv.gsph = sc.spheres;
  im.tv.lsph = lsph;
  im.tv.gaux = sc.aux;
  im.tv.laux = laux;
  im.tv.gmat = sc.materials;
  im.tv.lmat = lmat;
  im.tv.n_lds = sc.n_nodes_lds;
  im.tv.n_sph_lds = sc.n_spheres_lds;
  im.tv.n_aux_lds = sc.n_aux_lds;
  im.tv.n_mat_lds = sc.n_materials_lds;
  // a tree that is in LDS as a whole is walked by LDS addresses (see stage_node)
  im.tv.root = (kAllLds && !kWide && !(sc.root & kRefLeaf)) ? lds_node_address(lnodes, sc.root) : sc.root;
  im.tv.first_leaf = sc.first_leaf;
  im.tv.lfirst = lfirst;
  im.tv.bound = sc.bound;
  im.tv.pad_bound = uniform_f32(pad_for_bound(sc.bound));
  im.tv.time = uniform_frame(time_frame(t_min, time_scale > 0.0f ? time_scale : time_scale_for(sc.bound)));
  im.tv.single_leaves = sc.single_leaves != 0u;
  im.cam.p = lcam;
  im.trig.tab = ltrig;
  im.ring = ring;
  im.stack_base = stack_base;
  return im;
}

}  // namespace rta
