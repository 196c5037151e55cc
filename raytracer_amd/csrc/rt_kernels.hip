// gfx950 kernels of the render hot path and their launchers.
#include "rt_launch.h"

#include "rt_device.h"

namespace rta {

// LDS image of a workgroup: [top-of-tree nodes][sphere slots][per-lane stacks].
__device__ __forceinline__ TreeView stage_tree(const SceneDev& sc, float4* smem, uint32_t*& stack_base, uint32_t wg) {
  float4* lnodes = smem;
  double2* lsph = reinterpret_cast<double2*>(smem + 2 * sc.n_nodes_lds);
  stack_base = reinterpret_cast<uint32_t*>(lsph + 2 * sc.n_spheres_lds);
  for (uint32_t i = threadIdx.x; i < 2 * sc.n_nodes_lds; i += wg) lnodes[i] = sc.nodes[i];
  for (uint32_t i = threadIdx.x; i < 2 * sc.n_spheres_lds; i += wg) lsph[i] = sc.spheres[i];
  __syncthreads();
  TreeView tv;
  tv.gnodes = sc.nodes;
  tv.lnodes = lnodes;
  tv.gsph = sc.spheres;
  tv.lsph = lsph;
  tv.n_nodes = sc.n_nodes;
  tv.n_lds = sc.n_nodes_lds;
  tv.n_sph_lds = sc.n_spheres_lds;
  return tv;
}

// ------------------------------------------------------------ trace batch ----
__global__ __launch_bounds__(kWorkgroup) void trace_batch_kernel(SceneDev sc, const double* __restrict__ org, const double* __restrict__ dir,
                                                                 size_t n, double t_min, double t_max, int32_t* __restrict__ prim,
                                                                 double* __restrict__ t_out, double* __restrict__ normal,
                                                                 uint8_t* __restrict__ front, uint32_t* __restrict__ visits) {
  extern __shared__ float4 smem[];
  uint32_t* stack_base;
  TreeView tv = stage_tree(sc, smem, stack_base, kWorkgroup);
  uint32_t* stack = stack_base + threadIdx.x;
  for (size_t i = (size_t)blockIdx.x * kWorkgroup + threadIdx.x; i < n; i += (size_t)gridDim.x * kWorkgroup) {
    D3 o = d3(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
    D3 d = normalize(d3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));  // Ray::new, src/ray.rs:16-23
    double t;
    uint32_t nv = 0;
    int slot = traverse(tv, stack, kWorkgroup, o, d, t_min, t_max, t, nv);
    if (slot >= 0) {
      Hit h = make_hit(tv, sc.aux, slot, o, d, t);
      prim[i] = (int32_t)h.prim;
      t_out[i] = t;
      normal[3 * i] = h.normal.x;
      normal[3 * i + 1] = h.normal.y;
      normal[3 * i + 2] = h.normal.z;
      front[i] = h.front ? 1 : 0;
    } else {
      prim[i] = -1;
      t_out[i] = __builtin_inf();
      normal[3 * i] = normal[3 * i + 1] = normal[3 * i + 2] = 0.0;
      front[i] = 0;
    }
    if (visits) visits[i] = nv;
  }
}

// ---------------------------------------------------------- scatter batch ----
__global__ __launch_bounds__(256) void scatter_batch_kernel(const double2* __restrict__ materials, const uint32_t* __restrict__ mat,
                                                            const double* __restrict__ in_dir, const double* __restrict__ p,
                                                            const double* __restrict__ normal, const uint8_t* __restrict__ front,
                                                            const double* __restrict__ u, size_t n, uint8_t* __restrict__ scattered,
                                                            double* __restrict__ attenuation, double* __restrict__ out_dir) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  D3 d = normalize(d3(in_dir[3 * i], in_dir[3 * i + 1], in_dir[3 * i + 2]));
  Hit h;
  h.p = d3(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
  h.normal = d3(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
  h.front = front[i] != 0;
  h.material = mat[i];
  h.prim = 0;
  D3 att, od;
  bool ok = scatter(materials, h.material, d, h, u[2 * i], u[2 * i + 1], att, od);
  scattered[i] = ok ? 1 : 0;
  if (!ok) att = od = d3(0.0, 0.0, 0.0);
  attenuation[3 * i] = att.x;
  attenuation[3 * i + 1] = att.y;
  attenuation[3 * i + 2] = att.z;
  out_dir[3 * i] = od.x;
  out_dir[3 * i + 1] = od.y;
  out_dir[3 * i + 2] = od.z;
}

// ------------------------------------------------------------- slab batch ----
// boxes: n x 2 float4 (already rounded outwards by the host).
__global__ __launch_bounds__(256) void slab_batch_kernel(const float4* __restrict__ boxes, const double* __restrict__ org,
                                                         const double* __restrict__ dir, const double* __restrict__ interval, size_t n,
                                                         uint8_t* __restrict__ hit) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  D3 o = d3(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
  D3 d = normalize(d3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
  RayF32 q = ray_f32(o, d);
  double t_min = interval[2 * i], t_max = interval[2 * i + 1];
  float tminf = f32_below(t_min);
  float tmaxf = (t_max < 3.0e38) ? f32_above(t_max) : __builtin_inff();
  hit[i] = node_test(q, d.x < 0.0, d.y < 0.0, d.z < 0.0, boxes[2 * i], boxes[2 * i + 1], tminf, tmaxf) ? 1 : 0;
}

// ----------------------------------------------------------- primary rays ----
__global__ __launch_bounds__(256) void primary_rays_kernel(CameraDev cam, uint32_t k0, uint32_t k1, const uint32_t* __restrict__ ijs,
                                                           size_t n, double* __restrict__ org, double* __restrict__ dir) {
  size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  D3 o, d;
  camera_ray(cam, k0, k1, ijs[3 * k], ijs[3 * k + 1], ijs[3 * k + 2], o, d);
  org[3 * k] = o.x;
  org[3 * k + 1] = o.y;
  org[3 * k + 2] = o.z;
  dir[3 * k] = d.x;
  dir[3 * k + 1] = d.y;
  dir[3 * k + 2] = d.z;
}

// ------------------------------------------------------------------ render ----
// The per-pixel closure of Camera::render (src/camera.rs:144-153) with
// Camera::ray_colour (src/camera.rs:98-112) unrolled into a loop.
//
// Persistent workgroups: each stages the tree in LDS once, then its waves
// pull work items (an 8x8 pixel tile x a chunk of the sample range) from a
// global counter.  A lane owns one pixel of the tile; it starts its next
// sample as soon as its current path ends, so lanes of a wave do not wait for
// each other between samples.  Each lane adds its samples in index order, so
// the result does not depend on how items are scheduled.
__global__ __launch_bounds__(kWorkgroup) void render_kernel(RenderParams P) {
  extern __shared__ float4 smem[];
  uint32_t* stack_base;
  const TreeView tv = stage_tree(P.sc, smem, stack_base, kWorkgroup);
  uint32_t* stack = stack_base + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t tile_w = P.x1 - P.x0;

  for (;;) {
    uint32_t item = 0;
    if (lane == 0) item = (uint32_t)atomicAdd(&P.counters[0], 1ull);
    item = __builtin_amdgcn_readfirstlane(item);
    if (item >= P.n_items) break;
    const uint32_t tile = item / P.n_chunks, chunk = item - tile * P.n_chunks;
    const uint32_t ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const uint32_t i = P.x0 + tx * 8u + (lane & 7u), j = P.y0 + ty * 8u + (lane >> 3);
    const bool valid = i < P.x1 && j < P.y1;
    const uint32_t s_first = P.spp_begin + chunk * P.chunk_spp;
    const uint32_t s_last = min(s_first + P.chunk_spp, P.spp_end);
    const uint32_t pixel = j * P.cam.width + i;

    uint32_t s = valid ? s_first : s_last;
    uint32_t n_rays = 0, n_samples = 0, n_visits = 0;
    D3 acc = d3(0.0, 0.0, 0.0);
    D3 o = d3(0.0, 0.0, 0.0), d = d3(0.0, 0.0, 1.0), thr = d3(1.0, 1.0, 1.0);
    uint32_t depth_left = 0, bounce = 0;
    while (depth_left != 0 || s < s_last) {
      if (depth_left == 0) {  // start the next sample of this pixel
        camera_ray(P.cam, P.seed_lo, P.seed_hi, i, j, s, o, d);
        thr = d3(1.0, 1.0, 1.0);
        depth_left = P.cam.max_depth;
        bounce = 0;
        ++n_samples;
        if (depth_left == 0) {  // max_depth 0: every sample is black (src/camera.rs:99-101)
          ++s;
          continue;
        }
      }
      double t;
      const int slot = traverse(tv, stack, kWorkgroup, o, d, 1e-3, __builtin_inf(), t, n_visits);  // src/camera.rs:102
      ++n_rays;
      if (slot < 0) {  // sky, src/camera.rs:103-107
        double a = (d.y + 1.0) / 2.0;
        D3 sky = d3((1.0 - a) + a * 0.5, (1.0 - a) + a * 0.7, (1.0 - a) + a * 1.0);
        acc = acc + thr * sky;
        depth_left = 0;
        ++s;
        continue;
      }
      const Hit h = make_hit(tv, P.sc.aux, slot, o, d, t);
      double u0, u1;
      philox_pair(P.seed_lo, P.seed_hi, pixel, s, bounce, DOM_SCATTER, u0, u1);
      D3 att, nd;
      if (!scatter(P.sc.materials, h.material, d, h, u0, u1, att, nd)) {  // absorbed, src/camera.rs:108-110
        depth_left = 0;
        ++s;
        continue;
      }
      thr = thr * att;
      o = h.p;  // HitRecord::ray, src/hit.rs:29-31
      d = nd;
      ++bounce;
      if (--depth_left == 0) ++s;  // depth exhausted: contributes nothing (src/camera.rs:99-101)
    }

    if (valid) {
      if (P.direct) {
        double* px = P.out + 3 * ((size_t)j * P.cam.width + i);
        px[0] += acc.x;
        px[1] += acc.y;
        px[2] += acc.z;
      } else {
        double* px = P.out + 3 * ((size_t)chunk * tile_w * (P.y1 - P.y0) + (size_t)(j - P.y0) * tile_w + (i - P.x0));
        px[0] = acc.x;
        px[1] = acc.y;
        px[2] = acc.z;
      }
    }
    // one counter update per wave and item
    for (int off = 32; off > 0; off >>= 1) {
      n_rays += __shfl_down(n_rays, off);
      n_samples += __shfl_down(n_samples, off);
    }
    if (lane == 0) {
      atomicAdd(&P.counters[1], (unsigned long long)n_rays);
      atomicAdd(&P.counters[2], (unsigned long long)n_samples);
    }
  }
}

// Adds the chunks' partial sums to the image in chunk order.
__global__ __launch_bounds__(256) void reduce_chunks_kernel(const double* __restrict__ partial, double* __restrict__ out, uint32_t n_chunks,
                                                            uint32_t x0, uint32_t y0, uint32_t tile_w, uint32_t tile_h, uint32_t width) {
  size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t n = (size_t)tile_w * tile_h;
  if (k >= n) return;
  uint32_t y = (uint32_t)(k / tile_w), x = (uint32_t)(k - (size_t)y * tile_w);
  double r = 0.0, g = 0.0, b = 0.0;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    const double* p = partial + 3 * ((size_t)c * n + k);
    r += p[0];
    g += p[1];
    b += p[2];
  }
  double* px = out + 3 * ((size_t)(y0 + y) * width + (x0 + x));
  px[0] += r;
  px[1] += g;
  px[2] += b;
}

// --------------------------------------------------------------- launchers ----
size_t tree_lds_bytes(const SceneDev& sc, uint32_t wg) {
  return (size_t)sc.n_nodes_lds * 32 + (size_t)sc.n_spheres_lds * 32 + (size_t)sc.stack_cap * wg * 4;
}

static hipError_t allow_lds(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t launch_trace_batch(const SceneDev& sc, const double* org, const double* dir, size_t n, double t_min, double t_max, int32_t* prim,
                              double* t, double* normal, uint8_t* front, uint32_t* visits, int max_blocks, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  size_t lds = tree_lds_bytes(sc, kWorkgroup);
  hipError_t e = allow_lds((const void*)trace_batch_kernel, lds);
  if (e != hipSuccess) return e;
  size_t blocks = (n + kWorkgroup - 1) / kWorkgroup;
  if (blocks > (size_t)max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(trace_batch_kernel, dim3((unsigned)blocks), dim3(kWorkgroup), lds, stream, sc, org, dir, n, t_min, t_max, prim, t, normal,
                     front, visits);
  return hipGetLastError();
}

hipError_t launch_scatter_batch(const SceneDev& sc, const uint32_t* mat, const double* in_dir, const double* p, const double* normal,
                                const uint8_t* front, const double* u, size_t n, uint8_t* scattered, double* attenuation, double* out_dir,
                                hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(scatter_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, sc.materials, mat, in_dir, p, normal,
                     front, u, n, scattered, attenuation, out_dir);
  return hipGetLastError();
}

hipError_t launch_slab_batch(const float* boxes, const double* org, const double* dir, const double* interval, size_t n, uint8_t* hit,
                             hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(slab_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float4*)boxes, org, dir, interval, n,
                     hit);
  return hipGetLastError();
}

hipError_t launch_primary_rays(const CameraDev& cam, uint64_t seed, const uint32_t* ijs, size_t n, double* org, double* dir,
                               hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(primary_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, cam, (uint32_t)seed,
                     (uint32_t)(seed >> 32), ijs, n, org, dir);
  return hipGetLastError();
}

hipError_t launch_render(const RenderParams& P, int blocks, hipStream_t stream) {
  size_t lds = tree_lds_bytes(P.sc, kWorkgroup);
  hipError_t e = allow_lds((const void*)render_kernel, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(render_kernel, dim3((unsigned)blocks), dim3(kWorkgroup), lds, stream, P);
  return hipGetLastError();
}

hipError_t launch_reduce_chunks(const double* partial, double* out, uint32_t n_chunks, uint32_t x0, uint32_t y0, uint32_t tile_w,
                                uint32_t tile_h, uint32_t width, hipStream_t stream) {
  size_t n = (size_t)tile_w * tile_h;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, partial, out, n_chunks, x0, y0, tile_w,
                     tile_h, width);
  return hipGetLastError();
}

}  // namespace rta
