// A wavefront path tracer over the same device functions as the render kernel, kept for measurement:
// one kernel per stage with ray queues in HBM, so that its counters (lane use, HBM traffic, occupancy) can be
// laid next to the persistent kernel's.  It renders the same paths (same Philox counters), hence the same frame
// up to the order of additions; rt_render* never calls it.
//
//   generate   camera rays of a batch of paths (all pixels x a few samples)        -> queue A
//   trace      closest hit of every queued ray (one ray per lane, to completion)     -> hit records
//   shade      sky for rays that left the scene, else hit record + scatter; rays that go on are packed into the
//              other queue (ballot + prefix count per wave, one atomic per workgroup) -> queue B, per-path results
//   resolve    adds each pixel's path results in sample order to the frame
//
// Queues are arrays of one component each (coalesced): origin, direction, throughput (nine doubles), path id.
#include "rt_launch.h"

#include <algorithm>
#include <vector>

#include "rt_device.h"

namespace rta {

struct WfQueue {
  double* v[9];   // ox oy oz dx dy dz tx ty tz, `capacity` entries each
  uint32_t* path; // path id: sample_in_batch * n_pixels + pixel
};
struct WfParams {
  SceneDev sc;
  CameraDev cam;
  WfQueue in, out;
  int32_t* hit_slot;
  double* hit_t;
  double* result;         // 3 x capacity: radiance of each finished path (zero until then)
  unsigned int* counts;   // [0]: entries of `in`, [1]: entries of `out`, [2..3]: rays traced (64-bit)
  uint32_t capacity, n_pixels, batch_spp, spp_begin;
  uint32_t seed_lo, seed_hi, bounce;
};

__global__ __launch_bounds__(256) void wf_generate_kernel(WfParams P) {
  const uint32_t n = P.n_pixels * P.batch_spp;
  for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
    const uint32_t s_local = k / P.n_pixels, pixel = k - s_local * P.n_pixels;
    const uint32_t j = pixel / P.cam.width, i = pixel - j * P.cam.width;
    D3 o, d;
    camera_ray(CamArgs{P.cam}, P.cam.width, P.cam.defocus != 0, P.seed_lo, P.seed_hi, i, j, P.spp_begin + s_local, o, d);
    P.in.v[0][k] = o.x;
    P.in.v[1][k] = o.y;
    P.in.v[2][k] = o.z;
    P.in.v[3][k] = d.x;
    P.in.v[4][k] = d.y;
    P.in.v[5][k] = d.z;
    P.in.v[6][k] = 1.0;
    P.in.v[7][k] = 1.0;
    P.in.v[8][k] = 1.0;
    P.in.path[k] = k;
    P.result[3 * (size_t)k] = P.result[3 * (size_t)k + 1] = P.result[3 * (size_t)k + 2] = 0.0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    P.counts[0] = n;
    P.counts[1] = 0;
  }
}

template <bool kAllLds, bool kWide>
__global__ __launch_bounds__(kWorkgroup) void wf_trace_kernel(WfParams P) {
  extern __shared__ float4 smem[];
  const LdsImage im = stage_scene<kAllLds, kWide>(P.sc, nullptr, smem, kWorkgroup, 1e-3);
  const TreeView& tv = im.tv;
  const uint32_t n = P.counts[0];
  const uint32_t gtid = blockIdx.x * kWorkgroup + threadIdx.x;
  for (uint32_t k = gtid; k < n; k += gridDim.x * kWorkgroup) {
    const D3 o = d3(P.in.v[0][k], P.in.v[1][k], P.in.v[2][k]), d = d3(P.in.v[3][k], P.in.v[4][k], P.in.v[5][k]);
    Trav tr;
    uint32_t sp0 = 0u;
    if constexpr (!kWide) sp0 = stack_origin(im.stack_base + threadIdx.x);
    trav_begin<kWide>(tr, tv, o, d, __builtin_inf(), sp0);
    uint32_t n_steps = 0, n_tests = trav_first<kAllLds>(tr, tv, 1e-3);
    if constexpr (kWide)
      trav_run_wide(tr, tv, HybridStack{im.stack_base + threadIdx.x, P.sc.gstack + gtid, P.sc.lds_stack, P.sc.gstack_stride}, 1e-3, n_steps, n_tests);
    else
      trav_run<kAllLds>(tr, tv, im.stack_base + threadIdx.x, kWorkgroup, 1e-3, n_steps, n_tests);
    P.hit_slot[k] = tr.best;
    P.hit_t[k] = tr.t_best;
  }
  if (gtid == 0) atomicAdd(reinterpret_cast<unsigned long long*>(P.counts + 2), (unsigned long long)n);
}

constexpr uint32_t kShadeBlock = 1024;
__global__ __launch_bounds__(kShadeBlock) void wf_shade_kernel(WfParams P) {
  __shared__ uint32_t wave_count[kShadeBlock / 64];
  __shared__ uint32_t block_base;
  const uint32_t n = P.counts[0];
  TreeView tv{};  // scene data from global memory (L2): no LDS image in this kernel
  tv.gsph = P.sc.spheres;
  tv.gaux = P.sc.aux;
  tv.gmat = P.sc.materials;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t rounds = (n + gridDim.x * kShadeBlock - 1) / (gridDim.x * kShadeBlock);
  for (uint32_t r = 0; r < rounds; ++r) {  // every lane makes every round, so that the ballots and barriers below see whole workgroups
    const uint32_t k = (r * gridDim.x + blockIdx.x) * kShadeBlock + threadIdx.x;
    bool goes_on = false;
    D3 no, nd, nthr;
    uint32_t path = 0;
    if (k < n) {
      const D3 o = d3(P.in.v[0][k], P.in.v[1][k], P.in.v[2][k]), d = d3(P.in.v[3][k], P.in.v[4][k], P.in.v[5][k]);
      const D3 thr = d3(P.in.v[6][k], P.in.v[7][k], P.in.v[8][k]);
      path = P.in.path[k];
      const int slot = P.hit_slot[k];
      if (slot < 0) {  // the sky, src/camera.rs:103-107
        const double a = (d.y + 1.0) / 2.0;
        double* res = P.result + 3 * (size_t)path;
        res[0] = thr.x * ((1.0 - a) + a * 0.5);
        res[1] = thr.y * ((1.0 - a) + a * 0.7);
        res[2] = thr.z * ((1.0 - a) + a * 1.0);
      } else {
        const Hit h = make_hit<false>(tv, slot, o, d, P.hit_t[k]);
        const uint32_t s_local = path / P.n_pixels, pixel = path - s_local * P.n_pixels;
        double u0, u1;
        philox_pair(P.seed_lo, P.seed_hi, pixel, P.spp_begin + s_local, P.bounce, DOM_SCATTER, u0, u1);
        D3 att;
        if (scatter<false>(tv, h.material, d, h, u0, u1, att, nd) && P.bounce + 1u < P.cam.max_depth) {
          goes_on = true;
          no = h.p;
          nthr = thr * att;
        }
      }
    }
    // pack the rays that go on: position = the workgroup's base (one atomic per workgroup and round) + the counts of
    // the waves below this one + the number of such lanes below this one
    const unsigned long long m = __ballot(goes_on);
    if (lane == 0) wave_count[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t total = 0;
      for (uint32_t w = 0; w < kShadeBlock / 64; ++w) {
        const uint32_t c = wave_count[w];
        wave_count[w] = total;
        total += c;
      }
      block_base = total ? atomicAdd(&P.counts[1], total) : 0u;
    }
    __syncthreads();
    const uint32_t base = block_base + wave_count[wave];
    __syncthreads();  // the counts are rewritten in the next round
    if (goes_on) {
      const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      P.out.v[0][at] = no.x;
      P.out.v[1][at] = no.y;
      P.out.v[2][at] = no.z;
      P.out.v[3][at] = nd.x;
      P.out.v[4][at] = nd.y;
      P.out.v[5][at] = nd.z;
      P.out.v[6][at] = nthr.x;
      P.out.v[7][at] = nthr.y;
      P.out.v[8][at] = nthr.z;
      P.out.path[at] = path;
    }
  }
}

// Between bounces: the packed queue becomes the input, the other one is emptied.
__global__ void wf_swap_kernel(unsigned int* counts) {
  counts[0] = counts[1];
  counts[1] = 0;
}

__global__ __launch_bounds__(256) void wf_resolve_kernel(const double* __restrict__ result, double* __restrict__ frame, uint32_t n_pixels,
                                                         uint32_t batch_spp) {
  const uint32_t pixel = blockIdx.x * 256 + threadIdx.x;
  if (pixel >= n_pixels) return;
  double r = 0.0, g = 0.0, b = 0.0;
  for (uint32_t s = 0; s < batch_spp; ++s) {
    const double* p = result + 3 * ((size_t)s * n_pixels + pixel);
    r += p[0];
    g += p[1];
    b += p[2];
  }
  frame[3 * (size_t)pixel] += r;
  frame[3 * (size_t)pixel + 1] += g;
  frame[3 * (size_t)pixel + 2] += b;
}

size_t wavefront_workspace_bytes(uint32_t capacity) { return (size_t)capacity * (2 * (9 * 8 + 4) + 4 + 8 + 24) + 256; }

// Renders samples [spp_begin, spp_end) of the whole frame into `frame` (device, added to) with the wavefront
// pipeline; `workspace` holds wavefront_workspace_bytes(capacity) bytes, capacity >= width x height.
// rays_out (host) receives the number of rays traced.  Waits for the stream.
hipError_t wavefront_render(const SceneDev& sc_in, const CameraDev& cam, uint64_t seed, uint32_t spp_begin, uint32_t spp_end, double* frame,
                            void* workspace, uint32_t capacity, int num_cus, hipStream_t stream, uint64_t* rays_out, uint32_t* launches_out) {
  WfParams P{};
  P.sc = sc_in;
  P.cam = cam;
  P.capacity = capacity;
  P.n_pixels = cam.width * cam.height;
  P.seed_lo = (uint32_t)seed;
  P.seed_hi = (uint32_t)(seed >> 32);
  char* w = static_cast<char*>(workspace);
  P.counts = reinterpret_cast<unsigned int*>(w);
  w += 256;
  for (WfQueue* q : {&P.in, &P.out}) {
    for (int c = 0; c < 9; ++c) {
      q->v[c] = reinterpret_cast<double*>(w);
      w += (size_t)capacity * 8;
    }
  }
  P.hit_t = reinterpret_cast<double*>(w);
  w += (size_t)capacity * 8;
  P.result = reinterpret_cast<double*>(w);
  w += (size_t)capacity * 24;
  P.in.path = reinterpret_cast<uint32_t*>(w);
  w += (size_t)capacity * 4;
  P.out.path = reinterpret_cast<uint32_t*>(w);
  w += (size_t)capacity * 4;
  P.hit_slot = reinterpret_cast<int32_t*>(w);

  const bool all_lds = !P.sc.wide;
  void (*trace)(WfParams) = P.sc.wide ? wf_trace_kernel<false, true> : wf_trace_kernel<true, false>;
  if (!P.sc.wide && !P.sc.stack_in_lds) return hipErrorInvalidValue;  // (trees too deep for LDS stacks: not covered by the experiment)
  (void)all_lds;
  const size_t lds = tree_lds_bytes(P.sc);
  hipError_t e = lds > 64 * 1024 ? hipFuncSetAttribute((const void*)trace, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) : hipSuccess;
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(P.counts, 0, 256, stream);
  if (e != hipSuccess) return e;
  const uint32_t max_batch = capacity / P.n_pixels;
  uint32_t launches = 0;
  std::vector<unsigned int> host_counts(4);
  for (uint32_t s0 = spp_begin; s0 < spp_end; s0 += max_batch) {
    P.batch_spp = std::min(max_batch, spp_end - s0);
    P.spp_begin = s0;
    const uint32_t n = P.n_pixels * P.batch_spp;
    hipLaunchKernelGGL(wf_generate_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 256 * 32)), dim3(256), 0, stream, P);
    ++launches;
    for (uint32_t bounce = 0; bounce < cam.max_depth; ++bounce) {
      P.bounce = bounce;
      hipLaunchKernelGGL(trace, dim3(num_cus), dim3(kWorkgroup), lds, stream, P);
      hipLaunchKernelGGL(wf_shade_kernel, dim3(num_cus * 2), dim3(kShadeBlock), 0, stream, P);
      hipLaunchKernelGGL(wf_swap_kernel, dim3(1), dim3(1), 0, stream, P.counts);
      launches += 3;
      std::swap(P.in, P.out);
      if ((bounce & 3u) == 3u) {  // every few bounces: has the queue run dry?
        e = hipMemcpyAsync(host_counts.data(), P.counts, 16, hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        if (host_counts[0] == 0) break;
      }
    }
    hipLaunchKernelGGL(wf_resolve_kernel, dim3((P.n_pixels + 255) / 256), dim3(256), 0, stream, P.result, frame, P.n_pixels, P.batch_spp);
    ++launches;
  }
  e = hipMemcpyAsync(host_counts.data(), P.counts, 16, hipMemcpyDeviceToHost, stream);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return e;
  *rays_out = (uint64_t)host_counts[2] | ((uint64_t)host_counts[3] << 32);
  *launches_out = launches;
  return hipGetLastError();
}

}  // namespace rta
