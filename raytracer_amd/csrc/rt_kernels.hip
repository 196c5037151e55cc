// gfx950 kernels of the render hot path and their launchers.
#include "rt_launch.h"

#include <mutex>

#include "rt_device.h"

namespace rta {

// ------------------------------------------------------------ trace batch ----
template <bool kAllLds, bool kGStack, bool kWide = false>
__global__ __launch_bounds__(kWorkgroup) void trace_batch_kernel(SceneDev sc, const double* __restrict__ org, const double* __restrict__ dir,
                                                                 size_t n, double t_min, double t_max, int32_t* __restrict__ prim,
                                                                 double* __restrict__ t_out, double* __restrict__ normal,
                                                                 uint8_t* __restrict__ front, uint32_t* __restrict__ counts) {
  extern __shared__ float4 smem[];
  const LdsImage im = stage_scene<kAllLds, kWide>(sc, nullptr, smem, kWorkgroup, t_min);
  const TreeView& tv = im.tv;
  const uint32_t gtid = blockIdx.x * kWorkgroup + threadIdx.x;
  for (size_t i = (size_t)blockIdx.x * kWorkgroup + threadIdx.x; i < n; i += (size_t)gridDim.x * kWorkgroup) {
    D3 o = d3(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
    D3 d = normalize(d3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));  // Ray::new, src/ray.rs:16-23
    Trav tr;
    uint32_t sp0 = 0u;
    if constexpr (!kWide) sp0 = kGStack ? 0u : stack_origin(im.stack_base + threadIdx.x);
    trav_begin<kWide>(tr, tv, o, d, t_max, sp0);
    uint32_t n_steps = 0, n_tests = trav_first<kAllLds>(tr, tv, t_min);
    if constexpr (kWide)
      trav_run_wide(tr, tv, HybridStack{im.stack_base + threadIdx.x, sc.gstack + gtid, sc.lds_stack, sc.gstack_stride}, t_min, n_steps, n_tests);
    else if constexpr (kGStack)
      trav_run<kAllLds>(tr, tv, sc.gstack + gtid, sc.gstack_stride, t_min, n_steps, n_tests);
    else
      trav_run<kAllLds>(tr, tv, im.stack_base + threadIdx.x, kWorkgroup, t_min, n_steps, n_tests);
    if (tr.best >= 0) {
      Hit h = make_hit<kAllLds>(tv, tr.best, o, d, tr.t_best);
      prim[i] = (int32_t)h.prim;
      t_out[i] = tr.t_best;
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
    if (counts) {
      counts[2 * i] = n_steps;
      counts[2 * i + 1] = n_tests;
    }
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
  TreeView tv{};  // only the global material table is used
  tv.gmat = materials;
  D3 d = normalize(d3(in_dir[3 * i], in_dir[3 * i + 1], in_dir[3 * i + 2]));
  Hit h;
  h.p = d3(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
  h.normal = d3(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
  h.front = front[i] != 0;
  h.material = mat[i];
  h.prim = 0;
  D3 att, od;
  bool ok = scatter<false>(tv, h.material, d, h, u[2 * i], u[2 * i + 1], att, od);
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
// boxes: n x 2 float4 = (lo.xyz, bound) (hi.xyz, -), already rounded outwards by the host.
__global__ __launch_bounds__(256) void slab_batch_kernel(const float4* __restrict__ boxes, const double* __restrict__ org,
                                                         const double* __restrict__ dir, const double* __restrict__ interval, size_t n,
                                                         uint8_t* __restrict__ hit) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  D3 o = d3(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
  D3 d = normalize(d3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
  const float4 lo = boxes[2 * i], hi = boxes[2 * i + 1];
  double t_min = interval[2 * i], t_max = interval[2 * i + 1];
  // the ray's time frame starts at the interval's start; scaled for a scene of the box's own extent
  const TimeFrame frame = time_frame(t_min, time_scale_for(lo.w));
  RayBox q = ray_box(o, d, pad_for_bound(lo.w), frame);
  float tn;
  hit[i] = box_test(q, f32x2{lo.x, hi.x}, f32x2{lo.y, hi.y}, f32x2{lo.z, hi.z}, 0.0f, tau_above(frame, t_max), tn) ? 1 : 0;
}

// ----------------------------------------------------------- primary rays ----
__global__ __launch_bounds__(256) void primary_rays_kernel(CameraDev cam, uint32_t k0, uint32_t k1, const uint32_t* __restrict__ ijs,
                                                           size_t n, double* __restrict__ org, double* __restrict__ dir) {
  size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  D3 o, d;
  camera_ray(CamArgs{cam}, cam.width, cam.defocus != 0, k0, k1, ijs[3 * k], ijs[3 * k + 1], ijs[3 * k + 2], o, d);
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
// Persistent workgroups, one per CU: each stages the scene in LDS once, then
// its lanes pull units of work (a pixel x a chunk of the sample range) from a
// global counter, 8x8 pixel tiles in order.  One lane walks through a unit's
// samples in index order, so the sums do not depend on scheduling.
//
// Every lane is a small state machine that wants exactly one of four code
// blocks next: CAMERA (finish a ray that left the scene, start the next sample),
// INNER (inner BVH nodes: test both child boxes), LEAF (the spheres of one leaf)
// or SHADE (closest hit known: scatter, then the next ray or the end of the
// path).  Path and traversal lengths differ from lane to lane, so instead of
// running every lane's traversal to completion before shading, the wave counts
// with ballots how many lanes want each block: traversal rounds (INNER if
// `inner_min` lanes want it, then LEAF if `leaf_min` lanes want it) repeat while
// either runs; otherwise the block most lanes are waiting for runs once.  Which
// block runs when never changes a lane's own sequence of operations, so results
// do not depend on the policy.
enum { BLK_CAMERA = 0, BLK_INNER = 1, BLK_LEAF = 2, BLK_SHADE = 3 };

// Lanes of the wave for which p holds, as a 32-bit scalar (the barrier keeps the
// compiler from carrying the count around as a 64-bit value).
__device__ __forceinline__ uint32_t count_lanes(bool p) {
  uint32_t n = (uint32_t)__builtin_popcountll(__ballot(p));
  asm volatile("" : "+s"(n));
  return n;
}

__device__ __forceinline__ uint32_t count_mask(unsigned long long m) {
  uint32_t n = (uint32_t)__builtin_popcountll(m);
  asm volatile("" : "+s"(n));
  return n;
}

// kProf: also count, per code block, how often it ran and with how many active lanes.
#define RTA_PROF(k)                                              \
  if (kProf) {                                                    \
    const unsigned long long m_ = __ballot(1);                    \
    if (lane == (uint32_t)__ffsll((long long)m_) - 1u) {          \
      prof_runs[k] += 1;                                          \
      prof_lanes[k] += (uint32_t)__popcll(m_);                    \
    }                                                             \
  }
// Shader-clock stamps around a block (profiling variant only).
#define RTA_T0()              \
  unsigned long long t0_ = 0; \
  if (kProf) t0_ = __builtin_amdgcn_s_memtime();
#define RTA_T1(k) \
  if (kProf) prof_cycles[k] += __builtin_amdgcn_s_memtime() - t0_;

template <bool kAllLds, bool kProf, bool kGStack, bool kSingle, bool kWide = false>
__global__ __launch_bounds__(kWorkgroup) void render_kernel(RenderParams P) {  // 16 waves per CU: at most 128 VGPRs
  extern __shared__ float4 smem[];
  const double t_min = 1e-3;  // src/camera.rs:102
  const LdsImage im = stage_scene<kAllLds, kWide>(P.sc, &P.cam, smem, kWorkgroup, t_min, P.time_scale);
  const TreeView& tv = im.tv;
  // The lane's stack column is recomputed where it is used (one add), not kept in a register.
  auto stack_of_lane = [&]() {
    if constexpr (kWide)
      return HybridStack{im.stack_base + threadIdx.x, P.sc.gstack + (blockIdx.x * kWorkgroup + threadIdx.x), P.sc.lds_stack, P.sc.gstack_stride};
    else if constexpr (kGStack)
      return P.sc.gstack + (blockIdx.x * kWorkgroup + threadIdx.x);
    else
      return im.stack_base + threadIdx.x;
  };
  const uint32_t stack_stride = kGStack ? P.sc.gstack_stride : kWorkgroup;
#define RTA_STACK stack_of_lane()
  auto empty_stack = [&]() {  // the stack pointer a traversal starts with
    if constexpr (kWide)
      return 0u;
    else
      return stack_origin(stack_of_lane());
  };
  const uint32_t lane = threadIdx.x & 63u;
  constexpr uint32_t kNoUnit = 0xffffffffu;
  // Camera rays start from their tile's entry list (rt_launch.h).  A wave keeps the lists of its last kCutRing batches in
  // LDS; a lane remembers which of them its unit belongs to in the top two bits of `unit`, kCutRing meaning none: a lane
  // still at work on a unit when the wave's ring comes round to that unit's place loses its list and starts its
  // remaining camera rays at the root.
  constexpr uint32_t kUnitChunk = 0x3fffffffu, kSlotShift = 30u;
  constexpr uint32_t kSlotWords = cut_tile_words(kWide);  // a batch's lists
  lds_u32* const my_ring = im.ring + (threadIdx.x >> 6) * ring_words_per_wave(kWide);
  uint32_t wave_slot = kCutRing, wave_serial = 0;  // the ring place of the wave's current batch; batches taken so far

  // A unit of work is one pixel's samples of one chunk of the sample range.  A wave takes a batch
  // (some or all chunks of one 8x8 tile) at a time from the global counter and its lanes share out
  // the batch's units (chunk by chunk, 64 pixels each) as they become free, so a lane whose pixel is cheap takes more units
  // instead of idling, and all lanes of a wave look at the same part of the scene.  The samples
  // of a unit are summed by one lane in index order and the chunks of a pixel are added in chunk
  // order afterwards, so the sums do not depend on which lane gets which unit or when.
  uint32_t unit = kNoUnit;      // the chunk index of the lane's unit (and in its top two bits where the unit's entry list is, see cut_slot), or kNoUnit
  uint32_t pij = 0;             // the unit's pixel: i | j << 16
  uint32_t s = 0, s_last = 0;   // next sample and end of the unit's sample range (s == s_last: unit finished)
#define RTA_I (pij & 0xffffu)
#define RTA_J (pij >> 16)
  uint32_t n_rays = 0, n_samples = 0, n_steps = 0, n_tests = 0;  // the last two are only kept by the profiling variant
  uint32_t prof_runs[4] = {0, 0, 0, 0}, prof_lanes[4] = {0, 0, 0, 0};
  unsigned long long prof_cycles[5] = {0, 0, 0, 0, 0};  // per block, and [4] = the whole kernel
  const unsigned long long kernel_t0 = kProf ? __builtin_amdgcn_s_memtime() : 0ull;
  D3 acc = d3(0.0, 0.0, 0.0), thr = d3(1.0, 1.0, 1.0);
  uint32_t depth_left = 0;
  Trav tr;
  // What a lane wants next is encoded in tr.cur: an inner node, a leaf, kRefDone (shade),
  // kRefCamera (start the next sample, fetching a unit first if need be) or kRefIdle (no units left).
  tr.cur = kRefCamera;

  // Where a finished unit's sum goes: added to the image, or stored in the unit's slot of the partial-sum buffer.
  auto flush = [&]() {
    const uint32_t i = RTA_I, j = RTA_J;
    if (P.direct) {
      double* px = P.out + 3 * ((size_t)j * P.cam.width + i);
      px[0] += acc.x;
      px[1] += acc.y;
      px[2] += acc.z;
    } else {
      // the partial sums hold the call's own tile rows only, packed: row (tile row / row_stride) * 8 + row within the tile
      const uint32_t tile_w = P.x1 - P.x0, rel = j - P.y0;
      const uint32_t row = P.row_stride == 1u ? rel : ((rel >> 3) / P.row_stride) * 8u + (rel & 7u);
      double* px = P.out + 3 * ((size_t)(unit & kUnitChunk) * tile_w * P.local_rows + (size_t)row * tile_w + (i - P.x0));
      px[0] = acc.x;
      px[1] = acc.y;
      px[2] = acc.z;
    }
  };

  // The wave alternates between a cold part (one SHADE or CAMERA block, which rewrite the whole
  // lane state) and a hot loop that only walks the tree (INNER and LEAF blocks touch the cursor,
  // the stack and the closest hit, nothing else), so the hot loop carries few live values around.
  // The wave's current batch: units [wave_next, wave_end) are still to be handed out; wave_base is the
  // first unit of the batch's tile and wave_origin the tile's top left pixel, i | j << 16.
  uint32_t wave_next = 0, wave_end = 0, wave_base = 0, wave_origin = 0;
  int block = BLK_CAMERA;
  for (;;) {
    {
      RTA_T0()
      if (block == BLK_SHADE) {
        if (tr.cur == kRefDone && tr.best >= 0) {  // a ray that missed everything is finished by the camera block
          RTA_PROF(BLK_SHADE)
          ++n_rays;
          const HitData hd = fetch_hit_data<kAllLds>(tv, tr.best);  // (asked for before the random numbers are worked out, used after)
          double u0, u1;
          philox_pair(P.seed_lo, P.seed_hi, RTA_J * P.cam.width + RTA_I, s, P.cam.max_depth - depth_left, DOM_SCATTER, u0, u1);
          const Hit h = make_hit_from(hd, tr.o, tr.d, tr.t_best);
          D3 att, nd;
          // None = absorbed (src/camera.rs:108-110); depth exhausted: the path contributes nothing (src/camera.rs:99-101).
          // Everything below is written whether the path goes on or not (what a finished path leaves behind is
          // overwritten when the lane's next sample starts), so the new ray is worked out where it is kept.
          const bool scattered = scatter<kAllLds>(tv, h.material, tr.d, h, u0, u1, att, nd, im.trig);
          thr = thr * att;
          --depth_left;
          trav_begin<kWide>(tr, tv, h.p, nd, __builtin_inf(), empty_stack());  // HitRecord::ray, src/hit.rs:29-31
          {
            const uint32_t first = trav_first<kAllLds>(tr, tv, t_min);
            if (kProf) n_tests += first;
          }
          if (!scattered || depth_left == 0) {
            ++s;
            tr.cur = kRefCamera;
          }
        }
      }
      if (block == BLK_CAMERA) {
        // The wave hands out the units of its current batch in order; when they run out it takes
        // the next batch from the global counter.  (Wave-uniform bookkeeping, done by all lanes.)
        // A ray that left the scene adds the sky (src/camera.rs:103-107) and ends its path here, in the
        // same block that starts the next one, instead of waiting for a shade round first.
        if (tr.cur == kRefDone && tr.best < 0) {
          ++n_rays;
          double a = (tr.d.y + 1.0) / 2.0;
          D3 sky = d3((1.0 - a) + a * 0.5, (1.0 - a) + a * 0.7, (1.0 - a) + a * 1.0);
          acc = acc + thr * sky;
          ++s;
          tr.cur = kRefCamera;
        }
        const uint32_t n_need = count_mask(__ballot(tr.cur == kRefCamera) & __ballot(s >= s_last));
        const uint32_t old_next = wave_next, old_avail = wave_end - wave_next, old_base = wave_base, old_origin = wave_origin, old_slot = wave_slot;
        uint32_t new_first = 0;
        if (n_need > old_avail) {
          uint32_t b = 0;
          if (lane == 0) b = (uint32_t)min(atomicAdd(&P.counters[0], 1ull), (unsigned long long)P.n_batches);
          b = __builtin_amdgcn_readfirstlane(b);  // b == n_batches: nothing left (tile n_tiles starts at n_units)
          const uint32_t t = b / P.batches_per_tile, c0 = (b - t * P.batches_per_tile) * P.batch_chunks;
          const uint32_t ty = t / P.tiles_x, tx = t - ty * P.tiles_x;  // ty: the call's own tile rows, row_stride apart in the rectangle
          wave_base = t * P.n_chunks * 64u;
          wave_origin = (P.x0 + tx * 8u) | ((P.y0 + (ty * P.row_stride + P.row_first) * 8u) << 16);
          const uint32_t first = min(wave_base + c0 * 64u, P.n_units);
          wave_end = min(wave_base + min(c0 + P.batch_chunks, P.n_chunks) * 64u, P.n_units);
          wave_next = min(first + (n_need - old_avail), wave_end);
          new_first = first;
          if (P.cuts) {
            wave_slot = wave_serial % kCutRing;
            ++wave_serial;
            if (unit != kNoUnit && (unit >> kSlotShift) == wave_slot) unit |= kCutRing << kSlotShift;  // (kCutRing == 3: both bits set)
            for (uint32_t w = lane; w < kSlotWords; w += 64u) {
              uint32_t e = t < P.n_tiles ? P.cuts[(size_t)t * kSlotWords + w] : kRefDone;
              if constexpr (kAllLds && !kWide)
                if (!(e & kRefLeaf)) e = lds_node_address(tv.lnodes, e);
              my_ring[wave_slot * kSlotWords + w] = e;
            }
            // (the lanes read it back below and in later rounds: LDS operations of one wave complete in order)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
        } else {
          wave_next += n_need;
        }
        if (tr.cur == kRefCamera) {
          RTA_PROF(BLK_CAMERA)
          if (s >= s_last) {  // the unit is finished (or there was none yet): hand in its sum and take the next one
            if (unit != kNoUnit) flush();
            const unsigned long long need = __ballot(1);
            const uint32_t k = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            const bool from_old = k < old_avail;
            const uint32_t rel = from_old ? old_next + k - old_base : new_first + (k - old_avail) - wave_base;  // unit within its tile
            if (from_old || wave_base + rel < wave_end) {
              const uint32_t origin = from_old ? old_origin : wave_origin;
              // 64 consecutive units are the tile's 8x8 pixels of one chunk, or one quarter of the tile (4x4 pixels) in four
              // chunks: the lanes of a wave then look at a quarter of the solid angle
              uint32_t chunk = rel >> 6, io = rel & 7u, jo = (rel >> 3) & 7u;
              if (P.quad_tiles) {
                io = (rel & 3u) | ((rel >> 4) & 4u);
                jo = ((rel >> 2) & 3u) | ((rel >> 5) & 4u);
                chunk = ((rel >> 4) & 3u) | ((rel >> 8) << 2);
              }
              const uint32_t i = (origin & 0xffffu) + io, j = (origin >> 16) + jo;
              unit = chunk | ((from_old ? old_slot : wave_slot) << kSlotShift);
              pij = i | (j << 16);
              acc = d3(0.0, 0.0, 0.0);
              if (i < P.x1 && j < P.y1) {
                s = P.spp_begin + chunk * P.chunk_spp;
                s_last = min(s + P.chunk_spp, P.spp_end);
                n_samples += s_last - s;
              } else {  // the tile sticks out of the rectangle: an empty unit
                unit = kNoUnit;
                s = s_last = 0;
              }
            } else {
              unit = kNoUnit;
              tr.cur = kRefIdle;
            }
          }
          if (tr.cur == kRefCamera && s < s_last) {  // Camera::get_ray, src/camera.rs:147-148
            D3 o, d;
            camera_ray(im.cam, P.cam.width, P.cam.defocus != 0, P.seed_lo, P.seed_hi, RTA_I, RTA_J, s, o, d, im.trig);
            thr = d3(1.0, 1.0, 1.0);
            depth_left = P.cam.max_depth;
            trav_begin<kWide>(tr, tv, o, d, __builtin_inf(), empty_stack());
            {
              const uint32_t first = trav_first<kAllLds>(tr, tv, t_min);
              if (kProf) n_tests += first;
            }
            if ((unit >> kSlotShift) < kCutRing) {  // the walk starts at the tile's entry list: the nearest entry now, the others set aside
              const lds_u32* list = my_ring + (unit >> kSlotShift) * kSlotWords;
              if constexpr (kWide) {
                // the entries whose own box the ray can hit are set aside, the farthest first; then the nearest is taken up.
                // A group of four entries is read and tested like a node (rt_launch.h).
                tr.cur = kRefDone;
#pragma unroll
                for (int g = (int)kWideCutEntries / 4 - 1; g >= 0; --g) {
                  const lds_u32* group = list + g * kWideCutGroupWords;
                  if (group[24] != kRefDone) {  // entries fill a list from the front: the group has any
                    const WideData w = wide_fetch_lds(tr, (lds_char*)group);
                    const WideTimes x = wide_times(tr, w);
                    if (x.h3) trav_push(tr, RTA_STACK, w.refs.w);
                    if (x.h2) trav_push(tr, RTA_STACK, w.refs.z);
                    if (x.h1) trav_push(tr, RTA_STACK, w.refs.y);
                    if (x.h0) trav_push(tr, RTA_STACK, w.refs.x);
                  }
                }
                trav_pop(tr, RTA_STACK, 0u);
              } else {
                const u32x4 e = *(const lds_u32x4*)list;
                tr.cur = e.x;
                if (e.w != kRefDone) trav_push(tr, RTA_STACK, stack_stride, e.w);
                if (e.z != kRefDone) trav_push(tr, RTA_STACK, stack_stride, e.z);
                if (e.y != kRefDone) trav_push(tr, RTA_STACK, stack_stride, e.y);
              }
            }
          }
        }
      }
      RTA_T1(block)
    }

    bool finished = false;
    for (;;) {
      // Traversal rounds while enough lanes want them: the innermost loop, which only moves the
      // cursor, the stack and the closest hit.  A round runs the inner-node block (up to three steps per
      // lane) if inner_min lanes are at inner nodes and then the leaf block if leaf_min lanes are
      // at leaves; it ends when neither is the case.
      // (the lanes at inner nodes as a mask too: the block's execution mask is that mask, not a second compare)
      unsigned long long m_inner = __ballot(!(tr.cur & kRefLeaf));
      uint32_t n_inner = count_mask(m_inner);
      for (;;) {
        bool ran = false;
        if (n_inner >= P.inner_min) {
          RTA_T0()
          if (__builtin_amdgcn_inverse_ballot_w64(m_inner)) {
            RTA_PROF(BLK_INNER)
            uint32_t steps;
            if constexpr (kWide) {
              trav_wide_step(tr, tv, RTA_STACK);
              steps = 1u;
            } else {
              steps = trav_inner_steps<kAllLds>(tr, tv, RTA_STACK, stack_stride);
            }
            if (kProf) n_steps += steps;
          }
          RTA_T1(BLK_INNER)
          ran = true;
        }
        if (count_lanes((tr.cur & kRefLeaf) && tr.cur < kRefIdle) >= P.leaf_min) {
          RTA_T0()
          if ((tr.cur & kRefLeaf) && tr.cur < kRefIdle) {
            RTA_PROF(BLK_LEAF)
            uint32_t tests = 0;
            trav_leaf_step<kAllLds, kSingle>(tr, tv, RTA_STACK, stack_stride, t_min, tests);
            if (kProf) n_tests += tests;
          }
          RTA_T1(BLK_LEAF)
          ran = true;
        }
        m_inner = __ballot(!(tr.cur & kRefLeaf));
        n_inner = count_mask(m_inner);
        if (!ran) break;
      }
      // Too few: run the block most lanes are waiting for.
      const uint32_t n_leaf = count_lanes((tr.cur & kRefLeaf) && tr.cur < kRefIdle);
      // (ballots of plain compares, combined as masks on the scalar side: a compound condition would be
      // built per lane first)
      const unsigned long long m_done = __ballot(tr.cur == kRefDone), m_hit = __ballot(tr.best >= 0);
      const uint32_t n_shade = count_mask(m_done & m_hit);
      const uint32_t n_camera = count_mask(__ballot(tr.cur == kRefCamera) | (m_done & ~m_hit));
      if ((n_inner | n_leaf | n_shade | n_camera) == 0u) {
        finished = true;
        break;
      }
      int pick = BLK_INNER;
      uint32_t most = n_inner;
      if (n_leaf > most) {
        most = n_leaf;
        pick = BLK_LEAF;
      }
      if (n_shade > most) {
        most = n_shade;
        pick = BLK_SHADE;
      }
      if (n_camera > most) pick = BLK_CAMERA;
      if (pick > BLK_LEAF) {  // SHADE: the cold part runs it
        block = pick;
        break;
      }
      if (pick == BLK_CAMERA) {
        block = pick;
        break;
      }
      // uniform ifs in sequence rather than if/else: the compiler then merges each block's results
      // into the lane state in place instead of copying the state around the pair
      if (pick == BLK_INNER || n_inner >= P.both_min) {
        RTA_T0()
        if (!(tr.cur & kRefLeaf)) {
          RTA_PROF(BLK_INNER)
          if constexpr (kWide)
            trav_wide_step(tr, tv, RTA_STACK);
          else
            trav_inner_step<kAllLds>(tr, tv, RTA_STACK, stack_stride);
          if (kProf) ++n_steps;
        }
        RTA_T1(BLK_INNER)
      }
      if (pick == BLK_LEAF || n_leaf >= P.both_min) {
        RTA_T0()
        if ((tr.cur & kRefLeaf) && tr.cur < kRefIdle) {
          RTA_PROF(BLK_LEAF)
          uint32_t tests = 0;
          trav_leaf_step<kAllLds, kSingle>(tr, tv, RTA_STACK, stack_stride, t_min, tests);
          if (kProf) n_tests += tests;
        }
        RTA_T1(BLK_LEAF)
      }
    }
    if (finished) break;
  }
  if (kProf) prof_cycles[4] = __builtin_amdgcn_s_memtime() - kernel_t0;

  // one counter update per wave
  unsigned long long c_rays = n_rays, c_samples = n_samples, c_steps = n_steps, c_tests = n_tests;
  for (int off = 32; off > 0; off >>= 1) {
    c_rays += __shfl_down(c_rays, off);
    c_samples += __shfl_down(c_samples, off);
    c_steps += __shfl_down(c_steps, off);
    c_tests += __shfl_down(c_tests, off);
  }
  if (lane == 0) {
    atomicAdd(&P.counters[1], c_rays);
    atomicAdd(&P.counters[2], c_samples);
    atomicAdd(&P.counters[3], c_steps);
    atomicAdd(&P.counters[4], c_tests);
  }
  if (kProf) {  // each run was counted by one elected lane: sum over the wave
    for (int k = 0; k < 4; ++k) {
      unsigned long long r = prof_runs[k], l = prof_lanes[k];
      for (int off = 32; off > 0; off >>= 1) {
        r += __shfl_down(r, off);
        l += __shfl_down(l, off);
      }
      if (lane == 0) {
        atomicAdd(&P.counters[5 + 2 * k], r);
        atomicAdd(&P.counters[6 + 2 * k], l);
      }
    }
    if (lane == 0)
      for (int k = 0; k < 5; ++k) atomicAdd(&P.counters[13 + k], prof_cycles[k]);
  }
}

// Adds the chunks' partial sums to the image in chunk order.  The partial sums hold `local_rows` packed rows of
// `tile_w` pixels per chunk: packed row r is row (r / 8 * row_stride + row_first) * 8 + r % 8 of the rectangle.
__global__ __launch_bounds__(256) void reduce_chunks_kernel(const double* __restrict__ partial, double* __restrict__ out, uint32_t n_chunks,
                                                            uint32_t x0, uint32_t y0, uint32_t y1, uint32_t tile_w, uint32_t local_rows,
                                                            uint32_t row_first, uint32_t row_stride, uint32_t width) {
  size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t n = (size_t)tile_w * local_rows;
  if (k >= n) return;
  const uint32_t r = (uint32_t)(k / tile_w), x = (uint32_t)(k - (size_t)r * tile_w);
  const uint32_t y = y0 + ((r >> 3) * row_stride + row_first) * 8u + (r & 7u);
  if (y >= y1) return;  // the last tile row sticks out of the rectangle
  double sr = 0.0, sg = 0.0, sb = 0.0;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    const double* p = partial + 3 * ((size_t)c * n + k);
    sr += p[0];
    sg += p[1];
    sb += p[2];
  }
  double* px = out + 3 * ((size_t)y * width + (x0 + x));
  px[0] += sr;
  px[1] += sg;
  px[2] += sb;
}

// out[k] += sum of the n_src frames src[0..n_src)[k], added in that order, for k in [begin, end): the frames of a
// sample split on several GPUs brought together (src may be memory of peer devices).
struct FrameList {
  const double* src[16];
};
__global__ __launch_bounds__(256) void add_frames_kernel(double* __restrict__ out, FrameList frames, uint32_t n_src, size_t begin, size_t end) {
  for (size_t k = begin + (size_t)blockIdx.x * 256 + threadIdx.x; k < end; k += (size_t)gridDim.x * 256) {
    double v = out[k];
    for (uint32_t f = 0; f < n_src; ++f) v += frames.src[f][k];
    out[k] = v;
  }
}

// ---------------------------------------------------------------- quantise ----
// write_colour (src/camera.rs:85-95) for every component of the frame: divide by the
// sample count, gamma 2 (sqrt), clamp to [0,1], scale by 255, truncate.  Division and
// square root are correctly rounded in f64, so the bytes equal the host's.  Two
// components per lane and step: 16-byte loads, 2-byte stores.
__global__ __launch_bounds__(256) void quantise_kernel(const double* __restrict__ rgb_sum, size_t n_components, double spp,
                                                       uint8_t* __restrict__ out) {
  const size_t n_pairs = n_components / 2;
  const f64x2* in2 = reinterpret_cast<const f64x2*>(rgb_sum);
  auto q = [spp](double x) -> uint32_t {
    double g = sqrt(x / spp);
    double v = 255.0 * fmin(fmax(g, 0.0), 1.0);  // fmin/fmax drop a NaN operand: NaN -> 0, as Interval::clamp then `as u8` give
    return !(v > 0.0) ? 0u : v >= 255.0 ? 255u : (uint32_t)v;
  };
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n_pairs; k += (size_t)gridDim.x * 256) {
    const f64x2 v = in2[k];
    reinterpret_cast<uint16_t*>(out)[k] = (uint16_t)(q(v.x) | (q(v.y) << 8));
  }
  if ((n_components & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[n_components - 1] = (uint8_t)q(rgb_sum[n_components - 1]);
}

// --------------------------------------------------------- static workspace ----
// Device memory for render calls that must not allocate (rt_render_device_noalloc): the launch's counters and its
// per-chunk sums.  One copy per GPU, at its own address, held from the time the code object is loaded until the
// process ends.  `scratch` names the array in device code; host code may only use the address that
// hipGetSymbolAddress returns for it on each device.
constexpr size_t kScratchBytes = 32u << 20;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxScratchDevices = 64;
hipError_t static_scratch(int device, void** ptr, size_t* bytes) {
  static std::mutex mu;
  static void* address[kMaxScratchDevices] = {};
  if (device < 0 || device >= kMaxScratchDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  if (!address[device]) {  // first use on this device: look the array's address up there
    int current = -1;
    hipError_t e = hipGetDevice(&current);
    if (e != hipSuccess) return e;
    if (current != device) {
      e = hipSetDevice(device);
      if (e != hipSuccess) return e;
    }
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (current != device) (void)hipSetDevice(current);
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    address[device] = p;
  }
  *ptr = address[device];
  *bytes = kScratchBytes;
  return hipSuccess;
}

// --------------------------------------------------------------- launchers ----
size_t lds_node_bytes() { return kLdsNodeQuads * 16; }
size_t lds_wide_node_bytes() { return kWideLdsStride; }
size_t lds_fixed_bytes(bool wide) { return CAM_DOUBLES * 8 + kTrigEntries * 16 + 32 + ring_words(wide) * 4; }  // camera constants, sin/cos table, the record of the sphere outside the tree, the waves' entry lists
size_t tree_lds_bytes(const SceneDev& sc) {
  if (sc.wide)
    return lds_fixed_bytes(true) + (size_t)sc.n_nodes_lds * kWideLdsStride + (size_t)sc.n_spheres_lds * 32 + (size_t)sc.n_aux_lds * 16 +
           (size_t)sc.n_materials_lds * 32 + (size_t)sc.lds_stack * kWorkgroup * 4;
  return lds_fixed_bytes(false) + (size_t)sc.n_nodes_lds * lds_node_bytes() + (size_t)sc.n_spheres_lds * 32 + (size_t)sc.n_aux_lds * 16 +
         (size_t)sc.n_materials_lds * 32 + (sc.stack_in_lds ? (size_t)(sc.stack_cap + kStackSpareRows) * kWorkgroup * 4 : 0);
}

static bool all_in_lds(const SceneDev& sc) {
  return sc.n_nodes_lds == sc.n_nodes && sc.n_spheres_lds == sc.n_spheres && sc.n_aux_lds == sc.n_spheres && sc.n_materials_lds == sc.n_materials;
}

static hipError_t allow_lds(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t launch_trace_batch(const SceneDev& sc, const double* org, const double* dir, size_t n, double t_min, double t_max, int32_t* prim,
                              double* t, double* normal, uint8_t* front, uint32_t* counts, int max_blocks, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  size_t lds = tree_lds_bytes(sc);
  // A scene is either wholly in LDS (two-wide tree) or walked through the four-wide tree.
  if (!sc.wide && !all_in_lds(sc)) return hipErrorInvalidValue;
  auto fn = sc.wide ? trace_batch_kernel<false, false, true> : sc.stack_in_lds ? trace_batch_kernel<true, false> : trace_batch_kernel<true, true>;
  hipError_t e = allow_lds((const void*)fn, lds);
  if (e != hipSuccess) return e;
  size_t blocks = (n + kWorkgroup - 1) / kWorkgroup;
  if (blocks > (size_t)max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kWorkgroup), lds, stream, sc, org, dir, n, t_min, t_max, prim, t, normal, front, counts);
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

// The variant for the scene: two-wide tree wholly in LDS (stacks in LDS, or in global memory for very deep
// trees) or four-wide tree; leaves of one sphere each or not; the instrumented form exists for one-sphere
// leaves and LDS stacks only.
static hipError_t render_variant(const SceneDev& sc, bool profile, void (**out)(RenderParams)) {
  const bool single = sc.single_leaves != 0u;
  void (*fn)(RenderParams);
  if (sc.wide)
    fn = profile && single ? render_kernel<false, true, false, true, true>
         : single          ? render_kernel<false, false, false, true, true>
                           : render_kernel<false, false, false, false, true>;
  else if (!all_in_lds(sc))
    return hipErrorInvalidValue;
  else if (sc.stack_in_lds)
    fn = profile && single ? render_kernel<true, true, false, true> : single ? render_kernel<true, false, false, true> : render_kernel<true, false, false, false>;
  else
    fn = single ? render_kernel<true, false, true, true> : render_kernel<true, false, true, false>;
  *out = fn;
  return hipSuccess;
}

hipError_t launch_render(const RenderParams& P, int blocks, bool profile, hipStream_t stream) {
  size_t lds = tree_lds_bytes(P.sc);
  void (*fn)(RenderParams);
  hipError_t e = render_variant(P.sc, profile, &fn);
  if (e != hipSuccess) return e;
  e = allow_lds((const void*)fn, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kWorkgroup), lds, stream, P);
  return hipGetLastError();
}

// Has the runtime load the code of the scene's render kernel (and set its LDS limit) now rather than at the first render: the
// reference's timed span (src/camera.rs:140-156) starts after the scene is built, and so should a first frame's.
hipError_t preload_render(const SceneDev& sc) {
  void (*fn)(RenderParams);
  hipError_t e = render_variant(sc, false, &fn);
  if (e != hipSuccess) return e;
  e = allow_lds((const void*)fn, tree_lds_bytes(sc));
  if (e != hipSuccess) return e;
  hipFuncAttributes attr;
  return hipFuncGetAttributes(&attr, (const void*)fn);
}

hipError_t launch_reduce_chunks(const double* partial, double* out, uint32_t n_chunks, uint32_t x0, uint32_t y0, uint32_t y1, uint32_t tile_w,
                                uint32_t local_rows, uint32_t row_first, uint32_t row_stride, uint32_t width, hipStream_t stream) {
  size_t n = (size_t)tile_w * local_rows;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, partial, out, n_chunks, x0, y0, y1, tile_w,
                     local_rows, row_first, row_stride, width);
  return hipGetLastError();
}

hipError_t launch_add_frames(double* out, const double* const* src, uint32_t n_src, size_t begin, size_t end, hipStream_t stream) {
  if (end <= begin || n_src == 0) return hipSuccess;
  if (n_src > kMaxAddFrames) return hipErrorInvalidValue;
  FrameList fl{};
  for (uint32_t f = 0; f < n_src; ++f) fl.src[f] = src[f];
  size_t blocks = (end - begin + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(add_frames_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, out, fl, n_src, begin, end);
  return hipGetLastError();
}

hipError_t launch_quantise(const double* rgb_sum, size_t n_components, uint64_t spp, uint8_t* out, hipStream_t stream) {
  if (n_components == 0) return hipSuccess;
  size_t blocks = (n_components / 2 + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(quantise_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rgb_sum, n_components, (double)spp, out);
  return hipGetLastError();
}

}  // namespace rta
