// Kernel parameter blocks and launch functions shared by the host glue
// (rt_api.cpp) and the device code (rt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace rta {

constexpr uint32_t kWorkgroup = 512;  // threads per workgroup of the tree-walking kernels

// ------------------------------------------------------------ parameters ----
struct SceneDev {
  const float4* nodes;      // 2 x float4 per node (DevNode)
  const double2* spheres;   // 2 x double2 per slot (DevSphere: cx,cy | cz,r)
  const double2* aux;       // 1 x double2 per slot: inv_r | (material, prim) packed in the second double's bits
  const double2* materials; // 4 x double2 per material: (albedo.x, albedo.y) (albedo.z, param) (inv_param, r0) (type bits, 0)
  uint32_t n_nodes;
  uint32_t n_spheres;
  uint32_t n_nodes_lds;    // nodes [0, n_nodes_lds) are read from LDS
  uint32_t n_spheres_lds;  // slots [0, n_spheres_lds) are read from LDS (0 or n_spheres)
  uint32_t stack_cap;      // entries per lane
};

struct CameraDev {
  double center[3], pixel00[3], du[3], dv[3], disk_u[3], disk_v[3];
  uint32_t width, height;
  uint32_t defocus;  // 1 when defocus_angle > 0
  uint32_t max_depth;
};

struct RenderParams {
  SceneDev sc;
  CameraDev cam;
  uint32_t x0, y0, x1, y1;  // tile
  uint32_t tiles_x, n_tiles;
  uint32_t n_chunks, chunk_spp;
  uint32_t spp_begin, spp_end;
  uint32_t seed_lo, seed_hi;
  uint32_t n_items;
  uint32_t direct;  // 1: out is the image and is added to; 0: out is the partial-sum buffer and is overwritten
  double* out;
  unsigned long long* counters;  // [0] next work item, [1] rays, [2] samples
};


// Dynamic LDS a tree-walking kernel needs for this scene.
size_t tree_lds_bytes(const SceneDev& sc, uint32_t wg);

// All launchers enqueue on `stream` and return the launch status; pointers are device pointers.
hipError_t launch_trace_batch(const SceneDev& sc, const double* org, const double* dir, size_t n, double t_min, double t_max, int32_t* prim,
                              double* t, double* normal, uint8_t* front, uint32_t* visits, int max_blocks, hipStream_t stream);
hipError_t launch_scatter_batch(const SceneDev& sc, const uint32_t* mat, const double* in_dir, const double* p, const double* normal,
                                const uint8_t* front, const double* u, size_t n, uint8_t* scattered, double* attenuation, double* out_dir,
                                hipStream_t stream);
hipError_t launch_slab_batch(const float* boxes, const double* org, const double* dir, const double* interval, size_t n, uint8_t* hit,
                             hipStream_t stream);
hipError_t launch_primary_rays(const CameraDev& cam, uint64_t seed, const uint32_t* ijs, size_t n, double* org, double* dir,
                               hipStream_t stream);
hipError_t launch_render(const RenderParams& P, int blocks, hipStream_t stream);
hipError_t launch_reduce_chunks(const double* partial, double* out, uint32_t n_chunks, uint32_t x0, uint32_t y0, uint32_t tile_w,
                                uint32_t tile_h, uint32_t width, hipStream_t stream);

}  // namespace rta
