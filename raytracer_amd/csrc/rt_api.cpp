// C ABI of libraytracer_amd.so (declared in include/rt.h): argument checks,
// device memory, and the sequencing of kernel launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "bvh_builder.h"
#include "rt.h"
#include "rt_host.h"
#include "rt_build.h"
#include "rt_launch.h"

using namespace rta;

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) {
  g_error = msg;
  return code;
}

#define RT_HIP(call)                                                                                         \
  do {                                                                                                       \
    hipError_t e_ = (call);                                                                                  \
    if (e_ != hipSuccess) return fail(RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));        \
  } while (0)

// Frees device allocations when a call leaves early.
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
  template <class T>
  T* as() const {
    return static_cast<T*>(p);
  }
};

struct DeviceGuard {  // makes the scene's device current for the duration of a call
  int prev = -1;
  hipError_t status;
  explicit DeviceGuard(int dev) {
    status = hipGetDevice(&prev);
    if (status == hipSuccess && prev != dev) status = hipSetDevice(dev);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// Lanes (of 64) that must want the inner-node block for a wave to run it without looking at the other blocks.
uint32_t g_inner_min = 10;
// Lanes that must want the other traversal block (inner or leaf) for it to run in the same round as the one that is due.
uint32_t g_both_min = 8;
// Lanes waiting at leaves that make the leaf block run between inner-node rounds.
uint32_t g_leaf_min = 10;
// Samples of one pixel that make a unit of work (the sample range is cut into chunks of about this size).
uint32_t g_chunk_spp = 64;
// Fewest units of work a call is cut into (32 per lane of the chip).
uint64_t g_target_units = 1u << 25;
uint64_t g_batches_per_wave = 32;
// Test hook: scenes created while this is set keep their traversal stacks in global memory, as very deep trees must.
bool g_force_global_stack = false;

}  // namespace

struct rt_scene {
  int device = 0;
  int num_cus = 0;
  HostBvh host;
  size_t n_spheres = 0, n_materials = 0;
  std::vector<uint8_t> material_type;
  void *d_nodes = nullptr, *d_spheres = nullptr, *d_aux = nullptr, *d_materials = nullptr;
  SceneDev dev{};
  rt_scene_info info{};
};

struct rt_host_bvh {
  HostBvh bvh;
};

static void dump_bvh(const HostBvh& b, double* boxes, int32_t* kind, uint64_t* skip_or_count, uint64_t* leaf_indices) {
  for (size_t i = 0; i < b.nodes.size(); ++i) {
    const HostNode& h = b.nodes[i];
    if (boxes) {
      std::memcpy(boxes + 6 * i, h.box.min, 24);
      std::memcpy(boxes + 6 * i + 3, h.box.max, 24);
    }
    if (kind) kind[i] = h.kind;
    if (skip_or_count) skip_or_count[i] = h.kind == 3 ? h.count : h.skip;
  }
  if (leaf_indices) std::memcpy(leaf_indices, b.leaf_indices.data(), b.leaf_indices.size() * 8);
}

static int make_camera_dev(const rt_config* cfg, CameraDev* cam, const char* who) {
  if (!cfg) return fail(RT_ERR_INVALID, std::string(who) + ": cfg is null");
  std::string why = config_check(*cfg);
  if (!why.empty()) return fail(RT_ERR_INVALID, std::string(who) + ": " + why);
  rt_camera c;
  camera_derive(*cfg, &c);
  std::memcpy(cam->center, c.center, 24);
  std::memcpy(cam->pixel00, c.pixel00_loc, 24);
  std::memcpy(cam->du, c.pixel_delta_u, 24);
  std::memcpy(cam->dv, c.pixel_delta_v, 24);
  std::memcpy(cam->disk_u, c.defocus_disk_u, 24);
  std::memcpy(cam->disk_v, c.defocus_disk_v, 24);
  cam->width = cfg->image_width;
  cam->height = c.image_height;
  cam->defocus = cfg->defocus_angle <= 0.0 ? 0u : 1u;  // src/camera.rs:127
  cam->max_depth = (uint32_t)cfg->max_depth;
  return RT_OK;
}

extern "C" {

const char* rt_last_error(void) { return g_error.c_str(); }

int rt_device_count(int* count) {
  if (!count) return fail(RT_ERR_INVALID, "rt_device_count: null argument");
  RT_HIP(hipGetDeviceCount(count));
  return RT_OK;
}
int rt_device_set(int ordinal) {
  RT_HIP(hipSetDevice(ordinal));
  return RT_OK;
}
int rt_device_get(int* ordinal) {
  if (!ordinal) return fail(RT_ERR_INVALID, "rt_device_get: null argument");
  RT_HIP(hipGetDevice(ordinal));
  return RT_OK;
}

void rt_config_default(rt_config* cfg) {
  if (cfg) config_default(cfg);
}

int rt_camera_derive(const rt_config* cfg, rt_camera* out) {
  if (!cfg || !out) return fail(RT_ERR_INVALID, "rt_camera_derive: null argument");
  std::string why = config_check(*cfg);
  if (!why.empty()) return fail(RT_ERR_INVALID, "rt_camera_derive: " + why);
  camera_derive(*cfg, out);
  return RT_OK;
}

int rt_scene_create(const rt_sphere* spheres, size_t n, const rt_material* materials, size_t nm, rt_scene** out) {
  return rt_scene_create_ex(spheres, n, materials, nm, RT_BUILD_HOST_SAH, out);
}

int rt_scene_create_ex(const rt_sphere* spheres, size_t n, const rt_material* materials, size_t nm, uint32_t builder, rt_scene** out) {
  if (!out) return fail(RT_ERR_INVALID, "rt_scene_create: out is null");
  if (builder > RT_BUILD_DEVICE_LBVH) return fail(RT_ERR_INVALID, "rt_scene_create: unknown builder");
  *out = nullptr;
  if ((n && !spheres) || (nm && !materials)) return fail(RT_ERR_INVALID, "rt_scene_create: null array");
  if (n >= kMaxSlots) return fail(RT_ERR_LIMIT, "rt_scene_create: at most 16777199 spheres");
  if (nm >= (1ull << 30)) return fail(RT_ERR_LIMIT, "rt_scene_create: at most 2^30-1 materials");
  for (size_t i = 0; i < n; ++i) {
    const rt_sphere& s = spheres[i];
    for (int k = 0; k < 3; ++k)
      if (!std::isfinite(s.center[k]) || std::fabs(s.center[k]) > 1e15)
        return fail(RT_ERR_INVALID, "rt_scene_create: sphere " + std::to_string(i) + " has a non-finite or huge centre");
    if (!std::isfinite(s.radius) || std::fabs(s.radius) > 1e15 || s.radius == 0.0)
      return fail(RT_ERR_INVALID, "rt_scene_create: sphere " + std::to_string(i) + " has a zero, non-finite or huge radius");
    if (s.material >= nm) return fail(RT_ERR_INVALID, "rt_scene_create: sphere " + std::to_string(i) + " names a material out of range");
  }
  for (size_t i = 0; i < nm; ++i) {
    const rt_material& m = materials[i];
    if (m.type > RT_DIELECTRIC) return fail(RT_ERR_INVALID, "rt_scene_create: material " + std::to_string(i) + " has an unknown type");
    if (!std::isfinite(m.albedo[0]) || !std::isfinite(m.albedo[1]) || !std::isfinite(m.albedo[2]) || !std::isfinite(m.param))
      return fail(RT_ERR_INVALID, "rt_scene_create: material " + std::to_string(i) + " has a non-finite field");
  }

  rt_scene* sc = new (std::nothrow) rt_scene();
  if (!sc) return fail(RT_ERR_LIMIT, "rt_scene_create: out of host memory");
  struct Cleanup {
    rt_scene* s;
    ~Cleanup() {
      if (s) rt_scene_destroy(s);
    }
  } cleanup{sc};

  RT_HIP(hipGetDevice(&sc->device));
  hipDeviceProp_t prop;
  RT_HIP(hipGetDeviceProperties(&prop, sc->device));
  sc->num_cus = prop.multiProcessorCount;

  sc->n_spheres = n;
  sc->n_materials = nm;

  // Material table, 32 bytes each (layout documented at scatter() in rt_device.h); the type
  // travels with the material index in each sphere slot's shading word.
  std::vector<double> mats(4 * nm, 0.0);
  sc->material_type.resize(nm);
  for (size_t i = 0; i < nm; ++i) {
    const rt_material& m = materials[i];
    double* d = &mats[4 * i];
    sc->material_type[i] = (uint8_t)m.type;
    if (m.type == RT_DIELECTRIC) {
      d[0] = m.param;
      d[1] = 1.0 / m.param;                            // ir.powi(-1), src/material.rs:52
      double r0 = (1.0 - m.param) / (1.0 + m.param);  // Schlick, src/material.rs:47-48
      d[2] = r0 * r0;
    } else {
      d[0] = m.albedo[0];
      d[1] = m.albedo[1];
      d[2] = m.albedo[2];
      d[3] = m.type == RT_METAL ? m.param : 0.0;
    }
  }

  // The tree: built on the host with the reference's algorithm, or on the device (two or more
  // spheres; a tree too deep for the traversal stack falls back to the host builder).
  auto t0 = std::chrono::steady_clock::now();
  const size_t max_depth = (kLdsPerCu - lds_fixed_bytes() - 64 - 16 * 1024) / (kWorkgroup * 4);
  FlatBvh flat;
  DeviceTree tree;
  bool on_device = false;
  if (builder == RT_BUILD_DEVICE_LBVH && n >= 2) {
    hipError_t e = device_build_lbvh(spheres, n, sc->material_type.data(), nm, &tree, nullptr);
    sc->d_nodes = tree.d_nodes;
    sc->d_spheres = tree.d_spheres;
    sc->d_aux = tree.d_aux;
    RT_HIP(e);
    if (tree.depth <= max_depth) {
      on_device = true;
    } else {
      (void)hipFree(sc->d_nodes);
      (void)hipFree(sc->d_spheres);
      (void)hipFree(sc->d_aux);
      sc->d_nodes = sc->d_spheres = sc->d_aux = nullptr;
    }
  }
  if (!on_device) {
    sc->host = build_bvh(spheres, n);
    flat = flatten_bvh(sc->host, spheres, n);
    for (auto& a : flat.aux) a.material |= (uint32_t)sc->material_type[a.material] << 30;
  }
  auto t1 = std::chrono::steady_clock::now();
  const size_t n_dev_nodes = on_device ? tree.n_nodes : flat.nodes.size();
  const uint32_t tree_depth = on_device ? tree.depth : flat.depth;

  // LDS plan for one workgroup per CU: the per-lane stacks are mandatory; then, while they
  // fit, the top of the tree, the sphere records, the slots' shading data and the materials.
  // A tree too deep for that keeps its stacks in global memory (allocated per call) and LDS holds scene data only.
  const uint32_t depth = tree_depth ? tree_depth : 1;
  const size_t reserve = lds_fixed_bytes() + 64;
  const bool stack_in_lds = depth <= max_depth && !g_force_global_stack;
  const size_t stack_bytes = stack_in_lds ? (size_t)depth * kWorkgroup * 4 : 0;
  size_t room = kLdsPerCu - stack_bytes - reserve;
  const size_t node_bytes = n_dev_nodes * sizeof(DevNode), sph_bytes = n * sizeof(DevSphere);
  const size_t aux_bytes = n * sizeof(DevSphereAux), mat_bytes = nm * 32;
  uint32_t n_nodes_lds = (uint32_t)std::min(n_dev_nodes, room / lds_node_bytes());
  room -= n_nodes_lds * lds_node_bytes();
  uint32_t n_sph_lds = 0, n_aux_lds = 0, n_mat_lds = 0;
  if (n_nodes_lds == n_dev_nodes) {
    if (sph_bytes <= room) {
      n_sph_lds = (uint32_t)n;
      room -= sph_bytes;
      if (aux_bytes <= room) {
        n_aux_lds = (uint32_t)n;
        room -= aux_bytes;
        if (mat_bytes <= room) n_mat_lds = (uint32_t)nm;
      }
    }
  }

  if (!on_device) {
    RT_HIP(hipMalloc(&sc->d_nodes, node_bytes ? node_bytes : 64));
    RT_HIP(hipMalloc(&sc->d_spheres, sph_bytes ? sph_bytes : 32));
    RT_HIP(hipMalloc(&sc->d_aux, aux_bytes ? aux_bytes : 16));
    if (node_bytes) RT_HIP(hipMemcpy(sc->d_nodes, flat.nodes.data(), node_bytes, hipMemcpyHostToDevice));
    if (sph_bytes) RT_HIP(hipMemcpy(sc->d_spheres, flat.spheres.data(), sph_bytes, hipMemcpyHostToDevice));
    if (aux_bytes) RT_HIP(hipMemcpy(sc->d_aux, flat.aux.data(), aux_bytes, hipMemcpyHostToDevice));
  }
  RT_HIP(hipMalloc(&sc->d_materials, mat_bytes ? mat_bytes : 32));
  if (mat_bytes) RT_HIP(hipMemcpy(sc->d_materials, mats.data(), mat_bytes, hipMemcpyHostToDevice));

  sc->dev.nodes = static_cast<const float4*>(sc->d_nodes);
  sc->dev.spheres = static_cast<const double2*>(sc->d_spheres);
  sc->dev.aux = static_cast<const double2*>(sc->d_aux);
  sc->dev.materials = static_cast<const double2*>(sc->d_materials);
  sc->dev.n_nodes = (uint32_t)n_dev_nodes;
  sc->dev.n_spheres = (uint32_t)n;
  sc->dev.n_materials = (uint32_t)nm;
  sc->dev.n_nodes_lds = n_nodes_lds;
  sc->dev.n_spheres_lds = n_sph_lds;
  sc->dev.n_aux_lds = n_aux_lds;
  sc->dev.n_materials_lds = n_mat_lds;
  sc->dev.stack_cap = depth;
  sc->dev.stack_in_lds = stack_in_lds ? 1u : 0u;
  sc->dev.root = on_device ? tree.root : flat.root;
  sc->dev.bound = on_device ? tree.bound : flat.bound;
  bool single = true;  // the device builder makes one-sphere leaves only
  if (!on_device) {
    auto one = [](uint32_t ref) { return !(ref & kRefLeaf) || ref == kRefNone || ((ref >> 24) & 0x7fu) == 1u; };
    single = one(flat.root);
    for (const DevNode& nd : flat.nodes) single = single && one(nd.ref0) && one(nd.ref1);
  }
  sc->dev.single_leaves = single ? 1u : 0u;

  sc->info.n_spheres = n;
  sc->info.n_nodes = on_device ? 2 * n - 1 : sc->host.nodes.size();
  sc->info.n_device_nodes = n_dev_nodes;
  sc->info.depth = on_device ? tree.depth : sc->host.depth;
  sc->info.builder = on_device ? RT_BUILD_DEVICE_LBVH : RT_BUILD_HOST_SAH;
  sc->info.n_nodes_in_lds = n_nodes_lds;
  sc->info.spheres_in_lds = (n_sph_lds ? 1u : 0u) | (n_aux_lds ? 2u : 0u) | (n_mat_lds ? 4u : 0u);
  sc->info.device = (uint32_t)sc->device;
  sc->info.device_bytes = node_bytes + sph_bytes + aux_bytes + mat_bytes;
  sc->info.build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  sc->info.lds_bytes = tree_lds_bytes(sc->dev);

  cleanup.s = nullptr;
  *out = sc;
  return RT_OK;
}

int rt_scene_destroy(rt_scene* sc) {
  if (!sc) return RT_OK;
  DeviceGuard g(sc->device);
  if (sc->d_nodes) (void)hipFree(sc->d_nodes);
  if (sc->d_spheres) (void)hipFree(sc->d_spheres);
  if (sc->d_aux) (void)hipFree(sc->d_aux);
  if (sc->d_materials) (void)hipFree(sc->d_materials);
  delete sc;
  return RT_OK;
}

int rt_scene_get_info(const rt_scene* sc, rt_scene_info* out) {
  if (!sc || !out) return fail(RT_ERR_INVALID, "rt_scene_get_info: null argument");
  *out = sc->info;
  return RT_OK;
}

int rt_scene_bvh_dump(const rt_scene* sc, double* boxes, int32_t* kind, uint64_t* skip_or_count, uint64_t* leaf_indices) {
  if (!sc) return fail(RT_ERR_INVALID, "rt_scene_bvh_dump: scene is null");
  if (sc->info.builder != RT_BUILD_HOST_SAH) return fail(RT_ERR_INVALID, "rt_scene_bvh_dump: the scene's tree was built on the device; there is no host tree to dump");
  dump_bvh(sc->host, boxes, kind, skip_or_count, leaf_indices);
  return RT_OK;
}

int rt_bvh_build(const rt_sphere* spheres, size_t n, rt_host_bvh** out) {
  if (!out || (n && !spheres)) return fail(RT_ERR_INVALID, "rt_bvh_build: null argument");
  rt_host_bvh* b = new (std::nothrow) rt_host_bvh();
  if (!b) return fail(RT_ERR_LIMIT, "rt_bvh_build: out of host memory");
  b->bvh = build_bvh(spheres, n);
  *out = b;
  return RT_OK;
}
int rt_bvh_destroy(rt_host_bvh* b) {
  delete b;
  return RT_OK;
}
int rt_bvh_counts(const rt_host_bvh* b, uint64_t* n_nodes, uint64_t* n_leaf_indices, uint32_t* depth) {
  if (!b) return fail(RT_ERR_INVALID, "rt_bvh_counts: bvh is null");
  if (n_nodes) *n_nodes = b->bvh.nodes.size();
  if (n_leaf_indices) *n_leaf_indices = b->bvh.leaf_indices.size();
  if (depth) *depth = b->bvh.depth;
  return RT_OK;
}
int rt_bvh_dump(const rt_host_bvh* b, double* boxes, int32_t* kind, uint64_t* skip_or_count, uint64_t* leaf_indices) {
  if (!b) return fail(RT_ERR_INVALID, "rt_bvh_dump: bvh is null");
  dump_bvh(b->bvh, boxes, kind, skip_or_count, leaf_indices);
  return RT_OK;
}

int rt_trace_batch(const rt_scene* sc, const double* org, const double* dir, size_t n, double t_min, double t_max, int32_t* prim,
                   double* t, double* normal, uint8_t* front_face) {
  if (!sc) return fail(RT_ERR_INVALID, "rt_trace_batch: scene is null");
  if (n == 0) return RT_OK;
  if (!org || !dir || !prim || !t || !normal || !front_face) return fail(RT_ERR_INVALID, "rt_trace_batch: null buffer");
  if (std::isnan(t_min) || std::isnan(t_max)) return fail(RT_ERR_INVALID, "rt_trace_batch: NaN interval");
  DeviceGuard g(sc->device);
  RT_HIP(g.status);
  DevBuf d_org, d_dir, d_prim, d_t, d_n, d_f;
  RT_HIP(d_org.alloc(n * 24));
  RT_HIP(d_dir.alloc(n * 24));
  RT_HIP(d_prim.alloc(n * 4));
  RT_HIP(d_t.alloc(n * 8));
  RT_HIP(d_n.alloc(n * 24));
  RT_HIP(d_f.alloc(n));
  RT_HIP(hipMemcpy(d_org.p, org, n * 24, hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(d_dir.p, dir, n * 24, hipMemcpyHostToDevice));
  SceneDev dev = sc->dev;
  DevBuf gstack;
  if (!dev.stack_in_lds) {  // a tree too deep for LDS stacks: one column of global memory per thread of the launch
    const size_t blocks = std::min<size_t>((n + kWorkgroup - 1) / kWorkgroup, (size_t)sc->num_cus);
    dev.gstack_stride = (uint32_t)(blocks * kWorkgroup);
    RT_HIP(gstack.alloc((size_t)dev.gstack_stride * dev.stack_cap * 4));
    dev.gstack = gstack.as<uint32_t>();
  }
  RT_HIP(launch_trace_batch(dev, d_org.as<double>(), d_dir.as<double>(), n, t_min, t_max, d_prim.as<int32_t>(), d_t.as<double>(),
                            d_n.as<double>(), d_f.as<uint8_t>(), nullptr, sc->num_cus, nullptr));
  RT_HIP(hipDeviceSynchronize());
  RT_HIP(hipMemcpy(prim, d_prim.p, n * 4, hipMemcpyDeviceToHost));
  RT_HIP(hipMemcpy(t, d_t.p, n * 8, hipMemcpyDeviceToHost));
  RT_HIP(hipMemcpy(normal, d_n.p, n * 24, hipMemcpyDeviceToHost));
  RT_HIP(hipMemcpy(front_face, d_f.p, n, hipMemcpyDeviceToHost));
  return RT_OK;
}

int rt_scatter_batch(const rt_scene* sc, const uint32_t* material, const double* in_dir, const double* p, const double* normal,
                     const uint8_t* front_face, const double* u, size_t n, uint8_t* scattered, double* attenuation, double* out_dir) {
  if (!sc) return fail(RT_ERR_INVALID, "rt_scatter_batch: scene is null");
  if (n == 0) return RT_OK;
  if (!material || !in_dir || !p || !normal || !front_face || !u || !scattered || !attenuation || !out_dir)
    return fail(RT_ERR_INVALID, "rt_scatter_batch: null buffer");
  std::vector<uint32_t> tagged(n);  // material index with its type in the top two bits, as the kernels carry it
  for (size_t i = 0; i < n; ++i) {
    if (material[i] >= sc->n_materials) return fail(RT_ERR_INVALID, "rt_scatter_batch: material index out of range");
    tagged[i] = material[i] | ((uint32_t)sc->material_type[material[i]] << 30);
  }
  DeviceGuard g(sc->device);
  RT_HIP(g.status);
  DevBuf d_m, d_in, d_p, d_n, d_f, d_u, d_s, d_a, d_o;
  RT_HIP(d_m.alloc(n * 4));
  RT_HIP(d_in.alloc(n * 24));
  RT_HIP(d_p.alloc(n * 24));
  RT_HIP(d_n.alloc(n * 24));
  RT_HIP(d_f.alloc(n));
  RT_HIP(d_u.alloc(n * 16));
  RT_HIP(d_s.alloc(n));
  RT_HIP(d_a.alloc(n * 24));
  RT_HIP(d_o.alloc(n * 24));
  RT_HIP(hipMemcpy(d_m.p, tagged.data(), n * 4, hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(d_in.p, in_dir, n * 24, hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(d_p.p, p, n * 24, hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(d_n.p, normal, n * 24, hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(d_f.p, front_face, n, hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(d_u.p, u, n * 16, hipMemcpyHostToDevice));
  RT_HIP(launch_scatter_batch(sc->dev, d_m.as<uint32_t>(), d_in.as<double>(), d_p.as<double>(), d_n.as<double>(), d_f.as<uint8_t>(),
                              d_u.as<double>(), n, d_s.as<uint8_t>(), d_a.as<double>(), d_o.as<double>(), nullptr));
  RT_HIP(hipDeviceSynchronize());
  RT_HIP(hipMemcpy(scattered, d_s.p, n, hipMemcpyDeviceToHost));
  RT_HIP(hipMemcpy(attenuation, d_a.p, n * 24, hipMemcpyDeviceToHost));
  RT_HIP(hipMemcpy(out_dir, d_o.p, n * 24, hipMemcpyDeviceToHost));
  return RT_OK;
}

int rt_slab_batch(const double* boxes, const double* org, const double* dir, const double* interval, size_t n, uint8_t* hit) {
  if (n == 0) return RT_OK;
  if (!boxes || !org || !dir || !interval || !hit) return fail(RT_ERR_INVALID, "rt_slab_batch: null buffer");
  // Round the boxes outwards to f32 exactly as scene creation does.
  std::vector<float> fb(8 * n, 0.0f);
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      double lo = boxes[6 * i + k], hi = boxes[6 * i + 3 + k];
      float fl = (float)lo, fh = (float)hi;
      if ((double)fl > lo) fl = std::nextafterf(fl, -INFINITY);
      if ((double)fh < hi) fh = std::nextafterf(fh, INFINITY);
      fb[8 * i + k] = fl;
      fb[8 * i + 4 + k] = fh;
      fb[8 * i + 3] = std::fmax(fb[8 * i + 3], std::fmax(std::fabs(fl), std::fabs(fh)));  // the pad's bound on |coordinate|
    }
  DevBuf d_b, d_o, d_d, d_i, d_h;
  RT_HIP(d_b.alloc(n * 32));
  RT_HIP(d_o.alloc(n * 24));
  RT_HIP(d_d.alloc(n * 24));
  RT_HIP(d_i.alloc(n * 16));
  RT_HIP(d_h.alloc(n));
  RT_HIP(hipMemcpy(d_b.p, fb.data(), n * 32, hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(d_o.p, org, n * 24, hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(d_d.p, dir, n * 24, hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(d_i.p, interval, n * 16, hipMemcpyHostToDevice));
  RT_HIP(launch_slab_batch(d_b.as<float>(), d_o.as<double>(), d_d.as<double>(), d_i.as<double>(), n, d_h.as<uint8_t>(), nullptr));
  RT_HIP(hipDeviceSynchronize());
  RT_HIP(hipMemcpy(hit, d_h.p, n, hipMemcpyDeviceToHost));
  return RT_OK;
}

int rt_primary_rays(const rt_config* cfg, uint64_t seed, const uint32_t* ijs, size_t n, double* org, double* dir) {
  CameraDev cam;
  int rc = make_camera_dev(cfg, &cam, "rt_primary_rays");
  if (rc) return rc;
  if (n == 0) return RT_OK;
  if (!ijs || !org || !dir) return fail(RT_ERR_INVALID, "rt_primary_rays: null buffer");
  DevBuf d_i, d_o, d_d;
  RT_HIP(d_i.alloc(n * 12));
  RT_HIP(d_o.alloc(n * 24));
  RT_HIP(d_d.alloc(n * 24));
  RT_HIP(hipMemcpy(d_i.p, ijs, n * 12, hipMemcpyHostToDevice));
  RT_HIP(launch_primary_rays(cam, seed, d_i.as<uint32_t>(), n, d_o.as<double>(), d_d.as<double>(), nullptr));
  RT_HIP(hipDeviceSynchronize());
  RT_HIP(hipMemcpy(org, d_o.p, n * 24, hipMemcpyDeviceToHost));
  RT_HIP(hipMemcpy(dir, d_d.p, n * 24, hipMemcpyDeviceToHost));
  return RT_OK;
}

int rt_render_device(const rt_scene* sc, const rt_config* cfg, uint64_t seed, uint32_t spp_begin, uint32_t spp_end, const rt_rect* tile,
                     double* rgb_sum_dev, void* stream_, rt_render_stats* stats) {
  if (!sc) return fail(RT_ERR_INVALID, "rt_render: scene is null");
  if (!rgb_sum_dev) return fail(RT_ERR_INVALID, "rt_render: rgb_sum is null");
  RenderParams P{};
  int rc = make_camera_dev(cfg, &P.cam, "rt_render");
  if (rc) return rc;
  if (spp_end < spp_begin) return fail(RT_ERR_INVALID, "rt_render: spp_end < spp_begin");
  rt_rect r = tile ? *tile : rt_rect{0, 0, P.cam.width, P.cam.height};
  if (r.x1 > P.cam.width || r.y1 > P.cam.height || r.x0 > r.x1 || r.y0 > r.y1) return fail(RT_ERR_INVALID, "rt_render: tile outside the image");
  if (stats) std::memset(stats, 0, sizeof(*stats));
  const uint32_t tw = r.x1 - r.x0, th = r.y1 - r.y0, spp = spp_end - spp_begin;
  if (tw == 0 || th == 0 || spp == 0) return RT_OK;
  if (P.cam.max_depth == 0) {  // every sample is black (src/camera.rs:99-101): nothing to add to the frame
    if (stats) stats->samples = (uint64_t)tw * th * spp;
    return RT_OK;
  }

  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DeviceGuard g(sc->device);
  RT_HIP(g.status);

  // The sample range is cut into chunks: a unit of work is one pixel's samples of one chunk, and
  // the lanes of a wave share out the units of one 8x8 tile, so a lane with a cheap pixel takes
  // more of them.  Enough chunks to keep lanes busy (about g_chunk_spp samples each, at most 16),
  // more when the rectangle has few pixels, fewer when the partial sums would not fit in 4 GB.
  // The count depends on the request only.
  const uint64_t kTargetLanes = g_target_units;
  uint64_t pixels = (uint64_t)tw * th;
  uint64_t want = std::max<uint64_t>(std::min<uint64_t>(16, spp / std::max<uint32_t>(1, g_chunk_spp)), (kTargetLanes + pixels - 1) / pixels);
  want = std::min<uint64_t>(want, std::max<uint64_t>(1, (4ull << 30) / (pixels * 24)));
  uint32_t n_chunks = (uint32_t)std::min<uint64_t>(spp, std::max<uint64_t>(1, want));
  uint32_t chunk_spp = (spp + n_chunks - 1) / n_chunks;
  n_chunks = (spp + chunk_spp - 1) / chunk_spp;

  P.sc = sc->dev;
  P.x0 = r.x0;
  P.y0 = r.y0;
  P.x1 = r.x1;
  P.y1 = r.y1;
  P.tiles_x = (tw + 7) / 8;
  P.n_tiles = P.tiles_x * ((th + 7) / 8);
  P.n_chunks = n_chunks;
  P.chunk_spp = chunk_spp;
  P.spp_begin = spp_begin;
  P.spp_end = spp_end;
  P.seed_lo = (uint32_t)seed;
  P.seed_hi = (uint32_t)(seed >> 32);
  uint64_t n_items = (uint64_t)P.n_tiles * n_chunks;
  // the unit counter is read as 32 bits and every lane overshoots it by one at the end
  if (n_items * 64 >= 0xfff00000ull) return fail(RT_ERR_LIMIT, "rt_render: too many work units; render a smaller tile or sample range");
  if (P.cam.width > 0xffffu || P.cam.height > 0xffffu) return fail(RT_ERR_LIMIT, "rt_render: images larger than 65535 pixels on a side are not supported");
  P.n_units = (uint32_t)(n_items * 64);
  // A wave takes some chunks of one tile at a time: all of them when there are many tiles per wave,
  // fewer when there are not, so that every wave gets about g_batches_per_wave batches.
  const uint64_t n_waves = (uint64_t)sc->num_cus * (kWorkgroup / 64);
  P.batch_chunks = (uint32_t)std::min<uint64_t>(n_chunks, std::max<uint64_t>(1, n_items / (g_batches_per_wave * n_waves)));
  P.batches_per_tile = (n_chunks + P.batch_chunks - 1) / P.batch_chunks;
  P.n_batches = P.n_tiles * P.batches_per_tile;
  P.direct = n_chunks == 1 ? 1u : 0u;
  P.inner_min = g_inner_min;
  P.both_min = g_both_min;
  P.leaf_min = g_leaf_min;

  DevBuf counters, partial, gstack;
  if (!P.sc.stack_in_lds) {
    P.sc.gstack_stride = (uint32_t)sc->num_cus * kWorkgroup;
    RT_HIP(gstack.alloc((size_t)P.sc.gstack_stride * P.sc.stack_cap * 4));
    P.sc.gstack = gstack.as<uint32_t>();
  }
  RT_HIP(counters.alloc(kRenderCounters * 8));
  RT_HIP(hipMemsetAsync(counters.p, 0, kRenderCounters * 8, stream));
  P.counters = counters.as<unsigned long long>();
  if (P.direct) {
    P.out = rgb_sum_dev;
  } else {
    RT_HIP(partial.alloc((size_t)n_chunks * pixels * 24));
    P.out = partial.as<double>();
  }

  // One persistent workgroup per CU (fewer if there is less work than that).
  int blocks = sc->num_cus;
  uint64_t wgs_needed = (n_items + (kWorkgroup / 64) - 1) / (kWorkgroup / 64);
  if ((uint64_t)blocks > wgs_needed) blocks = (int)wgs_needed;

  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  RT_HIP(hipEventCreate(&ev0));
  struct EvGuard {
    hipEvent_t& a;
    hipEvent_t& b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } evg{ev0, ev1};
  RT_HIP(hipEventCreate(&ev1));
  RT_HIP(hipEventRecord(ev0, stream));
  const bool profile = std::getenv("RT_PROFILE_BLOCKS") != nullptr;
  RT_HIP(launch_render(P, blocks, profile, stream));
  uint32_t launches = 1;
  if (!P.direct) {
    RT_HIP(launch_reduce_chunks(partial.as<double>(), rgb_sum_dev, n_chunks, r.x0, r.y0, tw, th, P.cam.width, stream));
    ++launches;
  }
  RT_HIP(hipEventRecord(ev1, stream));

  // The workspace is freed when this function returns, so the stream must have drained by then.
  RT_HIP(hipStreamSynchronize(stream));
  if (stats) {
    unsigned long long c[kRenderCounters];
    RT_HIP(hipMemcpy(c, counters.p, sizeof(c), hipMemcpyDeviceToHost));
    float ms = 0.0f;
    RT_HIP(hipEventElapsedTime(&ms, ev0, ev1));
    stats->rays = c[1];
    stats->samples = c[2];
    stats->kernel_ms = ms;
    stats->launches = launches;
    stats->node_steps = c[3];
    stats->sphere_tests = c[4];
    if (profile) {
      const char* names[4] = {"camera", "inner", "leaf", "shade"};  // BLK_* order
      for (int k = 0; k < 4; ++k)
        std::fprintf(stderr, "rt profile: %-6s runs %llu mean active lanes %.1f cycles/run %.0f share of wave time %.3f\n", names[k],
                     c[5 + 2 * k], c[5 + 2 * k] ? (double)c[6 + 2 * k] / (double)c[5 + 2 * k] : 0.0,
                     c[5 + 2 * k] ? (double)c[13 + k] / (double)c[5 + 2 * k] : 0.0, c[17] ? (double)c[13 + k] / (double)c[17] : 0.0);
      // Every wave is busy for (nearly) the whole launch, so stamped cycles / (waves x time) is the stamp clock's rate.
      std::fprintf(stderr, "rt profile: kernel %.2f ms, %d waves, stamped cycles per wave-second %.3f GHz\n", ms, blocks * (int)(kWorkgroup / 64),
                   (double)c[17] / (blocks * (double)(kWorkgroup / 64)) / (ms * 1e-3) / 1e9);
    }
  }
  return RT_OK;
}

int rt_render(const rt_scene* sc, const rt_config* cfg, uint64_t seed, uint32_t spp_begin, uint32_t spp_end, const rt_rect* tile,
              double* rgb_sum, rt_render_stats* stats) {
  if (!sc) return fail(RT_ERR_INVALID, "rt_render: scene is null");
  if (!rgb_sum) return fail(RT_ERR_INVALID, "rt_render: rgb_sum is null");
  CameraDev cam;
  int rc = make_camera_dev(cfg, &cam, "rt_render");
  if (rc) return rc;
  DeviceGuard g(sc->device);
  RT_HIP(g.status);
  const size_t n = (size_t)cam.width * cam.height * 3;
  DevBuf img;
  RT_HIP(img.alloc(n * 8));
  RT_HIP(hipMemset(img.p, 0, n * 8));
  rc = rt_render_device(sc, cfg, seed, spp_begin, spp_end, tile, img.as<double>(), nullptr, stats);
  if (rc) return rc;
  std::vector<double> host(n);
  RT_HIP(hipMemcpy(host.data(), img.p, n * 8, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) rgb_sum[i] += host[i];
  return RT_OK;
}

int rt_debug_set(const char* name, double value) {
  if (!name) return fail(RT_ERR_INVALID, "rt_debug_set: name is null");
  if (std::strcmp(name, "inner_min") == 0) {
    g_inner_min = (uint32_t)value;
    return RT_OK;
  }
  if (std::strcmp(name, "leaf_min") == 0) {
    g_leaf_min = (uint32_t)value;
    return RT_OK;
  }
  if (std::strcmp(name, "batches_per_wave") == 0) {
    g_batches_per_wave = std::max<uint64_t>(1, (uint64_t)value);
    return RT_OK;
  }
  if (std::strcmp(name, "both_min") == 0) {
    g_both_min = (uint32_t)value;
    return RT_OK;
  }
  if (std::strcmp(name, "force_global_stack") == 0) {
    g_force_global_stack = value != 0.0;
    return RT_OK;
  }
  if (std::strcmp(name, "target_units") == 0) {
    if (!(value >= 1.0 && value <= 1e12)) return fail(RT_ERR_INVALID, "rt_debug_set: target_units must be in 1 .. 1e12");
    g_target_units = (uint64_t)value;
    return RT_OK;
  }
  if (std::strcmp(name, "chunk_spp") == 0) {
    if (!(value >= 1.0 && value <= 1e9)) return fail(RT_ERR_INVALID, "rt_debug_set: chunk_spp must be in 1 .. 1e9");
    g_chunk_spp = (uint32_t)value;
    return RT_OK;
  }
  return fail(RT_ERR_INVALID, std::string("rt_debug_set: unknown tunable ") + name);
}

int rt_quantise(const double* rgb_sum, size_t n_pixels, uint64_t spp, uint8_t* out) {
  if (n_pixels && (!rgb_sum || !out)) return fail(RT_ERR_INVALID, "rt_quantise: null buffer");
  quantise(rgb_sum, n_pixels, spp, out);
  return RT_OK;
}

int rt_quantise_device(const double* rgb_sum_dev, size_t n_pixels, uint64_t spp, uint8_t* out_dev, void* stream) {
  if (n_pixels && (!rgb_sum_dev || !out_dev)) return fail(RT_ERR_INVALID, "rt_quantise_device: null buffer");
  if (((uintptr_t)rgb_sum_dev & 15) || ((uintptr_t)out_dev & 1)) return fail(RT_ERR_INVALID, "rt_quantise_device: rgb_sum must be 16-byte and out 2-byte aligned");
  if (spp == 0) return fail(RT_ERR_INVALID, "rt_quantise_device: samples_per_pixel is 0");
  RT_HIP(launch_quantise(rgb_sum_dev, 3 * n_pixels, spp, out_dev, (hipStream_t)stream));
  RT_HIP(hipStreamSynchronize((hipStream_t)stream));
  return RT_OK;
}

int rt_write_ppm_bytes(int fd, const uint8_t* rgb, uint32_t width, uint32_t height) {
  if (!rgb && width && height) return fail(RT_ERR_INVALID, "rt_write_ppm_bytes: rgb is null");
  if (!write_ppm_bytes(fd, rgb, width, height)) return fail(RT_ERR_IO, "rt_write_ppm_bytes: write failed");
  return RT_OK;
}

int rt_render_ppm(const rt_scene* sc, const rt_config* cfg, uint64_t seed, int fd, rt_render_stats* stats) {
  if (!sc) return fail(RT_ERR_INVALID, "rt_render_ppm: scene is null");
  CameraDev cam;
  int rc = make_camera_dev(cfg, &cam, "rt_render_ppm");
  if (rc) return rc;
  if (cfg->samples_per_pixel == 0 || cfg->samples_per_pixel > 0xffffffffull) return fail(RT_ERR_INVALID, "rt_render_ppm: samples_per_pixel must be in 1 .. 2^32-1");
  DeviceGuard g(sc->device);
  RT_HIP(g.status);
  const size_t n_px = (size_t)cam.width * cam.height;
  DevBuf img, bytes;
  RT_HIP(img.alloc(n_px * 24));
  RT_HIP(bytes.alloc(n_px * 3 + 2));
  RT_HIP(hipMemset(img.p, 0, n_px * 24));
  rc = rt_render_device(sc, cfg, seed, 0, (uint32_t)cfg->samples_per_pixel, nullptr, img.as<double>(), nullptr, stats);
  if (rc) return rc;
  RT_HIP(launch_quantise(img.as<double>(), 3 * n_px, cfg->samples_per_pixel, bytes.as<uint8_t>(), nullptr));
  std::vector<uint8_t> host(3 * n_px);
  RT_HIP(hipMemcpy(host.data(), bytes.p, 3 * n_px, hipMemcpyDeviceToHost));
  if (!write_ppm_bytes(fd, host.data(), cam.width, cam.height)) return fail(RT_ERR_IO, "rt_render_ppm: write failed");
  return RT_OK;
}

int rt_write_ppm(int fd, const double* rgb_sum, uint32_t width, uint32_t height, uint64_t spp) {
  if (!rgb_sum && width && height) return fail(RT_ERR_INVALID, "rt_write_ppm: rgb_sum is null");
  if (!write_ppm(fd, rgb_sum, width, height, spp)) return fail(RT_ERR_IO, "rt_write_ppm: write failed");
  return RT_OK;
}

int rt_scene_rtiow(uint64_t seed, rt_sphere* spheres, rt_material* materials) {
  if (!spheres || !materials) return fail(RT_ERR_INVALID, "rt_scene_rtiow: null buffer");
  scene_rtiow(seed, spheres, materials);
  return RT_OK;
}

int rt_scene_field(uint64_t seed, size_t n_small, double p_lambertian, double p_metal, double max_fuzz, rt_sphere* spheres,
                   rt_material* materials) {
  if (!spheres || !materials) return fail(RT_ERR_INVALID, "rt_scene_field: null buffer");
  if (!(p_lambertian >= 0.0) || !(p_metal >= 0.0) || p_lambertian + p_metal > 1.0 + 1e-12)
    return fail(RT_ERR_INVALID, "rt_scene_field: probabilities must be non-negative and sum to at most 1");
  scene_field(seed, n_small, p_lambertian, p_metal, max_fuzz, spheres, materials);
  return RT_OK;
}

}  // extern "C"
