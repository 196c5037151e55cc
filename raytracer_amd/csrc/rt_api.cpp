// C ABI of libraytracer_amd.so (declared in include/rt.h): argument checks,
// device memory, and the sequencing of kernel launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "bvh_builder.h"
#include "rt.h"
#ifdef RTA_EXPERIMENTS
#include "rt_experiments.h"
#endif
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

// Runs the body of an entry point; no C++ exception leaves the library.
template <class F>
int guarded(const char* who, F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    try {
      g_error = std::string(who) + ": out of host memory";
    } catch (...) {
    }
    return RT_ERR_LIMIT;
  } catch (const std::exception& e) {
    try {
      g_error = std::string(who) + ": " + e.what();
    } catch (...) {
    }
    return RT_ERR_INVALID;
  } catch (...) {
    try {
      g_error = std::string(who) + ": unknown failure";
    } catch (...) {
    }
    return RT_ERR_INVALID;
  }
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

struct DeviceGuard {  // makes a device current for the duration of a call
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

// Tunables (rt_debug_set).  Atomics: a render call reads each of them once, when it starts.
// Lanes (of 64) that must want the inner-node block for a wave to run it without looking at the other blocks.
std::atomic<uint32_t> g_inner_min{0};  // 0 (only as the initial value): 10 for the two-wide walk, 8 for the four-wide one
// Lanes that must want the other traversal block (inner or leaf) for it to run in the same round as the one that is due.
std::atomic<uint32_t> g_both_min{8};
// Lanes waiting at leaves that make the leaf block run between inner-node rounds.
std::atomic<uint32_t> g_leaf_min{8};
// Samples of one pixel that make a unit of work (the sample range is cut into chunks of about this size).
std::atomic<uint32_t> g_chunk_spp{64};
std::atomic<uint32_t> g_first_leaf{1};  // 0: keep a leaf under the root inside the walk (for measurements)
std::atomic<uint32_t> g_quad_tiles{1};
// Longest entry list a tile's camera rays start from (rt_launch.h); 0: every ray starts at the root.
std::atomic<uint32_t> g_tile_cuts{kWideCutEntries};
// Fewest units of work a call is cut into (32 per lane of the chip).
std::atomic<uint64_t> g_target_units{1u << 25};
std::atomic<uint64_t> g_batches_per_wave{64};
// Test hook: scenes created while this is set keep their traversal stacks in global memory, as very deep trees must.
std::atomic<bool> g_force_global_stack{false};
// Test hook: rt_render_multi brings the frames of a sample split together through a copy on the first device even
// where the devices can read each other's memory.
std::atomic<bool> g_force_staged_sum{false};
// Entries of a lane's traversal stack that the four-wide walk keeps in LDS (deeper ones go to global memory).
std::atomic<uint32_t> g_wide_lds_stack{8};
// Test hook: scenes created while this is set are walked through the four-wide tree even if they fit in LDS two-wide.
std::atomic<bool> g_force_wide{false};
#ifdef RTA_EXPERIMENTS
// Paths the wavefront harness keeps in flight (whole samples of the frame, at least one).
std::atomic<uint64_t> g_wavefront_paths{32u << 20};
#endif

// RT_BUILD_AUTO: scenes of this many spheres or more get their tree from the device's surface-area build, which makes the host
// builder's tree (1.7 against 2.3 ms here, 2.1 against 24 ms at 10^4 spheres, 11 against 680 ms at 10^6).
constexpr size_t kAutoDeviceBuild = 1024;

// Partial sums of an allocating render call never exceed this; the sample range is cut into fewer chunks instead.
constexpr uint64_t kWorkspaceLimit = 4ull << 30;

// Device memory a render call needs beyond its inputs and outputs.
struct Workspace {
  unsigned long long* counters = nullptr;  // kRenderCounters words
  double* partial = nullptr;               // per-chunk sums
  size_t partial_cap = 0;
  uint32_t* gstack = nullptr;  // traversal stacks of trees too deep for LDS
  size_t gstack_cap = 0;
  uint32_t* cuts = nullptr;  // the tiles' entry lists
  size_t cuts_cap = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool fixed = false;  // carved out of the library's static device array: cannot grow, owns nothing
  ~Workspace() {
    if (fixed) return;
    if (counters) (void)hipFree(counters);
    if (partial) (void)hipFree(partial);
    if (gstack) (void)hipFree(gstack);
    if (cuts) (void)hipFree(cuts);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
  }
};

// One copy of a scene on one device.
struct Replica {
  int device = 0;
  int num_cus = 0;
  void *d_nodes = nullptr, *d_spheres = nullptr, *d_aux = nullptr, *d_materials = nullptr;
  SceneDev dev{};
  // Workspaces of finished calls, kept for the next ones (a call takes one for itself, so calls may overlap).
  std::mutex pool_mu;
  std::vector<std::unique_ptr<Workspace>> idle;
  // rt_render_multi / rt_render_ppm: this device's frame, a second frame for sums staged through a copy, the
  // quantised bytes, and pinned host memory for the way back.  Grow-only; guarded by rt_scene::multi_mu.
  double* frame = nullptr;
  size_t frame_cap = 0;
  double* frame2 = nullptr;
  size_t frame2_cap = 0;
  uint8_t* bytes = nullptr;
  size_t bytes_cap = 0;
  void* pinned = nullptr;
  size_t pinned_cap = 0;
  hipStream_t stream = nullptr;

  ~Replica() {
    DeviceGuard g(device);
    idle.clear();
    for (void* p : {d_nodes, d_spheres, d_aux, d_materials, (void*)frame, (void*)frame2, (void*)bytes})
      if (p) (void)hipFree(p);
    if (pinned) (void)hipHostFree(pinned);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

hipError_t grow_device(void** p, size_t* cap, size_t bytes) {
  if (bytes <= *cap) return hipSuccess;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess) *cap = bytes;
  return e;
}

}  // namespace

struct rt_scene {
  std::vector<std::unique_ptr<Replica>> reps;  // reps[0] is the scene's home: the single-device entry points use it
  std::vector<std::vector<uint8_t>> peer;      // peer[a][b]: device of replica a can read the memory of replica b's
  HostBvh host;
  size_t n_spheres = 0, n_materials = 0;
  std::vector<uint8_t> material_type;   // by index into the device's material table
  std::vector<uint32_t> table_index;    // caller's material index -> index into the device's material table
  rt_scene_info info{};
  std::mutex multi_mu;  // one multi-device call per scene at a time (they share the replicas' frames)
};

struct rt_host_bvh {
  HostBvh bvh;
};

namespace {

void dump_bvh(const HostBvh& b, double* boxes, int32_t* kind, uint64_t* skip_or_count, uint64_t* leaf_indices) {
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

int make_camera_dev(const rt_config* cfg, CameraDev* cam, const char* who) {
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


// Looks in the top levels of a two-wide tree (nodes in breadth-first order) for a one-sphere leaf whose box has at least half the surface
// area of the whole tree's, and takes it out: its sibling takes its parent's place and the boxes recorded for the
// ancestors shrink to what is left below them.  *first_leaf = the sphere's slot, *walk_root = where the walk starts;
// *edited = the node array with those changes (left empty when the leaf hangs directly under the root: no node changes,
// the walk just starts at the sibling).  Nothing is written when there is no such leaf.
void take_out_first_leaf(const DevNode* tree_nodes, size_t n_nodes, uint32_t root, std::vector<DevNode>* edited, uint32_t* first_leaf,
                         uint32_t* walk_root) {
  if ((root & kRefLeaf) || root >= n_nodes) return;
  struct Box {
    float lo[3], hi[3];
  };
  auto child_box = [](const DevNode& d, int c) {
    return Box{{d.x[2 * c], d.y[2 * c], d.z[2 * c]}, {d.x[2 * c + 1], d.y[2 * c + 1], d.z[2 * c + 1]}};
  };
  auto set_child_box = [](DevNode& d, int c, const Box& b) {
    d.x[2 * c] = b.lo[0], d.x[2 * c + 1] = b.hi[0];
    d.y[2 * c] = b.lo[1], d.y[2 * c + 1] = b.hi[1];
    d.z[2 * c] = b.lo[2], d.z[2 * c + 1] = b.hi[2];
  };
  auto merged = [](const Box& a, const Box& b) {
    Box m;
    for (int k = 0; k < 3; ++k) m.lo[k] = std::min(a.lo[k], b.lo[k]), m.hi[k] = std::max(a.hi[k], b.hi[k]);
    return m;
  };
  auto area = [](const Box& b) {
    const double dx = (double)b.hi[0] - b.lo[0], dy = (double)b.hi[1] - b.lo[1], dz = (double)b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
  };
  auto one_sphere = [](uint32_t ref) { return (ref & kRefLeaf) && ref != kRefNone && ((ref >> 24) & 0x7fu) == 1u; };
  const DevNode& top = tree_nodes[root];
  const double whole = area(merged(child_box(top, 0), child_box(top, 1)));
  // depth-first over the top four levels; `path` holds (node, child slot taken) from the root down
  constexpr int kLevels = 4;
  struct Step {
    uint32_t node;
    int slot;
  };
  std::vector<Step> path, found;
  int found_slot = -1;
  std::function<bool(uint32_t)> visit = [&](uint32_t n) {
    const DevNode& d = tree_nodes[n];
    for (int c = 0; c < 2; ++c) {
      const uint32_t ref = c == 0 ? d.ref0 : d.ref1;
      if (one_sphere(ref) && area(child_box(d, c)) >= 0.5 * whole) {
        found = path;
        found.push_back(Step{n, c});
        found_slot = c;
        return true;
      }
    }
    if ((int)path.size() + 1 >= kLevels) return false;
    for (int c = 0; c < 2; ++c) {
      const uint32_t ref = c == 0 ? d.ref0 : d.ref1;
      if (!(ref & kRefLeaf) && ref < n_nodes) {
        path.push_back(Step{n, c});
        const bool hit = visit(ref);
        path.pop_back();
        if (hit) return true;
      }
    }
    return false;
  };
  if (!visit(root)) return;
  const Step parent = found.back();
  const DevNode& P = tree_nodes[parent.node];
  const uint32_t leaf_ref = found_slot == 0 ? P.ref0 : P.ref1, sibling_ref = found_slot == 0 ? P.ref1 : P.ref0;
  *first_leaf = leaf_ref & 0x00ffffffu;
  if (found.size() == 1) {  // directly under the root
    *walk_root = sibling_ref;
    return;
  }
  edited->assign(tree_nodes, tree_nodes + n_nodes);
  std::vector<DevNode>& nodes = *edited;
  // the grandparent's slot for the parent takes the sibling, with the sibling's box
  const Step grand = found[found.size() - 2];
  (grand.slot == 0 ? nodes[grand.node].ref0 : nodes[grand.node].ref1) = sibling_ref;
  set_child_box(nodes[grand.node], grand.slot, child_box(P, 1 - found_slot));
  // further up, the box recorded for the child on the path is the union of what that child now holds
  for (size_t k = found.size() - 2; k-- > 0;) {
    const DevNode& below = nodes[found[k + 1].node];
    set_child_box(nodes[found[k].node], found[k].slot, merged(child_box(below, 0), child_box(below, 1)));
  }
  *walk_root = root;
}

// ------------------------------------------------------------ scene set-up ----
// Uploads one copy of the scene to the current device (or builds the tree there) and plans its LDS image.
int make_replica(Replica& rp, const rt_sphere* spheres, size_t n, size_t nm, size_t n_used, const std::vector<double>& mats,
                 const std::vector<uint8_t>& material_type, bool on_device_wanted, DeviceBuilder device_builder, const FlatBvh* flat_in, size_t max_depth, WideBvh* wide_cache,
                 bool* on_device_out, uint32_t* depth_out, size_t* n_dev_nodes_out) {
  RT_HIP(hipGetDevice(&rp.device));
  hipDeviceProp_t prop;
  RT_HIP(hipGetDeviceProperties(&prop, rp.device));
  rp.num_cus = prop.multiProcessorCount;

  DeviceTree tree;
  bool on_device = false;
  if (on_device_wanted) {
    hipError_t e = device_build(spheres, n, material_type.data(), nm, device_builder, &tree, nullptr);
    rp.d_nodes = tree.d_nodes;
    rp.d_spheres = tree.d_spheres;
    rp.d_aux = tree.d_aux;
    RT_HIP(e);
    // (a tree deeper than LDS has stack rows for is walked with its stacks in global memory, like a host tree; the builder itself numbers
    // levels in eight bits)
    if (tree.depth <= 250) {
      on_device = true;
    } else {  // the caller falls back to the host builder
      (void)hipFree(rp.d_nodes);
      (void)hipFree(rp.d_spheres);
      (void)hipFree(rp.d_aux);
      rp.d_nodes = rp.d_spheres = rp.d_aux = nullptr;
      *on_device_out = false;
      return RT_OK;
    }
  }
  const FlatBvh& flat = *flat_in;
  const size_t n_dev_nodes = on_device ? tree.n_nodes : flat.nodes.size();
  // (levels of inner nodes: what the traversal stack must hold; the device builder counts the leaves' level too)
  const uint32_t tree_depth = on_device ? (tree.depth > 1 ? tree.depth - 1 : 1) : flat.depth;

  // LDS plan for one workgroup per CU: the per-lane stacks are mandatory; then, while they
  // fit, the top of the tree, the sphere records, the slots' shading data and the materials.
  // A tree too deep for that keeps its stacks in global memory and LDS holds scene data only.
  const uint32_t depth = tree_depth ? tree_depth : 1;
  const size_t reserve = lds_fixed_bytes(false) + 64;
  const bool stack_in_lds = depth <= max_depth && !g_force_global_stack.load();
  const size_t node_bytes = n_dev_nodes * sizeof(DevNode), sph_bytes = n * sizeof(DevSphere);
  const size_t aux_bytes = n * sizeof(DevSphereAux), mat_bytes = nm * 32, used_mat_bytes = n_used * 32;  // nm: records in the table
  // Camera rays that start from an entry list of k nodes (rt_launch.h) need k - 2 stack entries more than the tree is
  // deep: lists of four if the scene still fits in LDS with two more rows of stacks, else of three or two.
  uint32_t n_nodes_lds = 0, n_sph_lds = 0, n_aux_lds = 0, n_mat_lds = 0, cut_rows = 0;
  for (uint32_t extra : {2u, 1u, 0u}) {
    const size_t stack_bytes = stack_in_lds ? (size_t)(depth + kStackSpareRows + extra) * kWorkgroup * 4 : 0;
    cut_rows = extra;
    n_nodes_lds = n_sph_lds = n_aux_lds = n_mat_lds = 0;
    if (stack_bytes + reserve > kLdsPerCu) continue;
    size_t room = kLdsPerCu - stack_bytes - reserve;
    n_nodes_lds = (uint32_t)std::min(n_dev_nodes, room / lds_node_bytes());
    room -= n_nodes_lds * lds_node_bytes();
    if (n_nodes_lds == n_dev_nodes) {
      if (sph_bytes <= room) {
        n_sph_lds = (uint32_t)n;
        room -= sph_bytes;
        if (aux_bytes <= room) {
          n_aux_lds = (uint32_t)n;
          room -= aux_bytes;
          if (used_mat_bytes <= room) n_mat_lds = (uint32_t)n_used;
        }
      }
    }
    if (!stack_in_lds || (n_nodes_lds == n_dev_nodes && n_sph_lds == n && n_aux_lds == n && n_mat_lds == n_used)) break;
  }

  // A sphere as large as the scene (the reference scene's ground) is a one-sphere leaf near the top of the host's tree:
  // its box holds every ray's origin, so every walk tests it first, and it inflates the boxes of its ancestors to the
  // whole scene.  Such a leaf is taken out of the tree the device walks (take_out_first_leaf) and tested when a query
  // starts instead: the closest hit is the same, found with a node step and a leaf round fewer per ray.
  uint32_t first_leaf = kRefNone, walk_root = on_device ? tree.root : flat.root;
  std::vector<DevNode> edited;  // the host tree's nodes with the leaf taken out, if that changed any node
  if (g_first_leaf.load() && !on_device) {
    take_out_first_leaf(flat.nodes.data(), flat.nodes.size(), flat.root, &edited, &first_leaf, &walk_root);
  } else if (g_first_leaf.load() && tree.n_nodes > 1) {
    // the device builder's tree: its top four levels (breadth-first: the first 15 nodes at most) are looked at and
    // rewritten here; a leaf directly under the root is dropped by copying its sibling's node over the root's
    std::vector<DevNode> top(std::min<size_t>(tree.n_nodes, 15)), changed;
    RT_HIP(hipMemcpy(top.data(), rp.d_nodes, top.size() * sizeof(DevNode), hipMemcpyDeviceToHost));
    uint32_t slot = kRefNone, start = tree.root;
    take_out_first_leaf(top.data(), top.size(), tree.root, &changed, &slot, &start);
    if (slot != kRefNone && changed.empty() && !(start & kRefLeaf) && start < top.size()) {
      changed = top;
      changed[tree.root] = top[start];
    }
    if (slot != kRefNone && !changed.empty()) {
      RT_HIP(hipMemcpy(rp.d_nodes, changed.data(), changed.size() * sizeof(DevNode), hipMemcpyHostToDevice));
      first_leaf = slot;
    }
  }
  const DevNode* host_nodes = on_device ? nullptr : (edited.empty() ? flat.nodes.data() : edited.data());
  const bool wide_peel = first_leaf != kRefNone && !(walk_root & kRefLeaf);  // else the four-wide tree keeps the leaf

  // A scene that fits in LDS as a whole is walked through its two-wide tree; any other through the four-wide
  // tree with quantised boxes (half the bytes per ray), collapsed here from the two-wide nodes.
  const bool fits = n_nodes_lds == n_dev_nodes && n_sph_lds == n && n_aux_lds == n && n_mat_lds == n_used;
  const bool wide = (!fits || g_force_wide.load()) && n_dev_nodes > 0;
  uint32_t lds_stack = 0;
  size_t n_final_nodes = n_dev_nodes;
  uint32_t wide_stack_entries = 0;
  float wide_bound = 0.0f;
  void* d_wide = nullptr;
  if (wide) {
    if (on_device) {  // the device builder's nodes are collapsed where they are
      uint32_t n_wide = 0, levels = 0;
      hipError_t e = device_collapse_wide(rp.d_nodes, (uint32_t)n_dev_nodes, &d_wide, &n_wide, &levels, nullptr);
      (void)hipFree(rp.d_nodes);
      rp.d_nodes = d_wide;  // owned by the replica from here on, also if the call failed half-way
      RT_HIP(e);
      n_final_nodes = n_wide;
      wide_stack_entries = 3 * levels + 1 + (kWideCutEntries - 2);  // at most three children set aside per level; a camera ray's entry list
      wide_bound = tree.bound;
    } else {
      if (wide_cache->nodes.empty()) *wide_cache = collapse_wide(wide_peel ? host_nodes : flat.nodes.data(), n_dev_nodes, wide_peel ? walk_root : flat.root);
      n_final_nodes = wide_cache->nodes.size();
      wide_stack_entries = wide_cache->stack_entries + (kWideCutEntries - 2);
      wide_bound = std::fmax(wide_cache->bound, flat.bound);
    }
    lds_stack = g_force_global_stack.load() ? 0u : std::min<uint32_t>(wide_stack_entries, g_wide_lds_stack.load());
    const size_t wide_room = kLdsPerCu - (lds_fixed_bytes(true) + 64) - (size_t)lds_stack * kWorkgroup * 4;
    size_t wide_left = wide_room;
    n_nodes_lds = (uint32_t)std::min(n_final_nodes, wide_left / lds_wide_node_bytes());
    wide_left -= n_nodes_lds * lds_wide_node_bytes();
    n_sph_lds = n_aux_lds = n_mat_lds = 0;
    if (n_nodes_lds == n_final_nodes && sph_bytes <= wide_left) {  // room to spare: the spheres, their shading data, the materials
      n_sph_lds = (uint32_t)n;
      wide_left -= sph_bytes;
      if (aux_bytes <= wide_left) {
        n_aux_lds = (uint32_t)n;
        wide_left -= aux_bytes;
        if (used_mat_bytes <= wide_left) n_mat_lds = (uint32_t)n_used;
      }
    }
    if (!on_device) {
      RT_HIP(hipMalloc(&rp.d_nodes, n_final_nodes * sizeof(WideNode)));
      RT_HIP(hipMemcpy(rp.d_nodes, wide_cache->nodes.data(), n_final_nodes * sizeof(WideNode), hipMemcpyHostToDevice));
    }
  } else if (!on_device) {
    RT_HIP(hipMalloc(&rp.d_nodes, node_bytes ? node_bytes : 64));
    if (node_bytes) RT_HIP(hipMemcpy(rp.d_nodes, host_nodes, node_bytes, hipMemcpyHostToDevice));
  }
  if (!on_device) {
    RT_HIP(hipMalloc(&rp.d_spheres, sph_bytes ? sph_bytes : 32));
    RT_HIP(hipMalloc(&rp.d_aux, aux_bytes ? aux_bytes : 16));
    if (sph_bytes) RT_HIP(hipMemcpy(rp.d_spheres, flat.spheres.data(), sph_bytes, hipMemcpyHostToDevice));
    if (aux_bytes) RT_HIP(hipMemcpy(rp.d_aux, flat.aux.data(), aux_bytes, hipMemcpyHostToDevice));
  }
  RT_HIP(hipMalloc(&rp.d_materials, mat_bytes ? mat_bytes : 32));
  if (mat_bytes) RT_HIP(hipMemcpy(rp.d_materials, mats.data(), mat_bytes, hipMemcpyHostToDevice));

  SceneDev& dev = rp.dev;
  dev.nodes = static_cast<const float4*>(rp.d_nodes);
  dev.spheres = static_cast<const double2*>(rp.d_spheres);
  dev.aux = static_cast<const double2*>(rp.d_aux);
  dev.materials = static_cast<const double2*>(rp.d_materials);
  dev.n_nodes = (uint32_t)n_final_nodes;
  dev.n_spheres = (uint32_t)n;
  dev.n_materials = (uint32_t)n_used;  // the records a render can touch
  dev.n_nodes_lds = n_nodes_lds;
  dev.n_spheres_lds = n_sph_lds;
  dev.n_aux_lds = n_aux_lds;
  dev.n_materials_lds = n_mat_lds;
  dev.stack_cap = wide ? wide_stack_entries : depth + cut_rows;
  dev.cut_k = wide ? kWideCutEntries : cut_rows + 2;
  dev.stack_in_lds = wide ? 1u : (stack_in_lds ? 1u : 0u);
  dev.wide = wide ? 1u : 0u;
  dev.lds_stack = lds_stack;
  dev.root = wide ? 0u : (on_device ? tree.root : flat.root);
  dev.bound = wide ? wide_bound : (on_device ? tree.bound : flat.bound);
  bool single = true;  // the device builder makes one-sphere leaves only
  if (!on_device) {
    auto one = [](uint32_t ref) { return !(ref & kRefLeaf) || ref == kRefNone || ((ref >> 24) & 0x7fu) == 1u; };
    single = one(flat.root);
    for (const DevNode& nd : flat.nodes) single = single && one(nd.ref0) && one(nd.ref1);
  }
  dev.single_leaves = single ? 1u : 0u;
  dev.first_leaf = wide && !wide_peel ? kRefNone : first_leaf;
  if (!wide && first_leaf != kRefNone) dev.root = walk_root;
  *on_device_out = on_device;
  *depth_out = on_device ? tree.depth : tree_depth;
  *n_dev_nodes_out = n_final_nodes;
  return RT_OK;
}

int create_scene(const rt_sphere* spheres, size_t n, const rt_material* materials, size_t nm, uint32_t builder, const int32_t* devices,
                 size_t n_devices, bool current_device_only, rt_scene** out) {
  if (!out) return fail(RT_ERR_INVALID, "rt_scene_create: out is null");
  if (builder > RT_BUILD_DEVICE_SAH) return fail(RT_ERR_INVALID, "rt_scene_create: unknown builder");
  if (builder == RT_BUILD_AUTO) builder = n >= kAutoDeviceBuild ? RT_BUILD_DEVICE_SAH : RT_BUILD_HOST_SAH;
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

  // The devices that get a copy: the current one, or the caller's list (an ordinal may repeat: two copies on one GPU).
  std::vector<int> devs;
  if (current_device_only) {
    int cur = 0;
    RT_HIP(hipGetDevice(&cur));
    devs.push_back(cur);
  } else {
    int count = 0;
    RT_HIP(hipGetDeviceCount(&count));
    if (!devices) {
      for (int d = 0; d < count; ++d) devs.push_back(d);
    } else {
      if (n_devices == 0) return fail(RT_ERR_INVALID, "rt_scene_create_multi: empty device list");
      for (size_t k = 0; k < n_devices; ++k) {
        if (devices[k] < 0 || devices[k] >= count)
          return fail(RT_ERR_INVALID, "rt_scene_create_multi: device " + std::to_string(devices[k]) + " does not exist (" + std::to_string(count) + " visible)");
        devs.push_back(devices[k]);
      }
    }
    if (devs.empty()) return fail(RT_ERR_HIP, "rt_scene_create_multi: no HIP device");
    if (devs.size() > kMaxAddFrames) return fail(RT_ERR_LIMIT, "rt_scene_create_multi: at most 16 devices");
  }

  std::unique_ptr<rt_scene> sc(new rt_scene());
  sc->n_spheres = n;
  sc->n_materials = nm;

  // The device's material table holds each distinct material once (the reference scene's 160 glass spheres
  // share one record), those that spheres use first: only they are staged in LDS.
  struct Key {
    uint64_t w[5];
    bool operator==(const Key& o) const { return std::memcmp(w, o.w, sizeof(w)) == 0; }
  };
  struct KeyHash {
    size_t operator()(const Key& k) const {
      uint64_t h = 0x9E3779B97F4A7C15ull;
      for (uint64_t v : k.w) h = (h ^ v) * 0xD6E8FEB86659FD93ull + (h >> 29);
      return (size_t)h;
    }
  };
  auto key_of = [](const rt_material& m) {
    Key k{};
    k.w[0] = m.type;
    if (m.type == RT_DIELECTRIC) {
      std::memcpy(&k.w[1], &m.param, 8);
    } else {
      std::memcpy(&k.w[1], m.albedo, 24);
      if (m.type == RT_METAL) std::memcpy(&k.w[4], &m.param, 8);
    }
    return k;
  };
  // (an open-addressing table over the records met so far: a node-based map spent 80 ms here on a scene of 10^6 materials.
  // Beyond 65 536 materials nothing is looked for: a table that size is far from fitting in LDS whatever it shares, and the
  // search itself, 45 ms at 10^6, would be most of what scene creation costs; the table is then the caller's, in its order.)
  const bool dedup = nm <= 65536;
  std::vector<const rt_material*> table;
  std::vector<rt_sphere> placed;  // the spheres with their materials' table indices
  size_t n_used = nm;
  if (dedup) {
    std::vector<Key> table_key;
    size_t n_slots = 64;
    int slot_bits = 6;
    while (n_slots < 2 * nm) n_slots *= 2, ++slot_bits;
    std::vector<uint32_t> slot(n_slots, 0u);  // 0: free, else a table index + 1
    table.reserve(nm);
    table_key.reserve(nm);
    sc->table_index.assign(nm, 0xffffffffu);
    auto place = [&](size_t m) {
      if (sc->table_index[m] != 0xffffffffu) return;
      const Key k = key_of(materials[m]);
      size_t h = (size_t)(((uint64_t)KeyHash{}(k) * 0x9E3779B97F4A7C15ull) >> (64 - slot_bits));  // the top bits: the best mixed
      while (slot[h] && !(table_key[slot[h] - 1u] == k)) h = (h + 1) & (n_slots - 1);
      if (!slot[h]) {
        slot[h] = (uint32_t)table.size() + 1u;
        table.push_back(&materials[m]);
        table_key.push_back(k);
      }
      sc->table_index[m] = slot[h] - 1u;
    };
    placed.assign(spheres, spheres + n);
    for (size_t i = 0; i < n; ++i) {
      place(spheres[i].material);
      placed[i].material = sc->table_index[spheres[i].material];
    }
    n_used = table.size();
    for (size_t m = 0; m < nm; ++m) place(m);  // materials no sphere uses: kept for rt_scatter_batch, never staged
    spheres = placed.data();
  } else {
    table.resize(nm);
    sc->table_index.resize(nm);
    for (size_t m = 0; m < nm; ++m) table[m] = &materials[m], sc->table_index[m] = (uint32_t)m;
  }
  const size_t n_table = table.size();

  // Table records, 32 bytes each (layout documented at scatter() in rt_device.h); the type
  // travels with the material index in each sphere slot's shading word.
  std::vector<double> mats(4 * n_table, 0.0);
  sc->material_type.resize(n_table);
  for (size_t i = 0; i < n_table; ++i) {
    const rt_material& m = *table[i];
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

  // The tree: built on the host with the reference's algorithm (once, then uploaded to every device), or on each
  // device (two or more spheres; a tree too deep for the traversal stack falls back to the host builder).
  for (int d : devs) {  // (the runtime sets a device up at the first calls that need it, 0.15 s in a fresh process: not part of the build time either)
    DeviceGuard g(d);
    RT_HIP(g.status);
    hipDeviceProp_t prop;
    void* p = nullptr;
    const uint64_t word = 0;
    RT_HIP(hipGetDeviceProperties(&prop, d));
    RT_HIP(hipMalloc(&p, 64));
    (void)hipMemcpy(p, &word, 8, hipMemcpyHostToDevice);
    (void)hipFree(p);
  }
  auto t0 = std::chrono::steady_clock::now();
  const size_t max_depth = (kLdsPerCu - lds_fixed_bytes(false) - 64 - 16 * 1024) / (kWorkgroup * 4);
  bool want_device = (builder == RT_BUILD_DEVICE_LBVH || builder == RT_BUILD_DEVICE_SAH) && n >= 2;
  const DeviceBuilder device_builder = builder == RT_BUILD_DEVICE_SAH ? DeviceBuilder::kBinnedSah : DeviceBuilder::kClustering;
  FlatBvh flat;
  WideBvh wide_cache;
  bool have_flat = false;
  bool on_device = false;
  uint32_t tree_depth = 0;
  size_t n_dev_nodes = 0;
  double build_ms = 0.0;
  for (size_t k = 0; k < devs.size(); ++k) {
    DeviceGuard g(devs[k]);
    RT_HIP(g.status);
    std::unique_ptr<Replica> rp(new Replica());
    for (int attempt = 0; attempt < 2; ++attempt) {
      if (!want_device && !have_flat) {
        sc->host = build_bvh(spheres, n);
        flat = flatten_bvh(sc->host, spheres, n);
        for (auto& a : flat.aux) a.material |= (uint32_t)sc->material_type[a.material] << 30;
        have_flat = true;
      }
      bool od = false;
      int rc = make_replica(*rp, spheres, n, n_table, n_used, mats, sc->material_type, want_device, device_builder, &flat, max_depth, &wide_cache, &od, &tree_depth, &n_dev_nodes);
      if (rc) return rc;
      if (want_device && !od) {  // too deep: every replica takes the host tree
        want_device = false;
        continue;
      }
      on_device = od;
      break;
    }
    if (k == 0) build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // (not part of the build time: a process pays it once per kernel variant) the render kernel's code is loaded now, so that
    // the first frame costs what the later ones do
    if (n > 0) RT_HIP(preload_render(rp->dev));
    sc->reps.push_back(std::move(rp));
  }

  // Which replicas' devices can read each other's memory (xGMI / PCIe peer access), enabled once here.
  const size_t R = sc->reps.size();
  sc->peer.assign(R, std::vector<uint8_t>(R, 0));
  for (size_t a = 0; a < R; ++a) {
    DeviceGuard g(sc->reps[a]->device);
    for (size_t b = 0; b < R; ++b) {
      const int da = sc->reps[a]->device, db = sc->reps[b]->device;
      if (da == db) {
        sc->peer[a][b] = 1;
        continue;
      }
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, da, db) != hipSuccess || !can) continue;
      hipError_t e = hipDeviceEnablePeerAccess(db, 0);
      if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) sc->peer[a][b] = 1;
      (void)hipGetLastError();
    }
  }

  const Replica& r0 = *sc->reps[0];
  sc->info.n_spheres = n;
  sc->info.n_nodes = on_device ? 2 * n - 1 : sc->host.nodes.size();
  sc->info.n_device_nodes = n_dev_nodes;
  sc->info.depth = on_device ? tree_depth : sc->host.depth;
  sc->info.builder = on_device ? builder : (uint32_t)RT_BUILD_HOST_SAH;
  sc->info.n_nodes_in_lds = r0.dev.n_nodes_lds;
  sc->info.spheres_in_lds = (r0.dev.n_spheres_lds ? 1u : 0u) | (r0.dev.n_aux_lds ? 2u : 0u) | (r0.dev.n_materials_lds ? 4u : 0u);
  sc->info.device = (uint32_t)r0.device;
  sc->info.device_bytes = n_dev_nodes * 64 + n * sizeof(DevSphere) + n * sizeof(DevSphereAux) + n_table * 32;  // both node kinds take 64 bytes
  sc->info.build_ms = build_ms;
  sc->info.lds_bytes = tree_lds_bytes(r0.dev);
  sc->info.n_devices = (uint32_t)R;
  for (size_t a = 0; a < R; ++a)
    for (size_t b = 0; b < R; ++b) {
      if (a < 8 && b < 8 && sc->peer[a][b]) sc->info.peer_matrix |= 1ull << (a * 8 + b);
      if (sc->reps[a]->device == sc->reps[b]->device) continue;
      ++sc->info.peer_pairs;
      if (sc->peer[a][b]) ++sc->info.peer_pairs_enabled;
    }
  sc->info.cut_entries = r0.dev.cut_k;

  *out = sc.release();
  return RT_OK;
}

// ------------------------------------------------------------- render core ----
// How a call's sample range is cut into chunks.  A unit of work is one pixel's samples of one chunk, and
// the lanes of a wave share out the units of one 8x8 tile, so a lane with a cheap pixel takes
// more of them.  Enough chunks to keep lanes busy (about chunk_spp samples each, at most 16),
// more when the rectangle has few pixels, fewer when the partial sums would not fit in `limit` bytes.
// A function of the request only (sample count and the pixels of the whole rectangle): every shard of a
// rectangle is cut the same way, so a pixel's sum does not depend on how its frame was shared out.
struct ChunkPlan {
  uint32_t n_chunks, chunk_spp;
};
ChunkPlan plan_chunks(uint32_t spp, uint64_t pixels, uint64_t limit) {
  const uint64_t target = g_target_units.load();
  const uint64_t fit = std::max<uint64_t>(1, limit / (pixels * 24));  // chunks whose partial sums fit in the workspace
  uint64_t want = std::max<uint64_t>(std::min<uint64_t>(16, spp / std::max<uint32_t>(1, g_chunk_spp.load())), (target + pixels - 1) / pixels);
  // With a multiple of four chunks the 64 lanes of a wave start on a quarter of a tile in four chunks (quad_tiles), which
  // keeps their rays closer together: worth 4-9 % on scenes that are walked out of L1/L2.  Eight chunks where the
  // samples allow (at least four samples each), else four.
  if (g_quad_tiles.load() && spp >= 16) want = std::max<uint64_t>(want, spp >= 32 ? 8 : 4);
  want = std::min<uint64_t>(std::min<uint64_t>(want, fit), spp);
  auto cut = [&](uint32_t chunk_spp) { return ChunkPlan{(spp + chunk_spp - 1) / chunk_spp, chunk_spp}; };
  ChunkPlan plan = cut((uint32_t)((spp + want - 1) / std::max<uint64_t>(1, want)));
  if (g_quad_tiles.load() && want >= 4 && plan.n_chunks % 4 != 0) {  // a chunk size next to it that gives a multiple of four
    for (int d : {1, -1, 2, -2, 3, -3}) {
      const long c = (long)plan.chunk_spp + d;
      if (c < 1) continue;
      const ChunkPlan other = cut((uint32_t)c);
      if (other.n_chunks % 4 == 0 && other.n_chunks <= fit && other.n_chunks >= 4) {
        plan = other;
        break;
      }
    }
  }
  return plan;
}

uint32_t shard_tile_rows(const rt_shard& sh) {  // tile rows (8 pixel rows) of the rectangle that belong to the shard
  const uint32_t all = (sh.rect.y1 - sh.rect.y0 + 7) / 8;
  return sh.row_first < all ? (all - sh.row_first + sh.row_stride - 1) / sh.row_stride : 0;
}

int check_shard(const CameraDev& cam, const rt_shard& sh, const char* who) {
  const rt_rect& r = sh.rect;
  if (r.x1 > cam.width || r.y1 > cam.height || r.x0 > r.x1 || r.y0 > r.y1) return fail(RT_ERR_INVALID, std::string(who) + ": tile outside the image");
  if (sh.spp_end < sh.spp_begin) return fail(RT_ERR_INVALID, std::string(who) + ": spp_end < spp_begin");
  if (sh.row_stride == 0 || sh.row_first >= sh.row_stride) return fail(RT_ERR_INVALID, std::string(who) + ": row_first must be below row_stride");
  if (cam.width > 0xffffu || cam.height > 0xffffu) return fail(RT_ERR_LIMIT, std::string(who) + ": images larger than 65535 pixels on a side are not supported");
  return RT_OK;
}

// Takes a workspace from the replica's pool (or makes one) and grows it to the call's needs.
int acquire_workspace(Replica& rp, size_t partial_bytes, size_t gstack_bytes, size_t cuts_bytes, std::unique_ptr<Workspace>* out) {
  std::unique_ptr<Workspace> ws;
  {
    std::lock_guard<std::mutex> lock(rp.pool_mu);
    if (!rp.idle.empty()) {
      ws = std::move(rp.idle.back());
      rp.idle.pop_back();
    }
  }
  if (!ws) {
    ws.reset(new Workspace());
    RT_HIP(hipMalloc((void**)&ws->counters, kRenderCounters * 8));
    RT_HIP(hipEventCreate(&ws->ev0));
    RT_HIP(hipEventCreate(&ws->ev1));
  }
  RT_HIP(grow_device((void**)&ws->partial, &ws->partial_cap, partial_bytes));
  RT_HIP(grow_device((void**)&ws->gstack, &ws->gstack_cap, gstack_bytes));
  RT_HIP(grow_device((void**)&ws->cuts, &ws->cuts_cap, cuts_bytes));
  *out = std::move(ws);
  return RT_OK;
}
void release_workspace(Replica& rp, std::unique_ptr<Workspace> ws) {
  std::lock_guard<std::mutex> lock(rp.pool_mu);
  if (rp.idle.size() < 4) rp.idle.push_back(std::move(ws));
}

struct Launch {  // what enqueue_render worked out, for the caller to size the workspace and to fill the stats
  RenderParams P;
  int blocks = 0;
  size_t partial_bytes = 0, gstack_bytes = 0, cuts_bytes = 0;
  uint32_t cut_k = 0;                 // entries per tile's entry list (0: none are made)
  double beam_rho0 = 0.0, beam_h = 0.0;  // see launch_tile_cuts
  uint64_t samples = 0;
  bool nothing = false;  // no pixel, no sample or max_depth 0: no launch
};

// Everything about a launch that depends on the request only.
int plan_render(const Replica& rp, const CameraDev& cam, uint64_t seed, const rt_shard& sh, uint64_t workspace_limit, Launch* L) {
  RenderParams& P = L->P;
  P = RenderParams{};
  P.cam = cam;
  const rt_rect& r = sh.rect;
  const uint32_t tw = r.x1 - r.x0, th = r.y1 - r.y0, spp = sh.spp_end - sh.spp_begin;
  const uint32_t rows = shard_tile_rows(sh);
  // pixels of the shard: whole tile rows, clipped at the rectangle's last row
  uint64_t shard_pixel_rows = 0;
  for (uint32_t k = 0; k < rows; ++k) {
    const uint32_t first = (k * sh.row_stride + sh.row_first) * 8u;
    shard_pixel_rows += std::min(8u, th - first);
  }
  L->samples = (uint64_t)tw * shard_pixel_rows * spp;
  if (tw == 0 || rows == 0 || spp == 0 || cam.max_depth == 0) {  // max_depth 0: every sample is black (src/camera.rs:99-101)
    L->nothing = true;
    return RT_OK;
  }
  const ChunkPlan plan = plan_chunks(spp, (uint64_t)tw * th, workspace_limit);
  P.sc = rp.dev;
  P.x0 = r.x0;
  P.y0 = r.y0;
  P.x1 = r.x1;
  P.y1 = r.y1;
  P.row_first = sh.row_first;
  P.row_stride = sh.row_stride;
  P.local_rows = rows * 8u;
  P.tiles_x = (tw + 7) / 8;
  P.n_tiles = P.tiles_x * rows;
  P.n_chunks = plan.n_chunks;
  P.chunk_spp = plan.chunk_spp;
  P.spp_begin = sh.spp_begin;
  P.spp_end = sh.spp_end;
  P.seed_lo = (uint32_t)seed;
  P.seed_hi = (uint32_t)(seed >> 32);
  const uint64_t n_items = (uint64_t)P.n_tiles * plan.n_chunks;
  // the unit counter is read as 32 bits and every lane overshoots it by one at the end
  if (n_items * 64 >= 0xfff00000ull) return fail(RT_ERR_LIMIT, "rt_render: too many work units; render a smaller tile or sample range");
  P.n_units = (uint32_t)(n_items * 64);
  // A wave takes some chunks of one tile at a time: all of them when there are many tiles per wave,
  // fewer when there are not, so that every wave gets about batches_per_wave batches.
  const uint64_t n_waves = (uint64_t)rp.num_cus * (kWorkgroup / 64);
  P.batch_chunks = (uint32_t)std::min<uint64_t>(plan.n_chunks, std::max<uint64_t>(1, n_items / (g_batches_per_wave.load() * n_waves)));
  if (g_quad_tiles.load() && plan.n_chunks % 4 == 0) {
    P.quad_tiles = 1;
    P.batch_chunks = std::max(4u, P.batch_chunks & ~3u);
  }
  P.batches_per_tile = (plan.n_chunks + P.batch_chunks - 1) / P.batch_chunks;
  P.n_batches = P.n_tiles * P.batches_per_tile;
  P.direct = plan.n_chunks == 1 ? 1u : 0u;
  {  // the node tests' time scale: the scene's extent and the camera's distance from the origin, with a factor 16 to spare
    float extent = std::max(rp.dev.bound, 1e-30f);
    for (int k = 0; k < 3; ++k) extent = std::max(extent, (float)std::fabs(cam.center[k]));
    int e = 0;
    (void)std::frexp(extent, &e);  // extent < 2^e
    P.time_scale = std::ldexp(1.0f, -(e + 5));
  }
  P.inner_min = g_inner_min.load();
  if (P.inner_min == 0) P.inner_min = P.sc.wide ? 8u : 10u;  // the four-wide walk waits on its loads: it gains from running with fewer lanes
  P.both_min = g_both_min.load();
  P.leaf_min = g_leaf_min.load();
  if (P.sc.wide) {  // the stack entries beyond those kept in LDS
    if (P.sc.stack_cap > P.sc.lds_stack) {
      P.sc.gstack_stride = (uint32_t)rp.num_cus * kWorkgroup;
      L->gstack_bytes = (size_t)P.sc.gstack_stride * (P.sc.stack_cap - P.sc.lds_stack) * 4;
    }
  } else if (!P.sc.stack_in_lds) {
    P.sc.gstack_stride = (uint32_t)rp.num_cus * kWorkgroup;
    L->gstack_bytes = (size_t)P.sc.gstack_stride * P.sc.stack_cap * 4;
  }
  L->partial_bytes = P.direct ? 0 : (size_t)plan.n_chunks * tw * P.local_rows * 24;
  L->cut_k = std::min(g_tile_cuts.load(), rp.dev.cut_k);
  if (L->cut_k >= 1) {
    // |dx U + dy V| for (dx, dy) in the unit disk, and |a du + b dv| for |a|, |b| <= 1/2
    auto dot3 = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    const double uu = dot3(cam.disk_u, cam.disk_u), vv = dot3(cam.disk_v, cam.disk_v), uv = std::fabs(dot3(cam.disk_u, cam.disk_v));
    L->beam_rho0 = cam.defocus ? std::sqrt(std::max(uu, vv) + uv) * (1.0 + 1e-9) : 0.0;
    L->beam_h = 0.5 * std::sqrt(dot3(cam.du, cam.du) + dot3(cam.dv, cam.dv) + 2.0 * std::fabs(dot3(cam.du, cam.dv))) * (1.0 + 1e-9);
    L->cuts_bytes = (size_t)P.n_tiles * cut_tile_words(rp.dev.wide != 0u) * 4;
  }
  // One persistent workgroup per CU (fewer if there is less work than that).
  L->blocks = rp.num_cus;
  const uint64_t wgs_needed = (n_items + (kWorkgroup / 64) - 1) / (kWorkgroup / 64);
  if ((uint64_t)L->blocks > wgs_needed) L->blocks = (int)wgs_needed;
  return RT_OK;
}

// Enqueues the launch on `stream` with the workspace given (large enough, on the current device).
// Returns the number of kernels launched in *launches.
int enqueue_render(Launch& L, Workspace& ws, double* out_dev, hipStream_t stream, bool profile, uint32_t* launches) {
  RenderParams& P = L.P;
  RT_HIP(hipMemsetAsync(ws.counters, 0, kRenderCounters * 8, stream));
  P.counters = ws.counters;
  P.sc.gstack = ws.gstack;
  P.out = P.direct ? out_dev : ws.partial;
  if (ws.ev0) RT_HIP(hipEventRecord(ws.ev0, stream));
  *launches = 1;
  P.cuts = nullptr;
  if (L.cut_k >= 1 && ws.cuts) {
    RT_HIP(launch_tile_cuts(P.sc, P.cam, L.beam_rho0, L.beam_h, P.x0, P.y0, P.tiles_x, P.n_tiles, P.row_first, P.row_stride, L.cut_k, ws.cuts, stream));
    P.cuts = ws.cuts;
    ++*launches;
  }
  RT_HIP(launch_render(P, L.blocks, profile, stream));
  if (!P.direct) {
    RT_HIP(launch_reduce_chunks(ws.partial, out_dev, P.n_chunks, P.x0, P.y0, P.y1, P.x1 - P.x0, P.local_rows, P.row_first, P.row_stride, P.cam.width,
                                stream));
    ++*launches;
  }
  if (ws.ev1) RT_HIP(hipEventRecord(ws.ev1, stream));
  return RT_OK;
}

// After the stream has drained: the launch's counters into the caller's stats.
int read_stats(const Launch& L, const Workspace& ws, uint32_t launches, bool profile, rt_render_stats* stats) {
  unsigned long long c[kRenderCounters];
  RT_HIP(hipMemcpy(c, ws.counters, sizeof(c), hipMemcpyDeviceToHost));
  float ms = 0.0f;
  if (ws.ev0 && ws.ev1) RT_HIP(hipEventElapsedTime(&ms, ws.ev0, ws.ev1));
  stats->rays = c[1];
  stats->samples = c[2];
  stats->kernel_ms = ms;
  stats->launches = launches;
  stats->node_steps = c[3];
  stats->sphere_tests = c[4];
  if (profile) {
    if (c[5] + c[7] + c[9] + c[11] == 0) {
      std::fprintf(stderr, "rt profile: this scene's kernel variant (leaves of several spheres, or stacks in global memory) has no instrumented form\n");
      return RT_OK;
    }
    const char* names[4] = {"camera", "inner", "leaf", "shade"};  // BLK_* order
    for (int k = 0; k < 4; ++k)
      std::fprintf(stderr, "rt profile: %-6s runs %llu mean active lanes %.1f cycles/run %.0f share of wave time %.3f\n", names[k],
                   c[5 + 2 * k], c[5 + 2 * k] ? (double)c[6 + 2 * k] / (double)c[5 + 2 * k] : 0.0,
                   c[5 + 2 * k] ? (double)c[13 + k] / (double)c[5 + 2 * k] : 0.0, c[17] ? (double)c[13 + k] / (double)c[17] : 0.0);
    // Every wave is busy for (nearly) the whole launch, so stamped cycles / (waves x time) is the stamp clock's rate.
    std::fprintf(stderr, "rt profile: kernel %.2f ms, %d waves, stamped cycles per wave-second %.3f GHz\n", ms, L.blocks * (int)(kWorkgroup / 64),
                 (double)c[17] / (L.blocks * (double)(kWorkgroup / 64)) / (ms * 1e-3) / 1e9);
  }
  return RT_OK;
}

// One shard on one replica, start to finish: the body of the single-device render entry points and
// of each device's share of a multi-device call.  The replica's device must be current.
int render_shard(Replica& rp, const CameraDev& cam, uint64_t seed, const rt_shard& sh, double* out_dev, hipStream_t stream,
                 rt_render_stats* stats) {
  if (stats) std::memset(stats, 0, sizeof(*stats));
  Launch L;
  int rc = plan_render(rp, cam, seed, sh, kWorkspaceLimit, &L);
  if (rc) return rc;
  if (L.nothing) {
    if (stats && cam.max_depth == 0) stats->samples = L.samples;
    return RT_OK;
  }
  std::unique_ptr<Workspace> ws;
  rc = acquire_workspace(rp, L.partial_bytes, L.gstack_bytes, L.cuts_bytes, &ws);
  if (rc) return rc;
  const bool profile = std::getenv("RT_PROFILE_BLOCKS") != nullptr;
  uint32_t launches = 0;
  rc = enqueue_render(L, *ws, out_dev, stream, profile, &launches);
  // The workspace goes back to the pool when this function returns, so the stream must have drained by then.
  hipError_t e = hipStreamSynchronize(stream);
  if (rc == RT_OK && e != hipSuccess) rc = fail(RT_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
  if (rc == RT_OK && stats) rc = read_stats(L, *ws, launches, profile, stats);
  if (rc == RT_OK) release_workspace(rp, std::move(ws));  // after a failure the workspace is dropped, not reused
  return rc;
}

void partition_request(uint32_t partition, uint32_t n, const rt_rect& r, uint32_t s0, uint32_t s1, rt_shard* out) {
  const uint32_t spp = s1 - s0, base = spp / n, extra = spp % n;
  uint32_t s = s0;
  for (uint32_t g = 0; g < n; ++g) {
    rt_shard& sh = out[g];
    sh.rect = r;
    if (partition == RT_PARTITION_TILES) {  // tile rows dealt round-robin: cheap sky rows and costly ground rows mix on every device
      sh.row_first = g;
      sh.row_stride = n;
      sh.spp_begin = s0;
      sh.spp_end = s1;
    } else {  // contiguous sample ranges, sizes differing by at most one
      sh.row_first = 0;
      sh.row_stride = 1;
      sh.spp_begin = s;
      s += base + (g < extra ? 1u : 0u);
      sh.spp_end = s;
    }
  }
}

// Runs job(g) for g in [0, n) on one host thread each (the calling thread takes g = 0) and returns the first
// failure's code; a job's error text is handed over to the calling thread.  Where a thread cannot be
// created the job runs on the calling thread.
template <class F>
int for_each_replica(size_t n, F job) {
  std::vector<int> rc(n, RT_OK);
  std::vector<std::string> msg(n);
  auto run = [&](size_t g) {
    rc[g] = guarded("rt_render_multi", [&]() { return job(g); });
    if (rc[g]) msg[g] = g_error;
  };
  std::vector<std::thread> th;
  th.reserve(n);
  std::vector<size_t> inline_jobs;
  for (size_t g = 1; g < n; ++g) {
    try {
      th.emplace_back(run, g);
    } catch (const std::system_error&) {
      inline_jobs.push_back(g);
    }
  }
  run(0);
  for (size_t g : inline_jobs) run(g);
  for (auto& t : th) t.join();
  for (size_t g = 0; g < n; ++g)
    if (rc[g]) return fail(rc[g], msg[g]);
  return RT_OK;
}

// Every replica renders its shard into its own zeroed frame (rp.frame, whole image).  tiles: afterwards each
// replica's frame holds its tile rows.  samples: afterwards the frames are summed, in replica order, into
// replica 0's frame.  The caller holds sc->multi_mu.
int render_replicas(rt_scene* sc, const CameraDev& cam, uint64_t seed, uint32_t partition, const rt_rect& r, uint32_t s0, uint32_t s1,
                    std::vector<rt_shard>* shards_out, rt_render_stats* stats) {
  const size_t R = sc->reps.size();
  std::vector<rt_shard> shards(R);
  partition_request(partition, (uint32_t)R, r, s0, s1, shards.data());
  for (const rt_shard& sh : shards) {
    int rc = check_shard(cam, sh, "rt_render_multi");
    if (rc) return rc;
  }
  const size_t frame_bytes = (size_t)cam.width * cam.height * 24;
  std::vector<rt_render_stats> st(R);
  int rc = for_each_replica(R, [&](size_t g) -> int {
    Replica& rp = *sc->reps[g];
    DeviceGuard dg(rp.device);
    RT_HIP(dg.status);
    if (!rp.stream) RT_HIP(hipStreamCreateWithFlags(&rp.stream, hipStreamNonBlocking));
    RT_HIP(grow_device((void**)&rp.frame, &rp.frame_cap, frame_bytes));
    RT_HIP(hipMemsetAsync(rp.frame, 0, frame_bytes, rp.stream));
    return render_shard(rp, cam, seed, shards[g], rp.frame, rp.stream, &st[g]);
  });
  if (rc) return rc;
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    for (const rt_render_stats& s : st) {
      stats->rays += s.rays;
      stats->samples += s.samples;
      stats->kernel_ms = std::max(stats->kernel_ms, s.kernel_ms);  // the devices run side by side
      stats->launches += s.launches;
      stats->node_steps += s.node_steps;
      stats->sphere_tests += s.sphere_tests;
    }
  }
  if (partition == RT_PARTITION_SAMPLES && R > 1) {
    const size_t n_values = (size_t)cam.width * cam.height * 3;
    bool all_peers = !g_force_staged_sum.load();
    for (size_t a = 0; a < R && all_peers; ++a)
      for (size_t b = 0; b < R; ++b) all_peers = all_peers && sc->peer[a][b];
    if (all_peers) {
      // Every device reads its slice of all frames over its own links and writes the slice's sum into replica 0's
      // frame: the traffic is spread over all point-to-point links instead of converging on one device.
      // (Replica 0's own contribution is already in place: the others are added to it in replica order.)
      std::vector<const double*> src;
      for (size_t g = 1; g < R; ++g) src.push_back(sc->reps[g]->frame);
      rc = for_each_replica(R, [&](size_t g) -> int {
        Replica& rp = *sc->reps[g];
        DeviceGuard dg(rp.device);
        RT_HIP(dg.status);
        const size_t begin = n_values * g / R, end = n_values * (g + 1) / R;
        RT_HIP(launch_add_frames(sc->reps[0]->frame, src.data(), (uint32_t)src.size(), begin, end, rp.stream));
        RT_HIP(hipStreamSynchronize(rp.stream));
        return RT_OK;
      });
      if (rc) return rc;
    } else {
      // No peer access: the frames are copied to replica 0's device one after the other and added there.
      Replica& r0 = *sc->reps[0];
      DeviceGuard dg(r0.device);
      RT_HIP(dg.status);
      RT_HIP(grow_device((void**)&r0.frame2, &r0.frame2_cap, frame_bytes));
      for (size_t g = 1; g < R; ++g) {
        RT_HIP(hipMemcpyPeerAsync(r0.frame2, r0.device, sc->reps[g]->frame, sc->reps[g]->device, frame_bytes, r0.stream));
        const double* one[1] = {r0.frame2};
        RT_HIP(launch_add_frames(r0.frame, one, 1, 0, n_values, r0.stream));
      }
      RT_HIP(hipStreamSynchronize(r0.stream));
    }
  }
  if (shards_out) *shards_out = std::move(shards);
  return RT_OK;
}

// The library's static device array (rt_kernels.hip) as a workspace: counters in its first 256 bytes, per-chunk sums after them.
int static_workspace(int device, Workspace* ws) {
  void* base = nullptr;
  size_t bytes = 0;
  RT_HIP(static_scratch(device, &base, &bytes));
  ws->fixed = true;
  ws->counters = static_cast<unsigned long long*>(base);
  ws->partial = reinterpret_cast<double*>(static_cast<char*>(base) + 256);
  ws->partial_cap = bytes - 256;
  return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_error.c_str(); }

int rt_device_count(int* count) {
  return guarded("rt_device_count", [&]() -> int {
    if (!count) return fail(RT_ERR_INVALID, "rt_device_count: null argument");
    RT_HIP(hipGetDeviceCount(count));
    return RT_OK;
  });
}
int rt_device_set(int ordinal) {
  return guarded("rt_device_set", [&]() -> int {
    RT_HIP(hipSetDevice(ordinal));
    return RT_OK;
  });
}
int rt_device_get(int* ordinal) {
  return guarded("rt_device_get", [&]() -> int {
    if (!ordinal) return fail(RT_ERR_INVALID, "rt_device_get: null argument");
    RT_HIP(hipGetDevice(ordinal));
    return RT_OK;
  });
}

void rt_config_default(rt_config* cfg) {
  if (cfg) config_default(cfg);
}

int rt_camera_derive(const rt_config* cfg, rt_camera* out) {
  return guarded("rt_camera_derive", [&]() -> int {
    if (!cfg || !out) return fail(RT_ERR_INVALID, "rt_camera_derive: null argument");
    std::string why = config_check(*cfg);
    if (!why.empty()) return fail(RT_ERR_INVALID, "rt_camera_derive: " + why);
    camera_derive(*cfg, out);
    return RT_OK;
  });
}

int rt_scene_create(const rt_sphere* spheres, size_t n, const rt_material* materials, size_t nm, rt_scene** out) {
  return rt_scene_create_ex(spheres, n, materials, nm, RT_BUILD_AUTO, out);
}

int rt_scene_create_ex(const rt_sphere* spheres, size_t n, const rt_material* materials, size_t nm, uint32_t builder, rt_scene** out) {
  return guarded("rt_scene_create", [&]() -> int { return create_scene(spheres, n, materials, nm, builder, nullptr, 0, true, out); });
}

int rt_scene_create_multi(const rt_sphere* spheres, size_t n, const rt_material* materials, size_t nm, uint32_t builder, const int32_t* devices,
                          size_t n_devices, rt_scene** out) {
  return guarded("rt_scene_create_multi",
                 [&]() -> int { return create_scene(spheres, n, materials, nm, builder, devices, n_devices, false, out); });
}

int rt_scene_destroy(rt_scene* sc) {
  return guarded("rt_scene_destroy", [&]() -> int {
    delete sc;
    return RT_OK;
  });
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
  return guarded("rt_bvh_build", [&]() -> int {
    if (!out || (n && !spheres)) return fail(RT_ERR_INVALID, "rt_bvh_build: null argument");
    std::unique_ptr<rt_host_bvh> b(new rt_host_bvh());
    b->bvh = build_bvh(spheres, n);
    *out = b.release();
    return RT_OK;
  });
}
// The device-built hierarchy in the reference's layout: nodes in pre-order, a split's second child `skip` nodes on, leaves
// with their spheres; where the device build cut a would-be leaf of the reference in the middle, the leaf is put together again.
int rt_bvh_build_device(const rt_sphere* spheres, size_t n, rt_host_bvh** out) {
  return guarded("rt_bvh_build_device", [&]() -> int {
    if (!out || (n && !spheres)) return fail(RT_ERR_INVALID, "rt_bvh_build_device: null argument");
    if (n >= kMaxSlots) return fail(RT_ERR_LIMIT, "rt_bvh_build_device: at most 16777199 spheres");
    std::unique_ptr<rt_host_bvh> b(new rt_host_bvh());
    if (n < 2) {  // nothing to split
      b->bvh = build_bvh(spheres, n);
      *out = b.release();
      return RT_OK;
    }
    HostHierarchy h;
    RT_HIP(device_build_hierarchy(spheres, n, &h, nullptr));
    HostBvh& t = b->bvh;
    t.nodes.reserve(2 * n);
    t.leaf_indices.reserve(n);
    auto sphere_box = [&](uint32_t i) {  // Sphere::aabb, src/sphere.rs:50-52, as build_bvh computes it
      Box bx;
      for (int a = 0; a < 3; ++a) {
        const double p0 = spheres[i].center[a] - spheres[i].radius, p1 = spheres[i].center[a] + spheres[i].radius;
        bx.min[a] = std::fmin(p0, p1);
        bx.max[a] = std::fmax(p0, p1);
      }
      return bx;
    };
    auto node_box = [&](uint32_t id) {
      Box bx;
      for (int a = 0; a < 3; ++a) bx.min[a] = h.boxes[6 * (size_t)id + a], bx.max[a] = h.boxes[6 * (size_t)id + 3 + a];
      return bx;
    };
    constexpr uint32_t kLeaf = 0x80000000u;
    std::function<void(uint32_t)> gather = [&](uint32_t ref) {  // the spheres below ref, left to right
      if (ref & kLeaf) {
        t.leaf_indices.push_back(h.order[ref & ~kLeaf]);
        return;
      }
      gather(h.left[ref]);
      gather(h.right[ref]);
    };
    std::function<void(uint32_t, uint32_t)> emit = [&](uint32_t ref, uint32_t level) {
      t.depth = std::max(t.depth, level);
      if (ref & kLeaf) {
        const uint32_t prim = h.order[ref & ~kLeaf];
        t.nodes.push_back(HostNode{sphere_box(prim), 3, 0, t.leaf_indices.size(), 1});
        t.leaf_indices.push_back(prim);
      } else if (h.axis[ref] > 2) {
        const uint64_t first = t.leaf_indices.size();
        gather(ref);
        t.nodes.push_back(HostNode{node_box(ref), 3, 0, first, t.leaf_indices.size() - first});
      } else {
        const size_t me = t.nodes.size();
        t.nodes.push_back(HostNode{node_box(ref), (int32_t)h.axis[ref], 0, 0, 0});
        emit(h.left[ref], level + 1);
        t.nodes[me].skip = t.nodes.size() - me;  // src/bvh.rs:126-130
        emit(h.right[ref], level + 1);
      }
    };
    emit(h.root, 1);
    *out = b.release();
    return RT_OK;
  });
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
  return guarded("rt_trace_batch", [&]() -> int {
    if (!sc) return fail(RT_ERR_INVALID, "rt_trace_batch: scene is null");
    if (n == 0) return RT_OK;
    if (!org || !dir || !prim || !t || !normal || !front_face) return fail(RT_ERR_INVALID, "rt_trace_batch: null buffer");
    if (std::isnan(t_min) || std::isnan(t_max)) return fail(RT_ERR_INVALID, "rt_trace_batch: NaN interval");
    const Replica& rp = *sc->reps[0];
    DeviceGuard g(rp.device);
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
    SceneDev dev = rp.dev;
    DevBuf gstack;
    const uint32_t global_entries = dev.wide ? (dev.stack_cap > dev.lds_stack ? dev.stack_cap - dev.lds_stack : 0u) : (dev.stack_in_lds ? 0u : dev.stack_cap);
    if (global_entries) {  // stack entries that do not live in LDS: one column of global memory per thread of the launch
      const size_t blocks = std::min<size_t>((n + kWorkgroup - 1) / kWorkgroup, (size_t)rp.num_cus);
      dev.gstack_stride = (uint32_t)(blocks * kWorkgroup);
      RT_HIP(gstack.alloc((size_t)dev.gstack_stride * global_entries * 4));
      dev.gstack = gstack.as<uint32_t>();
    }
    RT_HIP(launch_trace_batch(dev, d_org.as<double>(), d_dir.as<double>(), n, t_min, t_max, d_prim.as<int32_t>(), d_t.as<double>(),
                              d_n.as<double>(), d_f.as<uint8_t>(), nullptr, rp.num_cus, nullptr));
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(prim, d_prim.p, n * 4, hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(t, d_t.p, n * 8, hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(normal, d_n.p, n * 24, hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(front_face, d_f.p, n, hipMemcpyDeviceToHost));
    return RT_OK;
  });
}

int rt_scatter_batch(const rt_scene* sc, const uint32_t* material, const double* in_dir, const double* p, const double* normal,
                     const uint8_t* front_face, const double* u, size_t n, uint8_t* scattered, double* attenuation, double* out_dir) {
  return guarded("rt_scatter_batch", [&]() -> int {
    if (!sc) return fail(RT_ERR_INVALID, "rt_scatter_batch: scene is null");
    if (n == 0) return RT_OK;
    if (!material || !in_dir || !p || !normal || !front_face || !u || !scattered || !attenuation || !out_dir)
      return fail(RT_ERR_INVALID, "rt_scatter_batch: null buffer");
    std::vector<uint32_t> tagged(n);  // material index with its type in the top two bits, as the kernels carry it
    for (size_t i = 0; i < n; ++i) {
      if (material[i] >= sc->n_materials) return fail(RT_ERR_INVALID, "rt_scatter_batch: material index out of range");
      const uint32_t at = sc->table_index[material[i]];
      tagged[i] = at | ((uint32_t)sc->material_type[at] << 30);
    }
    const Replica& rp = *sc->reps[0];
    DeviceGuard g(rp.device);
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
    RT_HIP(launch_scatter_batch(rp.dev, d_m.as<uint32_t>(), d_in.as<double>(), d_p.as<double>(), d_n.as<double>(), d_f.as<uint8_t>(),
                                d_u.as<double>(), n, d_s.as<uint8_t>(), d_a.as<double>(), d_o.as<double>(), nullptr));
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(scattered, d_s.p, n, hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(attenuation, d_a.p, n * 24, hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(out_dir, d_o.p, n * 24, hipMemcpyDeviceToHost));
    return RT_OK;
  });
}

int rt_slab_batch(const double* boxes, const double* org, const double* dir, const double* interval, size_t n, uint8_t* hit) {
  return guarded("rt_slab_batch", [&]() -> int {
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
  });
}

int rt_primary_rays(const rt_config* cfg, uint64_t seed, const uint32_t* ijs, size_t n, double* org, double* dir) {
  return guarded("rt_primary_rays", [&]() -> int {
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
  });
}

int rt_render_shard_device(const rt_scene* sc, const rt_config* cfg, uint64_t seed, const rt_shard* shard, double* rgb_sum_dev, void* stream,
                           rt_render_stats* stats) {
  return guarded("rt_render", [&]() -> int {
    if (!sc) return fail(RT_ERR_INVALID, "rt_render: scene is null");
    if (!rgb_sum_dev) return fail(RT_ERR_INVALID, "rt_render: rgb_sum is null");
    if (!shard) return fail(RT_ERR_INVALID, "rt_render: shard is null");
    CameraDev cam;
    int rc = make_camera_dev(cfg, &cam, "rt_render");
    if (rc) return rc;
    rc = check_shard(cam, *shard, "rt_render");
    if (rc) return rc;
    Replica& rp = *sc->reps[0];
    DeviceGuard g(rp.device);
    RT_HIP(g.status);
    return render_shard(rp, cam, seed, *shard, rgb_sum_dev, static_cast<hipStream_t>(stream), stats);
  });
}

int rt_render_device(const rt_scene* sc, const rt_config* cfg, uint64_t seed, uint32_t spp_begin, uint32_t spp_end, const rt_rect* tile,
                     double* rgb_sum_dev, void* stream, rt_render_stats* stats) {
  if (!cfg) return fail(RT_ERR_INVALID, "rt_render: cfg is null");
  rt_camera c;
  int rc = rt_camera_derive(cfg, &c);
  if (rc) return rc;
  rt_shard sh{tile ? *tile : rt_rect{0, 0, cfg->image_width, c.image_height}, 0, 1, spp_begin, spp_end};
  return rt_render_shard_device(sc, cfg, seed, &sh, rgb_sum_dev, stream, stats);
}

namespace {
// One host thread at a time enqueues work on a GPU's static workspace.  (This keeps two callers from interleaving their memsets and
// launches; it cannot see a call whose kernels are still running on another stream after it returned: that remains the caller's to avoid.)
std::atomic_flag g_static_busy[64] = {};
struct StaticBusy {
  std::atomic_flag* f = nullptr;
  ~StaticBusy() {
    if (f) f->clear(std::memory_order_release);
  }
};
}  // namespace

int rt_render_device_noalloc(const rt_scene* sc, const rt_config* cfg, uint64_t seed, uint32_t spp_begin, uint32_t spp_end,
                             const rt_rect* tile, double* rgb_sum_dev, void* stream, rt_render_stats* stats) {
  return guarded("rt_render_device_noalloc", [&]() -> int {
    if (!sc) return fail(RT_ERR_INVALID, "rt_render_device_noalloc: scene is null");
    if (!rgb_sum_dev) return fail(RT_ERR_INVALID, "rt_render_device_noalloc: rgb_sum is null");
    CameraDev cam;
    int rc = make_camera_dev(cfg, &cam, "rt_render_device_noalloc");
    if (rc) return rc;
    const rt_shard sh{tile ? *tile : rt_rect{0, 0, cam.width, cam.height}, 0, 1, spp_begin, spp_end};
    rc = check_shard(cam, sh, "rt_render_device_noalloc");
    if (rc) return rc;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    Replica& rp = *sc->reps[0];
    DeviceGuard g(rp.device);
    RT_HIP(g.status);
    StaticBusy busy;
    if (rp.device >= 0 && rp.device < 64) {
      if (g_static_busy[rp.device].test_and_set(std::memory_order_acquire))
        return fail(RT_ERR_INVALID, "rt_render_device_noalloc: another call is enqueueing on this GPU's static workspace; calls on one GPU must not overlap");
      busy.f = &g_static_busy[rp.device];
    }
    Workspace ws;
    rc = static_workspace(rp.device, &ws);
    if (rc) return rc;
    // Cut into chunks exactly as rt_render_device does, so that the sums come out the same; a request whose
    // per-chunk sums or traversal stacks do not fit in the static array is refused, not rendered differently.
    Launch L;
    rc = plan_render(rp, cam, seed, sh, kWorkspaceLimit, &L);
    if (rc) return rc;
    if (L.nothing) {
      if (stats && cam.max_depth == 0) stats->samples = L.samples;
      return RT_OK;
    }
    if (L.partial_bytes > ws.partial_cap)
      return fail(RT_ERR_LIMIT, "rt_render_device_noalloc: the request needs " + std::to_string(L.partial_bytes) +
                                    " bytes of per-chunk sums, the static workspace holds " + std::to_string(ws.partial_cap) +
                                    "; render a smaller tile or sample range per call, or use rt_render_device");
    // Behind the per-chunk sums: the traversal stacks' overflow columns (scenes walked through the four-wide tree, very deep
    // trees), which must fit, and the tiles' entry lists, which are left out if they do not (the frame is the same without).
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    char* const base = reinterpret_cast<char*>(ws.counters);
    const size_t total = ws.partial_cap + 256;
    size_t used = 256 + up(L.partial_bytes);
    if (L.gstack_bytes) {
      if (used + L.gstack_bytes > total)
        return fail(RT_ERR_LIMIT, "rt_render_device_noalloc: the request needs " + std::to_string(L.partial_bytes) + " bytes of per-chunk sums and " +
                                      std::to_string(L.gstack_bytes) + " bytes of traversal stacks, the static workspace holds " +
                                      std::to_string(total - 256) + "; render a smaller tile or sample range per call, or use rt_render_device");
      ws.gstack = reinterpret_cast<uint32_t*>(base + used);
      used += up(L.gstack_bytes);
    }
    if (L.cuts_bytes && used + L.cuts_bytes <= total) ws.cuts = reinterpret_cast<uint32_t*>(base + used);
    uint32_t launches = 0;
    rc = enqueue_render(L, ws, rgb_sum_dev, static_cast<hipStream_t>(stream), false, &launches);
    if (rc) return rc;
    if (stats) {  // the counters can only be read once the kernels have finished
      RT_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
      rc = read_stats(L, ws, launches, false, stats);
    }
    return rc;
  });
}

#ifdef RTA_EXPERIMENTS  // built into libraytracer_amd_experiments.so only
int rt_experiment_wavefront_device(const rt_scene* sc, const rt_config* cfg, uint64_t seed, uint32_t spp_begin, uint32_t spp_end,
                                   double* rgb_sum_dev, rt_render_stats* stats) {
  return guarded("rt_experiment_wavefront_device", [&]() -> int {
    if (!sc || !rgb_sum_dev) return fail(RT_ERR_INVALID, "rt_experiment_wavefront_device: null argument");
    CameraDev cam;
    int rc = make_camera_dev(cfg, &cam, "rt_experiment_wavefront_device");
    if (rc) return rc;
    if (spp_end < spp_begin) return fail(RT_ERR_INVALID, "rt_experiment_wavefront_device: spp_end < spp_begin");
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (spp_end == spp_begin || cam.max_depth == 0) return RT_OK;
    const Replica& rp = *sc->reps[0];
    DeviceGuard g(rp.device);
    RT_HIP(g.status);
    const uint64_t n_pixels = (uint64_t)cam.width * cam.height;
    if (n_pixels > (1u << 26)) return fail(RT_ERR_LIMIT, "rt_experiment_wavefront_device: at most 2^26 pixels");
    // paths in flight: as many whole samples of the frame as fit in the budget (32 million by default), at least one
    const uint64_t batch = std::max<uint64_t>(1, std::min<uint64_t>(spp_end - spp_begin, g_wavefront_paths.load() / n_pixels));
    if (batch * n_pixels > 0xfff00000ull) return fail(RT_ERR_LIMIT, "rt_experiment_wavefront_device: too many paths in flight");
    const uint32_t capacity = (uint32_t)(batch * n_pixels);
    SceneDev dev = rp.dev;
    DevBuf work, gstack;
    RT_HIP(work.alloc(wavefront_workspace_bytes(capacity)));
    if (dev.wide && dev.stack_cap > dev.lds_stack) {
      dev.gstack_stride = (uint32_t)rp.num_cus * kWorkgroup;
      RT_HIP(gstack.alloc((size_t)dev.gstack_stride * (dev.stack_cap - dev.lds_stack) * 4));
      dev.gstack = gstack.as<uint32_t>();
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    RT_HIP(hipEventCreate(&ev0));
    RT_HIP(hipEventCreate(&ev1));
    RT_HIP(hipEventRecord(ev0, nullptr));
    uint64_t rays = 0;
    uint32_t launches = 0;
    hipError_t e = wavefront_render(dev, cam, seed, spp_begin, spp_end, rgb_sum_dev, work.p, capacity, rp.num_cus, nullptr, &rays, &launches);
    hipError_t e2 = hipEventRecord(ev1, nullptr);
    if (e == hipSuccess) e = e2;
    if (e == hipSuccess) e = hipEventSynchronize(ev1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev0, ev1);
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    RT_HIP(e);
    if (stats) {
      stats->rays = rays;
      stats->samples = n_pixels * (spp_end - spp_begin);
      stats->kernel_ms = ms;
      stats->launches = launches;
    }
    return RT_OK;
  });
}
#endif

int rt_render(const rt_scene* sc, const rt_config* cfg, uint64_t seed, uint32_t spp_begin, uint32_t spp_end, const rt_rect* tile,
              double* rgb_sum, rt_render_stats* stats) {
  return guarded("rt_render", [&]() -> int {
    if (!sc) return fail(RT_ERR_INVALID, "rt_render: scene is null");
    if (!rgb_sum) return fail(RT_ERR_INVALID, "rt_render: rgb_sum is null");
    CameraDev cam;
    int rc = make_camera_dev(cfg, &cam, "rt_render");
    if (rc) return rc;
    DeviceGuard g(sc->reps[0]->device);
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
  });
}

int rt_partition(const rt_config* cfg, uint32_t spp_begin, uint32_t spp_end, const rt_rect* tile, uint32_t partition, uint32_t n_devices,
                 rt_shard* shards) {
  return guarded("rt_partition", [&]() -> int {
    if (!shards) return fail(RT_ERR_INVALID, "rt_partition: shards is null");
    if (partition > RT_PARTITION_SAMPLES) return fail(RT_ERR_INVALID, "rt_partition: unknown partition");
    if (n_devices == 0) return fail(RT_ERR_INVALID, "rt_partition: n_devices is 0");
    CameraDev cam;
    int rc = make_camera_dev(cfg, &cam, "rt_partition");
    if (rc) return rc;
    const rt_rect r = tile ? *tile : rt_rect{0, 0, cam.width, cam.height};
    const rt_shard whole{r, 0, 1, spp_begin, spp_end};
    rc = check_shard(cam, whole, "rt_partition");
    if (rc) return rc;
    partition_request(partition, n_devices, r, spp_begin, spp_end, shards);
    return RT_OK;
  });
}

int rt_render_multi(const rt_scene* sc_, const rt_config* cfg, uint64_t seed, uint32_t spp_begin, uint32_t spp_end, const rt_rect* tile,
                    uint32_t partition, double* rgb_sum, rt_render_stats* stats) {
  return guarded("rt_render_multi", [&]() -> int {
    rt_scene* sc = const_cast<rt_scene*>(sc_);  // the replicas' frame buffers are scratch state behind multi_mu
    if (!sc) return fail(RT_ERR_INVALID, "rt_render_multi: scene is null");
    if (!rgb_sum) return fail(RT_ERR_INVALID, "rt_render_multi: rgb_sum is null");
    if (partition > RT_PARTITION_SAMPLES) return fail(RT_ERR_INVALID, "rt_render_multi: unknown partition");
    CameraDev cam;
    int rc = make_camera_dev(cfg, &cam, "rt_render_multi");
    if (rc) return rc;
    const rt_rect r = tile ? *tile : rt_rect{0, 0, cam.width, cam.height};
    const rt_shard whole{r, 0, 1, spp_begin, spp_end};
    rc = check_shard(cam, whole, "rt_render_multi");
    if (rc) return rc;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    std::lock_guard<std::mutex> lock(sc->multi_mu);
    std::vector<rt_shard> shards;
    rc = render_replicas(sc, cam, seed, partition, r, spp_begin, spp_end, &shards, stats);
    if (rc) return rc;
    const uint32_t tw = r.x1 - r.x0;
    if (tw == 0 || r.y1 == r.y0) return RT_OK;
    const size_t row_bytes = (size_t)cam.width * 24;
    // Back to the host through pinned memory, each replica its own rows, and added into the caller's frame.
    const size_t R = partition == RT_PARTITION_TILES ? sc->reps.size() : 1;  // samples: replica 0 holds the sum
    return for_each_replica(R, [&](size_t g) -> int {
      Replica& rp = *sc->reps[g];
      DeviceGuard dg(rp.device);
      RT_HIP(dg.status);
      const rt_shard& sh = partition == RT_PARTITION_TILES ? shards[g] : whole;
      const uint32_t rows = shard_tile_rows(sh);
      size_t need = 0;
      for (uint32_t k = 0; k < rows; ++k) need += std::min(8u, (r.y1 - r.y0) - (k * sh.row_stride + sh.row_first) * 8u) * row_bytes;
      if (need > rp.pinned_cap) {
        if (rp.pinned) (void)hipHostFree(rp.pinned);
        rp.pinned = nullptr;
        rp.pinned_cap = 0;
        RT_HIP(hipHostMalloc(&rp.pinned, need, hipHostMallocDefault));
        rp.pinned_cap = need;
      }
      char* stage = static_cast<char*>(rp.pinned);
      size_t at = 0;
      for (uint32_t k = 0; k < rows; ++k) {  // one copy per band of consecutive rows
        const uint32_t y = r.y0 + (k * sh.row_stride + sh.row_first) * 8u, nrow = std::min(8u, r.y1 - y);
        RT_HIP(hipMemcpyAsync(stage + at, rp.frame + (size_t)y * cam.width * 3, nrow * row_bytes, hipMemcpyDeviceToHost, rp.stream));
        at += nrow * row_bytes;
      }
      RT_HIP(hipStreamSynchronize(rp.stream));
      at = 0;
      for (uint32_t k = 0; k < rows; ++k) {
        const uint32_t y = r.y0 + (k * sh.row_stride + sh.row_first) * 8u, nrow = std::min(8u, r.y1 - y);
        for (uint32_t dy = 0; dy < nrow; ++dy) {
          const double* src = reinterpret_cast<const double*>(stage + at + dy * row_bytes) + 3 * (size_t)r.x0;
          double* dst = rgb_sum + 3 * ((size_t)(y + dy) * cam.width + r.x0);
          for (size_t i = 0; i < 3 * (size_t)tw; ++i) dst[i] += src[i];
        }
        at += nrow * row_bytes;
      }
      return RT_OK;
    });
  });
}

int rt_debug_take_out_first_leaf(const void* nodes, size_t n_nodes, uint32_t root, void* nodes_out, uint32_t* first_leaf, uint32_t* walk_root) {
  return guarded("rt_debug_take_out_first_leaf", [&]() -> int {
    if ((!nodes && n_nodes) || !nodes_out || !first_leaf || !walk_root) return fail(RT_ERR_INVALID, "rt_debug_take_out_first_leaf: null argument");
    const DevNode* in = static_cast<const DevNode*>(nodes);
    std::vector<DevNode> edited;
    *first_leaf = kRefNone;
    *walk_root = root;
    take_out_first_leaf(in, n_nodes, root, &edited, first_leaf, walk_root);
    if (n_nodes) std::memcpy(nodes_out, edited.empty() ? in : edited.data(), n_nodes * sizeof(DevNode));
    return RT_OK;
  });
}

int rt_debug_set(const char* name, double value) {
  return guarded("rt_debug_set", [&]() -> int {
    if (!name) return fail(RT_ERR_INVALID, "rt_debug_set: name is null");
    const std::string key(name);
    // A wave's traversal loop ends when fewer than inner_min lanes want the inner block and fewer than leaf_min
    // the leaf block: with a threshold of 0 it would never end.
    auto lanes = [&](std::atomic<uint32_t>& knob, double lo) -> int {
      if (!(value >= lo && value <= 64.0) || value != std::floor(value))
        return fail(RT_ERR_INVALID, "rt_debug_set: " + key + " must be a whole number in " + std::to_string((int)lo) + " .. 64");
      knob.store((uint32_t)value);
      return RT_OK;
    };
    if (key == "inner_min") return lanes(g_inner_min, 1.0);
    if (key == "leaf_min") return lanes(g_leaf_min, 1.0);
    if (key == "both_min") return lanes(g_both_min, 0.0);
    if (key == "batches_per_wave") {
      if (!(value >= 1.0 && value <= 1e9)) return fail(RT_ERR_INVALID, "rt_debug_set: batches_per_wave must be in 1 .. 1e9");
      g_batches_per_wave.store((uint64_t)value);
      return RT_OK;
    }
    if (key == "force_global_stack") {
      g_force_global_stack.store(value != 0.0);
      return RT_OK;
    }
#ifdef RTA_EXPERIMENTS
    if (key == "wavefront_paths") {
      if (!(value >= 1.0 && value <= 4e9)) return fail(RT_ERR_INVALID, "rt_debug_set: wavefront_paths must be in 1 .. 4e9");
      g_wavefront_paths.store((uint64_t)value);
      return RT_OK;
    }
#endif
    if (key == "force_wide") {
      g_force_wide.store(value != 0.0);
      return RT_OK;
    }
    if (key == "wide_lds_stack") {
      if (!(value >= 0.0 && value <= 30.0)) return fail(RT_ERR_INVALID, "rt_debug_set: wide_lds_stack must be in 0 .. 30");
      g_wide_lds_stack.store((uint32_t)value);
      return RT_OK;
    }
    if (key == "force_staged_sum") {
      g_force_staged_sum.store(value != 0.0);
      return RT_OK;
    }
    if (key == "target_units") {
      if (!(value >= 1.0 && value <= 1e12)) return fail(RT_ERR_INVALID, "rt_debug_set: target_units must be in 1 .. 1e12");
      g_target_units.store((uint64_t)value);
      return RT_OK;
    }
    if (key == "first_leaf") {  // takes effect for scenes created afterwards
      g_first_leaf.store(value != 0.0 ? 1u : 0u);
      return RT_OK;
    }
    if (key == "tile_cuts") {
      if (!(value >= 0.0 && value <= (double)kWideCutEntries) || value != std::floor(value))
        return fail(RT_ERR_INVALID, "rt_debug_set: tile_cuts must be a whole number in 0 .. " + std::to_string(kWideCutEntries));
      g_tile_cuts.store((uint32_t)value);
      return RT_OK;
    }
    if (key == "quad_tiles") {
      g_quad_tiles.store(value != 0.0 ? 1u : 0u);
      return RT_OK;
    }
    if (key == "chunk_spp") {
      if (!(value >= 1.0 && value <= 1e9)) return fail(RT_ERR_INVALID, "rt_debug_set: chunk_spp must be in 1 .. 1e9");
      g_chunk_spp.store((uint32_t)value);
      return RT_OK;
    }
    return fail(RT_ERR_INVALID, "rt_debug_set: unknown tunable " + key);
  });
}

int rt_quantise(const double* rgb_sum, size_t n_pixels, uint64_t spp, uint8_t* out) {
  if (n_pixels && (!rgb_sum || !out)) return fail(RT_ERR_INVALID, "rt_quantise: null buffer");
  quantise(rgb_sum, n_pixels, spp, out);
  return RT_OK;
}

int rt_quantise_device(const double* rgb_sum_dev, size_t n_pixels, uint64_t spp, uint8_t* out_dev, void* stream) {
  return guarded("rt_quantise_device", [&]() -> int {
    if (n_pixels && (!rgb_sum_dev || !out_dev)) return fail(RT_ERR_INVALID, "rt_quantise_device: null buffer");
    if (((uintptr_t)rgb_sum_dev & 15) || ((uintptr_t)out_dev & 1)) return fail(RT_ERR_INVALID, "rt_quantise_device: rgb_sum must be 16-byte and out 2-byte aligned");
    if (spp == 0) return fail(RT_ERR_INVALID, "rt_quantise_device: samples_per_pixel is 0");
    RT_HIP(launch_quantise(rgb_sum_dev, 3 * n_pixels, spp, out_dev, (hipStream_t)stream));
    RT_HIP(hipStreamSynchronize((hipStream_t)stream));
    return RT_OK;
  });
}

int rt_write_ppm_bytes(int fd, const uint8_t* rgb, uint32_t width, uint32_t height) {
  return guarded("rt_write_ppm_bytes", [&]() -> int {
    if (!rgb && width && height) return fail(RT_ERR_INVALID, "rt_write_ppm_bytes: rgb is null");
    if (!write_ppm_bytes(fd, rgb, width, height)) return fail(RT_ERR_IO, "rt_write_ppm_bytes: write failed");
    return RT_OK;
  });
}

int rt_render_ppm(const rt_scene* sc_, const rt_config* cfg, uint64_t seed, int fd, rt_render_stats* stats) {
  return guarded("rt_render_ppm", [&]() -> int {
    rt_scene* sc = const_cast<rt_scene*>(sc_);
    if (!sc) return fail(RT_ERR_INVALID, "rt_render_ppm: scene is null");
    CameraDev cam;
    int rc = make_camera_dev(cfg, &cam, "rt_render_ppm");
    if (rc) return rc;
    if (cfg->samples_per_pixel == 0 || cfg->samples_per_pixel > 0xffffffffull) return fail(RT_ERR_INVALID, "rt_render_ppm: samples_per_pixel must be in 1 .. 2^32-1");
    if (cam.width > 0xffffu || cam.height > 0xffffu) return fail(RT_ERR_LIMIT, "rt_render_ppm: images larger than 65535 pixels on a side are not supported");
    const uint32_t spp = (uint32_t)cfg->samples_per_pixel;
    const size_t n_px = (size_t)cam.width * cam.height;
    const rt_rect r{0, 0, cam.width, cam.height};
    std::lock_guard<std::mutex> lock(sc->multi_mu);
    // Tile rows dealt over the scene's devices: each renders, quantises and sends back its own rows, so the
    // text does not depend on how many devices there are.
    std::vector<rt_shard> shards;
    rc = render_replicas(sc, cam, seed, RT_PARTITION_TILES, r, 0, spp, &shards, stats);
    if (rc) return rc;
    std::vector<uint8_t> host(3 * n_px);
    rc = for_each_replica(sc->reps.size(), [&](size_t g) -> int {
      Replica& rp = *sc->reps[g];
      DeviceGuard dg(rp.device);
      RT_HIP(dg.status);
      size_t cap = rp.bytes_cap;
      RT_HIP(grow_device((void**)&rp.bytes, &cap, 3 * n_px + 2));
      rp.bytes_cap = cap;
      RT_HIP(launch_quantise(rp.frame, 3 * n_px, spp, rp.bytes, rp.stream));
      const rt_shard& sh = shards[g];
      const uint32_t rows = shard_tile_rows(sh);
      for (uint32_t k = 0; k < rows; ++k) {
        const uint32_t y = (k * sh.row_stride + sh.row_first) * 8u, nrow = std::min(8u, cam.height - y);
        const size_t off = (size_t)y * cam.width * 3;
        RT_HIP(hipMemcpyAsync(host.data() + off, rp.bytes + off, (size_t)nrow * cam.width * 3, hipMemcpyDeviceToHost, rp.stream));
      }
      RT_HIP(hipStreamSynchronize(rp.stream));
      return RT_OK;
    });
    if (rc) return rc;
    if (!write_ppm_bytes(fd, host.data(), cam.width, cam.height)) return fail(RT_ERR_IO, "rt_render_ppm: write failed");
    return RT_OK;
  });
}

int rt_write_ppm(int fd, const double* rgb_sum, uint32_t width, uint32_t height, uint64_t spp) {
  return guarded("rt_write_ppm", [&]() -> int {
    if (!rgb_sum && width && height) return fail(RT_ERR_INVALID, "rt_write_ppm: rgb_sum is null");
    if (!write_ppm(fd, rgb_sum, width, height, spp)) return fail(RT_ERR_IO, "rt_write_ppm: write failed");
    return RT_OK;
  });
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
