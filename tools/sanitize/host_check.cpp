// Host-side code of the library and the CPU oracle under the sanitizers (no GPU, no HIP):
//   bvh_builder.cpp  threaded SAH build (1 and 16 threads must give the same tree), flatten, four-wide collapse
//   rt_host.cpp      camera_derive, quantise, write_ppm (threaded), the seeded scene generators
//   oracle           its tree, a small threaded render, trace_batch
// Built and run by tools/sanitize/run.sh with -fsanitize=address,undefined and again with -fsanitize=thread.
//   host_check [max_spheres]     (default 1000000)
#include <fcntl.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "bvh_builder.h"
#include "rt.h"
#include "rt_host.h"

extern "C" {
struct OrcSphere;
void* orc_scene_create(const void* spheres, size_t n, const void* mats, size_t nm);
void orc_scene_destroy(void* h);
size_t orc_bvh_node_count(void* h);
void orc_render(void* h, const void* cfg, uint64_t seed, uint32_t s0, uint32_t s1, const uint32_t* tile4, int mode, int nthreads, double* rgb_sum,
                uint64_t* stats);
void orc_trace_batch(void* h, const double* org, const double* dir, size_t n, double tmin, double tmax, int32_t* prim, double* t, double* p,
                     double* normal, uint8_t* front, uint64_t* stats);
}

using namespace rta;

static int failures = 0;
#define CHECK(c)                                                   \
  do {                                                             \
    if (!(c)) {                                                    \
      std::fprintf(stderr, "host_check: %s failed (line %d)\n", #c, __LINE__); \
      ++failures;                                                  \
    }                                                              \
  } while (0)

static bool same_tree(const HostBvh& a, const HostBvh& b) {
  if (a.nodes.size() != b.nodes.size() || a.leaf_indices != b.leaf_indices || a.depth != b.depth) return false;
  for (size_t i = 0; i < a.nodes.size(); ++i) {
    const HostNode &x = a.nodes[i], &y = b.nodes[i];
    if (std::memcmp(&x.box, &y.box, sizeof(Box)) || x.kind != y.kind || (x.kind == 3 ? x.count != y.count || x.first != y.first : x.skip != y.skip))
      return false;
  }
  return true;
}

static void check_scene(const char* name, const std::vector<rt_sphere>& s) {
  setenv("RT_HOST_THREADS", "1", 1);
  const HostBvh one = build_bvh(s.data(), s.size());
  setenv("RT_HOST_THREADS", "16", 1);
  const HostBvh many = build_bvh(s.data(), s.size());
  CHECK(same_tree(one, many));
  CHECK(one.nodes.size() >= 1 && one.leaf_indices.size() == s.size());
  const FlatBvh flat = flatten_bvh(many, s.data(), s.size());
  CHECK(flat.spheres.size() == s.size() && flat.aux.size() == s.size());
  size_t wide_nodes = 0;
  if (!(flat.root & kRefLeaf) && flat.root != kRefNone) {
    const WideBvh wide = collapse_wide(flat.nodes.data(), flat.nodes.size(), flat.root);
    wide_nodes = wide.nodes.size();
    CHECK(wide_nodes >= 1 && wide.stack_entries >= 1);
  }
  std::fprintf(stderr, "host_check: %-8s %8zu spheres: %zu nodes, depth %u, %zu device nodes, %zu four-wide nodes; 1 and 16 threads agree\n", name, s.size(),
               one.nodes.size(), one.depth, flat.nodes.size(), wide_nodes);
}

int main(int argc, char** argv) {
  const size_t max_spheres = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1000000;
  // scenes
  std::vector<rt_sphere> rtiow(488);
  std::vector<rt_material> rtiow_m(488);
  scene_rtiow(1, rtiow.data(), rtiow_m.data());
  check_scene("rtiow", rtiow);
  for (size_t n : {(size_t)40000, (size_t)1000000}) {  // (the builder's threads take part from 32 768 spheres on)
    if (n > max_spheres) continue;
    std::vector<rt_sphere> s(n + 1);
    std::vector<rt_material> m(n + 1);
    scene_field(3, n, 1.0 / 3.0, 1.0 / 3.0, 1.0, s.data(), m.data());
    check_scene("field", s);
  }
  {  // degenerate inputs: one sphere, none, many at one centre (leaves cut into chains)
    std::vector<rt_sphere> s(300, rt_sphere{{1.0, 2.0, 3.0}, 0.5, 0, 0});
    check_scene("coincide", s);
    s.resize(1);
    check_scene("single", s);
    const HostBvh empty = build_bvh(nullptr, 0);
    CHECK(empty.leaf_indices.empty());
  }
  // camera, quantiser, PPM text
  rt_config cfg;
  config_default(&cfg);
  CHECK(config_check(cfg).empty());
  cfg.image_width = 830;
  cfg.aspect_ratio = 16.0 / 9.0;
  rt_camera cam;
  camera_derive(cfg, &cam);
  CHECK(cam.image_height == 466);
  const uint32_t w = 257, h = 131;
  std::vector<double> sums((size_t)w * h * 3);
  for (size_t i = 0; i < sums.size(); ++i) sums[i] = (double)((i * 2654435761u) % 1000) / 100.0 - 1.0;
  sums[0] = __builtin_nan("");
  sums[1] = __builtin_inf();
  sums[2] = -__builtin_inf();
  std::vector<uint8_t> bytes(sums.size());
  quantise(sums.data(), (size_t)w * h, 7, bytes.data());
  CHECK(bytes[0] == 0 && bytes[1] == 255 && bytes[2] == 0);
  int fd = open("/dev/null", O_WRONLY);
  CHECK(fd >= 0 && write_ppm(fd, sums.data(), w, h, 7) && write_ppm_bytes(fd, bytes.data(), w, h));
  close(fd);
  // the oracle: tree, a threaded render, closest hits
  struct OrcMaterial {
    uint32_t type, pad;
    double albedo[3], param;
  };
  static_assert(sizeof(OrcMaterial) == sizeof(rt_material), "the oracle's records have the library's layout");
  void* scene = orc_scene_create(rtiow.data(), rtiow.size(), rtiow_m.data(), rtiow_m.size());
  CHECK(scene && orc_bvh_node_count(scene) == 975);
  rt_config small = cfg;
  small.image_width = 48;
  small.samples_per_pixel = 4;
  small.max_depth = 50;
  small.vfov = 20.0;
  small.lookfrom[0] = 13.0, small.lookfrom[1] = 2.0, small.lookfrom[2] = 3.0;
  small.lookat[0] = small.lookat[1] = small.lookat[2] = 0.0;
  small.defocus_angle = 0.6;
  small.focus_dist = 10.0;
  std::vector<double> img(48 * 27 * 3, 0.0);
  const uint32_t tile[4] = {0, 0, 48, 27};
  uint64_t st[3] = {0, 0, 0};
  orc_render(scene, &small, 12345, 0, 4, tile, 0, 8, img.data(), st);
  CHECK(st[0] >= 48 * 27 * 4);
  const double org[6] = {13, 2, 3, 0, 5, 0}, dir[6] = {-13, -2, -3, 0, -1, 0};
  int32_t prim[2];
  double t[2], nrm[6], p[6];
  uint8_t front[2];
  orc_trace_batch(scene, org, dir, 2, 1e-3, __builtin_inf(), prim, t, p, nrm, front, nullptr);
  CHECK(prim[0] >= 0 && prim[1] >= 0);
  orc_scene_destroy(scene);
  std::fprintf(stderr, failures ? "host_check: %d check(s) failed\n" : "host_check: all checks passed%.0d\n", failures);
  return failures ? 1 : 0;
}
