// The reference's `fn main` (src/main.rs:42-99) over the MI355X library:
// builds the 488-sphere scene, the BVH and the camera, renders, prints the PPM.
//
//   raytracer [--width 830] [--spp 750] [--depth 50] [--seed 1] [--render-seed 12345] > image.ppm
//
// The reference draws the scene from an OS-seeded generator, so no two runs of
// it give the same scene; here the scene is a function of --seed (Philox), the
// same function as rt_scene_rtiow(), whose records are used directly so that
// this driver, the tests and the benchmark all render the same spheres.
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "raytracer.hpp"

using namespace rt;

static std::shared_ptr<const Material> make_material(const rt_material& m) {
  DVec3 a{m.albedo[0], m.albedo[1], m.albedo[2]};
  switch (m.type) {
    case RT_LAMBERTIAN:
      return std::make_shared<Lambertian>(Lambertian().albedo(a));
    case RT_METAL:
      return std::make_shared<Metal>(Metal().albedo(a).fuzz(m.param));
    default:
      return std::make_shared<Dielectric>(Dielectric().ir(m.param));
  }
}

int main(int argc, char** argv) {
  uint32_t width = 830;
  uint64_t spp = 750, depth = 50, scene_seed = 1, render_seed = 12345;
  for (int i = 1; i + 1 < argc; i += 2) {
    const char* k = argv[i];
    uint64_t v = std::strtoull(argv[i + 1], nullptr, 10);
    if (!std::strcmp(k, "--width")) width = (uint32_t)v;
    else if (!std::strcmp(k, "--spp")) spp = v;
    else if (!std::strcmp(k, "--depth")) depth = v;
    else if (!std::strcmp(k, "--seed")) scene_seed = v;
    else if (!std::strcmp(k, "--render-seed")) render_seed = v;
    else {
      std::fprintf(stderr, "unknown option %s\n", k);
      return 2;
    }
  }
  try {
    // src/main.rs:43-82: ground, three large spheres, 22 x 22 small ones
    std::vector<rt_sphere> flat_s(488);
    std::vector<rt_material> flat_m(488);
    check(rt_scene_rtiow(scene_seed, flat_s.data(), flat_m.data()));
    std::vector<std::unique_ptr<Hit>> objects;
    for (const rt_sphere& s : flat_s) {
      objects.push_back(std::make_unique<Sphere>(DVec3{s.center[0], s.center[1], s.center[2]}, s.radius, make_material(flat_m[s.material])));
    }

    BVH world(std::move(objects));  // src/main.rs:84

    Config config = Config()  // src/main.rs:86-96
                        .aspect_ratio(16.0 / 9.0)
                        .image_width(width)
                        .vfov(20.0)
                        .lookfrom({13.0, 2.0, 3.0})
                        .lookat(DVec3::ZERO())
                        .vup(DVec3::Y())
                        .defocus_angle(0.6)
                        .focus_dist(10.0)
                        .samples_per_pixel(spp)
                        .max_depth(depth);

    Camera camera = config.camera();
    camera.seed = render_seed;
    camera.render(world);  // src/main.rs:98
  } catch (const Error& e) {
    std::fprintf(stderr, "raytracer: %s\n", e.what());
    return 1;
  }
  return 0;
}
