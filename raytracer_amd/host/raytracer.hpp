// C++ mirror of the reference's host API for the render path, over the C ABI
// of libraytracer_amd.so (include/rt.h).  Same type names, field names, builder
// style (setters take and return by value, src/util.rs:12-16) and call shape as
// the reference, so that src/main.rs reads the same in C++:
//
//   std::vector<std::unique_ptr<Hit>> objects;
//   objects.push_back(std::make_unique<Sphere>(DVec3{0, 1, 0}, 1.0, glass));
//   BVH world(std::move(objects));
//   Config().image_width(830).vfov(20.0) ... .camera().render(world);
//
// Where the reference panics this layer throws rt::Error (the C ABI below it
// never throws).  Header-only; link with -lraytracer_amd.
#pragma once

#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <limits>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "rt.h"

namespace rt {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

inline void check(int status) {
  if (status != RT_OK) throw Error(status, rt_last_error());
}

// glam::DVec3 as far as the path's callers use it.
struct DVec3 {
  double x = 0, y = 0, z = 0;
  static constexpr DVec3 ZERO() { return {0, 0, 0}; }
  static constexpr DVec3 Y() { return {0, 1, 0}; }
  DVec3 operator+(DVec3 o) const { return {x + o.x, y + o.y, z + o.z}; }
  DVec3 operator-(DVec3 o) const { return {x - o.x, y - o.y, z - o.z}; }
  DVec3 operator*(double s) const { return {x * s, y * s, z * s}; }
  double dot(DVec3 o) const { return x * o.x + y * o.y + z * o.z; }
  double length() const { return std::sqrt(dot(*this)); }
  DVec3 normalize() const { return *this * (1.0 / length()); }
};

// src/ray.rs:25-42
struct Interval {
  double min, max;
  bool surrounds(double v) const { return min < v && v < max; }
  double clamp(double v) const { return v < min ? min : (v > max ? max : v); }
};

// src/ray.rs:3-23: the direction is normalised on construction.
struct Ray {
  DVec3 origin, direction, reciprocal;
  Ray(DVec3 o, DVec3 d) : origin(o), direction(d.normalize()) {
    reciprocal = {1.0 / direction.x, 1.0 / direction.y, 1.0 / direction.z};
  }
  DVec3 at(double t) const { return origin + direction * t; }
};

// src/material.rs:6-19.  The trait's scatter() runs on the device; what the host
// side needs from a material is its flat record.
struct Material {
  virtual ~Material() = default;
  virtual rt_material gpu_material() const = 0;
};

struct Lambertian final : Material {
  DVec3 albedo_{};
  Lambertian albedo(DVec3 a) const {
    Lambertian m = *this;
    m.albedo_ = a;
    return m;
  }
  rt_material gpu_material() const override { return {RT_LAMBERTIAN, 0, {albedo_.x, albedo_.y, albedo_.z}, 0.0}; }
};

struct Metal final : Material {
  DVec3 albedo_{};
  double fuzz_ = 0.0;
  Metal albedo(DVec3 a) const {
    Metal m = *this;
    m.albedo_ = a;
    return m;
  }
  Metal fuzz(double f) const {
    Metal m = *this;
    m.fuzz_ = f;
    return m;
  }
  rt_material gpu_material() const override { return {RT_METAL, 0, {albedo_.x, albedo_.y, albedo_.z}, fuzz_}; }
};

struct Dielectric final : Material {
  double ir_ = 1.5;
  Dielectric ir(double v) const {
    Dielectric m = *this;
    m.ir_ = v;
    return m;
  }
  rt_material gpu_material() const override { return {RT_DIELECTRIC, 0, {0, 0, 0}, ir_}; }
};

// src/hit.rs:9-15 (the material is given by index into BVH::materials()).
struct HitRecord {
  bool front_face;
  const Material* material;
  DVec3 normal, p;
  double t;
  size_t primitive;
  Ray ray(DVec3 direction) const { return Ray(p, direction); }
};

// src/hit.rs:34-38.  hit()/aabb() are evaluated by the library; a primitive
// only has to say which sphere it is.
struct Sphere;
struct Hit {
  virtual ~Hit() = default;
  virtual const Sphere* gpu_sphere() const = 0;
};

// src/sphere.rs:12-27
struct Sphere final : Hit {
  DVec3 center;
  double radius;
  std::shared_ptr<const Material> material;
  Sphere(DVec3 c, double r, std::shared_ptr<const Material> m) : center(c), radius(r), material(std::move(m)) {}
  const Sphere* gpu_sphere() const override { return this; }
};

// src/bvh.rs:38-48: owns the primitives; the tree lives in the library.
class BVH {
 public:
  // Which GPUs hold a copy of the scene: the one that is current in the calling thread (what a process that was given
  // one GPU of a node wants, and the default), or every GPU the process can see (one process driving the whole node;
  // also chosen by RT_DEVICES=all in the environment).
  enum class Devices { Current, All };
  explicit BVH(std::vector<std::unique_ptr<Hit>> objects, Devices devices = Devices::Current) : objects_(std::move(objects)) {
    std::vector<rt_sphere> spheres;
    std::vector<rt_material> materials;
    std::unordered_map<const Material*, uint32_t> seen;  // an Arc shared by several spheres becomes one record
    spheres.reserve(objects_.size());
    for (const auto& o : objects_) {
      const Sphere* s = o->gpu_sphere();
      auto it = seen.find(s->material.get());
      if (it == seen.end()) {
        it = seen.emplace(s->material.get(), (uint32_t)materials.size()).first;
        materials.push_back(s->material->gpu_material());
        material_ptrs_.push_back(s->material.get());
      }
      spheres.push_back({{s->center.x, s->center.y, s->center.z}, s->radius, it->second, 0});
    }
    const char* env = std::getenv("RT_DEVICES");
    if (devices == Devices::All || (env && std::string(env) == "all")) {
      // one copy of the tree on every visible GPU (null device list); Camera::render shares a frame out over them
      check(rt_scene_create_multi(spheres.data(), spheres.size(), materials.data(), materials.size(), RT_BUILD_AUTO, nullptr, 0, &scene_));
    } else {
      int current = 0;
      check(rt_device_get(&current));
      const int32_t list[1] = {current};
      check(rt_scene_create_multi(spheres.data(), spheres.size(), materials.data(), materials.size(), RT_BUILD_AUTO, list, 1, &scene_));
    }
  }
  ~BVH() {
    if (scene_) rt_scene_destroy(scene_);
  }
  BVH(const BVH&) = delete;
  BVH& operator=(const BVH&) = delete;

  // src/bvh.rs:50-84
  std::optional<HitRecord> hit(const Ray& r, Interval ray_t) const {
    double org[3] = {r.origin.x, r.origin.y, r.origin.z}, dir[3] = {r.direction.x, r.direction.y, r.direction.z};
    int32_t prim;
    double t, n[3];
    uint8_t front;
    check(rt_trace_batch(scene_, org, dir, 1, ray_t.min, ray_t.max, &prim, &t, n, &front));
    if (prim < 0) return std::nullopt;
    const Sphere* s = objects_[(size_t)prim]->gpu_sphere();
    return HitRecord{front != 0, s->material.get(), {n[0], n[1], n[2]}, r.at(t), t, (size_t)prim};
  }

  const rt_scene* scene() const { return scene_; }
  size_t len() const { return objects_.size(); }

 private:
  std::vector<std::unique_ptr<Hit>> objects_;
  std::vector<const Material*> material_ptrs_;
  rt_scene* scene_ = nullptr;
};

class Camera;

// src/camera.rs:15-26: same fields, same defaults (taken from the library so they cannot drift).
struct Config {
  rt_config c;
  Config() { rt_config_default(&c); }
  static Config make() { return Config(); }
#define RT_SETTER(name, type)   \
  Config name(type v) const {   \
    Config n = *this;           \
    n.c.name = v;               \
    return n;                   \
  }
  RT_SETTER(aspect_ratio, double)
  RT_SETTER(image_width, uint32_t)
  RT_SETTER(samples_per_pixel, uint64_t)
  RT_SETTER(max_depth, uint64_t)
  RT_SETTER(vfov, double)
  RT_SETTER(defocus_angle, double)
  RT_SETTER(focus_dist, double)
#undef RT_SETTER
#define RT_SETTER3(name)                           \
  Config name(DVec3 v) const {                     \
    Config n = *this;                              \
    n.c.name[0] = v.x, n.c.name[1] = v.y, n.c.name[2] = v.z; \
    return n;                                      \
  }
  RT_SETTER3(lookfrom)
  RT_SETTER3(lookat)
  RT_SETTER3(vup)
#undef RT_SETTER3
  inline Camera camera() const;
};

// src/camera.rs:28-37,138-165
class Camera {
 public:
  Config config;
  rt_camera derived;
  uint32_t image_height;
  uint64_t seed = 0;  // the reference seeds from the OS; here the image is a function of the seed

  explicit Camera(const Config& cfg) : config(cfg) {
    check(rt_camera_derive(&cfg.c, &derived));
    image_height = derived.image_height;
  }

  // PPM text on stdout, elapsed render time on stderr, as the reference prints them.  The frame
  // stays on the device until it has been quantised; three bytes per pixel come back.
  rt_render_stats render(const BVH& bvh) const {
    rt_render_stats st{};
    std::fflush(stdout);
    check(rt_render_ppm(bvh.scene(), &config.c, seed, 1, &st));
    std::fprintf(stderr, "Completed in %.3f seconds\n", st.kernel_ms / 1e3);
    return st;
  }
};

inline Camera Config::camera() const { return Camera(*this); }

}  // namespace rt
