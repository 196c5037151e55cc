#include "rt_host.h"

#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <system_error>
#include <thread>
#include <vector>

namespace rta {

void philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
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

void philox_pair(uint64_t seed, uint32_t a, uint32_t b, uint32_t slot, uint32_t domain, double u[2]) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, ctr[4] = {a, b, slot, domain}, o[4];
  philox4x32_10(key, ctr, o);
  u[0] = (double)((((uint64_t)o[1] << 32) | o[0]) >> 11) * (1.0 / 9007199254740992.0);
  u[1] = (double)((((uint64_t)o[3] << 32) | o[2]) >> 11) * (1.0 / 9007199254740992.0);
}

void config_default(rt_config* c) {  // src/camera.rs:15-26
  std::memset(c, 0, sizeof(*c));
  c->aspect_ratio = 16.0 / 9.0;
  c->image_width = 400;
  c->samples_per_pixel = 10;
  c->max_depth = 10;
  c->vfov = 90.0;
  c->lookfrom[2] = -1.0;  // DVec3::NEG_Z
  c->vup[1] = 1.0;        // DVec3::Y
  c->defocus_angle = 0.0;
  c->focus_dist = 10.0;
}

std::string config_check(const rt_config& c) {
  if (c.image_width == 0) return "image_width is 0";
  if (!(c.aspect_ratio > 0.0) || !std::isfinite(c.aspect_ratio)) return "aspect_ratio must be positive and finite";
  double h = std::fmax((double)c.image_width / c.aspect_ratio, 1.0);
  if (h * (double)c.image_width >= 4294967296.0) return "image has 2^32 pixels or more";
  if (c.max_depth > 0xffffffffull) return "max_depth does not fit 32 bits";
  const double* v[3] = {c.lookfrom, c.lookat, c.vup};
  for (auto p : v)
    for (int k = 0; k < 3; ++k)
      if (!std::isfinite(p[k])) return "camera vectors must be finite";
  if (!std::isfinite(c.vfov) || !std::isfinite(c.defocus_angle) || !std::isfinite(c.focus_dist)) return "camera scalars must be finite";
  return "";
}

namespace {
struct V {
  double x, y, z;
};
inline V operator+(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V operator-(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V operator*(double s, V a) { return {s * a.x, s * a.y, s * a.z}; }
inline V operator/(V a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline V unit(V a) {
  double inv = 1.0 / std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  return inv * a;
}
inline V cross(V a, V b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline void put(double* p, V a) {
  p[0] = a.x;
  p[1] = a.y;
  p[2] = a.z;
}
}  // namespace

void camera_derive(const rt_config& c, rt_camera* out) {  // src/camera.rs:40-82
  const double kPi = 3.14159265358979323846;
  double hf = std::fmax((double)c.image_width / c.aspect_ratio, 1.0);
  uint32_t image_height = hf >= 4294967295.0 ? 4294967295u : (uint32_t)hf;  // `as u32` truncates
  V lookfrom{c.lookfrom[0], c.lookfrom[1], c.lookfrom[2]}, lookat{c.lookat[0], c.lookat[1], c.lookat[2]};
  V vup{c.vup[0], c.vup[1], c.vup[2]};
  V center = lookfrom;
  double theta = c.vfov * (kPi / 180.0);
  double h = std::tan(theta / 2.0);
  double viewport_height = 2.0 * h * c.focus_dist;
  double viewport_width = viewport_height * (double)c.image_width / (double)image_height;
  V w = unit(lookfrom - lookat);
  V u = unit(cross(vup, w));
  V v = cross(w, u);
  V viewport_u = viewport_width * u;
  V viewport_v = viewport_height * V{-v.x, -v.y, -v.z};
  V pixel_delta_u = viewport_u / (double)c.image_width;
  V pixel_delta_v = viewport_v / (double)image_height;
  V viewport_upper_left = center - c.focus_dist * w - (viewport_u + viewport_v) / 2.0;
  V pixel00_loc = viewport_upper_left + (pixel_delta_u + pixel_delta_v) / 2.0;
  double defocus_radius = c.focus_dist * std::tan((c.defocus_angle / 2.0) * (kPi / 180.0));
  std::memset(out, 0, sizeof(*out));
  out->image_height = image_height;
  put(out->center, center);
  put(out->pixel00_loc, pixel00_loc);
  put(out->pixel_delta_u, pixel_delta_u);
  put(out->pixel_delta_v, pixel_delta_v);
  put(out->defocus_disk_u, defocus_radius * u);
  put(out->defocus_disk_v, defocus_radius * v);
}

void quantise(const double* rgb_sum, size_t n_pixels, uint64_t spp, uint8_t* out) {  // src/camera.rs:85-95
  for (size_t i = 0; i < 3 * n_pixels; ++i) {
    double g = std::sqrt(rgb_sum[i] / (double)spp);
    double v = 255.0 * std::fmin(std::fmax(g, 0.0), 1.0);  // Interval::clamp, src/ray.rs:39-41
    out[i] = !(v > 0.0) ? 0 : v >= 255.0 ? 255 : (uint8_t)v;  // `as u8`: truncate, saturate, NaN -> 0
  }
}

namespace {
bool write_all(int fd, const char* p, size_t n) {
  while (n) {
    ssize_t w = ::write(fd, p, n);
    if (w <= 0) return false;
    p += w;
    n -= (size_t)w;
  }
  return true;
}
// "0".."255" as text, 4 bytes each (digits then padding), and their lengths.
struct ByteText {
  char text[256][4];
  uint8_t len[256];
  ByteText() {
    for (int v = 0; v < 256; ++v) {
      int n = std::snprintf(text[v], 4, "%d", v);
      len[v] = (uint8_t)n;
    }
  }
};
// Formats pixels [begin, end) as "r g b\n" lines into out (at most 12 bytes per pixel); returns the length.
size_t format_pixels(const ByteText& bt, const uint8_t* rgb, size_t begin, size_t end, char* out) {
  char* o = out;
  for (size_t i = begin; i < end; ++i) {
    const uint8_t* p = rgb + 3 * i;
    for (int c = 0; c < 3; ++c) {
      const uint8_t v = p[c];
      std::memcpy(o, bt.text[v], 4);  // copies padding too; the next write overwrites it
      o += bt.len[v];
      *o++ = c == 2 ? '\n' : ' ';
    }
  }
  return (size_t)(o - out);
}
}  // namespace

// Header of src/camera.rs:139, then one line per pixel, rows top to bottom (src/camera.rs:157-164).
// The pixels are formatted by a few threads into per-thread buffers that are written in order.
bool write_ppm_bytes(int fd, const uint8_t* rgb, uint32_t width, uint32_t height) {
  static const ByteText bt;
  char head[64];
  int hn = std::snprintf(head, sizeof(head), "P3\n%u %u\n255\n", width, height);
  if (!write_all(fd, head, (size_t)hn)) return false;
  const size_t n = (size_t)width * height;
  const size_t batch = 1 << 18;  // pixels per buffer: 3 MB of text at most
  unsigned hw = std::thread::hardware_concurrency();
  const size_t n_threads = std::max<size_t>(1, std::min<size_t>({(size_t)(hw ? hw : 1), (size_t)8, (n + batch - 1) / batch}));
  std::vector<std::vector<char>> bufs(n_threads, std::vector<char>(batch * 12 + 4));
  std::vector<size_t> lens(n_threads);
  for (size_t base = 0; base < n; base += batch * n_threads) {
    std::vector<std::thread> workers;
    workers.reserve(n_threads);
    size_t used = 0;
    for (size_t t = 0; t < n_threads; ++t) {
      const size_t b = base + t * batch;
      if (b >= n) break;
      const size_t e = std::min(n, b + batch);
      ++used;
      bool threaded = false;
      if (n_threads > 1) {
        try {
          workers.emplace_back([&, t, b, e]() { lens[t] = format_pixels(bt, rgb, b, e, bufs[t].data()); });
          threaded = true;
        } catch (const std::system_error&) {  // no thread to be had: format this batch here
        }
      }
      if (!threaded) lens[t] = format_pixels(bt, rgb, b, e, bufs[t].data());
    }
    for (auto& w : workers) w.join();
    for (size_t t = 0; t < used; ++t)
      if (!write_all(fd, bufs[t].data(), lens[t])) return false;
  }
  return true;
}

bool write_ppm(int fd, const double* rgb_sum, uint32_t width, uint32_t height, uint64_t spp) {
  const size_t n = (size_t)width * height;
  std::vector<uint8_t> bytes(3 * n);
  quantise(rgb_sum, n, spp, bytes.data());
  return write_ppm_bytes(fd, bytes.data(), width, height);
}

namespace {
enum { DOM_SCENE = 3 };
inline rt_material make_material(uint32_t type, double r, double g, double b, double param) {
  rt_material m;
  std::memset(&m, 0, sizeof(m));
  m.type = type;
  m.albedo[0] = r;
  m.albedo[1] = g;
  m.albedo[2] = b;
  m.param = param;
  return m;
}
inline rt_sphere make_sphere(double x, double y, double z, double r, uint32_t mat) {
  rt_sphere s;
  std::memset(&s, 0, sizeof(s));
  s.center[0] = x;
  s.center[1] = y;
  s.center[2] = z;
  s.radius = r;
  s.material = mat;
  return s;
}
// The three random materials of src/main.rs:64-79 for object `id`.
// kind_u picks the kind; the other draws come from slots 2.. of the object's counter.
inline rt_material random_material(uint64_t seed, uint32_t id, double kind_u, double p_lambertian, double p_metal, double max_fuzz) {
  double a[2], b[2], c[2];
  if (kind_u < p_lambertian) {  // albedo = random::<DVec3>() * random::<DVec3>(), src/main.rs:67
    philox_pair(seed, id, 0, 2, DOM_SCENE, a);
    philox_pair(seed, id, 0, 3, DOM_SCENE, b);
    philox_pair(seed, id, 0, 4, DOM_SCENE, c);
    return make_material(RT_LAMBERTIAN, a[0] * b[1], a[1] * c[0], b[0] * c[1], 0.0);
  }
  if (kind_u < p_lambertian + p_metal) {  // albedo in [0.5,1]^3, fuzz in [0,max_fuzz), src/main.rs:72-73
    philox_pair(seed, id, 0, 2, DOM_SCENE, a);
    philox_pair(seed, id, 0, 3, DOM_SCENE, b);
    return make_material(RT_METAL, 0.5 + 0.5 * a[0], 0.5 + 0.5 * a[1], 0.5 + 0.5 * b[0], max_fuzz * b[1]);
  }
  return make_material(RT_DIELECTRIC, 0.0, 0.0, 0.0, 1.5);  // src/main.rs:77
}
}  // namespace

void scene_rtiow(uint64_t seed, rt_sphere* spheres, rt_material* materials) {  // src/main.rs:42-82
  size_t n = 0;
  materials[n] = make_material(RT_LAMBERTIAN, 0.5, 0.5, 0.5, 0.0);
  spheres[n] = make_sphere(0.0, -1000.0, 0.0, 1000.0, (uint32_t)n);
  ++n;
  materials[n] = make_material(RT_DIELECTRIC, 0.0, 0.0, 0.0, 1.5);
  spheres[n] = make_sphere(0.0, 1.0, 0.0, 1.0, (uint32_t)n);
  ++n;
  materials[n] = make_material(RT_LAMBERTIAN, 0.4, 0.2, 0.1, 0.0);
  spheres[n] = make_sphere(-4.0, 1.0, 0.0, 1.0, (uint32_t)n);
  ++n;
  materials[n] = make_material(RT_METAL, 0.7, 0.6, 0.5, 0.0);
  spheres[n] = make_sphere(4.0, 1.0, 0.0, 1.0, (uint32_t)n);
  ++n;
  for (int a = -11; a < 11; ++a)
    for (int b = -11; b < 11; ++b) {
      uint32_t id = (uint32_t)n;
      double pos[2], kind[2];
      philox_pair(seed, id, 0, 0, DOM_SCENE, pos);
      philox_pair(seed, id, 0, 1, DOM_SCENE, kind);
      materials[n] = random_material(seed, id, kind[0], 1.0 / 3.0, 1.0 / 3.0, 1.0);  // gen_range(0..3), src/main.rs:64
      spheres[n] = make_sphere((double)a + 0.9 * pos[0], 0.2, (double)b + 0.9 * pos[1], 0.2, id);
      ++n;
    }
}

void scene_field(uint64_t seed, size_t n_small, double p_lambertian, double p_metal, double max_fuzz, rt_sphere* spheres,
                 rt_material* materials) {
  const double R = 1000.0;
  materials[0] = make_material(RT_LAMBERTIAN, 0.5, 0.5, 0.5, 0.0);
  spheres[0] = make_sphere(0.0, -R, 0.0, R, 0);
  size_t side = (size_t)std::ceil(std::sqrt((double)n_small));
  double half = std::floor((double)side / 2.0);
  for (size_t k = 0; k < n_small; ++k) {
    uint32_t id = (uint32_t)(k + 1);
    double pos[2], kind[2];
    philox_pair(seed, id, 0, 0, DOM_SCENE, pos);
    philox_pair(seed, id, 0, 1, DOM_SCENE, kind);
    double x = (double)(k % side) - half + 0.9 * pos[0], z = (double)(k / side) - half + 0.9 * pos[1];
    // rest the sphere on the ground sphere
    double y = std::sqrt(std::fmax(R * R - x * x - z * z, 0.0)) - R + 0.2;
    materials[id] = random_material(seed, id, kind[0], p_lambertian, p_metal, max_fuzz);
    spheres[id] = make_sphere(x, y, z, 0.2, id);
  }
}

}  // namespace rta
