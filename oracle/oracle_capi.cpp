// C entry points over rt_oracle.hpp for ctypes.  TEST INFRASTRUCTURE ONLY:
// loaded by tests/, the smoke check and the CPU-baseline leg of bench.py.
#include "rt_oracle.hpp"

#include <cstdio>

using namespace orc;

static inline V3 ld(const double* p) { return v3(p[0], p[1], p[2]); }
static inline void st(double* p, V3 v) {
  p[0] = v.x;
  p[1] = v.y;
  p[2] = v.z;
}

extern "C" {

void orc_philox4x32_10(const uint32_t* key, const uint32_t* ctr, uint32_t* out) { philox4x32_10(key, ctr, out); }
void orc_philox4x32(const uint32_t* key, const uint32_t* ctr, uint32_t* out, int rounds) { philox4x32(key, ctr, out, rounds); }

void orc_philox_pair(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t slot, uint32_t domain, double* u) {
  philox_pair(seed, pixel, sample, slot, domain, u);
}

// ---- AABB algebra (src/aabb.rs) ----
void orc_aabb_new(double* mn, double* mx) {
  AABB b = aabb_new();
  st(mn, b.min);
  st(mx, b.max);
}
int orc_aabb_is_empty(const double* mn, const double* mx) { return aabb_is_empty(AABB{ld(mn), ld(mx)}); }
double orc_aabb_surface_area(const double* mn, const double* mx) { return aabb_surface_area(AABB{ld(mn), ld(mx)}); }
void orc_aabb_centroid(const double* mn, const double* mx, double* c) { st(c, aabb_centroid(AABB{ld(mn), ld(mx)})); }
void orc_aabb_bounding_box(const double* pts, size_t n, double* mn, double* mx) {
  std::vector<V3> p(n);
  for (size_t i = 0; i < n; ++i) p[i] = ld(pts + 3 * i);
  AABB b = aabb_bounding_box(p.data(), n);
  st(mn, b.min);
  st(mx, b.max);
}
// boxes: n x 6 doubles (min xyz, max xyz)
void orc_aabb_union(const double* boxes, size_t n, double* mn, double* mx) {
  std::vector<AABB> b(n);
  for (size_t i = 0; i < n; ++i) b[i] = AABB{ld(boxes + 6 * i), ld(boxes + 6 * i + 3)};
  AABB r = aabb_union(b.data(), n);
  st(mn, r.min);
  st(mx, r.max);
}
void orc_aabb_intersection(const double* boxes, size_t n, double* mn, double* mx) {
  std::vector<AABB> b(n);
  for (size_t i = 0; i < n; ++i) b[i] = AABB{ld(boxes + 6 * i), ld(boxes + 6 * i + 3)};
  AABB r = aabb_intersection(b.data(), n);
  st(mn, r.min);
  st(mx, r.max);
}
// Batched slab test: boxes n x 6, origins/directions n x 3 (direction is
// normalised as Ray::new does), intervals n x 2.
void orc_slab_batch(const double* boxes, const double* org, const double* dir, const double* interval, size_t n, uint8_t* hit) {
  for (size_t i = 0; i < n; ++i) {
    AABB b{ld(boxes + 6 * i), ld(boxes + 6 * i + 3)};
    Ray r = ray_new(ld(org + 3 * i), ld(dir + 3 * i));
    hit[i] = aabb_ray_intersection(b, r, Interval{interval[2 * i], interval[2 * i + 1]}) ? 1 : 0;
  }
}

// ---- Sphere (src/sphere.rs) ----
void orc_sphere_aabb(const double* c, double r, double* mn, double* mx) {
  AABB b = sphere_aabb(Sphere{ld(c), r, 0});
  st(mn, b.min);
  st(mx, b.max);
}
void orc_sphere_clipped_aabb(const double* c, double r, int axis, double t1, double t2, double* mn, double* mx) {
  AABB b = sphere_clipped_aabb(Sphere{ld(c), r, 0}, axis, t1, t2);
  st(mn, b.min);
  st(mx, b.max);
}
// One sphere against n rays.
void orc_sphere_hit_batch(const double* c, double radius, const double* org, const double* dir, size_t n, double tmin, double tmax,
                          uint8_t* hit, double* t, double* p, double* normal, uint8_t* front) {
  Sphere s{ld(c), radius, 0};
  for (size_t i = 0; i < n; ++i) {
    Ray r = ray_new(ld(org + 3 * i), ld(dir + 3 * i));
    HitRecord hr{};
    bool h = sphere_hit(s, 0, r, Interval{tmin, tmax}, hr);
    hit[i] = h;
    if (h) {
      t[i] = hr.t;
      st(p + 3 * i, hr.p);
      st(normal + 3 * i, hr.normal);
      front[i] = hr.front_face;
    }
  }
}

// ---- Scene / BVH ----
void* orc_scene_create(const SphereIn* spheres, size_t n, const Material* mats, size_t nm) {
  Scene* sc = new Scene();
  std::vector<Sphere> prims(n);
  for (size_t i = 0; i < n; ++i) prims[i] = Sphere{ld(spheres[i].c), spheres[i].r, spheres[i].mat};
  sc->bvh = bvh_new(prims);
  sc->materials.assign(mats, mats + nm);
  return sc;
}
void orc_scene_destroy(void* h) { delete (Scene*)h; }
size_t orc_bvh_node_count(void* h) { return ((Scene*)h)->bvh.nodes.size(); }
size_t orc_bvh_leaf_index_count(void* h) {
  size_t n = 0;
  for (auto& nd : ((Scene*)h)->bvh.nodes) n += nd.indices.size();
  return n;
}
// Dump in pre-order: boxes (n x 6), kind (0..2 split axis, 3 leaf), skip
// (split) or index count (leaf), and the concatenated leaf indices.
void orc_bvh_dump(void* h, double* boxes, int32_t* kind, uint64_t* skip_or_count, uint64_t* leaf_indices) {
  const BVH& b = ((Scene*)h)->bvh;
  size_t li = 0;
  for (size_t i = 0; i < b.nodes.size(); ++i) {
    const Node& nd = b.nodes[i];
    st(boxes + 6 * i, nd.aabb.min);
    st(boxes + 6 * i + 3, nd.aabb.max);
    kind[i] = nd.leaf ? 3 : nd.axis;
    skip_or_count[i] = nd.leaf ? nd.indices.size() : nd.skip;
    for (size_t pi : nd.indices) leaf_indices[li++] = pi;
  }
}

// BVH::hit over a batch (src/bvh.rs:50-84).  stats: rays, node visits, sphere tests.
void orc_trace_batch(void* h, const double* org, const double* dir, size_t n, double tmin, double tmax, int32_t* prim, double* t,
                     double* p, double* normal, uint8_t* front, uint64_t* stats) {
  const Scene& sc = *(Scene*)h;
  TraceStats ts;
  for (size_t i = 0; i < n; ++i) {
    Ray r = ray_new(ld(org + 3 * i), ld(dir + 3 * i));
    HitRecord hr{};
    if (bvh_hit(sc.bvh, r, Interval{tmin, tmax}, hr, &ts)) {
      prim[i] = hr.prim;
      t[i] = hr.t;
      if (p) st(p + 3 * i, hr.p);
      st(normal + 3 * i, hr.normal);
      front[i] = hr.front_face;
    } else {
      prim[i] = -1;
      t[i] = INF;
      if (p) st(p + 3 * i, v3(0, 0, 0));
      st(normal + 3 * i, v3(0, 0, 0));
      front[i] = 0;
    }
  }
  if (stats) {
    stats[0] = ts.rays;
    stats[1] = ts.node_visits;
    stats[2] = ts.sphere_tests;
  }
}

// Material::scatter with explicit uniforms (direct maps): u is n x 2.
// in_dir is normalised as Ray::new does.  out_dir is the scattered ray's
// (normalised) direction; its origin is p.
void orc_scatter_batch(const Material* mats, const uint32_t* mat_index, const double* in_dir, const double* p, const double* normal,
                       const uint8_t* front, const double* u, size_t n, uint8_t* scattered, double* attenuation, double* out_dir) {
  for (size_t i = 0; i < n; ++i) {
    Ray r = ray_new(v3(0, 0, 0), ld(in_dir + 3 * i));
    HitRecord hr{front[i] != 0, mat_index[i], 0, ld(normal + 3 * i), ld(p + 3 * i), 0.0};
    ScatterDraw d{unit_from_uniforms(u[2 * i], u[2 * i + 1]), u[2 * i]};
    V3 att;
    Ray out;
    bool ok = scatter(mats[mat_index[i]], r, hr, d, att, out);
    scattered[i] = ok;
    if (ok) {
      st(attenuation + 3 * i, att);
      st(out_dir + 3 * i, out.direction);
    } else {
      st(attenuation + 3 * i, v3(0, 0, 0));
      st(out_dir + 3 * i, v3(0, 0, 0));
    }
  }
}

void orc_unit_from_uniforms(const double* u, size_t n, double* out) {
  for (size_t i = 0; i < n; ++i) st(out + 3 * i, unit_from_uniforms(u[2 * i], u[2 * i + 1]));
}
void orc_disk_from_uniforms(const double* u, size_t n, double* out) {
  for (size_t i = 0; i < n; ++i) st(out + 3 * i, disk_from_uniforms(u[2 * i], u[2 * i + 1]));
}
// The reference's rejection samplers driven by a sequential stream; one
// vector per (pixel=i, sample=0).
void orc_unit_rejection(uint64_t seed, size_t n, double* out) {
  for (size_t i = 0; i < n; ++i) {
    PathRng g{seed, (uint32_t)i, 0, SAMPLER_REJECTION};
    st(out + 3 * i, random_unit_rejection(g));
  }
}
void orc_disk_rejection(uint64_t seed, size_t n, double* out) {
  for (size_t i = 0; i < n; ++i) {
    PathRng g{seed, (uint32_t)i, 0, SAMPLER_REJECTION};
    st(out + 3 * i, random_disk_rejection(g));
  }
}
void orc_reflect(const double* v, const double* n, double* out) { st(out, reflect(ld(v), ld(n))); }
void orc_refract(const double* v, const double* n, double ratio, double* out) { st(out, refract(ld(v), ld(n), ratio)); }
int orc_near_zero(const double* v) { return near_zero(ld(v)); }

// ---- Camera ----
// out: image_height, then center, pixel00_loc, pixel_delta_u, pixel_delta_v,
// defocus_disk_u, defocus_disk_v (18 doubles).
void orc_camera_derive(const Config* cfg, uint32_t* image_height, double* out18) {
  Camera cam = config_camera(*cfg);
  *image_height = cam.image_height;
  st(out18 + 0, cam.center);
  st(out18 + 3, cam.pixel00_loc);
  st(out18 + 6, cam.pixel_delta_u);
  st(out18 + 9, cam.pixel_delta_v);
  st(out18 + 12, cam.defocus_disk_u);
  st(out18 + 15, cam.defocus_disk_v);
}
// Primary rays (direct sampler) for n (i, j, sample) triples.
void orc_primary_rays(const Config* cfg, uint64_t seed, const uint32_t* ijs, size_t n, double* org, double* dir) {
  Camera cam = config_camera(*cfg);
  for (size_t k = 0; k < n; ++k) {
    uint32_t i = ijs[3 * k], j = ijs[3 * k + 1], s = ijs[3 * k + 2];
    PathRng g{seed, j * cfg->image_width + i, s, SAMPLER_DIRECT};
    Ray r = get_ray(cam, i, j, g);
    st(org + 3 * k, r.origin);
    st(dir + 3 * k, r.direction);
  }
}

// Camera::render without the text output.  stats: rays, node visits, sphere tests.
void orc_render(void* h, const Config* cfg, uint64_t seed, uint32_t s0, uint32_t s1, const uint32_t* tile4, int mode, int nthreads,
                double* rgb_sum, uint64_t* stats) {
  Rect tile{tile4[0], tile4[1], tile4[2], tile4[3]};
  TraceStats ts = render(*(Scene*)h, *cfg, seed, s0, s1, tile, mode, nthreads, rgb_sum);
  if (stats) {
    stats[0] = ts.rays;
    stats[1] = ts.node_visits;
    stats[2] = ts.sphere_tests;
  }
}

void orc_quantise(const double* rgb_sum, size_t npix, uint64_t spp, uint8_t* out) {
  for (size_t i = 0; i < npix; ++i) quantise(rgb_sum + 3 * i, spp, out + 3 * i);
}

// src/camera.rs:139,157-164: "P3\n{w} {h}\n255\n" then one "r g b\n" per pixel, row major.
int orc_write_ppm(const char* path, const double* rgb_sum, uint32_t w, uint32_t h, uint64_t spp) {
  FILE* f = std::fopen(path, "w");
  if (!f) return 1;
  std::fprintf(f, "P3\n%u %u\n255\n", w, h);
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    uint8_t q[3];
    quantise(rgb_sum + 3 * i, spp, q);
    std::fprintf(f, "%u %u %u\n", q[0], q[1], q[2]);
  }
  std::fclose(f);
  return 0;
}

}  // extern "C"
