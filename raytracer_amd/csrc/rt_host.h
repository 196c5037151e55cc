// Host-only pieces of the path: camera set-up, output quantisation, the PPM
// text, the counter-based RNG used for scene generation.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "rt.h"

namespace rta {

// Philox4x32-10; counter = (a, b, slot, domain), key = seed.  Domain 3 is
// reserved for scene generation (the kernels use 0 and 1).
void philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]);
void philox_pair(uint64_t seed, uint32_t a, uint32_t b, uint32_t slot, uint32_t domain, double u[2]);

void config_default(rt_config* cfg);
// Returns "" or the reason the config is unusable.
std::string config_check(const rt_config& cfg);
void camera_derive(const rt_config& cfg, rt_camera* out);

void quantise(const double* rgb_sum, size_t n_pixels, uint64_t spp, uint8_t* out);
// Returns false on a write error.
bool write_ppm(int fd, const double* rgb_sum, uint32_t width, uint32_t height, uint64_t spp);
// The same text from bytes that are already quantised (3 per pixel).
bool write_ppm_bytes(int fd, const uint8_t* rgb, uint32_t width, uint32_t height);

void scene_rtiow(uint64_t seed, rt_sphere* spheres, rt_material* materials);
void scene_field(uint64_t seed, size_t n_small, double p_lambertian, double p_metal, double max_fuzz, rt_sphere* spheres,
                 rt_material* materials);

}  // namespace rta
