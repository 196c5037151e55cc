/* rt.h -- C ABI of the MI355X render hot path (libraytracer_amd.so).
 *
 * The reference (rfsmi/raytracer) is a single Rust binary with no FFI seam;
 * this header is the seam a maintainer would bind.  Each entry point names
 * the reference item it stands in for (paths relative to the reference
 * checkout).  Plain pointers and sizes only; all host buffers are owned by
 * the caller; the library copies what it keeps.
 *
 * Conventions
 *  - Every function returns 0 on success or an rt_status code; the message of
 *    the last failure on the calling thread is rt_last_error().  Nothing
 *    throws or aborts across this boundary (the reference panics instead:
 *    src/camera.rs:160, src/bvh.rs:191).
 *  - A scene lives on the HIP device that was current when it was created.
 *    It is immutable afterwards, so calls on one scene may come from several
 *    host threads.
 *  - Results depend only on (scene, config, seed, sample range, tile), never
 *    on scheduling.
 *  - There is no CPU fallback: without a usable HIP device every compute
 *    entry point fails with RT_ERR_HIP.
 */
#ifndef RAYTRACER_AMD_RT_H
#define RAYTRACER_AMD_RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID = 1, /* bad argument (null pointer, non-finite input, empty range ...) */
  RT_ERR_HIP = 2,     /* a HIP runtime call failed; rt_last_error() has the call and HIP's text */
  RT_ERR_IO = 3,      /* file output failed */
  RT_ERR_LIMIT = 4    /* input exceeds a documented limit */
} rt_status;

/* Material kinds: src/material.rs:10-19 (Lambertian, Metal, Dielectric). */
enum { RT_LAMBERTIAN = 0, RT_METAL = 1, RT_DIELECTRIC = 2 };

/* One `Arc<dyn Material>` flattened.  param is Metal::fuzz or Dielectric::ir. */
typedef struct rt_material {
  uint32_t type;
  uint32_t reserved;
  double albedo[3];
  double param;
} rt_material;

/* One `Sphere` (src/sphere.rs:12-16); material indexes the rt_material array. */
typedef struct rt_sphere {
  double center[3];
  double radius;
  uint32_t material;
  uint32_t reserved;
} rt_sphere;

/* `Config` (src/camera.rs:15-26): same ten fields, same meaning. */
typedef struct rt_config {
  double aspect_ratio;
  uint32_t image_width;
  uint32_t reserved;
  uint64_t samples_per_pixel;
  uint64_t max_depth;
  double vfov;
  double lookfrom[3];
  double lookat[3];
  double vup[3];
  double defocus_angle;
  double focus_dist;
} rt_config;

/* Derived `Camera` fields (src/camera.rs:28-37). */
typedef struct rt_camera {
  uint32_t image_height;
  uint32_t reserved;
  double center[3];
  double pixel00_loc[3];
  double pixel_delta_u[3];
  double pixel_delta_v[3];
  double defocus_disk_u[3];
  double defocus_disk_v[3];
} rt_camera;

/* Half-open pixel rectangle [x0,x1) x [y0,y1). */
typedef struct rt_rect {
  uint32_t x0, y0, x1, y1;
} rt_rect;

/* A share of a render request: the tile rows (8 pixel rows each, counted from rect.y0) of `rect` whose index is
 * row_first modulo row_stride, for samples [spp_begin, spp_end).  row_first 0 and row_stride 1 name the whole
 * rectangle.  Every share of a rectangle is summed exactly as the whole would be, so frames put together from
 * tile-row shares are bit for bit the unshared frame. */
typedef struct rt_shard {
  rt_rect rect;
  uint32_t row_first, row_stride;
  uint32_t spp_begin, spp_end;
} rt_shard;

/* What one render call did.  kernel_ms is measured with HIP events on the
 * stream the kernels ran on.  node_steps and sphere_tests are only counted by the instrumented
 * kernel (RT_PROFILE_BLOCKS set in the environment) and are 0 otherwise. */
typedef struct rt_render_stats {
  uint64_t rays;        /* closest-hit queries (primary + secondary), one per BVH::hit call of the reference */
  uint64_t samples;     /* camera paths started */
  float kernel_ms;      /* device time of the launch(es) of this call */
  uint32_t launches;    /* kernels launched by this call */
  uint64_t node_steps;  /* inner nodes visited (two child boxes tested at each) */
  uint64_t sphere_tests; /* Sphere::hit evaluations */
} rt_render_stats;

typedef struct rt_scene_info {
  uint64_t n_spheres;
  uint64_t n_nodes;         /* nodes of the host tree (splits + leaves) */
  uint64_t n_device_nodes;  /* device nodes: 64-byte two-wide nodes (one per split) for a scene that fits in LDS, else 128-byte four-wide nodes */
  uint32_t depth;           /* levels in the BVH (a single leaf has depth 1) */
  uint32_t n_nodes_in_lds;  /* device nodes of the top of the tree staged in LDS by the kernels */
  uint32_t spheres_in_lds;  /* bit 0: sphere records staged in LDS too; bit 1: their shading data; bit 2: the materials */
  uint32_t device;          /* HIP device ordinal the scene lives on */
  uint64_t device_bytes;    /* HBM held by the scene */
  double build_ms;          /* BVH build (+ flatten on the host path) */
  uint64_t lds_bytes;       /* LDS each workgroup of the tree-walking kernels uses for this scene */
  uint32_t builder;         /* RT_BUILD_* that made the tree (the device builder falls back to the host for very deep trees) */
  uint32_t n_devices;       /* copies of the scene (1 unless made by rt_scene_create_multi); `device` is the first one's */
  uint32_t peer_pairs;      /* ordered pairs (a, b) of copies on different GPUs ... */
  uint32_t peer_pairs_enabled; /* ... and those of them where a's GPU can read b's memory (peer access over xGMI or PCIe, enabled at
                               scene creation): equal to peer_pairs when a sample split is summed over the peer links, else through copies */
  uint64_t peer_matrix;     /* bit a * 8 + b set: the GPU of copy a can read the memory of copy b's (copies 0 .. 7; the diagonal is set) */
  uint32_t cut_entries;     /* longest entry list camera rays start from in this scene (rt_debug_set "tile_cuts") */
  uint32_t reserved;
} rt_scene_info;

typedef struct rt_scene rt_scene;

/* Message of the last failure on this thread ("" if none). */
const char* rt_last_error(void);

/* The library's current HIP device (scenes are created on it), for callers that do not share
 * the library's HIP runtime: a process whose other GPU code brings its own copy of the runtime
 * (a PyTorch wheel does) selects the device here as well as there.  Callers linked against the
 * same libamdhip64 can simply use hipSetDevice. */
int rt_device_count(int* count);
int rt_device_set(int ordinal);
int rt_device_get(int* ordinal);

/* `Config::new()` defaults (src/camera.rs:15-26). */
void rt_config_default(rt_config* cfg);

/* `Config::camera()` (src/camera.rs:40-82). Host only. */
int rt_camera_derive(const rt_config* cfg, rt_camera* out);

/* `BVH::new(objects)` (src/bvh.rs:44-48, builder :92-266 without the unfinished
 * spatial splits): copies the inputs, builds the reference's binned-SAH tree (RT_BUILD_AUTO below: on the
 * host for scenes below 1 024 spheres, on the device above), flattens it and uploads it to the current
 * HIP device. */
int rt_scene_create(const rt_sphere* spheres, size_t n_spheres, const rt_material* materials, size_t n_materials, rt_scene** out);

/* rt_scene_create with a choice of tree builder.  RT_BUILD_HOST_SAH is the reference's
 * binned-SAH tree, built by host threads (0.1 s for 10^5 spheres, 0.65 s for 10^6 on 16 cores).
 * RT_BUILD_DEVICE_SAH is the same algorithm on the GPU (4 ms and 11 ms): the same boxes, buckets,
 * costs and partitions, so the same tree node for node and the same render; where the reference
 * would keep several spheres in one leaf (spheres on top of each other: no split cheaper than
 * the leaf) it cuts the leaf in the middle until every leaf holds one sphere.
 * RT_BUILD_DEVICE_LBVH clusters the spheres bottom-up along their Morton order (13 ms for 10^6):
 * a different tree over the same spheres, so closest hits are the same (exact ties between
 * overlapping spheres aside) while a render takes 4 - 9 % longer.  RT_BUILD_AUTO takes the host
 * builder below 1 024 spheres and RT_BUILD_DEVICE_SAH from there on.  A scene built on the
 * device keeps no host tree: rt_scene_bvh_dump refuses it. */
enum { RT_BUILD_HOST_SAH = 0, RT_BUILD_DEVICE_LBVH = 1, RT_BUILD_AUTO = 2, RT_BUILD_DEVICE_SAH = 3 };
int rt_scene_create_ex(const rt_sphere* spheres, size_t n_spheres, const rt_material* materials, size_t n_materials, uint32_t builder,
                       rt_scene** out);
/* `BVH::new` for a host that owns several GPUs (src/main.rs:84 builds one tree; here every device gets a copy
 * of it, the host tree being built once).  devices: n_devices HIP ordinals, or NULL for every visible device
 * (at most 16; an ordinal may repeat).  The single-device entry points use the first copy; rt_render_multi and
 * rt_render_ppm use all of them. */
int rt_scene_create_multi(const rt_sphere* spheres, size_t n_spheres, const rt_material* materials, size_t n_materials, uint32_t builder,
                          const int32_t* devices, size_t n_devices, rt_scene** out);
int rt_scene_destroy(rt_scene* scene);
int rt_scene_get_info(const rt_scene* scene, rt_scene_info* out);

/* Host-only: the tree `BVH::new` builds, without a device (src/bvh.rs:44-48).
 * Used to compare the builder with the reference's algorithm node by node. */
typedef struct rt_host_bvh rt_host_bvh;
int rt_bvh_build(const rt_sphere* spheres, size_t n_spheres, rt_host_bvh** out);
/* The same tree from the device's builder (RT_BUILD_DEVICE_SAH, on the current HIP device), brought back in the same form:
 * `BVH::new` (src/bvh.rs:92-266) computed on the GPU, for comparing it with the reference's algorithm node by node. */
int rt_bvh_build_device(const rt_sphere* spheres, size_t n_spheres, rt_host_bvh** out);
int rt_bvh_destroy(rt_host_bvh* bvh);
int rt_bvh_counts(const rt_host_bvh* bvh, uint64_t* n_nodes, uint64_t* n_leaf_indices, uint32_t* depth);
/* Same layout as rt_scene_bvh_dump. */
int rt_bvh_dump(const rt_host_bvh* bvh, double* boxes, int32_t* kind, uint64_t* skip_or_count, uint64_t* leaf_indices);

/* The host tree in the reference's own order (pre-order; a split's left child
 * follows it and its right child is `skip` entries later: src/bvh.rs:125-132).
 * boxes: n_nodes x 6 doubles (min, max); kind: 0..2 split axis, 3 leaf;
 * skip_or_count: skip (split) or primitive count (leaf); leaf_indices: the
 * leaves' sphere indices concatenated in node order (n_spheres entries).
 * Any output pointer may be NULL. */
int rt_scene_bvh_dump(const rt_scene* scene, double* boxes, int32_t* kind, uint64_t* skip_or_count, uint64_t* leaf_indices);

/* `BVH::hit(&Ray, Interval)` over a batch (src/bvh.rs:50-84 with
 * src/aabb.rs:69-82, src/sphere.rs:29-48, src/hit.rs:18-27).  Rays are built
 * as `Ray::new` does (src/ray.rs:16-23): dir need not be normalised.  Host
 * buffers: org, dir, normal are n x 3 doubles.  prim is the index into the
 * array given to rt_scene_create, -1 where the reference returns None (then
 * t is +inf and normal 0). */
int rt_trace_batch(const rt_scene* scene, const double* org, const double* dir, size_t n, double t_min, double t_max, int32_t* prim,
                   double* t, double* normal, uint8_t* front_face);

/* `Material::scatter(&Ray, &HitRecord)` over a batch (src/material.rs:21-65)
 * with the randomness passed in: u is n x 2 uniforms in [0,1).  The unit
 * vector of `vector::random_unit()` is the direct map z = 1-2u0, phi = 2 pi u1;
 * the Dielectric's `random::<f64>()` is u0.  in_dir is normalised first, as the
 * incoming Ray would be.  scattered[i] = 0 where the reference returns None.
 * out_dir is the direction of the scattered Ray (unit length); its origin is p. */
int rt_scatter_batch(const rt_scene* scene, const uint32_t* material, const double* in_dir, const double* p, const double* normal,
                     const uint8_t* front_face, const double* u, size_t n, uint8_t* scattered, double* attenuation, double* out_dir);

/* The device's node test, `AABB::ray_intersection(..).is_some()`
 * (src/aabb.rs:69-82), over a batch.  The kernels test nodes in f32 against
 * boxes rounded outwards and padded so that the answer is never "miss" where
 * the reference's f64 test says "hit"; it may say "hit" for rays that pass
 * within rounding distance of a box.  boxes: n x 6 doubles; interval: n x 2. */
int rt_slab_batch(const double* boxes, const double* org, const double* dir, const double* interval, size_t n, uint8_t* hit);

/* Primary rays, `Camera::get_ray(i, j)` (src/camera.rs:114-136), for n
 * (i, j, sample) triples under the library's counter-based RNG. */
int rt_primary_rays(const rt_config* cfg, uint64_t seed, const uint32_t* ijs, size_t n, double* org, double* dir);

/* The parallel part of `Camera::render` (src/camera.rs:140-155): for every
 * pixel of `tile` (NULL = whole image) adds the radiance of samples
 * [spp_begin, spp_end) into rgb_sum, a row-major image_width x image_height x 3
 * array of doubles.  cfg->samples_per_pixel is not read here (the sample range
 * is explicit); cfg->max_depth is.  Sample s of pixel (i,j) is the same path
 * whatever the tile or range, so tiles and ranges can be split over GPUs.
 *
 * rt_render: rgb_sum is host memory; returns when it has been updated.
 * rt_render_device: rgb_sum is device memory on the scene's device; the
 * kernels run on `stream` (a hipStream_t, NULL = default stream) and the call
 * returns when they have finished (it waits for the stream).  stats may be
 * NULL. */
int rt_render(const rt_scene* scene, const rt_config* cfg, uint64_t seed, uint32_t spp_begin, uint32_t spp_end, const rt_rect* tile,
              double* rgb_sum, rt_render_stats* stats);
int rt_render_device(const rt_scene* scene, const rt_config* cfg, uint64_t seed, uint32_t spp_begin, uint32_t spp_end,
                     const rt_rect* tile, double* rgb_sum_dev, void* stream, rt_render_stats* stats);

/* rt_render_device for one share of a request (see rt_shard): what each GPU of a node, or each process of a
 * multi-process job, calls with its own share.  shard->rect must lie inside the image. */
int rt_render_shard_device(const rt_scene* scene, const rt_config* cfg, uint64_t seed, const rt_shard* shard, double* rgb_sum_dev,
                           void* stream, rt_render_stats* stats);

/* How a frame is shared out over the GPUs of a node (every (pixel, sample) of src/camera.rs:144-151 is independent):
 * RT_PARTITION_TILES   tile rows dealt round-robin, all samples; nothing is exchanged, the result is bit for bit
 *                      the single-GPU frame;
 * RT_PARTITION_SAMPLES contiguous sample ranges of the whole rectangle; the per-device frames are added in device
 *                      order (equal to the single-GPU frame up to the rounding of those additions). */
enum { RT_PARTITION_TILES = 0, RT_PARTITION_SAMPLES = 1 };

/* The shares rt_render_multi gives to n_devices devices (host only; shards must hold n_devices entries). */
int rt_partition(const rt_config* cfg, uint32_t spp_begin, uint32_t spp_end, const rt_rect* tile, uint32_t partition, uint32_t n_devices,
                 rt_shard* shards);

/* rt_render on every device of a scene made by rt_scene_create_multi (one host thread per device): the
 * parallel part of `Camera::render` (src/camera.rs:140-155) spread over the node.  Adds into rgb_sum (host memory)
 * like rt_render.  stats: rays and samples summed over the devices, kernel_ms the slowest device's.  With the
 * sample partition the frames are brought together over the devices' peer links (each device sums one slice
 * of all frames), or through copies to the first device where peers cannot read each other.  One such call per
 * scene runs at a time. */
int rt_render_multi(const rt_scene* scene, const rt_config* cfg, uint64_t seed, uint32_t spp_begin, uint32_t spp_end, const rt_rect* tile,
                    uint32_t partition, double* rgb_sum, rt_render_stats* stats);

/* rt_render_device for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference.  The launch's counters, per-chunk
 * sums, the traversal stacks' overflow columns (scenes walked through the four-wide tree: CUs x 4 KiB per stack
 * entry beyond the eight kept in LDS) and the tiles' entry lists live in a 32 MiB static device array that every
 * GPU the library runs on holds until the process ends; the call makes no hipMalloc, hipFree, event or other
 * allocating call.  The request is cut into chunks exactly as rt_render_device cuts it, so the frame is bit for
 * bit the same; a request whose per-chunk sums (n_chunks x pixels x 24 bytes; frames too large for that still
 * fit when their sample range is short enough to be one chunk) and stack columns together exceed 32 MiB - 256 is
 * refused with RT_ERR_LIMIT; entry lists that no longer fit are left out (same frame, rendered more slowly).  The
 * kernels are enqueued on `stream`; with stats == NULL the call returns without waiting for them, otherwise it
 * waits and fills stats (kernel_ms stays 0: no events).  All calls on one GPU share that GPU's array: calls that
 * overlap in time on one GPU race (two calls without stats on different streams already do).  Two host threads
 * inside the call at once on one GPU are refused (RT_ERR_INVALID); kernels still running after a call returned are
 * not detected -- serialise such calls on one stream or wait in between. */
int rt_render_device_noalloc(const rt_scene* scene, const rt_config* cfg, uint64_t seed, uint32_t spp_begin, uint32_t spp_end,
                             const rt_rect* tile, double* rgb_sum_dev, void* stream, rt_render_stats* stats);

/* Tuning knobs and test hooks; not part of the stable interface.  Known names:
 * "inner_min" (lanes of 64 that must want the inner-node block for a wave to run
 * it without counting the other blocks; until it is set: 10 for scenes walked in LDS, 8 for larger ones), "leaf_min" (lanes waiting at leaves that make the leaf block run between
 * inner-node rounds; 8), "both_min" (when the inner-node or the leaf
 * block is due, the other of the two runs as well if this many lanes want it), "chunk_spp", "target_units" and "batches_per_wave" (how a
 * call's sample range is cut into units of work), "quad_tiles" (1, the default: the 64 lanes of a wave start on a 4x4 block of
 * pixels in four chunks of the sample range instead of 8x8 pixels in one, when the number of chunks allows) -- none of these changes which
 * samples are rendered; "tile_cuts" (longest entry list the camera rays of a tile start from, 0 .. 8; 0: every ray
 * starts at the root; frames are the same for every value); "force_global_stack" (scenes created while it is 1 keep
 * their traversal stacks in global memory, the path that trees too deep for per-lane stacks in LDS -- more than about
 * 30 levels, see rt_scene_info.lds_bytes -- take); "first_leaf" (1, the default: in scenes created while it is 1, a
 * one-sphere leaf in the top four levels of the tree, from either builder, whose box has at least half the tree's
 * surface area is tested when a query starts and cut out of the device's copy of the tree; closest hits are the same); "force_staged_sum" (rt_render_multi adds sample-split frames through copies to the first
 * device even where peers can read each other).  inner_min and leaf_min must be whole numbers in 1 .. 64,
 * both_min in 0 .. 64; other values are refused.  The knobs are process-wide atomics: a render call reads them
 * once when it starts.  With RT_PROFILE_BLOCKS set in the environment, rt_render* runs the
 * instrumented variant of the render kernel and prints, per code block, runs,
 * mean active lanes and cycles to stderr (slower; results unchanged). */
int rt_debug_set(const char* name, double value);

/* Test hook, host only (no GPU call): the rewrite that scene creation applies to the device's copy of a two-wide tree.
 * `nodes`: n_nodes records of 64 bytes in breadth-first order -- f32 x[4], y[4], z[4] = (lo, hi) of child 0 then of
 * child 1 per axis, u32 ref0, ref1, split axis, 0; a ref is a node index, or a leaf: bit 31 | sphere count << 24 |
 * first slot.  Looks in the top four levels below `root` for a one-sphere leaf whose box has at least half the surface
 * area of the tree's and takes it out (see "first_leaf" above): *first_leaf = its slot (0xffffffff: none found),
 * *walk_root = where the walk starts, nodes_out = the nodes with the leaf's sibling in its parent's place and the
 * ancestors' boxes shrunk (a copy of `nodes` when nothing changes).  No counterpart in the reference. */
int rt_debug_take_out_first_leaf(const void* nodes, size_t n_nodes, uint32_t root, void* nodes_out, uint32_t* first_leaf, uint32_t* walk_root);


/* `write_colour` (src/camera.rs:85-95) for n pixels: out is n x 3 bytes. Host only. */
int rt_quantise(const double* rgb_sum, size_t n_pixels, uint64_t samples_per_pixel, uint8_t* out);

/* The text the reference prints (src/camera.rs:139,157-164): "P3\n{w} {h}\n255\n"
 * then one "r g b\n" line per pixel.  fd is an open file descriptor (1 = stdout). */
int rt_write_ppm(int fd, const double* rgb_sum, uint32_t width, uint32_t height, uint64_t samples_per_pixel);

/* rt_quantise on the device: rgb_sum_dev (16-byte aligned) and out_dev (2-byte
 * aligned, 3 bytes per pixel) are device memory; runs on `stream` and waits for
 * it.  The bytes equal rt_quantise's (division and square root are correctly
 * rounded in both). */
int rt_quantise_device(const double* rgb_sum_dev, size_t n_pixels, uint64_t samples_per_pixel, uint8_t* out_dev, void* stream);

/* The text of rt_write_ppm from bytes that are already quantised. */
int rt_write_ppm_bytes(int fd, const uint8_t* rgb, uint32_t width, uint32_t height);

/* All of `Camera::render` (src/camera.rs:138-165) in one call: renders
 * cfg->samples_per_pixel samples of every pixel into a frame kept on the device,
 * quantises it there, copies 3 bytes per pixel back and writes the PPM text to
 * fd.  The text is what rt_render + rt_write_ppm produce.  A scene made by
 * rt_scene_create_multi is rendered by all its devices (tile rows dealt round-robin);
 * the text is the same for any number of devices. */
int rt_render_ppm(const rt_scene* scene, const rt_config* cfg, uint64_t seed, int fd, rt_render_stats* stats);

/* Seeded restatement of the scene of src/main.rs:42-82 (4 fixed spheres and a
 * 22 x 22 grid; 488 spheres, 488 materials).  Both arrays must hold 488 entries. */
int rt_scene_rtiow(uint64_t seed, rt_sphere* spheres, rt_material* materials);

/* Synthetic fields for stress configurations: ground sphere plus n_small
 * spheres of radius 0.2 on a jittered square grid, materials drawn with the
 * given probabilities (lambertian, metal; the rest glass), metal fuzz uniform
 * in [0, max_fuzz).  Arrays must hold n_small + 1 entries. */
int rt_scene_field(uint64_t seed, size_t n_small, double p_lambertian, double p_metal, double max_fuzz, rt_sphere* spheres,
                   rt_material* materials);

#ifdef __cplusplus
}
#endif
#endif
