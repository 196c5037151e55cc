// Kernel parameter blocks and launch functions shared by the host glue
// (rt_api.cpp) and the device code (rt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace rta {

// Threads per workgroup of the tree-walking kernels: one workgroup of 16 waves
// per CU, so that a CU holds a single copy of the scene in its 160 KB of LDS.
constexpr uint32_t kWorkgroup = 1024;
constexpr size_t kLdsPerCu = 160 * 1024;

// ------------------------------------------------------------ parameters ----
// Device arrays of a scene.  Each array may be staged into LDS from its start:
// entries below the matching n_*_lds count are read from LDS, the rest from
// global memory.
struct SceneDev {
  const float4* nodes;      // 4 x 16 B per inner node (DevNode)
  const double2* spheres;   // 2 x 16 B per slot (DevSphere: cx,cy | cz,r)
  const double2* aux;       // 1 x 16 B per slot: 1/r | bits: material index (30) + material type (2), sphere index (32)
  const double2* materials; // 2 x 16 B per material: albedo xyz, fuzz | ir, 1/ir, r0
  uint32_t n_nodes;         // inner nodes
  uint32_t n_spheres;
  uint32_t n_materials;
  uint32_t n_nodes_lds, n_spheres_lds, n_aux_lds, n_materials_lds;
  uint32_t stack_cap;       // traversal stack entries per lane
  uint32_t stack_in_lds;    // 1: the stacks live in LDS; 0: the tree is too deep, they live in gstack
  uint32_t* gstack;         // global traversal stacks (set per launch): entry e of thread t at gstack[e * gstack_stride + t]
  uint32_t gstack_stride;   // threads of the launch
  uint32_t root;            // ref of the root: inner node 0, a leaf ref, or 0xffffffff for an empty scene
  uint32_t single_leaves;   // 1: every leaf holds exactly one sphere (the leaf test then needs no loop)
  uint32_t first_leaf;      // slot of a sphere that every query tests before it walks the tree from `root`, or 0xffffffff: a one-sphere leaf
                            // directly under the tree's root (a ground sphere as large as the scene) is taken out of the walk this way
  float bound;              // largest |coordinate| of any box plane in the tree
  // Four-wide tree (scenes that do not fit in LDS): nodes are 64-byte WideNode records, n_nodes_lds of them staged
  // in LDS; the first lds_stack entries of a lane's stack live in LDS, the other stack_cap - lds_stack in gstack.
  uint32_t wide;
  uint32_t lds_stack;
  uint32_t cut_k;           // entries of an entry list that stack_cap has room for (see launch_tile_cuts)
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
  // The call renders the tile rows (8 pixel rows each) of the rectangle whose index is row_first modulo row_stride
  // (all of them with 0 and 1: the single-GPU case); local_rows = 8 x the number of those tile rows.
  uint32_t row_first, row_stride, local_rows;
  uint32_t n_chunks, chunk_spp;
  uint32_t spp_begin, spp_end;
  uint32_t seed_lo, seed_hi;
  uint32_t n_units;    // 64 x tiles x chunks
  uint32_t batch_chunks, batches_per_tile, n_batches;  // a wave takes batch_chunks chunks of one tile at a time from the global counter
  uint32_t direct;     // 1: out is the image and is added to; 0: out is the partial-sum buffer and is overwritten
  uint32_t quad_tiles; // 1: 64 consecutive units of a tile are a 4x4 block of pixels x 4 chunks (n_chunks and batch_chunks are multiples of 4); 0: 8x8 pixels x 1 chunk
  uint32_t inner_min;  // a wave runs the inner-node block while at least this many of its lanes want it
  uint32_t leaf_min;   // inside a run of inner-node rounds the leaf block runs as soon as this many lanes wait at leaves
  uint32_t both_min;   // when a traversal block (inner or leaf) is due, the other one runs too if this many lanes want it
  float time_scale;    // power of two that brings the times of the scene, seen from the camera, below 1 (0: from the scene's extent alone)
  // Entry lists of the tiles (launch_tile_cuts), kCutEntries words per tile in the order the kernel numbers its tiles, or null:
  // every camera ray then starts its walk at the root.
  const uint32_t* cuts;
  double* out;
  unsigned long long* counters;  // see kRenderCounters
};

// counters: [0] next batch, [1] rays, [2] samples, [3] node steps, [4] sphere tests;
// profiling variant only: [5+2k] runs and [6+2k] active lanes of block k, [13+k] shader-clock
// cycles of block k, [17] cycles of whole work items.
constexpr int kRenderCounters = 18;
// Rows (one word per lane) of a two-wide tree's LDS stack below its bottom entry: see stack_slot in rt_device.h.
constexpr uint32_t kStackSpareRows = 1;

// Dynamic LDS a tree-walking kernel needs for this scene (includes the camera block).
size_t tree_lds_bytes(const SceneDev& sc);
// LDS footprint of one staged node / of the fixed part (camera constants, tables, the waves' entry lists: more of those for the four-wide walk).
size_t lds_node_bytes();
size_t lds_wide_node_bytes();
size_t lds_fixed_bytes(bool wide);

// Entry lists ("cuts") for camera rays.  All camera rays of an 8x8 pixel tile lie in a thin beam around the ray from
// the middle of the lens through the middle of the tile.  Before a render, launch_tile_cuts walks the tree once per tile
// with that beam and records up to `k` inner nodes (kCutEntries words per tile, the nearest first, unused words
// 0xffffffff) such that every sphere a camera ray of the tile can hit lies below one of them; the render kernel starts the
// tile's camera rays at those nodes instead of at the root.  A walk that starts with k - 1 entries set aside needs up to
// k - 2 more stack entries than a walk from the root.
constexpr uint32_t kCutEntries = 4;
constexpr uint32_t kCutRing = 3;  // tiles whose entry lists a wave keeps at hand (the current batch's and the two before it)
// Scenes walked through the four-wide tree (where a node step costs a trip to L1 or L2) get longer lists whose entries
// carry the node's own box (f32, rounded outwards).  A camera ray tests the boxes when it starts and sets aside only the
// entries it can hit.  A list is laid out as groups of four entries in the form of a staged four-wide node (28 words: per
// axis the four low planes and the four high planes, then the four references; an absent entry has an inverted infinite
// box and the reference 0xffffffff), so that a lane reads a group as it reads a node -- by direction, seven reads asked
// for at once -- and tests it with the node test.  Entries fill a list from the front.  (Measured on 10^4 .. 10^6 spheres:
// lists of 6 and of 12 within 1.5 % of 8; a list per quarter of a tile, which takes four times the LDS, 1 - 7 % slower.)
#ifndef RTA_WIDE_CUT_ENTRIES  // (overridden by tools/build_variant.sh for measurements)
#define RTA_WIDE_CUT_ENTRIES 8
#endif
constexpr uint32_t kWideCutEntries = RTA_WIDE_CUT_ENTRIES;
static_assert(kWideCutEntries % 4 == 0, "lists of the four-wide walk are made of groups of four entries");
constexpr uint32_t kWideCutGroupWords = 28;  // one group: 7 x 16 bytes, the staged form of a four-wide node
constexpr uint32_t cut_entries(bool wide) { return wide ? kWideCutEntries : kCutEntries; }
constexpr uint32_t cut_entry_words(bool wide) { return wide ? 8u : 1u; }
constexpr uint32_t cut_tile_words(bool wide) { return cut_entries(wide) * cut_entry_words(wide); }  // what launch_tile_cuts writes per tile
// rho0: bound on the distance of a camera ray's origin from the lens centre; h_pixel: bound on the distance between two
// points of the focus plane that lie half a pixel apart in each direction (scene units); tile t covers the pixels from
// (x0 + (t % tiles_x) * 8, y0 + ((t / tiles_x) * row_stride + row_first) * 8).
hipError_t launch_tile_cuts(const SceneDev& sc, const CameraDev& cam, double rho0, double h_pixel, uint32_t x0, uint32_t y0, uint32_t tiles_x,
                            uint32_t n_tiles, uint32_t row_first, uint32_t row_stride, uint32_t k, uint32_t* cuts, hipStream_t stream);

// All launchers enqueue on `stream` and return the launch status; pointers are device pointers.
hipError_t launch_trace_batch(const SceneDev& sc, const double* org, const double* dir, size_t n, double t_min, double t_max, int32_t* prim,
                              double* t, double* normal, uint8_t* front, uint32_t* counts, int max_blocks, hipStream_t stream);
// mat: material index in the low 30 bits, material type in the top 2.
hipError_t launch_scatter_batch(const SceneDev& sc, const uint32_t* mat, const double* in_dir, const double* p, const double* normal,
                                const uint8_t* front, const double* u, size_t n, uint8_t* scattered, double* attenuation, double* out_dir,
                                hipStream_t stream);
// boxes: n x 8 floats (lo.xyz, bound, hi.xyz, unused), rounded outwards by the caller; bound = largest |coordinate| of the box.
hipError_t launch_slab_batch(const float* boxes, const double* org, const double* dir, const double* interval, size_t n, uint8_t* hit,
                             hipStream_t stream);
hipError_t launch_primary_rays(const CameraDev& cam, uint64_t seed, const uint32_t* ijs, size_t n, double* org, double* dir,
                               hipStream_t stream);
// profile: run the variant that also fills the profiling counters.
hipError_t launch_render(const RenderParams& P, int blocks, bool profile, hipStream_t stream);
// Loads the code of the scene's render kernel ahead of the first render (scene creation calls it).
hipError_t preload_render(const SceneDev& sc);
hipError_t launch_reduce_chunks(const double* partial, double* out, uint32_t n_chunks, uint32_t x0, uint32_t y0, uint32_t y1, uint32_t tile_w,
                                uint32_t local_rows, uint32_t row_first, uint32_t row_stride, uint32_t width, hipStream_t stream);
// out[k] += src[0][k] + ... + src[n_src-1][k] (in that order) for k in [begin, end); at most kMaxAddFrames sources.
constexpr uint32_t kMaxAddFrames = 16;
hipError_t launch_add_frames(double* out, const double* const* src, uint32_t n_src, size_t begin, size_t end, hipStream_t stream);
// The library's static device array on `device` (one copy per GPU): its address there and its size.  Nothing is allocated.
hipError_t static_scratch(int device, void** ptr, size_t* bytes);
// out (bytes, 2-byte aligned) = write_colour of each of the n_components sums.
hipError_t launch_quantise(const double* rgb_sum, size_t n_components, uint64_t spp, uint8_t* out, hipStream_t stream);

// The wavefront pipeline kept for measurement (rt_wavefront.hip, linked into libraytracer_amd_experiments.so only): samples [spp_begin, spp_end) of the whole frame added
// into `frame`; workspace of wavefront_workspace_bytes(capacity) bytes, capacity a multiple of width x height.
size_t wavefront_workspace_bytes(uint32_t capacity);
hipError_t wavefront_render(const SceneDev& sc, const CameraDev& cam, uint64_t seed, uint32_t spp_begin, uint32_t spp_end, double* frame,
                            void* workspace, uint32_t capacity, int num_cus, hipStream_t stream, uint64_t* rays_out, uint32_t* launches_out);

}  // namespace rta
