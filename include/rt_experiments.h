/* Entry points of libraytracer_amd_experiments.so: the product library (include/rt.h, all of whose entry points this
 * library exports as well) plus the measurement harnesses that are kept out of the shipped library.  No counterpart in
 * the reference; callers are tools/compare_pipelines.py, tools/pmc_pipelines.sh and tests/test_gpu_wavefront.py. */
#ifndef RT_EXPERIMENTS_H
#define RT_EXPERIMENTS_H
#include "rt.h"
#ifdef __cplusplus
extern "C" {
#endif

/* The same samples of the whole frame rendered by a wavefront pipeline -- one kernel per stage (generate, trace,
 * shade, resolve) with ray queues in device memory, rays that go on packed with wave ballots -- so that its profile
 * can be laid next to the persistent kernel's.  Same paths, hence the same frame up to the order of additions.
 * rgb_sum_dev is device memory and is added to; runs on the default stream and waits for it.  rt_debug_set knows
 * one more name in this library: "wavefront_paths" (paths kept in flight). */
int rt_experiment_wavefront_device(const rt_scene* scene, const rt_config* cfg, uint64_t seed, uint32_t spp_begin, uint32_t spp_end,
                                   double* rgb_sum_dev, rt_render_stats* stats);

#ifdef __cplusplus
}
#endif
#endif
