// Host-callable launchers of the render kernels (kl_kernels.hip).
#pragma once

#include "Math/kl_render.h"
#include "kl_types.hpp"

namespace kl {

struct LaunchCfg {
    uint32_t cus;                    // compute units of the device
    uint32_t isect_blocks_per_cu;    // resident traversal workgroups per CU (bounded by LDS and VGPRs)
    uint32_t shade_blocks_per_cu;    // upper bound of the shade grid, in 256-thread workgroups per CU
};

// Every ray leaves generate with its closest hit so far in its queue entry: copied from the one-frame queue `seed`
// (segment capacity seed_segcap) in which each pixel's camera ray was traced once, if seed.od is set; else, with
// `flat`, the hit among the flat primitives; else a miss. q.tp may be null (rays only).
void launch_generate(hipStream_t s, const LaunchCfg &lc, const SceneDev &sc, const FrameDev &f, RayQueue q, uint32_t *count0,
                     bool flat = false, RayQueue seed = RayQueue{nullptr, nullptr, nullptr, nullptr}, uint32_t seed_segcap = 0);
// ints of global memory the traversal kernel needs for stack entries beyond its LDS part (per launch grid)
size_t intersect_overflow_ints(const LaunchCfg &lc);
bool scene_has_flat(const SceneDev &sc);
// the flat primitives for every ray of a queue; the render loop runs them inside generate / shade instead
void launch_intersect_flat(hipStream_t s, const LaunchCfg &lc, const SceneDev &sc, const FrameDev &f, RayQueue q,
                           const uint32_t *counts, Counters *cn, bool stats, uint32_t max_rays);
// closest hits into the queue entries (and q.uv); stats: 0 none, 1 the instrumented all-types kernel, 2 wave-level tallies in the
// kernel the scene would run anyway; flat_by_producer only says who tallies the flat tests in statistics
void launch_intersect(hipStream_t s, const LaunchCfg &lc, const SceneDev &sc, const FrameDev &f, RayQueue q,
                      const uint32_t *count, uint32_t *head, Counters *cn, int *overflow, bool bvh, int stats, uint32_t max_rays,
                      bool flat_by_producer = false);
// flat (with fused compaction): the pushed rays leave with their flat-primitive hit; otherwise with a miss
void launch_shade(hipStream_t s, const LaunchCfg &lc, const SceneDev &sc, const FrameDev &f, RayQueue qin,
                  const uint32_t *count_in, RayQueue qout, uint32_t *count_out, float4 *raw, uint32_t bounce, bool fused,
                  uint32_t max_rays, bool flat = false);
void launch_compact(hipStream_t s, const LaunchCfg &lc, const FrameDev &f, RayQueue qin, const uint32_t *count_in, RayQueue qout,
                    uint32_t *count_out, uint32_t max_rays);
void launch_accumulate(hipStream_t s, const LaunchCfg &lc, const FrameDev &f, const float4 *raw, float4 *acc, float4 *last_raw,
                       bool camera_change, uint32_t accum_mode);
void launch_resolve(hipStream_t s, const LaunchCfg &lc, const float4 *acc, uchar4 *out, uint32_t n, uint32_t tonemap);
void launch_unpack_hits(hipStream_t s, const SceneDev &sc, RayQueue q, uint32_t n, float *t, uint32_t *prim, float2 *uv);

// closest hits for caller rays without allocating anything: all intermediate storage is a file-scope device array
// (kl_kernels.hip). noalloc_init(device), with that device current, once per device before the first launch.
hipError_t noalloc_init(int device);
size_t noalloc_scratch_bytes();
hipError_t launch_trace_noalloc(hipStream_t s, int device, const LaunchCfg &lc, const SceneDev &sc, const FrameDev &f, const float *o,
                                const float *d, uint32_t n, float *t, uint32_t *prim, float2 *uv, bool bvh);

}  // namespace kl
