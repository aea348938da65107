// Host-callable launchers of the render kernels (kl_kernels.hip).
#pragma once

#include "Math/kl_render.h"
#include "kl_types.hpp"

namespace kl {

struct LaunchCfg {
    uint32_t cus;                    // compute units of the device
    uint32_t isect_blocks_per_cu;    // resident traversal workgroups per CU (bounded by LDS and VGPRs)
};

// seed_hits != null: also copies each pixel's camera-ray hit (found for a one-frame queue of segment capacity
// seed_segcap) into `hits` for every frame of the batch; q.tp may be null (rays only).
// flat (without seed hits): also writes the flat-primitive hit record of every camera ray into `hits`.
void launch_generate(hipStream_t s, const LaunchCfg &lc, const SceneDev &sc, const FrameDev &f, RayQueue q, uint32_t *count0,
                     const float4 *seed_hits = nullptr, uint32_t seed_segcap = 0, float4 *hits = nullptr, bool flat = false);
// ints of global memory the traversal kernel needs for stack entries beyond its LDS part (per launch grid)
size_t intersect_overflow_ints(const LaunchCfg &lc);
bool scene_has_flat(const SceneDev &sc);
void launch_intersect_flat(hipStream_t s, const LaunchCfg &lc, const SceneDev &sc, const FrameDev &f, RayQueue q, float4 *hits,
                           const uint32_t *counts, Counters *cn, bool stats, uint32_t max_rays);
void launch_intersect(hipStream_t s, const LaunchCfg &lc, const SceneDev &sc, const FrameDev &f, RayQueue q, float4 *hits,
                      const uint32_t *count, uint32_t *head, Counters *cn, int *overflow, bool bvh, bool stats, uint32_t max_rays,
                      bool flat_by_producer = false);   // the flat-primitive records in `hits` came from generate / shade
void launch_shade(hipStream_t s, const LaunchCfg &lc, const SceneDev &sc, const FrameDev &f, RayQueue qin, const float4 *hits,
                  const uint32_t *count_in, RayQueue qout, uint32_t *count_out, float4 *raw, uint32_t bounce, bool fused,
                  uint32_t max_rays, float4 *hits_out = nullptr);   // hits_out: flat-primitive records of the pushed rays (fused only)
void launch_compact(hipStream_t s, const LaunchCfg &lc, const FrameDev &f, RayQueue qin, const uint32_t *count_in, RayQueue qout,
                    uint32_t *count_out, uint32_t max_rays);
void launch_accumulate(hipStream_t s, const LaunchCfg &lc, const FrameDev &f, const float4 *raw, float4 *acc, float4 *last_raw,
                       bool camera_change, uint32_t accum_mode);
void launch_resolve(hipStream_t s, const LaunchCfg &lc, const float4 *acc, uchar4 *out, uint32_t n, uint32_t tonemap);
void launch_unpack_hits(hipStream_t s, const SceneDev &sc, const float4 *hits, uint32_t n, float *t, uint32_t *prim, float2 *uv);

}  // namespace kl
