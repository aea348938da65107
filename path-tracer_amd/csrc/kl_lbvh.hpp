// GPU-side BVH builder and refit (no reference counterpart): Morton order, clustering (or radix tree), collapse into the 4-wide
// quantised nodes of kl_types.hpp. Same output contract as the host builder (kl_bvh.hpp): it fills the
// 4-wide part of HostBVH (nodes4, root4, depth4), the per-type leaf orders and the flat-primitive ranges, so
// everything downstream of the builder (record packing, traversal) is the same code for both.
#pragma once

#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "kl_bvh.hpp"

namespace kl {

enum LbvhResult { LBVH_OK = 0, LBVH_TOO_DEEP = 1, LBVH_HIP_ERROR = -1 };

// What build_bvh_gpu hands over when it also packs the scene's records on the device: two hipMalloc'ed arrays
// of 64-byte records (layout: kl_types.hpp). Whoever takes a pointer out sets it to null here; what is left is freed.
struct GpuScene {
    void *d_rec = nullptr; size_t n_rec = 0;           // nodes, then spheres, quads, triangles in leaf order
    void *d_shrec = nullptr; size_t n_shrec = 0;       // one shading record per primitive, same order
    std::vector<uint32_t> flat_quads;                  // the flat quads' indices into the caller's quad array, in flat order
    GpuScene() = default;
    GpuScene(const GpuScene &) = delete;
    GpuScene &operator=(const GpuScene &) = delete;
    ~GpuScene() { if (d_rec) (void)hipFree(d_rec); if (d_shrec) (void)hipFree(d_shrec); }
};

// Builds on `stream`'s device. A leaf holds one primitive, or two that were siblings in the binary tree.
// LBVH_TOO_DEEP: the tree is deeper than the traversal stack allows (BVH_MAX_DEPTH 4-wide levels; possible when
// very many primitives share a Morton cell) and the caller builds on the host instead; LBVH_HIP_ERROR: a HIP
// call failed (`err` says which). Primitives whose box area is at least `flat_fraction` of the scene's (at most
// `max_flat`, and at most 1024 candidates: beyond that none) stay out of the tree.
// With `packed` (and the materials) the builder also writes the record and shading-record arrays on the GPU,
// from the primitives it has already copied there; `out` then carries only the statistics, the flat ranges and
// the root (no nodes4, no orders), and nothing but a handful of words returns to the host.
LbvhResult build_bvh_gpu(hipStream_t stream, const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq,
                   const kl_tri *t, uint32_t nt, HostBVH &out, float flat_fraction, uint32_t max_flat, std::string &err,
                   const kl_material *m = nullptr, uint32_t nm = 0, GpuScene *packed = nullptr);

// Refit: the tree of an uploaded scene kept as it is, every box recomputed for primitives that have moved. `rec` /
// `shrec` are the scene's record arrays on the device (n_nodes nodes first), `root` its root reference, `levels`
// the number of node levels (the build's max_depth). The new arrays hold the same primitives in the same order as
// at upload; their geometry (and the triangles' uvs) replaces the old, materials stay. One launch per level down
// (which level each node is on) and one per level up (boxes), all on `stream`; returns when they are done.
LbvhResult refit_scene_gpu(hipStream_t stream, float4 *rec, float4 *shrec, uint32_t n_nodes, int32_t root, uint32_t levels,
                           const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq, const kl_tri *t, uint32_t nt, std::string &err);

}  // namespace kl
