// GPU-side BVH builder (no reference counterpart): Morton-ordered LBVH, refit, collapse into the 4-wide
// quantised nodes of kl_types.hpp. Same output contract as the host builder (kl_bvh.hpp): it fills the
// 4-wide part of HostBVH (nodes4, root4, depth4), the per-type leaf orders and the flat-primitive ranges, so
// everything downstream of the builder (record packing, traversal) is the same code for both.
#pragma once

#include <string>

#include <hip/hip_runtime.h>

#include "kl_bvh.hpp"

namespace kl {

enum LbvhResult { LBVH_OK = 0, LBVH_TOO_DEEP = 1, LBVH_HIP_ERROR = -1 };

// Builds on `stream`'s device. A leaf holds one primitive, or two that were siblings in the binary tree. LBVH_TOO_DEEP: the tree is deeper than the
// traversal stack allows (BVH_MAX_DEPTH 4-wide levels; possible when very many primitives share a Morton
// cell) and the caller builds on the host instead; LBVH_HIP_ERROR: a HIP call failed (`err` says which).
// Primitives whose box area is at least `flat_fraction` of the scene's (at most `max_flat`, and at most
// 1024 candidates: beyond that none) stay out of the tree.
LbvhResult build_bvh_gpu(hipStream_t stream, const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq,
                   const kl_tri *t, uint32_t nt, HostBVH &out, float flat_fraction, uint32_t max_flat, std::string &err);

}  // namespace kl
