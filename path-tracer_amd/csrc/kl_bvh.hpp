// Host-side SAH BVH builder. Output is the flat 64-byte node array and the
// leaf ordering of every primitive type that the traversal kernel consumes
// (layout: kl_types.hpp).
#pragma once

#include <cstdint>
#include <vector>

#include "Math/kl_render.h"

namespace kl {

struct HostBVH {
    std::vector<float> nodes;            // 16 floats (64 bytes) per node
    int32_t root;                        // child reference of the whole scene
    std::vector<uint32_t> sph_order;     // leaf position -> index into the uploaded sphere array
    std::vector<uint32_t> quad_order;
    std::vector<uint32_t> tri_order;
    uint32_t flat_start[3], flat_count[3];   // flat primitives close each order array (indexed by PRIM_*)
    kl_bvh_stats stats;
    // 4-wide form of the same tree (collapse_bvh4): 16 words (64 bytes) per node, see kl_types.hpp
    std::vector<uint32_t> nodes4;
    int32_t root4;
    uint32_t depth4;
};

// Builds over all primitives. Guarantees: every leaf holds 1..8 primitives of one type; no leaf
// lies deeper than BVH_MAX_DEPTH inner nodes, so a traversal stack of that many entries cannot
// overflow; leaf boxes are widened by a few ulps of their coordinates so that rounding in the
// slab test cannot cull a primitive the exact arithmetic would reach.
// Primitives whose box area is at least `flat_fraction` of the scene box area (at most `max_flat`, largest
// first) are left out of the tree and appended to their order array ("flat" primitives): giant primitives
// wreck a SAH tree, and testing a handful of them coherently for every ray is cheaper than meeting them
// in divergent leaves. flat_fraction <= 0 disables this.
void build_bvh(const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq, const kl_tri *t, uint32_t nt,
               HostBVH &out, unsigned threads, float flat_fraction, uint32_t max_flat);

// Collapses the binary tree into 4-wide nodes with 8-bit child boxes quantised outwards on the node's own
// grid (64 bytes per node). Leaves and their order are untouched; boxes only get looser, never tighter, so
// traversal visits a superset of what the exact boxes admit and finds the same hits.
void collapse_bvh4(HostBVH &bvh);

// No tree: identity ordering, root = NODE_DONE (flat-loop kernel only).
void identity_order(uint32_t ns, uint32_t nq, uint32_t nt, HostBVH &out);

}  // namespace kl
