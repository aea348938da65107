// GPU-side BVH builder: see kl_lbvh.hpp.
//
//   bounds     : padded box and area of every primitive, scene box (wave reduction + ordered-int atomics)
//   keys       : 63-bit Morton-style code of the box centre, each bit halving the currently longest cell axis;
//                candidates for the flat list
//   sort       : hipCUB radix sort of (code, primitive); flat primitives carry the largest key and end up last
//   cluster    : binary tree by locally-ordered clustering over the sorted sequence (k_ploc_*), whose left-to-right
//                leaf order replaces the sorted order. With search radius 0 instead:
//     hierarchy: binary radix tree over the sorted codes (Karras 2012), ties broken by position
//     depth    : depth of every inner node (walk to the root; read-only data)
//     refit    : boxes of the inner nodes, ONE LAUNCH PER DEPTH LEVEL from the deepest up: a node reads boxes
//                that an earlier launch wrote, so no hand-off between workgroups happens inside a launch
//   orders     : per-type leaf orders (exclusive scans of the type flags over the leaf sequence)
//   collapse   : 4-wide nodes with 8-bit child boxes, one launch per 4-wide level from the root down; a node
//                opens its inner child with the largest box until it has four children (as the host collapse)
#include "kl_lbvh.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "kl_math.hpp"
#include "kl_types.hpp"

namespace kl {
namespace {

constexpr uint32_t FLAT_CAP = 1024;          // flat candidates the device list holds
constexpr uint32_t WALK_CAP = 4096;          // bound of every parent walk (a tree is never this deep)
struct FlatCand { uint32_t prim; float area; };

// float <-> unsigned with the same order (for atomicMin / atomicMax on floats of either sign)
__host__ __device__ inline uint32_t f2ord(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f; memcpy(&f, &u, 4);
    return f;
}

__device__ __forceinline__ void grow(float lo[3], float hi[3], const float p[3]) {
    for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], p[k]); hi[k] = fmaxf(hi[k], p[k]); }
}

// Boxes of the primitives, widened by a few ulps like the host builder's leaves: rounding in the slab test must
// not cull a primitive that exact arithmetic would reach. lo / hi come in as +inf / -inf.
__device__ __forceinline__ void pad_box(float lo[3], float hi[3]) {
    for (int k = 0; k < 3; k++) {
        const float m = fmaxf(fabsf(lo[k]), fabsf(hi[k]));
        const float pad = 1e-6f + 4e-6f * m;
        lo[k] -= pad; hi[k] += pad;
    }
}
__device__ __forceinline__ void sphere_box(const kl_sphere &x, float lo[3], float hi[3]) {
    for (int k = 0; k < 3; k++) { lo[k] = x.center[k] - x.radius; hi[k] = x.center[k] + x.radius; }
    pad_box(lo, hi);
}
__device__ __forceinline__ void quad_box(const kl_quad &x, float lo[3], float hi[3]) {
    grow(lo, hi, x.v0); grow(lo, hi, x.v1); grow(lo, hi, x.v2); grow(lo, hi, x.v3);
    pad_box(lo, hi);
}
__device__ __forceinline__ void tri_box(const kl_tri &x, float lo[3], float hi[3]) {
    grow(lo, hi, x.v0); grow(lo, hi, x.v1); grow(lo, hi, x.v2);
    pad_box(lo, hi);
}

__global__ __launch_bounds__(256) void k_lbvh_bounds(const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq, const kl_tri *t, uint32_t nt,
                                                     float4 *blo, float4 *bhi, uint32_t *scene) {
    const uint32_t n = ns + nq + nt;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const float inf = __builtin_huge_valf();
    float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
    if (g < n) {
        if (g < ns) sphere_box(s[g], lo, hi);
        else if (g < ns + nq) quad_box(q[g - ns], lo, hi);
        else tri_box(t[g - ns - nq], lo, hi);
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        const float area = (dx < 0 || dy < 0 || dz < 0) ? 0.0f : 2.0f * (dx * dy + dy * dz + dz * dx);
        blo[g] = make_float4(lo[0], lo[1], lo[2], area);
        bhi[g] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 3; k++) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    if ((threadIdx.x & 63u) == 0u)
        for (int k = 0; k < 3; k++) {
            if (lo[k] < inf) atomicMin(&scene[k], f2ord(lo[k]));
            if (hi[k] > -inf) atomicMax(&scene[3 + k], f2ord(hi[k]));
        }
}

// Which axis each of the 63 code bits halves, most significant first: always the axis along which the cell is
// longest at that point, so that cells stay close to cubes whatever the proportions of the scene box (plain
// interleaving gives a 1 : 1 : 4 room cells four times as long as wide at every level). axes[0..62] = axis of
// bit, axes[64..66] = bits each axis gets.
__global__ void k_lbvh_axes(const uint32_t *scene, uint8_t *axes) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float ext[3]; uint32_t nb[3] = { 0, 0, 0 };
    for (int k = 0; k < 3; k++) { ext[k] = ord2f(scene[3 + k]) - ord2f(scene[k]); if (!(ext[k] > 0.0f)) ext[k] = 0.0f; }
    for (int b = 0; b < 63; b++) {
        int a = 0;
        if (ext[1] > ext[a]) a = 1;
        if (ext[2] > ext[a]) a = 2;
        if (nb[a] >= 30u) {                       // an axis holds at most 30 bits: the next longest takes over
            int o1 = (a + 1) % 3, o2 = (a + 2) % 3;
            a = (nb[o1] < 30u && (ext[o1] >= ext[o2] || nb[o2] >= 30u)) ? o1 : o2;
        }
        axes[b] = (uint8_t)a; nb[a]++; ext[a] *= 0.5f;
    }
    for (int k = 0; k < 3; k++) axes[64 + k] = (uint8_t)nb[k];
}

__global__ __launch_bounds__(256) void k_lbvh_keys(const float4 *blo, const float4 *bhi, uint32_t n, const uint32_t *scene, const uint8_t *axes,
                                                   float flat_fraction, FlatCand *cand, uint32_t *ncand, unsigned long long *keys, uint32_t *vals) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    float slo[3], ext[3];
    for (int k = 0; k < 3; k++) { slo[k] = ord2f(scene[k]); ext[k] = ord2f(scene[3 + k]) - slo[k]; }
    const float sarea = 2.0f * (ext[0] * ext[1] + ext[1] * ext[2] + ext[2] * ext[0]);
    const float4 lo = blo[g], hi = bhi[g];
    if (flat_fraction > 0.0f && lo.w >= flat_fraction * sarea) {
        const uint32_t idx = atomicAdd(ncand, 1u);
        if (idx < FLAT_CAP) { cand[idx].prim = g; cand[idx].area = lo.w; }
    }
    const float c[3] = { 0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z) };
    // the box centre on a grid of 2^nb cells per axis, then the bits dealt out in the order k_lbvh_axes chose
    uint32_t ic[3], left[3];
    for (int k = 0; k < 3; k++) {
        const uint32_t nb = axes[64 + k];
        double u = ext[k] > 0.0f ? ((double)c[k] - (double)slo[k]) / (double)ext[k] : 0.0;
        u = !(u >= 0.0) ? 0.0 : (u > 1.0 ? 1.0 : u);
        const double cells = (double)(1ull << nb);
        double q = u * cells;
        if (q > cells - 1.0) q = cells - 1.0;
        ic[k] = (uint32_t)q; left[k] = nb;
    }
    unsigned long long key = 0ull;
    for (int b = 0; b < 63; b++) {
        const uint32_t a = axes[b];
        const uint32_t v = a == 0u ? ic[0] : (a == 1u ? ic[1] : ic[2]);
        const uint32_t l = (a == 0u ? left[0] : (a == 1u ? left[1] : left[2])) - 1u;
        key = (key << 1) | (unsigned long long)((v >> l) & 1u);
        if (a == 0u) left[0] = l; else if (a == 1u) left[1] = l; else left[2] = l;
    }
    keys[g] = key;
    vals[g] = g;
}
// flat primitives sort behind every Morton code (which has 63 bits)
__global__ void k_lbvh_mark_flat(const uint32_t *flat, uint32_t nflat, unsigned long long *keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nflat) keys[flat[i]] = ~0ull;
}

__global__ __launch_bounds__(256) void k_lbvh_typeflags(const uint32_t *vals, uint32_t n, uint32_t ns, uint32_t nq, uint32_t *f0, uint32_t *f1, uint32_t *f2) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t g = vals[p];
    f0[p] = g < ns ? 1u : 0u;
    f1[p] = (g >= ns && g < ns + nq) ? 1u : 0u;
    f2[p] = g >= ns + nq ? 1u : 0u;
}
// position p of the sorted sequence holds primitive vals[p]; its rank among the primitives of its type is its place in
// that type's leaf order, and a one-primitive leaf names it
__global__ __launch_bounds__(256) void k_lbvh_orders(const uint32_t *vals, uint32_t n, uint32_t ns, uint32_t nq, const uint32_t *r0, const uint32_t *r1,
                                                     const uint32_t *r2, uint32_t *ord_s, uint32_t *ord_q, uint32_t *ord_t, int *leafref) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t g = vals[p];
    if (g < ns) { ord_s[r0[p]] = g; leafref[p] = make_leaf(PRIM_SPHERE, r0[p], 1u); }
    else if (g < ns + nq) { ord_q[r1[p]] = g - ns; leafref[p] = make_leaf(PRIM_QUAD, r1[p], 1u); }
    else { ord_t[r2[p]] = g - ns - nq; leafref[p] = make_leaf(PRIM_TRI, r2[p], 1u); }
}

// length of the common prefix of the codes at sorted positions i and j, position bits appended so that equal
// codes still differ; -1 outside the sequence
__device__ __forceinline__ int delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz(i ^ j);
    return __clzll((long long)(a ^ b));
}
// inner node i of the binary radix tree over n sorted leaves. Children: >= 0 inner node, < 0 leaf at position ~c.
__global__ __launch_bounds__(256) void k_lbvh_hierarchy(const unsigned long long *keys, int n, int *left, int *right, int *parent_int, int *parent_leaf) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (lmax < (1 << 30) && delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    if (lo == gamma) { left[i] = ~gamma; parent_leaf[gamma] = i; } else { left[i] = gamma; parent_int[gamma] = i; }
    if (hi == gamma + 1) { right[i] = ~(gamma + 1); parent_leaf[gamma + 1] = i; } else { right[i] = gamma + 1; parent_int[gamma + 1] = i; }
    if (i == 0) parent_int[0] = -1;
}

__global__ __launch_bounds__(256) void k_lbvh_depth(const int *parent_int, uint32_t n_int, uint32_t *depth, uint32_t *maxdepth) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t d = 0;
    if (i < n_int) {
        int p = parent_int[i];
        while (p >= 0 && d < WALK_CAP) { d++; p = parent_int[p]; }
        depth[i] = d;
    }
    for (int off = 32; off > 0; off >>= 1) d = max(d, (uint32_t)__shfl_xor((int)d, off));
    if ((threadIdx.x & 63u) == 0u) atomicMax(maxdepth, d);
}

struct CBox { float lo[3], hi[3]; };
__device__ __forceinline__ CBox child_box(int c, const uint32_t *vals, const float4 *blo, const float4 *bhi, const float4 *ilo, const float4 *ihi) {
    float4 a, b;
    if (c < 0) { const uint32_t g = vals[~c]; a = blo[g]; b = bhi[g]; } else { a = ilo[c]; b = ihi[c]; }
    CBox x; x.lo[0] = a.x; x.lo[1] = a.y; x.lo[2] = a.z; x.hi[0] = b.x; x.hi[1] = b.y; x.hi[2] = b.z;
    return x;
}
__device__ __forceinline__ float box_area(const CBox &b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return (dx < 0 || dy < 0 || dz < 0) ? 0.0f : 2.0f * (dx * dy + dy * dz + dz * dx);
}
// boxes of the inner nodes of one depth level; their children's boxes were written by earlier launches
__global__ __launch_bounds__(256) void k_lbvh_refit(uint32_t level, const uint32_t *depth, uint32_t n_int, const int *left, const int *right,
                                                    const uint32_t *vals, const float4 *blo, const float4 *bhi, float4 *ilo, float4 *ihi) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_int || depth[i] != level) return;
    const CBox a = child_box(left[i], vals, blo, bhi, ilo, ihi), b = child_box(right[i], vals, blo, bhi, ilo, ihi);
    ilo[i] = make_float4(fminf(a.lo[0], b.lo[0]), fminf(a.lo[1], b.lo[1]), fminf(a.lo[2], b.lo[2]), 0.0f);
    ihi[i] = make_float4(fmaxf(a.hi[0], b.hi[0]), fmaxf(a.hi[1], b.hi[1]), fmaxf(a.hi[2], b.hi[2]), 0.0f);
}

// 4-wide nodes [begin, end): node idx is built from binary node src[idx]. Its inner children get consecutive
// 4-wide indices from `alloc` and are built by the next launch.
__global__ __launch_bounds__(256) void k_lbvh_collapse(uint32_t begin, uint32_t end, int *src, uint32_t *alloc, const int *left, const int *right,
                                                       const int *leafref, const uint32_t *vals, const float4 *blo, const float4 *bhi,
                                                       const float4 *ilo, const float4 *ihi, uint32_t *nodes4) {
    const uint32_t idx = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= end) return;
    const int b2 = src[idx];
    // a child is an inner binary node (ref >= 0, may be opened) or a finished leaf reference (ref = -1, the reference
    // itself in leaf[]). Two sibling leaves of one primitive type become ONE two-primitive leaf: they are neighbours
    // in the sorted sequence, hence in their type's leaf order, and testing two primitives costs less than visiting a
    // node whose only job is to tell them apart.
    CBox box[4]; int ref[4]; int leaf[4]; int nch = 2;
    auto child = [&](int c, int k) {
        box[k] = child_box(c, vals, blo, bhi, ilo, ihi);
        if (c < 0) { ref[k] = -1; leaf[k] = leafref[~c]; return; }
        const int l = left[c], r = right[c];
        if (l < 0 && r < 0) {
            const uint32_t a = (uint32_t)~leafref[~l], b = (uint32_t)~leafref[~r];       // (type << 29) | ((count-1) << 26) | start
            if ((a >> 29) == (b >> 29) && (b & 0x3FFFFFFu) == (a & 0x3FFFFFFu) + 1u) { ref[k] = -1; leaf[k] = ~(int)(a + (1u << 26)); return; }
        }
        ref[k] = c; leaf[k] = 0;
    };
    child(left[b2], 0); child(right[b2], 1);
    while (nch < 4) {                      // open the inner child with the largest box
        int best = -1; float best_area = -1.0f;
        for (int k = 0; k < nch; k++) {
            const float a = box_area(box[k]);
            if (ref[k] >= 0 && a > best_area) { best = k; best_area = a; }
        }
        if (best < 0) break;
        const int open = ref[best];
        child(left[open], best);
        child(right[open], nch);
        nch++;
    }
    uint32_t w[16];
    static_assert(sizeof(CBox) == 24, "node4_boxes reads the boxes as 6 floats each");
    node4_boxes(&box[0].lo[0], nch, w);
    uint32_t n_inner = 0;
    for (int k = 0; k < nch; k++) n_inner += ref[k] >= 0 ? 1u : 0u;
    uint32_t next = n_inner ? atomicAdd(alloc, n_inner) : 0u;
    for (int k = 0; k < nch; k++) {
        if (ref[k] >= 0) { src[next] = ref[k]; w[12 + k] = next++; }
        else w[12 + k] = (uint32_t)leaf[k];
    }
    for (int k = nch; k < 4; k++) w[12 + k] = w[12];     // an empty slot repeats child 0: always a valid reference
    uint4 *dst = (uint4 *)(nodes4 + (size_t)idx * 16);
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]); dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    dst[2] = make_uint4(w[8], w[9], w[10], w[11]); dst[3] = make_uint4(w[12], w[13], w[14], w[15]);
}

// ---- record packing (the device twin of the host loops in kl_upload_scene: same words, same arithmetic) ----
__global__ __launch_bounds__(256) void k_lbvh_pack_nodes(const uint32_t *nodes4, uint32_t count4, uint32_t base0, uint32_t base1, uint32_t base2, uint4 *rec) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count4) return;
    const uint4 *src = (const uint4 *)(nodes4 + (size_t)k * 16);
    uint4 c = src[3];
    uint32_t w[4] = { c.x, c.y, c.z, c.w };
    for (int j = 0; j < 4; j++) {
        const int32_t child = (int32_t)w[j];
        if (child < 0) {                                   // a leaf counts within its type: make it a record index
            const uint32_t lv = (uint32_t)~child, type = (lv >> 29) & 3u, cnt = ((lv >> 26) & 7u) + 1u, start = lv & 0x3FFFFFFu;
            w[j] = (uint32_t)make_leaf(type, (type == 0u ? base0 : (type == 1u ? base1 : base2)) + start, cnt);
        }
    }
    uint4 *dst = rec + (size_t)k * 4;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ void pack_shade(float4 *r, v3 n, const kl_material &mt, float s2z, float s2w, float4 s3) {
    r[0] = make_float4(n.x, n.y, n.z, __int_as_float(mt.type));
    r[1] = make_float4(mt.color[0], mt.color[1], mt.color[2], mt.emissive);
    r[2] = make_float4(mt.roughness, mt.ior, s2z, s2w);
    r[3] = s3;
}
__global__ __launch_bounds__(256) void k_lbvh_pack_prims(const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq, const kl_tri *t, uint32_t nt,
                                                         const uint32_t *ord_s, const uint32_t *ord_q, const uint32_t *ord_t, const kl_material *m,
                                                         uint32_t n_nodes, float4 *rec, float4 *shrec) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns + nq + nt) return;
    const float4 zero4 = make_float4(0, 0, 0, 0);
    float4 *r = rec + 4 * ((size_t)n_nodes + k), *sr = shrec + 4 * (size_t)k;
    if (k < ns) {
        const uint32_t i = ord_s[k];
        const kl_sphere &x = s[i];
        r[0] = make_float4(x.center[0], x.center[1], x.center[2], x.radius);
        r[1] = make_float4(__uint_as_float(i), __uint_as_float(x.material), 0.0f, 0.0f);
        r[2] = zero4; r[3] = zero4;
        pack_shade(sr, V(x.center[0], x.center[1], x.center[2]), m[x.material], x.radius, 0.0f, zero4);
    } else if (k < ns + nq) {
        const uint32_t i = ord_q[k - ns];
        const kl_quad &x = q[i];
        const kl_material &mt = m[x.material];
        r[0] = make_float4(x.v0[0], x.v0[1], x.v0[2], x.v1[0]);
        r[1] = make_float4(x.v1[1], x.v1[2], x.v2[0], x.v2[1]);
        r[2] = make_float4(x.v2[2], x.v3[0], x.v3[1], x.v3[2]);
        r[3] = make_float4(__uint_as_float(ns + i), __uint_as_float(x.material), __uint_as_float(mt.type == KL_TEXTURED ? 1u : 0u), 0.0f);
        const v3 v1 = V(x.v1[0], x.v1[1], x.v1[2]), v2 = V(x.v2[0], x.v2[1], x.v2[2]), v3_ = V(x.v3[0], x.v3[1], x.v3[2]);
        pack_shade(sr, normalize(cross(v2 - v1, v3_ - v1)), mt, 0.0f, 0.0f, zero4);          // RawFrame.glsl:319
    } else {
        const uint32_t i = ord_t[k - ns - nq];
        const kl_tri &x = t[i];
        const kl_material &mt = m[x.material];
        const v3 v0 = V(x.v0[0], x.v0[1], x.v0[2]);
        const v3 e1 = V(x.v1[0], x.v1[1], x.v1[2]) - v0, e2 = V(x.v2[0], x.v2[1], x.v2[2]) - v0;
        const v3 nor = normalize(cross(e1, e2));
        r[0] = make_float4(v0.x, v0.y, v0.z, e1.x);
        r[1] = make_float4(e1.y, e1.z, e2.x, e2.y);
        r[2] = make_float4(e2.z, __uint_as_float(ns + nq + i), __uint_as_float(x.material), __uint_as_float(mt.type == KL_TEXTURED ? 1u : 0u));
        r[3] = zero4;
        pack_shade(sr, nor, mt, x.uv0[0], x.uv0[1], make_float4(x.uv1[0], x.uv1[1], x.uv2[0], x.uv2[1]));
    }
}

// ---- refit: the uploaded tree kept, every box recomputed from moved primitives -------------------------------
// New geometry into the records and shading records of the primitives, each found through the id its record
// carries; ids, materials and uv flags stay. plo / phi = the primitives' boxes in record order.
__global__ __launch_bounds__(256) void k_refit_prims(const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq, const kl_tri *t, uint32_t nt,
                                                     uint32_t n_nodes, float4 *rec, float4 *shrec, float4 *plo, float4 *phi) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns + nq + nt) return;
    float4 *r = rec + 4 * ((size_t)n_nodes + k), *sr = shrec + 4 * (size_t)k;
    const float inf = __builtin_huge_valf();
    float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
    if (k < ns) {
        const uint32_t i = __float_as_uint(r[1].x);
        if (i < ns) {
            const kl_sphere &x = s[i];
            r[0] = make_float4(x.center[0], x.center[1], x.center[2], x.radius);
            sr[0] = make_float4(x.center[0], x.center[1], x.center[2], sr[0].w);
            sr[2].z = x.radius;
            sphere_box(x, lo, hi);
        }
    } else if (k < ns + nq) {
        const float4 tail = r[3];
        const uint32_t i = __float_as_uint(tail.x) - ns;
        if (i < nq) {
            const kl_quad &x = q[i];
            r[0] = make_float4(x.v0[0], x.v0[1], x.v0[2], x.v1[0]);
            r[1] = make_float4(x.v1[1], x.v1[2], x.v2[0], x.v2[1]);
            r[2] = make_float4(x.v2[2], x.v3[0], x.v3[1], x.v3[2]);
            const v3 v1 = V(x.v1[0], x.v1[1], x.v1[2]), v2 = V(x.v2[0], x.v2[1], x.v2[2]), v3_ = V(x.v3[0], x.v3[1], x.v3[2]);
            const v3 nor = normalize(cross(v2 - v1, v3_ - v1));          // RawFrame.glsl:319
            sr[0] = make_float4(nor.x, nor.y, nor.z, sr[0].w);
            quad_box(x, lo, hi);
        }
    } else {
        const float4 q2 = r[2];
        const uint32_t i = __float_as_uint(q2.y) - ns - nq;
        if (i < nt) {
            const kl_tri &x = t[i];
            const v3 v0 = V(x.v0[0], x.v0[1], x.v0[2]);
            const v3 e1 = V(x.v1[0], x.v1[1], x.v1[2]) - v0, e2 = V(x.v2[0], x.v2[1], x.v2[2]) - v0;
            const v3 nor = normalize(cross(e1, e2));
            r[0] = make_float4(v0.x, v0.y, v0.z, e1.x);
            r[1] = make_float4(e1.y, e1.z, e2.x, e2.y);
            r[2] = make_float4(e2.z, q2.y, q2.z, q2.w);
            sr[0] = make_float4(nor.x, nor.y, nor.z, sr[0].w);
            const float4 s2 = sr[2];
            sr[2] = make_float4(s2.x, s2.y, x.uv0[0], x.uv0[1]);
            sr[3] = make_float4(x.uv1[0], x.uv1[1], x.uv2[0], x.uv2[1]);
            tri_box(x, lo, hi);
        }
    }
    plo[k] = make_float4(lo[0], lo[1], lo[2], 0.0f);
    phi[k] = make_float4(hi[0], hi[1], hi[2], 0.0f);
}
__device__ __forceinline__ bool slot_empty(const uint4 &b, const uint4 &c, int k) {          // w6 = b.z (lo.x bytes), w9 = c.y (hi.x bytes)
    return ((b.z >> (8 * k)) & 255u) > ((c.y >> (8 * k)) & 255u);
}
// level of every node below the root, one launch per level from the root down
__global__ __launch_bounds__(256) void k_refit_levels(uint32_t level, uint32_t n_nodes, const uint4 *rec, uint32_t *levels) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes || levels[i] != level) return;
    const uint4 b = rec[4 * (size_t)i + 1], c = rec[4 * (size_t)i + 2], d = rec[4 * (size_t)i + 3];
    const uint32_t ref[4] = { d.x, d.y, d.z, d.w };
    for (int k = 0; k < 4; k++)
        if (!slot_empty(b, c, k) && (int32_t)ref[k] >= 0 && ref[k] < n_nodes) levels[ref[k]] = level + 1u;
}
// boxes of the nodes of one level, from the deepest level up: children's boxes come from earlier launches
__global__ __launch_bounds__(256) void k_refit_nodes(uint32_t level, uint32_t n_nodes, uint32_t n_prims, const uint32_t *levels, uint4 *rec,
                                                     const float4 *plo, const float4 *phi, float4 *nlo, float4 *nhi) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes || levels[i] != level) return;
    uint4 *node = rec + 4 * (size_t)i;
    const uint4 b = node[1], c = node[2], d = node[3];
    const uint32_t ref[4] = { d.x, d.y, d.z, d.w };
    const float inf = __builtin_huge_valf();
    float boxes[4][6];
    int nch = 0;
    for (int k = 0; k < 4; k++) {
        if (slot_empty(b, c, k)) break;                   // empty slots close the node
        float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
        if ((int32_t)ref[k] >= 0) {
            if (ref[k] < n_nodes) {
                const float4 a = nlo[ref[k]], z = nhi[ref[k]];
                lo[0] = a.x; lo[1] = a.y; lo[2] = a.z; hi[0] = z.x; hi[1] = z.y; hi[2] = z.z;
            }
        } else {
            const uint32_t lv = ~ref[k], cnt = ((lv >> 26) & 7u) + 1u, start = (lv & 0x3FFFFFFu) - n_nodes;
            for (uint32_t p = start; p < start + cnt && p < n_prims; p++) {
                const float4 a = plo[p], z = phi[p];
                lo[0] = fminf(lo[0], a.x); lo[1] = fminf(lo[1], a.y); lo[2] = fminf(lo[2], a.z);
                hi[0] = fmaxf(hi[0], z.x); hi[1] = fmaxf(hi[1], z.y); hi[2] = fmaxf(hi[2], z.z);
            }
        }
        for (int a = 0; a < 3; a++) { boxes[nch][a] = lo[a]; boxes[nch][3 + a] = hi[a]; }
        nch++;
    }
    if (nch == 0) return;
    uint32_t w[12];
    node4_boxes(&boxes[0][0], nch, w);
    node[0] = make_uint4(w[0], w[1], w[2], w[3]); node[1] = make_uint4(w[4], w[5], w[6], w[7]); node[2] = make_uint4(w[8], w[9], w[10], w[11]);
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) {
        lo[a] = boxes[0][a]; hi[a] = boxes[0][3 + a];
        for (int k = 1; k < nch; k++) { lo[a] = fminf(lo[a], boxes[k][a]); hi[a] = fmaxf(hi[a], boxes[k][3 + a]); }
    }
    nlo[i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
    nhi[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
}

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) -------------------------------------
// Bottom-up agglomeration over the Morton-ordered primitives: every cluster looks `radius` places to either side
// for the neighbour whose union with it has the smallest area; two clusters that choose each other merge into a
// new binary node; the survivors are compacted, still in Morton order, and the search repeats until one cluster
// is left. Candidate pairs are ranked by (area, distance in the sequence, a hash of the lower position): a total
// order that both ends of a pair evaluate alike, so the best pair of any neighbourhood is always mutual and a run
// of identical boxes still pairs off a third of its members per pass.
__device__ __forceinline__ uint32_t ploc_hash(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }
__global__ __launch_bounds__(256) void k_ploc_init(uint32_t n, const uint32_t *vals, const float4 *blo, const float4 *bhi, int *cid, float4 *clo, float4 *chi) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = vals[i];
    cid[i] = ~(int)i; clo[i] = blo[g]; chi[i] = bhi[g];
}
__global__ __launch_bounds__(256) void k_ploc_nn(uint32_t n, int radius, const float4 *clo, const float4 *chi, uint32_t *nn) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= (int)n) return;
    const float4 a = clo[i], b = chi[i];
    float best = __builtin_huge_valf(); int bj = -1; uint32_t bdist = 0, bh = 0;
    const int j0 = max(0, i - radius), j1 = min((int)n - 1, i + radius);
    for (int j = j0; j <= j1; j++) {
        if (j == i) continue;
        const float4 c = clo[j], d = chi[j];
        const float dx = fmaxf(b.x, d.x) - fminf(a.x, c.x), dy = fmaxf(b.y, d.y) - fminf(a.y, c.y), dz = fmaxf(b.z, d.z) - fminf(a.z, c.z);
        const float area = dx * dy + dy * dz + dz * dx;
        const uint32_t dist = (uint32_t)abs(j - i), h = ploc_hash((uint32_t)min(i, j));
        if (area < best || (area == best && (dist < bdist || (dist == bdist && h < bh)))) { best = area; bj = j; bdist = dist; bh = h; }
    }
    nn[i] = (uint32_t)bj;
}
// mutual pairs become a node at the lower position; cnt = leaves below a node, parents for the walk that orders the leaves
__global__ __launch_bounds__(256) void k_ploc_merge(uint32_t n, const uint32_t *nn, int *cid, float4 *clo, float4 *chi, uint32_t *n_nodes,
                                                    int *left, int *right, float4 *ilo, float4 *ihi, uint32_t *cnt, int *pint, int *pleaf, uint32_t *keep) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = nn[i];
    if (j >= n || nn[j] != i) { keep[i] = 1u; return; }
    if (j < i) { keep[i] = 0u; return; }
    const int l = cid[i], r = cid[j];
    const float4 a = clo[i], b = chi[i], c = clo[j], d = chi[j];
    const float4 lo = make_float4(fminf(a.x, c.x), fminf(a.y, c.y), fminf(a.z, c.z), 0.0f), hi = make_float4(fmaxf(b.x, d.x), fmaxf(b.y, d.y), fmaxf(b.z, d.z), 0.0f);
    const int node = (int)atomicAdd(n_nodes, 1u);
    left[node] = l; right[node] = r; ilo[node] = lo; ihi[node] = hi;
    cnt[node] = (l < 0 ? 1u : cnt[l]) + (r < 0 ? 1u : cnt[r]);
    if (l < 0) pleaf[~l] = node; else pint[l] = node;
    if (r < 0) pleaf[~r] = node; else pint[r] = node;
    pint[node] = -1;
    cid[i] = node; clo[i] = lo; chi[i] = hi;
    keep[i] = 1u;
}
__global__ __launch_bounds__(256) void k_ploc_compact(uint32_t n, const uint32_t *keep, const uint32_t *pos, const int *cid, const float4 *clo, const float4 *chi,
                                                      int *cid2, float4 *clo2, float4 *chi2, uint32_t *n_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (keep[i]) { const uint32_t p = pos[i]; cid2[p] = cid[i]; clo2[p] = clo[i]; chi2[p] = chi[i]; }
    if (i == n - 1) *n_out = pos[i] + keep[i];
}
// place of every leaf in a left-to-right walk of the finished tree = leaves under the left siblings along its way up
__global__ __launch_bounds__(256) void k_ploc_rank(uint32_t n, const int *pleaf, const int *pint, const int *left, const int *right, const uint32_t *cnt,
                                                   uint32_t *rank, uint32_t *maxdepth) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t d = 0;
    if (i < n) {
        uint32_t r = 0;
        int child = ~(int)i, p = pleaf[i];
        while (p >= 0 && d < WALK_CAP) {
            if (right[p] == child) { const int l = left[p]; r += l < 0 ? 1u : cnt[l]; }
            child = p; p = pint[p]; d++;
        }
        rank[i] = r;
    }
    for (int off = 32; off > 0; off >>= 1) d = max(d, (uint32_t)__shfl_xor((int)d, off));
    if ((threadIdx.x & 63u) == 0u) atomicMax(maxdepth, d);
}
__global__ __launch_bounds__(256) void k_ploc_reorder(uint32_t n, const uint32_t *rank, const uint32_t *vals, uint32_t *vals2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) vals2[rank[i]] = vals[i];
}
__global__ __launch_bounds__(256) void k_ploc_remap(uint32_t n_int, const uint32_t *rank, int *left, int *right) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_int) return;
    const int l = left[i], r = right[i];
    if (l < 0) left[i] = ~(int)rank[~l];
    if (r < 0) right[i] = ~(int)rank[~r];
}

struct Buf {
    void *p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <typename T> T *as() const { return (T *)p; }
};

inline uint32_t blocks_for(uint32_t n) { return n ? (n + 255u) / 256u : 1u; }

}  // namespace

LbvhResult build_bvh_gpu(hipStream_t stream, const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq,
                   const kl_tri *t, uint32_t nt, HostBVH &out, float flat_fraction, uint32_t max_flat, std::string &err,
                   const kl_material *m, uint32_t nm, GpuScene *packed) {
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t n = ns + nq + nt;
    identity_order(0, 0, 0, out);
    out.nodes4.clear(); out.root4 = NODE_DONE; out.depth4 = 0;
    out.stats.spheres = ns; out.stats.quads = nq; out.stats.tris = nt;
    if (n == 0) return LBVH_OK;
    hipError_t e = hipSuccess;
#define LB(call) do { e = (call); if (e != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e); return LBVH_HIP_ERROR; } } while (0)
    Buf d_s, d_q, d_t, d_blo, d_bhi, d_scene, d_axes, d_cand, d_ncand, d_keys[2], d_vals[2], d_tmp, d_flat;
    LB(d_s.alloc((size_t)ns * sizeof(kl_sphere))); LB(d_q.alloc((size_t)nq * sizeof(kl_quad))); LB(d_t.alloc((size_t)nt * sizeof(kl_tri)));
    if (ns) LB(hipMemcpyAsync(d_s.p, s, (size_t)ns * sizeof(kl_sphere), hipMemcpyHostToDevice, stream));
    if (nq) LB(hipMemcpyAsync(d_q.p, q, (size_t)nq * sizeof(kl_quad), hipMemcpyHostToDevice, stream));
    if (nt) LB(hipMemcpyAsync(d_t.p, t, (size_t)nt * sizeof(kl_tri), hipMemcpyHostToDevice, stream));
    LB(d_blo.alloc((size_t)n * 16)); LB(d_bhi.alloc((size_t)n * 16));
    LB(d_scene.alloc(6 * 4)); LB(d_axes.alloc(128)); LB(d_cand.alloc(FLAT_CAP * sizeof(FlatCand))); LB(d_ncand.alloc(4));
    for (int k = 0; k < 2; k++) { LB(d_keys[k].alloc((size_t)n * 8)); LB(d_vals[k].alloc((size_t)n * 4)); }
    const uint32_t scene0[6] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u };
    LB(hipMemcpyAsync(d_scene.p, scene0, sizeof scene0, hipMemcpyHostToDevice, stream));
    LB(hipMemsetAsync(d_ncand.p, 0, 4, stream));
    hipLaunchKernelGGL(k_lbvh_bounds, dim3(blocks_for(n)), dim3(256), 0, stream, d_s.as<kl_sphere>(), ns, d_q.as<kl_quad>(), nq, d_t.as<kl_tri>(), nt,
                       d_blo.as<float4>(), d_bhi.as<float4>(), d_scene.as<uint32_t>());
    hipLaunchKernelGGL(k_lbvh_axes, dim3(1), dim3(64), 0, stream, d_scene.as<uint32_t>(), d_axes.as<uint8_t>());
    hipLaunchKernelGGL(k_lbvh_keys, dim3(blocks_for(n)), dim3(256), 0, stream, d_blo.as<float4>(), d_bhi.as<float4>(), n, d_scene.as<uint32_t>(),
                       d_axes.as<uint8_t>(), max_flat ? flat_fraction : 0.0f, d_cand.as<FlatCand>(), d_ncand.as<uint32_t>(), d_keys[0].as<unsigned long long>(), d_vals[0].as<uint32_t>());
    // which candidates are flat: all of them, the max_flat largest, or (more candidates than the list holds) none
    uint32_t ncand = 0;
    LB(hipMemcpyAsync(&ncand, d_ncand.p, 4, hipMemcpyDeviceToHost, stream));
    LB(hipStreamSynchronize(stream));
    std::vector<uint32_t> flat;
    if (ncand > 0 && ncand <= FLAT_CAP) {
        std::vector<FlatCand> cand(ncand);
        LB(hipMemcpy(cand.data(), d_cand.p, ncand * sizeof(FlatCand), hipMemcpyDeviceToHost));
        std::sort(cand.begin(), cand.end(), [](const FlatCand &a, const FlatCand &b) { return a.area != b.area ? a.area > b.area : a.prim < b.prim; });
        if (cand.size() > max_flat) cand.resize(max_flat);
        for (const FlatCand &c : cand) flat.push_back(c.prim);
    }
    const uint32_t nflat = (uint32_t)flat.size();
    if (nflat) {
        LB(d_flat.alloc(nflat * 4));
        LB(hipMemcpyAsync(d_flat.p, flat.data(), nflat * 4, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_lbvh_mark_flat, dim3(blocks_for(nflat)), dim3(256), 0, stream, d_flat.as<uint32_t>(), nflat, d_keys[0].as<unsigned long long>());
    }
    const uint32_t n_tree = n - nflat;
    int ploc_radius = 12;                // 0: the radix tree over the codes instead of clustering
    if (const char *e = std::getenv("KL_PLOC_RADIUS")) ploc_radius = std::max(0, std::min(64, std::atoi(e)));
    // Morton order
    size_t tmp_bytes = 0;
    LB(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys[0].as<unsigned long long>(), d_keys[1].as<unsigned long long>(),
                                          d_vals[0].as<uint32_t>(), d_vals[1].as<uint32_t>(), (int)n, 0, 64, stream));
    size_t scan_bytes = 0;
    LB(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (int)n, stream));
    LB(d_tmp.alloc(std::max(tmp_bytes, scan_bytes)));
    LB(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, d_keys[0].as<unsigned long long>(), d_keys[1].as<unsigned long long>(),
                                          d_vals[0].as<uint32_t>(), d_vals[1].as<uint32_t>(), (int)n, 0, 64, stream));
    const unsigned long long *keys = d_keys[1].as<unsigned long long>();
    const uint32_t *vals = d_vals[1].as<uint32_t>();
    // tree over the first n_tree positions
    Buf d_left, d_right, d_pint, d_pleaf, d_depth, d_maxdepth, d_ilo, d_ihi, d_src, d_alloc, d_nodes4;
    uint32_t count4 = 0, depth4 = 0, maxdepth = 0;
    int32_t root4 = NODE_DONE, root2 = 0;
    const uint32_t n_int = n_tree >= 2 ? n_tree - 1 : 0;
    if (n_int) {
        LB(d_left.alloc((size_t)n_int * 4)); LB(d_right.alloc((size_t)n_int * 4)); LB(d_pint.alloc((size_t)n_int * 4)); LB(d_pleaf.alloc((size_t)n_tree * 4));
        LB(d_maxdepth.alloc(4)); LB(d_ilo.alloc((size_t)n_int * 16)); LB(d_ihi.alloc((size_t)n_int * 16));
        LB(d_src.alloc((size_t)n_int * 4)); LB(d_alloc.alloc(4)); LB(d_nodes4.alloc((size_t)n_int * 64));
        LB(hipMemsetAsync(d_maxdepth.p, 0, 4, stream));
    }
    if (n_int && ploc_radius > 0) {
        // agglomerative build; it also decides the leaf order (left to right in its tree), so it comes before the orders
        Buf d_cid[2], d_clo[2], d_chi[2], d_nn, d_keep, d_pos, d_cnt, d_count, d_rank;
        for (int k = 0; k < 2; k++) { LB(d_cid[k].alloc((size_t)n_tree * 4)); LB(d_clo[k].alloc((size_t)n_tree * 16)); LB(d_chi[k].alloc((size_t)n_tree * 16)); }
        LB(d_nn.alloc((size_t)n_tree * 4)); LB(d_keep.alloc((size_t)n_tree * 4)); LB(d_pos.alloc((size_t)n_tree * 4)); LB(d_cnt.alloc((size_t)n_int * 4));
        LB(d_count.alloc(4)); LB(d_rank.alloc((size_t)n_tree * 4));
        LB(hipMemsetAsync(d_alloc.p, 0, 4, stream));              // node counter during the clustering
        hipLaunchKernelGGL(k_ploc_init, dim3(blocks_for(n_tree)), dim3(256), 0, stream, n_tree, vals, d_blo.as<float4>(), d_bhi.as<float4>(),
                           d_cid[0].as<int>(), d_clo[0].as<float4>(), d_chi[0].as<float4>());
        uint32_t live = n_tree, passes = 0;
        int cur = 0;
        while (live > 1) {
            if (++passes > 8192) { err = "clustering does not converge"; return LBVH_TOO_DEEP; }
            hipLaunchKernelGGL(k_ploc_nn, dim3(blocks_for(live)), dim3(256), 0, stream, live, ploc_radius, d_clo[cur].as<float4>(), d_chi[cur].as<float4>(), d_nn.as<uint32_t>());
            hipLaunchKernelGGL(k_ploc_merge, dim3(blocks_for(live)), dim3(256), 0, stream, live, d_nn.as<uint32_t>(), d_cid[cur].as<int>(), d_clo[cur].as<float4>(),
                               d_chi[cur].as<float4>(), d_alloc.as<uint32_t>(), d_left.as<int>(), d_right.as<int>(), d_ilo.as<float4>(), d_ihi.as<float4>(),
                               d_cnt.as<uint32_t>(), d_pint.as<int>(), d_pleaf.as<int>(), d_keep.as<uint32_t>());
            LB(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, scan_bytes, d_keep.as<uint32_t>(), d_pos.as<uint32_t>(), (int)live, stream));
            hipLaunchKernelGGL(k_ploc_compact, dim3(blocks_for(live)), dim3(256), 0, stream, live, d_keep.as<uint32_t>(), d_pos.as<uint32_t>(), d_cid[cur].as<int>(),
                               d_clo[cur].as<float4>(), d_chi[cur].as<float4>(), d_cid[cur ^ 1].as<int>(), d_clo[cur ^ 1].as<float4>(), d_chi[cur ^ 1].as<float4>(),
                               d_count.as<uint32_t>());
            uint32_t next = 0;
            LB(hipMemcpyAsync(&next, d_count.p, 4, hipMemcpyDeviceToHost, stream));
            LB(hipStreamSynchronize(stream));
            if (next == 0 || next >= live) { err = "clustering pass merged nothing"; return LBVH_HIP_ERROR; }
            live = next; cur ^= 1;
        }
        LB(hipMemcpyAsync(&root2, d_cid[cur].p, 4, hipMemcpyDeviceToHost, stream));
        hipLaunchKernelGGL(k_ploc_rank, dim3(blocks_for(n_tree)), dim3(256), 0, stream, n_tree, d_pleaf.as<int>(), d_pint.as<int>(), d_left.as<int>(), d_right.as<int>(),
                           d_cnt.as<uint32_t>(), d_rank.as<uint32_t>(), d_maxdepth.as<uint32_t>());
        LB(hipMemcpyAsync(&maxdepth, d_maxdepth.p, 4, hipMemcpyDeviceToHost, stream));
        LB(hipStreamSynchronize(stream));
        if (maxdepth >= WALK_CAP) { err = "clustered tree deeper than the builder's walk bound"; return LBVH_TOO_DEEP; }
        if (root2 < 0 || (uint32_t)root2 >= n_int) { err = "clustering left no root"; return LBVH_HIP_ERROR; }
        // the sorted sequence rearranged into the tree's leaf order (the flat tail stays where it is)
        LB(hipMemcpyAsync(d_vals[0].p, d_vals[1].p, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
        hipLaunchKernelGGL(k_ploc_reorder, dim3(blocks_for(n_tree)), dim3(256), 0, stream, n_tree, d_rank.as<uint32_t>(), d_vals[1].as<uint32_t>(), d_vals[0].as<uint32_t>());
        hipLaunchKernelGGL(k_ploc_remap, dim3(blocks_for(n_int)), dim3(256), 0, stream, n_int, d_rank.as<uint32_t>(), d_left.as<int>(), d_right.as<int>());
        LB(hipStreamSynchronize(stream));            // the clustering's buffers go out of scope here
        vals = d_vals[0].as<uint32_t>();
    }
    // per-type leaf orders; d_keys[0] is free again and serves as scratch
    Buf d_f[3], d_r[3], d_ord[3], d_leafref;
    for (int k = 0; k < 3; k++) { LB(d_f[k].alloc((size_t)n * 4)); LB(d_r[k].alloc((size_t)n * 4)); }
    LB(d_ord[0].alloc((size_t)ns * 4)); LB(d_ord[1].alloc((size_t)nq * 4)); LB(d_ord[2].alloc((size_t)nt * 4));
    LB(d_leafref.alloc((size_t)n * 4));
    hipLaunchKernelGGL(k_lbvh_typeflags, dim3(blocks_for(n)), dim3(256), 0, stream, vals, n, ns, nq, d_f[0].as<uint32_t>(), d_f[1].as<uint32_t>(), d_f[2].as<uint32_t>());
    for (int k = 0; k < 3; k++)
        LB(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, scan_bytes, d_f[k].as<uint32_t>(), d_r[k].as<uint32_t>(), (int)n, stream));
    hipLaunchKernelGGL(k_lbvh_orders, dim3(blocks_for(n)), dim3(256), 0, stream, vals, n, ns, nq, d_r[0].as<uint32_t>(), d_r[1].as<uint32_t>(), d_r[2].as<uint32_t>(),
                       d_ord[0].as<uint32_t>(), d_ord[1].as<uint32_t>(), d_ord[2].as<uint32_t>(), d_leafref.as<int>());
    if (n_tree == 1) {
        LB(hipMemcpyAsync(&root4, d_leafref.p, 4, hipMemcpyDeviceToHost, stream));
        LB(hipStreamSynchronize(stream));
    } else if (n_tree >= 2) {
        if (ploc_radius <= 0) {
            // binary radix tree over the sorted codes, boxes by one refit launch per depth level
            LB(d_depth.alloc((size_t)n_int * 4));
            hipLaunchKernelGGL(k_lbvh_hierarchy, dim3(blocks_for(n_int)), dim3(256), 0, stream, keys, (int)n_tree, d_left.as<int>(), d_right.as<int>(),
                               d_pint.as<int>(), d_pleaf.as<int>());
            hipLaunchKernelGGL(k_lbvh_depth, dim3(blocks_for(n_int)), dim3(256), 0, stream, d_pint.as<int>(), n_int, d_depth.as<uint32_t>(), d_maxdepth.as<uint32_t>());
            LB(hipMemcpyAsync(&maxdepth, d_maxdepth.p, 4, hipMemcpyDeviceToHost, stream));
            LB(hipStreamSynchronize(stream));
            if (maxdepth >= WALK_CAP) { err = "radix tree deeper than the builder's walk bound"; return LBVH_TOO_DEEP; }
            for (uint32_t level = maxdepth + 1; level-- > 0;)
                hipLaunchKernelGGL(k_lbvh_refit, dim3(blocks_for(n_int)), dim3(256), 0, stream, level, d_depth.as<uint32_t>(), n_int, d_left.as<int>(), d_right.as<int>(),
                                   vals, d_blo.as<float4>(), d_bhi.as<float4>(), d_ilo.as<float4>(), d_ihi.as<float4>());
        }
        // collapse, level by level from the root (4-wide node 0 = the binary root)
        const uint32_t one = 1;
        LB(hipMemcpyAsync(d_src.p, &root2, 4, hipMemcpyHostToDevice, stream));
        LB(hipMemcpyAsync(d_alloc.p, &one, 4, hipMemcpyHostToDevice, stream));
        uint32_t begin = 0, end = 1;
        while (begin < end) {
            depth4++;
            if (depth4 > (uint32_t)BVH_MAX_DEPTH) { LB(hipStreamSynchronize(stream)); err = "tree deeper than the traversal stack allows"; return LBVH_TOO_DEEP; }
            hipLaunchKernelGGL(k_lbvh_collapse, dim3(blocks_for(end - begin)), dim3(256), 0, stream, begin, end, d_src.as<int>(), d_alloc.as<uint32_t>(),
                               d_left.as<int>(), d_right.as<int>(), d_leafref.as<int>(), vals, d_blo.as<float4>(), d_bhi.as<float4>(),
                               d_ilo.as<float4>(), d_ihi.as<float4>(), d_nodes4.as<uint32_t>());
            uint32_t total = 0;
            LB(hipMemcpyAsync(&total, d_alloc.p, 4, hipMemcpyDeviceToHost, stream));
            LB(hipStreamSynchronize(stream));
            if (total > n_int) { err = "4-wide node count exceeds the binary node count"; return LBVH_HIP_ERROR; }
            begin = end; end = total;
        }
        count4 = end; root4 = 0;
    }
    LB(hipGetLastError());
    if (packed && m && nm) {
        // records and shading records written here, from the primitives and orders already on the device
        Buf d_m, d_rec, d_shrec;
        const size_t n_rec = std::max<size_t>((size_t)count4 + n, 1), n_shrec = std::max<size_t>(n, 1);
        LB(d_m.alloc((size_t)nm * sizeof(kl_material)));
        LB(hipMemcpyAsync(d_m.p, m, (size_t)nm * sizeof(kl_material), hipMemcpyHostToDevice, stream));
        LB(d_rec.alloc(n_rec * 64)); LB(d_shrec.alloc(n_shrec * 64));
        if (count4)
            hipLaunchKernelGGL(k_lbvh_pack_nodes, dim3(blocks_for(count4)), dim3(256), 0, stream, d_nodes4.as<uint32_t>(), count4,
                               count4, count4 + ns, count4 + ns + nq, d_rec.as<uint4>());
        hipLaunchKernelGGL(k_lbvh_pack_prims, dim3(blocks_for(n)), dim3(256), 0, stream, d_s.as<kl_sphere>(), ns, d_q.as<kl_quad>(), nq, d_t.as<kl_tri>(), nt,
                           d_ord[0].as<uint32_t>(), d_ord[1].as<uint32_t>(), d_ord[2].as<uint32_t>(), d_m.as<kl_material>(), count4,
                           d_rec.as<float4>(), d_shrec.as<float4>());
        // the flat quads close the quad order: their indices go back for the caller's quad table
        uint32_t nfq = 0;
        for (uint32_t g : flat) nfq += (g >= ns && g < ns + nq) ? 1u : 0u;
        packed->flat_quads.resize(nfq);
        if (nfq) LB(hipMemcpyAsync(packed->flat_quads.data(), d_ord[1].as<uint32_t>() + (nq - nfq), (size_t)nfq * 4, hipMemcpyDeviceToHost, stream));
        LB(hipStreamSynchronize(stream));
        LB(hipGetLastError());
        packed->d_rec = d_rec.p; packed->n_rec = n_rec; d_rec.p = nullptr;          // ownership passes to the caller
        packed->d_shrec = d_shrec.p; packed->n_shrec = n_shrec; d_shrec.p = nullptr;
    } else {
        // results to the host
        out.nodes4.resize((size_t)count4 * 16);
        if (count4) LB(hipMemcpyAsync(out.nodes4.data(), d_nodes4.p, (size_t)count4 * 64, hipMemcpyDeviceToHost, stream));
        out.sph_order.resize(ns); out.quad_order.resize(nq); out.tri_order.resize(nt);
        if (ns) LB(hipMemcpyAsync(out.sph_order.data(), d_ord[0].p, (size_t)ns * 4, hipMemcpyDeviceToHost, stream));
        if (nq) LB(hipMemcpyAsync(out.quad_order.data(), d_ord[1].p, (size_t)nq * 4, hipMemcpyDeviceToHost, stream));
        if (nt) LB(hipMemcpyAsync(out.tri_order.data(), d_ord[2].p, (size_t)nt * 4, hipMemcpyDeviceToHost, stream));
        LB(hipStreamSynchronize(stream));
    }
#undef LB
    out.root4 = root4; out.depth4 = depth4;
    // the flat primitives sorted last, so they close each order array
    uint32_t fc[3] = { 0, 0, 0 };
    for (uint32_t g : flat) fc[g < ns ? PRIM_SPHERE : (g < ns + nq ? PRIM_QUAD : PRIM_TRI)]++;
    const uint32_t cnt[3] = { ns, nq, nt };
    for (int k = 0; k < 3; k++) { out.flat_count[k] = fc[k]; out.flat_start[k] = cnt[k] - fc[k]; }
    out.stats.nodes = count4; out.stats.leaves = n_tree; out.stats.max_depth = depth4; out.stats.max_leaf_size = n_tree ? (n_tree > 2 ? 2u : 1u) : 0u;
    out.stats.sah_cost = 0.0f;
    out.stats.node_bytes = (uint64_t)count4 * 64u;
    out.stats.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return LBVH_OK;
}

LbvhResult refit_scene_gpu(hipStream_t stream, float4 *rec, float4 *shrec, uint32_t n_nodes, int32_t root, uint32_t levels,
                           const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq, const kl_tri *t, uint32_t nt, std::string &err) {
    const uint32_t n = ns + nq + nt;
    if (n == 0) return LBVH_OK;
    hipError_t e = hipSuccess;
#define LB(call) do { e = (call); if (e != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e); return LBVH_HIP_ERROR; } } while (0)
    Buf d_s, d_q, d_t, d_plo, d_phi, d_nlo, d_nhi, d_lev;
    LB(d_s.alloc((size_t)ns * sizeof(kl_sphere))); LB(d_q.alloc((size_t)nq * sizeof(kl_quad))); LB(d_t.alloc((size_t)nt * sizeof(kl_tri)));
    if (ns) LB(hipMemcpyAsync(d_s.p, s, (size_t)ns * sizeof(kl_sphere), hipMemcpyHostToDevice, stream));
    if (nq) LB(hipMemcpyAsync(d_q.p, q, (size_t)nq * sizeof(kl_quad), hipMemcpyHostToDevice, stream));
    if (nt) LB(hipMemcpyAsync(d_t.p, t, (size_t)nt * sizeof(kl_tri), hipMemcpyHostToDevice, stream));
    LB(d_plo.alloc((size_t)n * 16)); LB(d_phi.alloc((size_t)n * 16));
    hipLaunchKernelGGL(k_refit_prims, dim3(blocks_for(n)), dim3(256), 0, stream, d_s.as<kl_sphere>(), ns, d_q.as<kl_quad>(), nq, d_t.as<kl_tri>(), nt,
                       n_nodes, rec, shrec, d_plo.as<float4>(), d_phi.as<float4>());
    if (n_nodes && root >= 0 && (uint32_t)root < n_nodes) {
        LB(d_nlo.alloc((size_t)n_nodes * 16)); LB(d_nhi.alloc((size_t)n_nodes * 16)); LB(d_lev.alloc((size_t)n_nodes * 4));
        LB(hipMemsetAsync(d_lev.p, 0xFF, (size_t)n_nodes * 4, stream));
        LB(hipMemsetAsync(d_lev.as<uint32_t>() + root, 0, 4, stream));
        for (uint32_t l = 0; l + 1 < levels; l++)
            hipLaunchKernelGGL(k_refit_levels, dim3(blocks_for(n_nodes)), dim3(256), 0, stream, l, n_nodes, (const uint4 *)rec, d_lev.as<uint32_t>());
        for (uint32_t l = levels; l-- > 0;)
            hipLaunchKernelGGL(k_refit_nodes, dim3(blocks_for(n_nodes)), dim3(256), 0, stream, l, n_nodes, n, d_lev.as<uint32_t>(), (uint4 *)rec,
                               d_plo.as<float4>(), d_phi.as<float4>(), d_nlo.as<float4>(), d_nhi.as<float4>());
    }
    LB(hipStreamSynchronize(stream));
    LB(hipGetLastError());
#undef LB
    return LBVH_OK;
}

}  // namespace kl
