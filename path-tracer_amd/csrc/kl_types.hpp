// Device-side data layout shared by the kernels (kl_kernels.hip), the BVH
// builder (kl_bvh.cpp) and the C-ABI implementation (kl_api.cpp).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace kl {

// ---- records ------------------------------------------------------------------------
// Everything the traversal touches lives in ONE array of 64-byte records (four float4 each):
// first the BVH nodes, then the spheres, the quads and the triangles in leaf order. A record is
// named by its index in that array, so its address is base + (index << 6): 32-bit arithmetic, one
// scalar base. The traversal loop fetches "the next record" with one group of loads whether the
// lane stands on an inner node or on a primitive.
//   node:   four children with 8-bit boxes, see "4-wide node" below
//   sphere: q0 = (c.xyz, r)          q1 = (id bits, material bits, 0, 0)
//   quad:   q0 = (v0.xyz, v1.x)      q1 = (v1.yz, v2.xy)   q2 = (v2.z, v3.xyz)   q3 = (id bits, material bits, uv flag, 0)
//   tri:    q0 = (v0.xyz, e1.x)      q1 = (e1.yz, e2.xy)   q2 = (e2.z, id bits, material bits, uv flag)   q3 = 0
// id = declaration-order primitive id (spheres, then quads, then triangles). uv flag = 1 if shading the primitive
// reads the hit's (u, v), i.e. its material is TEXTURED; only then does a hit store them.
//
// Shading reads a second array, one 64-byte SHADING record per primitive (index = record index - base[0]),
// holding everything the shade stage needs about the surface it hit, material included, so that one fetch
// follows the hit instead of a chain record -> material -> triangle uvs:
//   s0 = (n.xyz, material type bits)   n = the geometric normal in the shader's arithmetic: quads
//        normalize(cross(v2 - v1, v3 - v1)) (RawFrame.glsl:319), triangles normalize(cross(e1, e2));
//        spheres: the centre instead (their normal depends on the hit point)
//   s1 = (colour rgb, emissive strength)
//   s2 = (roughness, ior, uv0.xy)      spheres: (roughness, ior, radius, 0)
//   s3 = (uv1.xy, uv2.xy)              triangles only
// child >= 0: record index of an inner node. child < 0: leaf, ~child =
//   (type << 29) | ((count-1) << 26) | start: a run of `count` (1..8) primitives of one type whose
//   first record is `start`.
// A hit names its primitive by (type << 30) | record index.
enum : uint32_t { PRIM_SPHERE = 0, PRIM_QUAD = 1, PRIM_TRI = 2 };
constexpr uint32_t REF_MISS = 0xFFFFFFFFu;
__host__ __device__ inline uint32_t make_ref(uint32_t type, uint32_t record) { return (type << 30) | record; }
__host__ __device__ inline uint32_t ref_type(uint32_t r) { return r >> 30; }
__host__ __device__ inline uint32_t ref_record(uint32_t r) { return r & 0x3FFFFFFFu; }

constexpr int NODE_DONE = 0x7FFFFFFF;       // "stack empty" marker, never a valid node index
constexpr uint32_t LEAF_MAX = 8;
__host__ __device__ inline int make_leaf(uint32_t type, uint32_t start, uint32_t count) {
    return ~(int)((type << 29) | ((count - 1u) << 26) | start);
}
constexpr uint32_t MAX_RECORDS = 1u << 26;  // nodes + primitives
// 4-wide node, 16 words:
//   w0..w2 = node box lo (float)   w3..w5 = grid step of each axis (float, a power of two)
//   w6..w8 = child lo.x / lo.y / lo.z, one byte per child (byte k = child k)   w9..w11 = child hi.x / hi.y / hi.z
//   w12..w15 = child references (as in the binary node)
//   child box = lo + byte * step, rounded outwards at build time. An empty slot holds an inverted box
//   (lo bytes 255, hi bytes 0), which no ray enters, and repeats the reference of child 0 so that it is a
//   valid reference whatever rounding does to the box test.
// Words w0..w11 of a 4-wide node from its children's boxes: `boxes` = nch x 6 floats (lo.xyz, hi.xyz), nch = 1..4;
// the other slots become empty. Host and device builders and the refit all quantise through this one function.
__host__ __device__ inline void node4_boxes(const float *boxes, int nch, uint32_t w[12]) {
    float nlo[3], nhi[3];
    for (int a = 0; a < 3; a++) {
        nlo[a] = boxes[a]; nhi[a] = boxes[3 + a];
        for (int k = 1; k < nch; k++) { nlo[a] = fminf(nlo[a], boxes[6 * k + a]); nhi[a] = fmaxf(nhi[a], boxes[6 * k + 3 + a]); }
    }
    for (int k = 0; k < 12; k++) w[k] = 0u;
    double scale[3];
    for (int a = 0; a < 3; a++) {
        memcpy(&w[a], &nlo[a], 4);
        const double ext = (double)nhi[a] - (double)nlo[a];
        int e = ext > 0.0 ? (int)ceil(log2(ext / 255.0)) : -100;
        if (e < -100) e = -100;
        while (ldexp(255.0, e) < ext) e++;               // guard the rounding of log2
        scale[a] = ldexp(1.0, e);
        const float step = (float)scale[a];              // a power of two: exact
        memcpy(&w[3 + a], &step, 4);
    }
    for (int k = 0; k < 4; k++) {
        uint32_t qlo[3] = { 255u, 255u, 255u }, qhi[3] = { 0u, 0u, 0u };   // empty slot: inverted box, never entered
        if (k < nch)
            for (int a = 0; a < 3; a++) {
                const double l = floor(((double)boxes[6 * k + a] - (double)nlo[a]) / scale[a]);
                const double h = ceil(((double)boxes[6 * k + 3 + a] - (double)nlo[a]) / scale[a]);
                qlo[a] = (uint32_t)fmin(255.0, fmax(0.0, l));
                qhi[a] = (uint32_t)fmin(255.0, fmax(0.0, h));
            }
        for (int a = 0; a < 3; a++) { w[6 + a] |= qlo[a] << (8 * k); w[9 + a] |= qhi[a] << (8 * k); }
    }
}
#ifndef KL_BVH_MAX_DEPTH
#define KL_BVH_MAX_DEPTH 32
#endif
constexpr int BVH_MAX_DEPTH = KL_BVH_MAX_DEPTH;   // builder guarantee == traversal stack entries per lane

struct SceneDev {
    const float4 *rec;       // the record array
    int32_t root;            // child reference of the whole tree (node index or leaf), NODE_DONE if the tree is empty
    uint32_t n_nodes;
    uint32_t base[3];        // first record of each primitive type (indexed by PRIM_*)
    uint32_t count[3];       // primitives of each type
    // "flat" primitives: the few whose box spans a large part of the scene (walls, floor). They are kept
    // out of the tree, sit at the end of their type's records, and are tested for every ray by a coherent
    // pass (k_intersect_flat) whose result seeds the traversal's closest distance.
    uint32_t flat_start[3], flat_count[3];   // record index of the first flat primitive of a type, how many
    // per flat quad, the ray-independent part of the quad test evaluated once at upload (5 float4 each):
    //   F0 = (v0.xyz, nor[id])  F1 = (nor.xyz, k2)  F2 = (ka.xy, kb.xy)  F3 = (kg.xy, idu | idv << 2, 0)  F4 = (id bits, uv flag, 0, 0)
    const float4 *flatq;
    const float4 *shrec;     // shading records, 4 float4 per primitive
    // textures, RGBA8, rows as uploaded
    const uchar4 *tex[2];
    uint32_t tex_w[2], tex_h[2];
    // optional float environment (rgb in xyz), used instead of tex[KL_TEX_HDRI] when not null
    const float4 *envf;
    uint32_t envf_w, envf_h;
};

// ---- per-batch uniforms ---------------------------------------------------------
struct FrameDev {
    float pos[3], pc[3], pu[3], pv[3];  // camera position and projection frame (RawFrame.glsl:418-420)
    uint32_t width, height;             // full image
    uint32_t npl;                       // pixels owned by this context (local rows x width)
    uint32_t rank, world, band_rows;    // row-band partition
    uint32_t mode;
    uint32_t first_frame, nslots;       // slots in flight: spp2 consecutive slots per frame, frames from first_frame on
    uint32_t spp, spp2;                 // sub-pixel grid of RawFrame.glsl:37 (SPP) and its square: slot = frame slot * spp2 + (x * spp + y)
    uint32_t max_bounces, max_frames;
    float eps, max_dist, ao_length, hdri_strength;
    uint32_t nseg, segcap;              // queue segmentation, see RayQueue
    uint32_t all_uv;                    // 1: every hit stores its (u, v), not only those whose shading reads them (caller rays)
    // division by npl, width and band_rows as multiply-high + add + shift (exact for dividends below 2^31)
    uint32_t div_npl[2], div_width[2], div_band[2], div_spp2[2], div_spp[2];
};
// n / d for n < 2^31 with {m, s} = fastdiv_make(d): s = ceil(log2 d), m = floor(2^(32+s) / d) + 1 - 2^32
__host__ __device__ inline void fastdiv_make(uint32_t d, uint32_t out[2]) {
    uint32_t s = 0;
    while ((1ull << s) < d) s++;
    out[0] = (uint32_t)((((unsigned long long)1 << (32 + s)) / d) + 1ull - (1ull << 32));
    out[1] = s;
}
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, const uint32_t ms[2]) { return (__umulhi(n, ms[0]) + n) >> ms[1]; }
#endif

// ---- wavefront queues -------------------------------------------------------------
// One ray = 48 bytes over three arrays so that every access is a full-width vector access:
//   od[i] = (o.x, o.y, o.z, d.x)   dt[i] = (d.y, d.z, t, ref bits)   tp[i] = (throughput rgb, path id bits)
// (t, ref) is the ray's closest hit so far. The kernel that makes a ray writes the hit among the flat
// primitives there (or t = max_dist, ref = REF_MISS); the intersect stage overwrites the two words only if it
// finds something closer; shade reads them. The hit's (u, v) live in uv[i], 8 bytes that are written and read
// only for primitives whose material is textured (the records' uv flag) — most rays never touch them.
// path id = slot * npl + local pixel; it also indexes the raw-frame buffer.
//
// A queue is `nseg` independent segments of `segcap` entries: ray i of segment s lives at index
// s * segcap + i and each segment has its own fill count and its own dequeue head, so the tail
// atomics of compaction spread over nseg words instead of one. generate deals 64-path groups
// round-robin over the segments; a survivor always stays in its segment, hence a segment never
// holds more than it started with and segcap = ceil(paths / nseg) is enough at every bounce.
// The per-segment counters are CTR_STRIDE words apart (256 bytes): memory-side atomics on words of one
// cache line serialise as if they hit one word, so neighbouring segments must not share a line.
constexpr uint32_t CTR_STRIDE = 64;
struct RayQueue {
    float4 *od;
    float4 *dt;
    float4 *tp;
    float2 *uv;
};
struct Counters {            // 64-bit device counters, see kl_counters
    unsigned long long rays, nodes, sphere_tests, quad_tests, tri_tests, lane_active, lane_total, pad;
};

}  // namespace kl
