// Wavefront path-tracing kernels for gfx950 (wave64).
//
//   generate  : camera rays for every (frame slot, pixel) of a batch, each with its hit
//               among the flat primitives                                   -> ray queue 0
//   intersect : closest hit per ray, written into the ray's queue entry when the
//               tree holds something closer; persistent wave64 workgroups pull ray
//               ranges from a global counter and refill idle lanes; SAH-BVH
//               traversal with a per-lane stack in LDS (or a flat loop for
//               scenes of a few primitives)
//   shade     : surface reconstruction + BSDF / environment / texture; the
//               surviving rays are compacted with ballot + popcount into the
//               next queue, their flat-primitive hits with them (or left in
//               place for the separate compact kernel)
//   compact   : stand-alone stream compaction of a shaded queue
//   accumulate: running mean (or sum) over the frames of the batch, in order
//
// Arithmetic that restates the reference shader (resources/Shaders/RawFrame.glsl,
// cited as RF:line) keeps its operation order and is compiled unfused.
#include <mutex>

#include "kl_kernels.hpp"
#include "kl_math.hpp"

namespace kl {

// Queue traffic is streamed: every entry is written once (generate, shade) and read once or twice. Shade marks its queue
// loads and stores non-temporal, which leaves more of L2 to the records the traversal re-reads (+1 % on the bench workload);
// the traversal's own refill loads stay plain (non-temporal there measured -1 %).
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load(const float4 *p) { const nt_f4 v = __builtin_nontemporal_load((const nt_f4 *)p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void nt_store(float4 *p, float4 v) { nt_f4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, (nt_f4 *)p); }

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t mask_rank(unsigned long long m) {   // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// one atomic per wave: reserves popcount(pred) consecutive slots, returns this lane's slot
__device__ __forceinline__ uint32_t wave_reserve(uint32_t *counter, bool pred) {
    unsigned long long m = __ballot(pred);
    if (m == 0ull) return 0u;
    uint32_t n = (uint32_t)__popcll(m);
    uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
    uint32_t base = 0u;
    if (lane_id() == leader) base = atomicAdd(counter, n);
    base = __shfl(base, (int)leader);
    return base + mask_rank(m);
}

__device__ __forceinline__ v3 ldv(const float *p) { return V(p[0], p[1], p[2]); }

// local pixel -> global pixel for the row-band partition
__device__ __forceinline__ void local_to_pixel(const FrameDev &f, uint32_t li, uint32_t &x, uint32_t &y) {
    uint32_t ly = fastdiv(li, f.div_width);
    x = li - ly * f.width;
    uint32_t band = fastdiv(ly, f.div_band);
    y = (band * f.world + f.rank) * f.band_rows + (ly - band * f.band_rows);
}

// ------------------------------------------------------------------ intersectors
// RF:196-216; the field the shader calls Diameter is used as a radius
__device__ __forceinline__ bool sphere_intersect(v3 o, v3 d, float4 s, float eps, float &t) {
    v3 oc = o - V(s.x, s.y, s.z);
    float b = dot(oc, d);
    float delta = b * b - dot(oc, oc) + s.w * s.w;
    if (delta < 0.0f) return false;
    float sq = sqrtf(delta);
    if (-b - sq > eps) { t = -b - sq; return true; }
    else if (-b + sq > eps) { t = -b + sq; return true; }
    return false;
}
// RF:218-269 inverse-bilinear patch; uv out = (v, 1-u). May report "hit" with a NaN t,
// which the caller's comparison rejects exactly as RF:316 does.
__device__ __forceinline__ bool quad_intersect(v3 o, v3 d, v3 v0, v3 v1, v3 v2, v3 v3_, float best, float &t_out, float &uo, float &vo) {
    v3 a = v1 - v0, b = v3_ - v0, c = v2 - v0, p = o - v0;
    v3 nor = cross(a, b);
    float t = -dot(p, nor) / dot(d, nor);
    if (t < 0.0f) return false;
    if (!(t > 0.001f && t <= best)) return false;    // would fail RF:316 whatever (u, v) turn out to be
    v3 pos = p + d * t;
    v3 mor = V(fabsf(nor.x), fabsf(nor.y), fabsf(nor.z));
    int id = (mor.x > mor.y && mor.x > mor.z) ? 0 : ((mor.y > mor.z) ? 1 : 2);
    int idu = id == 0 ? 1 : (id == 1 ? 2 : 0);      // Quad_Face[id], RF:44
    int idv = id == 0 ? 2 : (id == 1 ? 0 : 1);      // Quad_Face[id+1]
    float kpx = comp(pos, idu), kpy = comp(pos, idv);
    float kax = comp(a, idu), kay = comp(a, idv);
    float kbx = comp(b, idu), kby = comp(b, idv);
    float kcx = comp(c, idu), kcy = comp(c, idv);
    float kgx = kcx - kbx - kax, kgy = kcy - kby - kay;
    float k0 = cross2d(kpx, kpy, kbx, kby);
    float k2 = cross2d(kcx - kbx, kcy - kby, kax, kay);
    float k1 = cross2d(kpx, kpy, kgx, kgy) - comp(nor, id);
    float u, v;
    if (fabsf(k2) < 0.00001f) {
        v = -k0 / k1;
        u = cross2d(kpx, kpy, kax, kay) / k1;
    } else {
        float w = k1 * k1 - 4.0f * k0 * k2;
        if (w < 0.0f) return false;
        w = sqrtf(w);
        float ik2 = 1.0f / (2.0f * k2);
        v = (-k1 - w) * ik2;
        if (v < 0.0f || v > 1.0f) v = (-k1 + w) * ik2;
        u = (kpx - kax * v) / (kbx + kgx * v);
    }
    if (u < 0.0f || u > 1.0f || v < 0.0f || v > 1.0f) return false;
    t_out = t; uo = v; vo = 1.0f - u;
    return true;
}
// two-sided Moeller-Trumbore on (v0, e1, e2) with explicit fused multiply-adds (no reference counterpart)
__device__ __forceinline__ bool tri_intersect(v3 o, v3 d, v3 v0, v3 e1, v3 e2, float &t, float &bu, float &bv) {
    v3 p = fcross(d, e2);
    float det = fdot(e1, p);
    float inv = 1.0f / det;
    v3 s = o - v0;
    float u = fdot(s, p) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    v3 q = fcross(s, e1);
    float v = fdot(d, q) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    t = fdot(e2, q) * inv; bu = u; bv = v;
    return true;
}

// uvf: bit 0 = the primitive's uv flag (shading reads u, v); bit 1 = this hit was taken by accept() (as opposed to
// handed over from a queue entry)
struct Hit { float t; uint32_t ref; float u, v; uint32_t id; uint32_t uvf; };
constexpr uint32_t HIT_UV = 1u, HIT_TAKEN = 2u;

// declaration-order id of a primitive reference (read from its record)
__device__ __forceinline__ uint32_t ref_id(const SceneDev &sc, uint32_t ref) {
    const float4 *p = sc.rec + 4u * ref_record(ref);
    const uint32_t type = ref_type(ref);
    return type == PRIM_SPHERE ? __float_as_uint(p[1].x) : (type == PRIM_QUAD ? __float_as_uint(p[3].x) : __float_as_uint(p[2].y));
}
// h.id of a hit taken over from a queue entry: not known, and fetched only if an exact tie asks for it
constexpr uint32_t ID_UNKNOWN = 0xFFFFFFFEu;

// acceptance rule of RF:302,316 (`t < best && t > 0.001`); on exactly equal t the lower
// declaration-order id wins, which is what the shader's in-order loop with a strict `<` does.
// `lazy`: h.id may be ID_UNKNOWN (h came from a queue entry, which has no room for the id).
__device__ __forceinline__ void accept(Hit &h, float t, uint32_t ref, uint32_t id, float u, float v, uint32_t uvf, const SceneDev *lazy = nullptr) {
    if (!(t > 0.001f)) return;
    bool take = t < h.t;
    if (t == h.t) {
        if (lazy && h.id == ID_UNKNOWN) h.id = ref_id(*lazy, h.ref);
        take = id < h.id;
    }
    if (take) { h.t = t; h.ref = ref; h.id = id; h.u = u; h.v = v; h.uvf = uvf | HIT_TAKEN; }
}
// primitive tests on the 64-byte records of kl_types.hpp
__device__ __forceinline__ void test_sphere(float4 r0, float4 r1, uint32_t i, v3 o, v3 d, float eps, Hit &h, const SceneDev *lazy = nullptr) {
    float t;
    if (sphere_intersect(o, d, r0, eps, t)) accept(h, t, make_ref(PRIM_SPHERE, i), __float_as_uint(r1.x), 0.0f, 0.0f, 0u, lazy);
}
__device__ __forceinline__ void test_quad(float4 a, float4 b, float4 c, float4 e, uint32_t i, v3 o, v3 d, Hit &h, const SceneDev *lazy = nullptr) {
    // the plane distance is the first thing RF:225 computes; a candidate that cannot pass the
    // acceptance test of RF:316 is dropped before the inverse-bilinear solve (same outcome)
    float t, u, v;
    if (quad_intersect(o, d, V(a.x, a.y, a.z), V(a.w, b.x, b.y), V(b.z, b.w, c.x), V(c.y, c.z, c.w), h.t, t, u, v))
        accept(h, t, make_ref(PRIM_QUAD, i), __float_as_uint(e.x), u, v, __float_as_uint(e.z), lazy);
}
__device__ __forceinline__ void test_tri(float4 a, float4 b, float4 c, uint32_t i, v3 o, v3 d, Hit &h, const SceneDev *lazy = nullptr) {
    float t, u, v;
    if (tri_intersect(o, d, V(a.x, a.y, a.z), V(a.w, b.x, b.y), V(b.z, b.w, c.x), t, u, v))
        accept(h, t, make_ref(PRIM_TRI, i), __float_as_uint(c.y), u, v, __float_as_uint(c.w), lazy);
}
// A 16-byte load from an address that is the same in every lane, through the constant address space: the
// compiler then uses the scalar cache and scalar registers instead of 64 identical vector loads. Only for data
// that no kernel writes (scene records and tables, uploaded before any launch).
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 uload(const float4 *p) {
    const vf4 v = *(const __attribute__((address_space(4))) vf4 *)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
// all primitives of one type in record range [first, first + n): uniform loads (flat loop, flat pass)
__device__ __forceinline__ void test_range(const SceneDev &sc, uint32_t type, uint32_t first, uint32_t n, v3 o, v3 d, float eps, Hit &h) {
    for (uint32_t r = first; r < first + n; r++) {
        const float4 *p = sc.rec + 4u * r;
        if (type == PRIM_SPHERE) test_sphere(uload(p), uload(p + 1), r, o, d, eps, h);
        else if (type == PRIM_QUAD) test_quad(uload(p), uload(p + 1), uload(p + 2), uload(p + 3), r, o, d, h);
        else test_tri(uload(p), uload(p + 1), uload(p + 2), r, o, d, h);
    }
}
__device__ __forceinline__ bool ray_finite(v3 o, v3 d) {
    const float inf = __builtin_huge_valf();
    return fabsf(o.x) < inf && fabsf(o.y) < inf && fabsf(o.z) < inf && fabsf(d.x) < inf && fabsf(d.y) < inf && fabsf(d.z) < inf;
}

// ------------------------------------------------------------------ intersect: flat primitives
// Flat quads: the quad test of RF:218-269 with its ray-independent terms (nor, ka, kb, kg, k2, projection
// axes) taken precomputed from upload, where they were evaluated with the same operations. All lanes walk the
// same quad, so every table value is a uniform (scalar) load and the axis selects cost nothing.
__device__ __forceinline__ void flat_quads(const SceneDev &sc, v3 o, v3 d, Hit &h) {
    const uint32_t nq = sc.flat_count[PRIM_QUAD];
    for (uint32_t k = 0; k < nq; k++) {
        const float4 F0 = uload(sc.flatq + 5 * k), F1 = uload(sc.flatq + 5 * k + 1), F2 = uload(sc.flatq + 5 * k + 2), F3 = uload(sc.flatq + 5 * k + 3);
        const float4 F4 = uload(sc.flatq + 5 * k + 4);
        const uint32_t id = __float_as_uint(F4.x);
        const v3 p = o - V(F0.x, F0.y, F0.z), nor = V(F1.x, F1.y, F1.z);
        const float t = -dot(p, nor) / dot(d, nor);                  // RF:225
        if (t < 0.0f) continue;                                      // RF:226
        if (!(t > 0.001f && t <= h.t)) continue;                     // cannot pass RF:316 whatever (u, v) are
        const v3 pos = p + d * t;
        const uint32_t axes = __float_as_uint(F3.z);
        const int idu = (int)(axes & 3u), idv = (int)((axes >> 2) & 3u);
        const float kpx = comp(pos, idu), kpy = comp(pos, idv);
        const float kax = F2.x, kay = F2.y, kbx = F2.z, kby = F2.w, kgx = F3.x, kgy = F3.y, k2 = F1.w;
        const float k0 = cross2d(kpx, kpy, kbx, kby);
        const float k1 = cross2d(kpx, kpy, kgx, kgy) - F0.w;
        float u, v;
        if (fabsf(k2) < 0.00001f) {
            v = -k0 / k1;
            u = cross2d(kpx, kpy, kax, kay) / k1;
        } else {
            float w = k1 * k1 - 4.0f * k0 * k2;
            if (w < 0.0f) continue;
            w = sqrtf(w);
            float ik2 = 1.0f / (2.0f * k2);
            v = (-k1 - w) * ik2;
            if (v < 0.0f || v > 1.0f) v = (-k1 + w) * ik2;
            u = (kpx - kax * v) / (kbx + kgx * v);
        }
        if (u < 0.0f || u > 1.0f || v < 0.0f || v > 1.0f) continue;
        accept(h, t, make_ref(PRIM_QUAD, sc.flat_start[PRIM_QUAD] + k), id, v, 1.0f - u, __float_as_uint(F4.y));
    }
}

// closest hit among the flat primitives (the scene-spanning ones kept out of the tree); the traversal kernel
// starts from it, so it begins with the ray already clipped by the walls
__device__ __forceinline__ Hit flat_hit(const SceneDev &sc, const FrameDev &f, v3 o, v3 d) {
    Hit h; h.t = f.max_dist; h.ref = REF_MISS; h.u = 0.0f; h.v = 0.0f; h.id = 0xFFFFFFFFu; h.uvf = 0u;
    if (ray_finite(o, d)) {
        test_range(sc, PRIM_SPHERE, sc.flat_start[PRIM_SPHERE], sc.flat_count[PRIM_SPHERE], o, d, f.eps, h);
        flat_quads(sc, o, d, h);
        test_range(sc, PRIM_TRI, sc.flat_start[PRIM_TRI], sc.flat_count[PRIM_TRI], o, d, f.eps, h);
    }
    return h;
}
// queue slot of path i when generate deals 64-path groups round-robin over the segments
__device__ __forceinline__ uint32_t dealt_slot(uint32_t i, uint32_t nseg, uint32_t segcap) {
    const uint32_t g = i >> 6;
    return (g % nseg) * segcap + (g / nseg) * 64u + (i & 63u);
}

// ------------------------------------------------------------------ generate
// RF:417-422 with the uv arithmetic of RF:428-429,436; one thread per (slot, pixel). Every ray leaves with the
// closest hit so far in its queue entry:
//   seed.od != null: every frame of a batch starts with the same camera rays per pixel (the shader's sub-pixel offsets,
//     RF:436, do not depend on the frame), so they were traced once, in the one-frame queue `seed` of segment capacity
//     `seed_segcap`, and its hit is copied into the entry of each frame's path;
//   else flat != 0: the camera ray is tested against the flat primitives here, while it is in registers;
//   else: a miss (an intersect kernel that looks at every primitive follows).
__global__ __launch_bounds__(256) void k_generate(SceneDev sc, FrameDev f, RayQueue q, uint32_t *count0, RayQueue seed,
                                                  uint32_t seed_segcap, uint32_t flat) {
    const uint32_t total = f.nslots * f.npl;
    const uint32_t groups = (total + 63u) >> 6;            // 64-path groups, dealt round-robin over the segments
    if (blockIdx.x == 0) {
        for (uint32_t s = threadIdx.x; s < f.nseg; s += blockDim.x) {
            uint32_t ng = s < groups ? (groups - s + f.nseg - 1u) / f.nseg : 0u;
            uint32_t n = ng * 64u;
            if (ng && (groups - 1u) % f.nseg == s) n -= groups * 64u - total;   // the last group may be partial
            count0[s * CTR_STRIDE] = n;
        }
    }
    const float rx = (float)f.width, ry = (float)f.height;
    const float m = fmaxf(rx, ry);
    const v3 pos = ldv(f.pos), pc = ldv(f.pc), pu = ldv(f.pu), pv = ldv(f.pv);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        uint32_t slot = fastdiv(i, f.div_npl), li = i - slot * f.npl, x, y;
        local_to_pixel(f, li, x, y);
        float ux = (((float)x + 0.5f) - 1.0f - rx / 2.0f) / m;
        float uy = (((float)y + 0.5f) - 1.0f - ry / 2.0f) / m;
        const uint32_t sub = slot - fastdiv(slot, f.div_spp2) * f.spp2;      // which sub-sample of the SPP x SPP grid (0 at SPP = 1)
        if (f.mode != KL_MODE_Z) {               // RF:436: uv - 0.5 * pixel_size + vec2(x / SPP, y / SPP) * pixel_size, x outer, y inner
            const uint32_t sx = fastdiv(sub, f.div_spp), sy = sub - sx * f.spp;
            float psx = 1.0f / rx, psy = 1.0f / ry;
            ux = (ux - 0.5f * psx) + ((float)sx / (float)f.spp) * psx;
            uy = (uy - 0.5f * psy) + ((float)sy / (float)f.spp) * psy;
        }
        v3 d = normalize(pc + pu * ux + pv * uy - pos);
        const uint32_t dst = dealt_slot(i, f.nseg, f.segcap);
        float t = f.max_dist; uint32_t ref = REF_MISS;
        if (seed.od) {
            const uint32_t src = dealt_slot(sub * f.npl + li, f.nseg, seed_segcap);    // this pixel's (sub-sample's) ray in the one-frame queue
            const float4 sdt = seed.dt[src];
            t = sdt.z; ref = __float_as_uint(sdt.w);
            q.uv[dst] = seed.uv[src];                                   // meaningful only for textured primitives; copied blindly
        } else if (flat) {
            const Hit h = flat_hit(sc, f, pos, d);
            t = h.t; ref = h.ref;
            if (h.uvf & HIT_UV) q.uv[dst] = make_float2(h.u, h.v);
        }
        q.od[dst] = make_float4(pos.x, pos.y, pos.z, d.x);
        q.dt[dst] = make_float4(d.y, d.z, t, __uint_as_float(ref));
        if (q.tp) q.tp[dst] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(i));
    }
}

// the two words of a queue entry that hold the closest hit, and the (u, v) of hits whose shading reads them
__device__ __forceinline__ void store_hit(const RayQueue &q, uint32_t i, const Hit &h) {
    ((float2 *)(q.dt + i))[1] = make_float2(h.t, __uint_as_float(h.ref));
    if (h.uvf & HIT_UV) q.uv[i] = make_float2(h.u, h.v);
}

// ------------------------------------------------------------------ intersect: flat loop
// RF:294-328 for scenes of a few primitives: every lane walks the same primitive list, so the
// geometry comes through uniform (scalar) loads and nothing but the ray and hit touch memory.
__global__ __launch_bounds__(256) void k_intersect_brute(SceneDev sc, FrameDev f, RayQueue q, const uint32_t *counts, Counters *cn) {
    // block b works on segment b % nseg together with the other blocks of that residue
    const uint32_t seg = blockIdx.x % f.nseg, sub = blockIdx.x / f.nseg, nsub = gridDim.x / f.nseg;
    const uint32_t count = counts[seg * CTR_STRIDE];
    if (sub == 0 && threadIdx.x == 0 && count) atomicAdd(&cn->rays, (unsigned long long)count);
    for (uint32_t r = sub * blockDim.x + threadIdx.x; r < count; r += nsub * blockDim.x) {
        const uint32_t i = seg * f.segcap + r;
        float4 od = q.od[i]; float4 dt = q.dt[i];
        v3 o = V(od.x, od.y, od.z), d = V(od.w, dt.x, dt.y);
        Hit h; h.t = f.max_dist; h.ref = REF_MISS; h.u = 0.0f; h.v = 0.0f; h.id = 0xFFFFFFFFu; h.uvf = 0u;
        if (ray_finite(o, d)) {
            test_range(sc, PRIM_SPHERE, sc.base[PRIM_SPHERE], sc.count[PRIM_SPHERE], o, d, f.eps, h);
            test_range(sc, PRIM_QUAD, sc.base[PRIM_QUAD], sc.count[PRIM_QUAD], o, d, f.eps, h);
            test_range(sc, PRIM_TRI, sc.base[PRIM_TRI], sc.count[PRIM_TRI], o, d, f.eps, h);
        }
        if (f.all_uv) h.uvf |= HIT_UV;
        store_hit(q, i, h);
    }
}

// The flat primitives for every ray of a queue, in queue order (uniform loads, no divergence). Writes the hit
// that k_intersect_bvh then starts from. The render loop does not need it: generate and shade run flat_hit
// themselves on the rays they produce; this kernel serves rays that come from elsewhere (caller rays, the
// stand-alone compaction).
__global__ __launch_bounds__(256) void k_intersect_flat(SceneDev sc, FrameDev f, RayQueue q, const uint32_t *counts, Counters *cn, uint32_t stats) {
    const uint32_t seg = blockIdx.x % f.nseg, sub = blockIdx.x / f.nseg, nsub = gridDim.x / f.nseg;
    const uint32_t count = counts[seg * CTR_STRIDE];
    if (stats && sub == 0 && threadIdx.x == 0 && count) {
        atomicAdd(&cn->sphere_tests, (unsigned long long)count * sc.flat_count[PRIM_SPHERE]);
        atomicAdd(&cn->quad_tests, (unsigned long long)count * sc.flat_count[PRIM_QUAD]);
        atomicAdd(&cn->tri_tests, (unsigned long long)count * sc.flat_count[PRIM_TRI]);
    }
    for (uint32_t r = sub * blockDim.x + threadIdx.x; r < count; r += nsub * blockDim.x) {
        const uint32_t i = seg * f.segcap + r;
        float4 od = q.od[i]; float4 dt = q.dt[i];
        Hit h = flat_hit(sc, f, V(od.x, od.y, od.z), V(od.w, dt.x, dt.y));
        if (f.all_uv) h.uvf |= HIT_UV;
        store_hit(q, i, h);
    }
}

// ------------------------------------------------------------------ intersect: BVH
#ifndef KL_ISECT_WAVES
#define KL_ISECT_WAVES 4
#endif
constexpr uint32_t ISECT_WAVES = KL_ISECT_WAVES;          // waves per workgroup
#ifndef KL_ISECT_CHUNK
#define KL_ISECT_CHUNK 64          // small chunks: a wave that ends up with the last rays of a segment holds few of them
#endif
constexpr uint32_t ISECT_CHUNK = KL_ISECT_CHUNK;   // rays a wave takes from its segment's counter at a time
#ifndef KL_ISECT_REFILL
#define KL_ISECT_REFILL 52
#endif
#ifndef KL_ISECT_LDS_STACK
#define KL_ISECT_LDS_STACK 10      // stack entries per lane kept in LDS; 10 + the staging area lets six workgroups share a CU
#endif
constexpr uint32_t ISECT_REFILL = KL_ISECT_REFILL;   // refill idle lanes when fewer than this many are active

// Every ray arrives with its closest hit so far in its queue entry (the flat-primitive hit its producer or
// k_intersect_flat found, or a miss) and starts from there.
//
// Record fetch. A lane needs the 64 bytes of "its" record every step. Fetched by the lane itself that is four
// 16-byte loads whose 64 lanes touch 64 different lines each: the L1 tag pipeline (about one line per clock
// per CU) becomes the limiter. Instead the four lanes of a quad fetch each other's records: load k of the
// wave brings, for every quad, the record of quad-lane k, one 16-byte slice per lane, so an instruction
// touches 16 lines instead of 64. The loads are LDS-DMA (`global_load_lds_dwordx4`: per-lane source, LDS
// destination = wave base + lane * 16), which lands each record contiguously; the lane then reads its own
// record back with four ds_read_b128. Regions of the four loads are 1040 bytes apart so that those reads
// are bank-conflict free.
//
// Stack. The first ISECT_LDS_STACK entries of a lane's traversal stack live in LDS ([wave][entry][lane]:
// consecutive lanes on consecutive banks); deeper entries (rare) spill to a per-wave slab in global memory.
constexpr uint32_t ISECT_LDS_STACK = KL_ISECT_LDS_STACK;
// a binary step pushes at most one entry per level, a 4-wide step at most three
constexpr uint32_t ISECT_STACK_MAX = 3u * BVH_MAX_DEPTH;
constexpr uint32_t ISECT_OVF_STACK = ISECT_STACK_MAX > ISECT_LDS_STACK ? ISECT_STACK_MAX - ISECT_LDS_STACK : 1;
constexpr uint32_t STAGE_REGION_F4 = 65;      // 64 float4 (1 KiB) of payload + 16 bytes of skew per region

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) int lds_int_t;      // explicit LDS pointers: stack accesses stay ds_read / ds_write
typedef __attribute__((address_space(1))) const void glb_void_t;

template <int K>
__device__ __forceinline__ void coop_fetch_one(const float4 *base, uint32_t my_offset, uint32_t slice, float4 *region) {
    // byte offset of quad-lane K's record, broadcast over the quad with a DPP quad_perm
    const uint32_t off = (uint32_t)__builtin_amdgcn_mov_dpp((int)my_offset, K * 0x55, 0xf, 0xf, false) + slice;
    __builtin_amdgcn_global_load_lds((glb_void_t *)((const char *)base + off), (lds_void_t *)region, 16, 0, 0);
}

#ifndef KL_ISECT_MIN_WAVES
#define KL_ISECT_MIN_WAVES 6       // waves per SIMD the register allocation must allow (<= 80 VGPRs)
#endif
// TYPES: which primitive types the tree holds (bit PRIM_*). The leaf branch of a kernel built for one type is that
// type's test alone; the test code of types a scene does not have costs registers even if it never runs (all three:
// 80 VGPRs and two spilled; triangles only: 54), and with them the scheduling freedom of the whole loop.
// STATS: 0 = none; 1 = per-lane tallies of nodes and primitive tests (the instrumented all-types instance); 2 = wave-level
// tallies only (ballot + popcount into scalar registers): lane-steps on nodes and on primitives and active lanes per step, cheap
// enough that the timed instance itself reports its own lane use and steps per ray.
template <int STATS, uint32_t TYPES>
__global__ __launch_bounds__(64 * ISECT_WAVES, KL_ISECT_MIN_WAVES) void k_intersect_bvh(SceneDev sc, FrameDev f, RayQueue q, uint32_t tally_flat,
                                                                   const uint32_t *counts, uint32_t *heads, Counters *cn, int *overflow) {
    __shared__ int stack[ISECT_WAVES][ISECT_LDS_STACK][64];
    __shared__ float4 stage[ISECT_WAVES][4][STAGE_REGION_F4];
    const uint32_t lane = lane_id();
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: LDS bases stay scalar
    lds_int_t *st = (lds_int_t *)&stack[wave_in_block][0][lane];
    int *ovf = overflow + ((size_t)(blockIdx.x * ISECT_WAVES + wave_in_block) * ISECT_OVF_STACK) * 64u + lane;
    // this workgroup serves queue segment blockIdx % nseg; its waves pull ray chunks from that segment's head
    const uint32_t seg = blockIdx.x % f.nseg;
    const uint32_t count = counts[seg * CTR_STRIDE];
    uint32_t *head = heads + seg * CTR_STRIDE;
    const uint32_t seg_base = seg * f.segcap;
    if (blockIdx.x < f.nseg && threadIdx.x == 0 && count) {
        atomicAdd(&cn->rays, (unsigned long long)count);
        if (STATS == 1 && tally_flat) {            // flat tests done by the kernel that produced the rays: tallied here
            atomicAdd(&cn->sphere_tests, (unsigned long long)count * sc.flat_count[PRIM_SPHERE]);
            atomicAdd(&cn->quad_tests, (unsigned long long)count * sc.flat_count[PRIM_QUAD]);
            atomicAdd(&cn->tri_tests, (unsigned long long)count * sc.flat_count[PRIM_TRI]);
        }
    }

    uint32_t cur = 0, end = 0;               // wave-uniform: rays [cur, end) reserved by this wave
    bool exhausted = false;                  // wave-uniform: the global counter has run past count
    bool active = false;
    uint32_t ray = 0; int node = NODE_DONE; int sp = 0;
    v3 o = V(0, 0, 0), d = V(0, 0, 0), id = V(0, 0, 0), nood = V(0, 0, 0);
    Hit h; h.t = 0; h.ref = REF_MISS; h.u = h.v = 0; h.id = 0xFFFFFFFFu; h.uvf = 0u;
    unsigned long long s_nodes = 0, s_sph = 0, s_quad = 0, s_tri = 0, s_act = 0, s_tot = 0;

    for (;;) {
        // ---- refill idle lanes from the wave's reserved range
        unsigned long long idle = __ballot(!active);
        if (idle != 0ull && !(exhausted && cur == end)) {
            float touch0 = 0.0f, touch1 = 0.0f; bool touched = false;
            if (cur == end) {
                uint32_t base = 0u;
                if (lane == 0) base = atomicAdd(head, ISECT_CHUNK);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= count) { exhausted = true; }
                else {
                    cur = base; end = min(base + ISECT_CHUNK, count);
                    // the whole chunk's queue lines are requested now, together with this refill's own loads (one
                    // wait for both): the chunk's later refills then find their rays in L2 instead of HBM
                    const uint32_t tr = seg_base + min(base + lane, count - 1u);
                    touch0 = ((const float *)(q.od + tr))[0];
                    touch1 = ((const float *)(q.dt + tr))[0];
                    touched = true;
                }
            }
            uint32_t avail = end - cur;
            uint32_t want = (uint32_t)__popcll(idle);
            uint32_t take = min(avail, want);
            uint32_t r = mask_rank(idle);
            if (!active && r < take) {
                ray = seg_base + cur + r;
                float4 od = q.od[ray]; float4 dt = q.dt[ray];
                o = V(od.x, od.y, od.z); d = V(od.w, dt.x, dt.y);
                // the entry's closest hit so far (its producer's flat-primitive hit, or a miss) is where the ray starts from;
                // that hit's id is fetched only if an exact tie asks for it
                h.t = dt.z; h.ref = __float_as_uint(dt.w); h.u = 0.0f; h.v = 0.0f; h.uvf = 0u;
                h.id = h.ref != REF_MISS ? ID_UNKNOWN : 0xFFFFFFFFu;
                // slab reciprocals; a zero component is nudged so 0 * inf never appears
                float sx = fabsf(d.x) > 1e-20f ? d.x : copysignf(1e-20f, d.x);
                float sy = fabsf(d.y) > 1e-20f ? d.y : copysignf(1e-20f, d.y);
                float sz = fabsf(d.z) > 1e-20f ? d.z : copysignf(1e-20f, d.z);
                // the hardware reciprocal (1 ulp) is enough here: leaf boxes are widened by 4e-6 of their coordinates
                // (tens of ulps) against rounding in the slab test, and the primitive tests, which decide the hit, do not
                // use these values. A correctly rounded divide costs ten instructions a component.
                id = V(__builtin_amdgcn_rcpf(sx), __builtin_amdgcn_rcpf(sy), __builtin_amdgcn_rcpf(sz));
                nood = V(-(o.x * id.x), -(o.y * id.y), -(o.z * id.z));
                sp = 0;
                node = sc.root;
                active = true;
                if (node == NODE_DONE || !ray_finite(o, d)) active = false;    // empty tree, or a NaN ray (hits nothing): the entry stands
            }
            cur += take;
            if (touched) asm volatile("" :: "v"(touch0), "v"(touch1));   // keeps the requests; their values are not used
        }
        unsigned long long am = __ballot(active);
        if (am == 0ull) { if (exhausted && cur == end) break; else continue; }
        const uint32_t thresh = (exhausted && cur == end) ? 1u : ISECT_REFILL;

        // ---- traverse until too few lanes are busy. Every iteration fetches ONE 64-byte record per
        // lane (an inner node or one primitive of the current leaf) with a single group of loads, so a
        // wave pays one memory latency per step whatever mix of nodes and primitives its lanes are on.
        do {
            const bool inner = node >= 0;
            const uint32_t lv = (uint32_t)~node;
            const uint32_t type = (lv >> 29) & 3u, start = lv & 0x3FFFFFFu;
            const uint32_t record = inner ? (uint32_t)node : start;
            // all 64 lanes take part (a quad serves its busy members); an idle lane names record 0
            const uint32_t my_offset = active ? record << 6 : 0u;
            const uint32_t slice = (lane & 3u) << 4;
            float4 *regions = &stage[wave_in_block][0][0];
            coop_fetch_one<0>(sc.rec, my_offset, slice, regions);
            coop_fetch_one<1>(sc.rec, my_offset, slice, regions + STAGE_REGION_F4);
            coop_fetch_one<2>(sc.rec, my_offset, slice, regions + 2 * STAGE_REGION_F4);
            coop_fetch_one<3>(sc.rec, my_offset, slice, regions + 3 * STAGE_REGION_F4);
            __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the four LDS-DMA loads have landed
            // the entry a pop would return is read together with the record, long before the pop needs it: a
            // step either pushes or pops, so it is still the top of the stack then. The LDS read is
            // unconditional (index clamped into the LDS part); entries beyond it are rare.
            const uint32_t tp = sp > 0 ? (uint32_t)(sp - 1) : 0u;
            int top = st[min(tp, ISECT_LDS_STACK - 1u) * 64u];
            if (active && tp >= ISECT_LDS_STACK) top = ovf[(tp - ISECT_LDS_STACK) * 64u];
            if (STATS == 2) {      // wave-level tallies: one ballot per class, counts stay in scalar registers
                constexpr uint32_t T = 1u << PRIM_TRI, S = 1u << PRIM_SPHERE, Q = 1u << PRIM_QUAD;
                s_nodes += (unsigned long long)__popcll(__ballot(active && inner));
                if (TYPES == T) s_tri += (unsigned long long)__popcll(__ballot(active && !inner));
                else {
                    if (TYPES & T) s_tri += (unsigned long long)__popcll(__ballot(active && !inner && type == PRIM_TRI));
                    if (TYPES & S) s_sph += (unsigned long long)__popcll(__ballot(active && !inner && type == PRIM_SPHERE));
                    if (TYPES & Q) s_quad += (unsigned long long)__popcll(__ballot(active && !inner && type == PRIM_QUAD));
                }
            }
            const float4 *mine = &stage[wave_in_block][lane & 3u][(lane >> 2) * 4u];
            const float4 n0 = mine[0], n1 = mine[1], n2 = mine[2], n3 = mine[3];
            if (active) {
                bool pop = false;
                auto push = [&](int ref) {
                    if ((uint32_t)sp < ISECT_LDS_STACK) st[sp * 64] = ref;
                    else ovf[((uint32_t)sp - ISECT_LDS_STACK) * 64u] = ref;
                    sp++;
                };
                if (inner) {
                    if (STATS == 1) s_nodes++;
                    // four child boxes on the node's 8-bit grid: plane = lo + byte * step, so along the ray
                    // t = byte * (step * 1/d) + (lo - o) * 1/d; near and far planes are picked by the sign of 1/d.
                    // An empty slot holds an inverted box and fails the test by itself.
                    const float Ax = n0.w * id.x, Ay = n1.x * id.y, Az = n1.y * id.z;
                    const float Bx = fmaf(n0.x, id.x, nood.x), By = fmaf(n0.y, id.y, nood.y), Bz = fmaf(n0.z, id.z, nood.z);
                    const uint32_t lox = __float_as_uint(n1.z), loy = __float_as_uint(n1.w), loz = __float_as_uint(n2.x);
                    const uint32_t hix = __float_as_uint(n2.y), hiy = __float_as_uint(n2.z), hiz = __float_as_uint(n2.w);
                    const bool px = id.x >= 0.0f, py = id.y >= 0.0f, pz = id.z >= 0.0f;
                    const uint32_t nx = px ? lox : hix, fx = px ? hix : lox;
                    const uint32_t ny = py ? loy : hiy, fy = py ? hiy : loy;
                    const uint32_t nz = pz ? loz : hiz, fz = pz ? hiz : loz;
                    int ref[4] = { __float_as_int(n3.x), __float_as_int(n3.y), __float_as_int(n3.z), __float_as_int(n3.w) };
                    float key[4];
                    const float inf = __builtin_huge_valf();
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float tnx = fmaf((float)((nx >> (8 * k)) & 0xFFu), Ax, Bx), tfx = fmaf((float)((fx >> (8 * k)) & 0xFFu), Ax, Bx);
                        const float tny = fmaf((float)((ny >> (8 * k)) & 0xFFu), Ay, By), tfy = fmaf((float)((fy >> (8 * k)) & 0xFFu), Ay, By);
                        const float tnz = fmaf((float)((nz >> (8 * k)) & 0xFFu), Az, Bz), tfz = fmaf((float)((fz >> (8 * k)) & 0xFFu), Az, Bz);
                        const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
                        const float tf = fminf(fminf(tfx, tfy), fminf(tfz, h.t));
                        key[k] = tn <= tf ? tn : inf;
                    }
                    // sort the four (entry distance, child) pairs, nearest first
                    auto cswap = [&](int a, int b) {
                        const bool sw = key[b] < key[a];
                        const float ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b];
                        const int ra = sw ? ref[b] : ref[a], rb = sw ? ref[a] : ref[b];
                        key[a] = ka; key[b] = kb; ref[a] = ra; ref[b] = rb;
                    };
                    cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
                    if (key[0] < inf) {
                        // farthest first, so that the nearest pending child pops first
                        if ((uint32_t)sp + 3u <= ISECT_LDS_STACK) {         // the usual case: all three fit the LDS part
                            lds_int_t *p = st + sp * 64;
                            if (key[3] < inf) { *p = ref[3]; p += 64; sp++; }
                            if (key[2] < inf) { *p = ref[2]; p += 64; sp++; }
                            if (key[1] < inf) { *p = ref[1]; sp++; }
                        } else {
                            if (key[3] < inf) push(ref[3]);
                            if (key[2] < inf) push(ref[2]);
                            if (key[1] < inf) push(ref[1]);
                        }
                        node = ref[0];
                    } else pop = true;
                } else {
                    constexpr uint32_t T = 1u << PRIM_TRI, S = 1u << PRIM_SPHERE, Q = 1u << PRIM_QUAD;
                    if (TYPES == T || ((TYPES & T) && type == PRIM_TRI)) { test_tri(n0, n1, n2, start, o, d, h, &sc); if (STATS == 1) s_tri++; }
                    else if (TYPES == S || ((TYPES & S) && type == PRIM_SPHERE)) { test_sphere(n0, n1, start, o, d, f.eps, h, &sc); if (STATS == 1) s_sph++; }
                    else if (TYPES & Q) { test_quad(n0, n1, n2, n3, start, o, d, h, &sc); if (STATS == 1) s_quad++; }
                    if ((lv >> 26) & 7u) node = ~(int)(lv + 1u - (1u << 26));   // next primitive of this leaf
                    else pop = true;
                }
                if (pop) {
                    if (sp > 0) {
                        sp--;
                        node = top;
                    } else {
                        // A ray that met nothing closer in the tree leaves its entry as it is (accept() marks every hit it
                        // takes; an exact tie that the entry's own hit wins fetches that hit's id but takes nothing). The
                        // address is formed here, from the ray index, rather than carried through the loop in two registers.
                        if (h.uvf & HIT_TAKEN) {
                            uint32_t slot = ray;
                            asm volatile("" : "+v"(slot));
                            if (f.all_uv) h.uvf |= HIT_UV;
                            store_hit(q, slot, h);
                        }
                        active = false;
                    }
                }
            }
            am = __ballot(active);
            if (STATS) { s_act += (unsigned long long)__popcll(am); s_tot += 64ull; }
        } while ((uint32_t)__popcll(am) >= thresh);
    }
    if (STATS) {
        // lane sums -> one atomic per wave per counter (the wave-level tallies are already whole-wave sums)
        if (STATS == 1)
            for (int off = 32; off > 0; off >>= 1) {
                s_nodes += __shfl_down(s_nodes, off); s_sph += __shfl_down(s_sph, off);
                s_quad += __shfl_down(s_quad, off); s_tri += __shfl_down(s_tri, off);
            }
        if (lane == 0) {
            atomicAdd(&cn->nodes, s_nodes); atomicAdd(&cn->sphere_tests, s_sph);
            atomicAdd(&cn->quad_tests, s_quad); atomicAdd(&cn->tri_tests, s_tri);
            atomicAdd(&cn->lane_active, s_act); atomicAdd(&cn->lane_total, s_tot);
        }
    }
}

// ------------------------------------------------------------------ shading helpers
struct Surf { v3 pos, n; bool inside; };

// the fields of the shader's Hit struct (RF:142-150) that shading reads, rebuilt from (ray, t, ref) and the
// first words of the primitive's shading record: s0.xyz = geometric normal (sphere: centre), radius (sphere only)
__device__ __forceinline__ Surf surface(v3 o, v3 d, float t, uint32_t ref, float4 s0, float radius) {
    Surf s;
    s.pos = o + d * t;                                             // RF:304,318
    if (ref_type(ref) == PRIM_SPHERE) {
        v3 ctr = V(s0.x, s0.y, s0.z);
        s.n = normalize(s.pos - ctr);                              // RF:305
        s.inside = length(o - ctr) <= radius;                      // RF:308
        if (s.inside) s.n = s.n * -1.0f;                           // RF:309
    } else {
        v3 nor = V(s0.x, s0.y, s0.z);                              // RF:319 for quads, stored at upload
        s.n = dot(nor, d) < 0.0f ? nor : -nor;                     // faceforward, RF:320
        s.inside = false;
    }
    return s;
}

// texture(sampler, uv).rgb with the sampler state of source/Rendering/Opengl/Texture.cpp:8-12:
// NEAREST, clamp to edge, rows as uploaded; a NaN coordinate selects texel 0
__device__ __forceinline__ v3 tex_fetch(const SceneDev &sc, int which, float u, float v) {
    uint32_t w = sc.tex_w[which], hgt = sc.tex_h[which];
    float fx = floorf(u * (float)w), fy = floorf(v * (float)hgt);
    uint32_t ix = !(fx >= 0.0f) ? 0u : (fx >= (float)w ? w - 1u : (uint32_t)fx);
    uint32_t iy = !(fy >= 0.0f) ? 0u : (fy >= (float)hgt ? hgt - 1u : (uint32_t)fy);
    uchar4 px = sc.tex[which][(size_t)iy * w + ix];
    const float k = 1.0f / 255.0f;
    return V((float)px.x * k, (float)px.y * k, (float)px.z * k);
}
// RF:330-339 with eulerToRot(0,-90,0) (RF:98-122) folded: d' = (d.x, c*d.y + s*d.z, -s*d.y + c*d.z)
__device__ __forceinline__ v3 environment(const SceneDev &sc, const FrameDev &f, v3 d) {
    const float c = -4.371138828673793e-08f, sn = -1.0f;
    float rx = (1.0f * d.x + 0.0f * d.y) + 0.0f * d.z;
    float ry = (0.0f * d.x + c * d.y) + sn * d.z;
    float rz = (0.0f * d.x + (-sn) * d.y) + c * d.z;
    float phi = atan2f_(ry, rx);
    float theta = acosf_(rz);
    float u = phi / 6.28318530718f + 0.5f;
    float v = theta / 3.14159265359f;
    if (sc.envf) {            // float environment: same NEAREST / clamp lookup, the texel as it is
        const uint32_t w = sc.envf_w, hgt = sc.envf_h;
        const float fx = floorf(u * (float)w), fy = floorf(v * (float)hgt);
        const uint32_t ix = !(fx >= 0.0f) ? 0u : (fx >= (float)w ? w - 1u : (uint32_t)fx);
        const uint32_t iy = !(fy >= 0.0f) ? 0u : (fy >= (float)hgt ? hgt - 1u : (uint32_t)fy);
        const float4 px = sc.envf[(size_t)iy * w + ix];
        return V(px.x, px.y, px.z) * f.hdri_strength;
    }
    return tex_fetch(sc, KL_TEX_HDRI, u, v) * f.hdri_strength;
}
// RF:273-277
__device__ __forceinline__ v3 hemisphere(f4 r) {
    float px = 2.0f * 3.14159265359f * r.x;
    float py = sqrtf(r.y);
    return normalize(V(sinf_(px) * py, cosf_(px) * py, sqrtf(1.0f - py * py)));
}
// RF:279-292
__device__ __forceinline__ v3 cone_roughness(v3 dir, float theta, f4 r) {
    v3 left = cross(dir, V(0.0f, 1.0f, 0.0f));
    left = length(left) > 0.1f ? normalize(left) : normalize(cross(dir, V(0.0f, 0.0f, 1.0f)));
    v3 up = normalize(cross(dir, left));
    float cos_theta = (1.0f - r.x) + r.x * cosf_(theta);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi = r.y * 2.0f * 3.14159265359f;
    return normalize(left * cosf_(phi) * sin_theta + up * sinf_(phi) * sin_theta + dir * cos_theta);
}
// RF:87-96
__device__ __forceinline__ float snell(float sin_theta, float iori, float iort) { return iori / iort * sin_theta; }
__device__ __forceinline__ float fresnel(float iori, float iort, float cosi, float cost) {
    float rpar = (iort * cosi - iori * cost) / (iort * cosi + iori * cost);
    float rper = (iori * cosi - iort * cost) / (iori * cosi + iort * cost);
    rpar *= rpar; rper *= rper;
    return (rpar + rper) / 2.0f;
}

constexpr uint32_t PATH_DEAD = 0xFFFFFFFFu;

// one queue entry (ray, closest hit, throughput and path), as shade reads it
struct ShadeIn { float4 od; float4 dt; float4 tp; };
__device__ __forceinline__ ShadeIn shade_load(const RayQueue &q, uint32_t i) {
    ShadeIn in; in.od = nt_load(q.od + i); in.dt = nt_load(q.dt + i); in.tp = nt_load(q.tp + i);
    return in;
}

// ------------------------------------------------------------------ shade
// One lane per traced ray. MODE selects the walk: RF:355-407 (path traced), RF:341-353 (AO),
// RF:409-415 (ZDepth). A finished path writes its colour to raw[path]; a continuing one is
// pushed to qout (fused compaction) or written back in place for k_compact. With `flat` (fused compaction
// only) the pushed ray leaves with its flat-primitive hit in its entry, otherwise with a miss.
//
// The kernel lives on memory latency and bandwidth: 48 bytes per ray come in and 48 go out, and the chain of
// dependent fetches is kept short: queue entry -> ONE shading record (surface and material together) -> texel. The next iteration's queue entry
// is requested before this iteration's work starts, and the queue-tail atomic is issued as soon as the lanes
// know whether they go on, so that its round trip runs under the flat-primitive tests.
#ifndef KL_SHADE_MIN_WAVES
#define KL_SHADE_MIN_WAVES 1
#endif
template <int MODE>
__global__ __launch_bounds__(256, KL_SHADE_MIN_WAVES) void k_shade(SceneDev sc_by_value, FrameDev f, RayQueue qin,
                                               const uint32_t *counts_in, RayQueue qout, uint32_t *counts_out,
                                               float4 *raw, uint32_t bounce, uint32_t fused, uint32_t flat) {
    // block b works on segment b % nseg; survivors stay in their segment
    const uint32_t seg = blockIdx.x % f.nseg, sub = blockIdx.x / f.nseg, nsub = gridDim.x / f.nseg;
    const uint32_t count = counts_in[seg * CTR_STRIDE];
    uint32_t *count_out = counts_out + seg * CTR_STRIDE;
    const uint32_t seg_base = seg * f.segcap;
    const uint32_t nwaves = (nsub * blockDim.x) >> 6;
    const uint32_t wave = (sub * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = lane_id();
    ShadeIn nxt;
    nxt.od = make_float4(0, 0, 0, 0); nxt.dt = make_float4(0, 0, 0, 0); nxt.tp = make_float4(0, 0, 0, 0);
    if (wave * 64u + lane < count) nxt = shade_load(qin, seg_base + wave * 64u + lane);
    for (uint32_t base = wave * 64u; base < count; base += nwaves * 64u) {
        // The scene description (32 words, the kernel's first argument) is read where it is used, by scalar loads from
        // the kernel-argument segment, instead of being held in scalar registers across the loop: together with the
        // frame description and the flat-quad table it does not fit there, and the compiler then parks scalars in
        // vector-register lanes and fetches them back one v_readlane at a time (58 per iteration, and 6 vector
        // registers: 84 -> 78, a sixth wave per SIMD). The empty asm makes the pointer opaque in every iteration,
        // so the loads cannot be hoisted back out of the loop.
        auto kernarg = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kernarg));
        const SceneDev &sc = *(const SceneDev *)(const __attribute__((address_space(4))) SceneDev *)kernarg;
        const uint32_t i = seg_base + base + lane;
        const bool valid = base + lane < count;
        const ShadeIn in = nxt;
        {   // the queue entry of the next iteration, in flight while this one is worked on
            const uint32_t nb = base + nwaves * 64u + lane;
            if (nb < count) nxt = shade_load(qin, seg_base + nb);
        }
        bool alive = false;
        v3 o = V(in.od.x, in.od.y, in.od.z), d = V(in.od.w, in.dt.x, in.dt.y), thr = V(in.tp.x, in.tp.y, in.tp.z);
        const uint32_t path = valid ? __float_as_uint(in.tp.w) : PATH_DEAD;
        const float t = in.dt.z; const uint32_t ref = __float_as_uint(in.dt.w);
        // the shading record of the primitive that was hit: the only fetch between the queue entry and the texel
        const bool hit = valid && !(t >= f.max_dist) && MODE != KL_MODE_Z;
        float4 s0 = make_float4(0, 0, 0, 0), s1 = s0, s2 = s0, s3 = s0;
        if (hit) {
            const float4 *sr = sc.shrec + 4u * (ref_record(ref) - sc.base[PRIM_SPHERE]);
            s0 = sr[0]; s1 = sr[1]; s2 = sr[2]; s3 = sr[3];
        }
        v3 col = V(0, 0, 0);
        bool done = true;
        if (valid) {
            uint32_t slot = fastdiv(path, f.div_npl), li = path - slot * f.npl, px, py;
            local_to_pixel(f, li, px, py);
            f4 r = rand4(px, py, f.first_frame + fastdiv(slot, f.div_spp2));     // every sub-sample of a frame sees the same numbers
            if (MODE == KL_MODE_Z) {
                if (!(t >= f.max_dist)) { float z = 1.0f - fminf(t / 50.0f, 0.9f); col = V(z, z, z); }
            } else if (MODE == KL_MODE_AO) {
                if (!(t >= f.max_dist)) {
                    if (bounce == 0) {
                        Surf s = surface(o, d, t, ref, s0, s2.z);
                        d = cone_roughness(s.n, 10.0f, r);            // RF:348: theta is 10 radians
                        o = s.pos + d * f.eps;                        // RF:349
                        done = false;
                    } else {
                        float dist = fminf(t / f.ao_length, 1.0f);    // RF:350
                        col = V(dist, dist, dist);
                    }
                }
            } else {
                if (t >= f.max_dist) {
                    col = thr * environment(sc, f, d);                // RF:362-364
                } else {
                    Surf s = surface(o, d, t, ref, s0, s2.z);
                    const int mtype = __float_as_int(s0.w);
                    const v3 mcol = V(s1.x, s1.y, s1.z);
                    if (mtype == KL_DIFFUSE || mtype == KL_TEXTURED) {           // RF:365-378
                        if (mtype == KL_TEXTURED) {
                            // the hit's (u, v): stored for textured quads (uv) and triangles (barycentrics); Hit_UV of a
                            // sphere is (0, 0)
                            const float2 huv = ref_type(ref) == PRIM_SPHERE ? make_float2(0.0f, 0.0f) : qin.uv[i];
                            float tu = huv.x, tv = huv.y;
                            if (ref_type(ref) == PRIM_TRI) {          // texture coordinates from the corners' (s2.zw, s3.xy, s3.zw)
                                float w = (1.0f - huv.x) - huv.y;
                                tu = (s2.z * w + s3.x * huv.x) + s3.z * huv.y;
                                tv = (s2.w * w + s3.y * huv.x) + s3.w * huv.y;
                            }
                            col = thr * tex_fetch(sc, KL_TEX_ALBEDO, tu, tv);
                        } else {
                            v3 tangent = normalize(cross(d, s.n));
                            v3 bitangent = normalize(cross(s.n, tangent));
                            v3 nl = hemisphere(r);
                            v3 local = normalize(tangent * nl.x + bitangent * nl.y + s.n * nl.z);
                            d = normalize(mix(reflect(d, s.n), local, s2.x));
                            thr = thr * mcol;
                            done = false;
                        }
                    } else if (mtype == KL_GLASS) {                               // RF:379-400
                        float cosi = fabsf(dot(s.n, d));
                        float sini = sqrtf(1.0f - cosi * cosi);
                        float iort = s2.y, iori = 1.0f;
                        if (s.inside) { iori = iort; iort = 1.0f; }
                        float sint = snell(sini, iori, iort);
                        float cost = sqrtf(1.0f - sint * sint);       // NaN past the critical angle: no TIR branch
                        float frsn = fresnel(iori, iort, cosi, cost);
                        if (r.x > frsn) {                             // false for NaN -> reflect
                            v3 bit = normalize(d - s.n * dot(s.n, d));
                            d = normalize(bit * sint - s.n * cost);
                            thr = thr * mcol;
                        } else {
                            d = reflect(d, s.n);
                        }
                        done = false;
                    } else if (mtype == KL_EMISSIVE) {                            // RF:401-403
                        col = thr * mcol * s1.w;
                    } else {
                        done = false;                                 // unknown type: the shader's if-chain falls through
                    }
                    if (!done) o = s.pos + d * f.eps;                 // RF:404
                }
                if (!done && bounce + 1 >= f.max_bounces) { done = true; col = V(0, 0, 0); }   // RF:406
            }
            alive = !done;
        }
        if (fused) {
            // one atomic per wave reserves the survivors' slots; issued before the colour store and the flat tests
            const unsigned long long m = __ballot(alive);
            uint32_t tail = 0u;
            const uint32_t leader = m ? (uint32_t)__ffsll((long long)m) - 1u : 0u;
            if (m != 0ull && lane == leader) tail = atomicAdd(count_out, (uint32_t)__popcll(m));
            if (valid && done) raw[path] = make_float4(col.x, col.y, col.z, 1.0f);
            Hit fh; fh.t = f.max_dist; fh.ref = REF_MISS; fh.u = fh.v = 0.0f; fh.id = 0xFFFFFFFFu; fh.uvf = 0u;
            if (alive && flat) fh = flat_hit(sc, f, o, d);
            if (m != 0ull) {
                const uint32_t slot_out = seg_base + (uint32_t)__shfl((int)tail, (int)leader) + mask_rank(m);
                if (alive) {
                    nt_store(qout.od + slot_out, make_float4(o.x, o.y, o.z, d.x));
                    nt_store(qout.dt + slot_out, make_float4(d.y, d.z, fh.t, __uint_as_float(fh.ref)));
                    nt_store(qout.tp + slot_out, make_float4(thr.x, thr.y, thr.z, __uint_as_float(path)));
                    if (fh.uvf & HIT_UV) qout.uv[slot_out] = make_float2(fh.u, fh.v);
                }
            }
        } else if (valid) {
            if (done) raw[path] = make_float4(col.x, col.y, col.z, 1.0f);
            qin.od[i] = make_float4(o.x, o.y, o.z, d.x);
            qin.dt[i] = make_float4(d.y, d.z, f.max_dist, __uint_as_float(REF_MISS));
            qin.tp[i] = make_float4(thr.x, thr.y, thr.z, __uint_as_float(alive ? path : PATH_DEAD));
        }
    }
}

// ------------------------------------------------------------------ compact
// stream compaction of a shaded queue: ballot + popcount, one atomic per wave
__global__ __launch_bounds__(256) void k_compact(FrameDev f, RayQueue qin, const uint32_t *counts_in, RayQueue qout, uint32_t *counts_out) {
    const uint32_t seg = blockIdx.x % f.nseg, sub = blockIdx.x / f.nseg, nsub = gridDim.x / f.nseg;
    const uint32_t count = counts_in[seg * CTR_STRIDE];
    const uint32_t seg_base = seg * f.segcap;
    const uint32_t nwaves = (nsub * blockDim.x) >> 6;
    const uint32_t wave = (sub * blockDim.x + threadIdx.x) >> 6;
    for (uint32_t base = wave * 64u; base < count; base += nwaves * 64u) {
        const uint32_t i = seg_base + base + lane_id();
        float4 tp = make_float4(0, 0, 0, 0);
        bool alive = false;
        if (base + lane_id() < count) { tp = qin.tp[i]; alive = __float_as_uint(tp.w) != PATH_DEAD; }
        uint32_t dst = seg_base + wave_reserve(counts_out + seg * CTR_STRIDE, alive);
        if (alive) { qout.od[dst] = qin.od[i]; qout.dt[dst] = qin.dt[i]; qout.tp[dst] = tp; }
    }
}

// what an RGBA8 colour attachment keeps of a value: clamp to [0,1], round to nearest (ties to even) on the 8-bit grid, read
// back as byte * (1/255); NaN stores 0 (llvmpipe's conversion, matched bit for bit against the reference run)
__device__ __forceinline__ float unorm8(float x) {
    x = !(x >= 0.0f) ? 0.0f : (x > 1.0f ? 1.0f : x);
    return rintf(x * 255.0f) * (1.0f / 255.0f);
}

// ------------------------------------------------------------------ accumulate
// resources/Shaders/Accumulation.glsl:16-24 applied to the frames of the batch in order.
// A frame at or past max_frames contributes the accumulation texel itself (RF:426,461-463).
__global__ __launch_bounds__(256) void k_accumulate(FrameDev f, const float4 *raw, float4 *acc, float4 *last_raw,
                                                    uint32_t camera_change, uint32_t accum_mode) {
    for (uint32_t li = blockIdx.x * blockDim.x + threadIdx.x; li < f.npl; li += gridDim.x * blockDim.x) {
        float4 a = acc[li];
        float4 r = make_float4(0, 0, 0, 1);
        const uint32_t nframes = fastdiv(f.nslots, f.div_spp2);
        for (uint32_t s = 0; s < nframes; s++) {
            const uint32_t frame = f.first_frame + s;
            const bool frozen = f.max_frames != 0u && frame >= f.max_frames;
            if (frozen) r = a;
            else if (f.spp2 == 1u) r = raw[(size_t)s * f.npl + li];
            else {
                // RF:433-440 / :445-453: col = 0; col += each sub-sample in x-outer, y-inner order; col /= float(SPP * SPP)
                float cx = 0.0f, cy = 0.0f, cz = 0.0f;
                for (uint32_t k = 0; k < f.spp2; k++) {
                    const float4 c = raw[((size_t)s * f.spp2 + k) * f.npl + li];
                    cx += c.x; cy += c.y; cz += c.z;
                }
                const float n = (float)f.spp2;
                r = make_float4(cx / n, cy / n, cz / n, 1.0f);
            }
            if (accum_mode == KL_ACCUM_SUM) {
                // a frozen frame leaves the mean where it is, i.e. adds no term to the sum
                if (!frozen) { a.x += r.x; a.y += r.y; a.z += r.z; }
                a.w = 1.0f;
            } else if (accum_mode == KL_ACCUM_MEAN_UNORM8) {
                // the application's own attachments (unsized GL_RGBA = RGBA8, FBT.cpp:6,21): the raw colour and the mean are
                // clamped and rounded to 8 bits whenever they are written
                r = make_float4(unorm8(r.x), unorm8(r.y), unorm8(r.z), 1.0f);
                if (frame <= 1u || !(camera_change && s == 0u)) {
                    const float w = (float)frame;
                    a = make_float4(unorm8((a.x * w + r.x) / (w + 1.0f)), unorm8((a.y * w + r.y) / (w + 1.0f)), unorm8((a.z * w + r.z) / (w + 1.0f)), 1.0f);
                } else a = r;
            } else if (frame <= 1u || !(camera_change && s == 0u)) {
                float w = (float)frame;
                a.x = (a.x * w + r.x) / (w + 1.0f);
                a.y = (a.y * w + r.y) / (w + 1.0f);
                a.z = (a.z * w + r.z) / (w + 1.0f);
                a.w = 1.0f;
            } else {
                a = r;
            }
        }
        acc[li] = a;
        last_raw[li] = r;
    }
}

// resources/Shaders/PostProcess.glsl:34-39: CLAMP is what line 38 hands to the 8-bit backbuffer,
// ACES evaluates the overwritten lines 35-37
__global__ __launch_bounds__(256) void k_resolve(const float4 *acc, uchar4 *out, uint32_t n, uint32_t tonemap) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 c = acc[i];
        if (tonemap == KL_TONEMAP_ACES) {
            float x = fmaxf(c.x, 0.0f), y = fmaxf(c.y, 0.0f), z = fmaxf(c.z, 0.0f);
            float r = (3.2404542f * x + -1.5371385f * y) + -0.4985314f * z;
            float g = (-0.9692660f * x + 1.8760108f * y) + 0.0415560f * z;
            float b = (0.0556434f * x + -0.2040259f * y) + 1.0572252f * z;
            r = fmaxf(r, 0.0f); g = fmaxf(g, 0.0f); b = fmaxf(b, 0.0f);
            c.x = (r * (2.51f * r + 0.03f)) / (r * (2.43f * r + 0.59f) + 0.14f);
            c.y = (g * (2.51f * g + 0.03f)) / (g * (2.43f * g + 0.59f) + 0.14f);
            c.z = (b * (2.51f * b + 0.03f)) / (b * (2.43f * b + 0.59f) + 0.14f);
            c.w = 1.0f;
        }
        // the 8-bit backbuffer keeps what an RGBA8 attachment keeps (unorm8 above: clamp, round to nearest, ties to even)
        out[i] = make_uchar4((unsigned char)rintf(unorm8(c.x) * 255.0f), (unsigned char)rintf(unorm8(c.y) * 255.0f),
                             (unsigned char)rintf(unorm8(c.z) * 255.0f), (unsigned char)rintf(unorm8(c.w) * 255.0f));
    }
}

// unpack the hits of kl_trace_rays' queue into (t, declaration-order id, uv)
__global__ void k_unpack_hits(SceneDev sc, RayQueue q, uint32_t n, float *t, uint32_t *prim, float2 *uv) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 dt = q.dt[i];
    uint32_t ref = __float_as_uint(dt.w), id = 0xFFFFFFFFu;
    if (ref != REF_MISS) id = ref_id(sc, ref);
    t[i] = dt.z; prim[i] = id;
    uv[i] = (ref != REF_MISS && ref_type(ref) != PRIM_SPHERE) ? q.uv[i] : make_float2(0.0f, 0.0f);   // the queue was traced with all_uv
}

// ------------------------------------------------------------------ launch wrappers
static inline uint32_t grid_for(uint32_t n, uint32_t cap) {
    uint32_t g = (n + 255u) / 256u;
    return g < 1u ? 1u : (g > cap ? cap : g);
}
// grids of the queue kernels are whole multiples of the segment count: block b serves segment b % nseg
static inline uint32_t seg_grid(const FrameDev &f, uint32_t max_rays, uint32_t cap_blocks, uint32_t rays_per_block) {
    uint32_t per_seg_rays = (max_rays + f.nseg - 1u) / f.nseg;
    uint32_t per_seg = (per_seg_rays + rays_per_block - 1u) / rays_per_block;
    uint32_t cap = cap_blocks / f.nseg;
    if (per_seg > cap) per_seg = cap;
    if (per_seg < 1u) per_seg = 1u;
    return per_seg * f.nseg;
}

void launch_generate(hipStream_t s, const LaunchCfg &lc, const SceneDev &sc, const FrameDev &f, RayQueue q, uint32_t *count0,
                     bool flat, RayQueue seed, uint32_t seed_segcap) {
    hipLaunchKernelGGL(k_generate, dim3(grid_for(f.nslots * f.npl, lc.cus * 8u)), dim3(256), 0, s, sc, f, q, count0, seed, seed_segcap,
                       flat ? 1u : 0u);
}
size_t intersect_overflow_ints(const LaunchCfg &lc) {
    return (size_t)lc.cus * lc.isect_blocks_per_cu * ISECT_WAVES * ISECT_OVF_STACK * 64u;
}
bool scene_has_flat(const SceneDev &sc) { return (sc.flat_count[0] + sc.flat_count[1] + sc.flat_count[2]) != 0u; }
// the coherent pass over the flat primitives; precedes launch_intersect when the tree is used and scene_has_flat()
void launch_intersect_flat(hipStream_t s, const LaunchCfg &lc, const SceneDev &sc, const FrameDev &f, RayQueue q,
                           const uint32_t *counts, Counters *cn, bool stats, uint32_t max_rays) {
    hipLaunchKernelGGL(k_intersect_flat, dim3(seg_grid(f, max_rays, lc.cus * 8u, 256u)), dim3(256), 0, s, sc, f, q, counts, cn, stats ? 1u : 0u);
}
void launch_intersect(hipStream_t s, const LaunchCfg &lc, const SceneDev &sc, const FrameDev &f, RayQueue q,
                      const uint32_t *counts, uint32_t *heads, Counters *cn, int *overflow, bool bvh, int stats, uint32_t max_rays,
                      bool flat_by_producer) {
    if (!bvh) {
        hipLaunchKernelGGL(k_intersect_brute, dim3(seg_grid(f, max_rays, lc.cus * 8u, 256u)), dim3(256), 0, s, sc, f, q, counts, cn);
        return;
    }
    const uint32_t tally_flat = (scene_has_flat(sc) && flat_by_producer) ? 1u : 0u;
    uint32_t blocks = seg_grid(f, max_rays, lc.cus * lc.isect_blocks_per_cu, ISECT_CHUNK * ISECT_WAVES);
    // the kernel built for the primitive types this tree holds (the instrumented variant: one kernel for all)
    uint32_t types = 0u;
    for (uint32_t t = 0; t < 3u; t++) types |= sc.count[t] > sc.flat_count[t] ? 1u << t : 0u;
#define KL_ISECT_LAUNCH(ST, TY) hipLaunchKernelGGL((k_intersect_bvh<ST, TY>), dim3(blocks), dim3(64 * ISECT_WAVES), 0, s, sc, f, q, tally_flat, counts, heads, cn, overflow)
    if (stats == 1) { KL_ISECT_LAUNCH(1, 7u); return; }
#define KL_ISECT_BY_TYPES(ST) switch (types) { \
        case 1u: KL_ISECT_LAUNCH(ST, 1u); break; case 2u: KL_ISECT_LAUNCH(ST, 2u); break; case 3u: KL_ISECT_LAUNCH(ST, 3u); break; \
        case 4u: KL_ISECT_LAUNCH(ST, 4u); break; case 5u: KL_ISECT_LAUNCH(ST, 5u); break; case 6u: KL_ISECT_LAUNCH(ST, 6u); break; \
        default: KL_ISECT_LAUNCH(ST, 7u); break; }
    if (stats == 2) { KL_ISECT_BY_TYPES(2) } else { KL_ISECT_BY_TYPES(0) }
#undef KL_ISECT_BY_TYPES
#undef KL_ISECT_LAUNCH
}
void launch_shade(hipStream_t s, const LaunchCfg &lc, const SceneDev &sc, const FrameDev &f, RayQueue qin,
                  const uint32_t *count_in, RayQueue qout, uint32_t *count_out, float4 *raw, uint32_t bounce, bool fused,
                  uint32_t max_rays, bool flat) {
    dim3 g(seg_grid(f, max_rays, lc.cus * lc.shade_blocks_per_cu, 256u)), b(256);
    const uint32_t fu = fused ? 1u : 0u, fl = (fused && flat) ? 1u : 0u;
    if (f.mode == KL_MODE_PT) hipLaunchKernelGGL(k_shade<KL_MODE_PT>, g, b, 0, s, sc, f, qin, count_in, qout, count_out, raw, bounce, fu, fl);
    else if (f.mode == KL_MODE_AO) hipLaunchKernelGGL(k_shade<KL_MODE_AO>, g, b, 0, s, sc, f, qin, count_in, qout, count_out, raw, bounce, fu, fl);
    else hipLaunchKernelGGL(k_shade<KL_MODE_Z>, g, b, 0, s, sc, f, qin, count_in, qout, count_out, raw, bounce, fu, fl);
}
void launch_compact(hipStream_t s, const LaunchCfg &lc, const FrameDev &f, RayQueue qin, const uint32_t *count_in, RayQueue qout,
                    uint32_t *count_out, uint32_t max_rays) {
    hipLaunchKernelGGL(k_compact, dim3(seg_grid(f, max_rays, lc.cus * 8u, 256u)), dim3(256), 0, s, f, qin, count_in, qout, count_out);
}
void launch_accumulate(hipStream_t s, const LaunchCfg &lc, const FrameDev &f, const float4 *raw, float4 *acc, float4 *last_raw,
                       bool camera_change, uint32_t accum_mode) {
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(f.npl, lc.cus * 8u)), dim3(256), 0, s, f, raw, acc, last_raw,
                       camera_change ? 1u : 0u, accum_mode);
}
void launch_resolve(hipStream_t s, const LaunchCfg &lc, const float4 *acc, uchar4 *out, uint32_t n, uint32_t tonemap) {
    hipLaunchKernelGGL(k_resolve, dim3(grid_for(n, lc.cus * 8u)), dim3(256), 0, s, acc, out, n, tonemap);
}
void launch_unpack_hits(hipStream_t s, const SceneDev &sc, RayQueue q, uint32_t n, float *t, uint32_t *prim, float2 *uv) {
    hipLaunchKernelGGL(k_unpack_hits, dim3((n + 255u) / 256u), dim3(256), 0, s, sc, q, n, t, prim, uv);
}

// ------------------------------------------------------------------ closest hits without any allocation
// kl_trace_rays_noalloc: the queue entries, the segment counters, the traversal's stack overflow slab and a
// Counters block live in one file-scope device array instead of buffers allocated per call. Every GPU the library
// runs on holds its own copy of the array (at its own address) until the process ends; calls on one GPU share it.
constexpr uint32_t NOALLOC_BLOCKS = 256;                       // traversal workgroups of a no-allocation launch
constexpr size_t NOALLOC_OD = 0;                                // float4[MAX]
constexpr size_t NOALLOC_DT = NOALLOC_OD + (size_t)KL_NOALLOC_MAX_RAYS * 16;
constexpr size_t NOALLOC_UV = NOALLOC_DT + (size_t)KL_NOALLOC_MAX_RAYS * 16;   // float2[MAX]
constexpr size_t NOALLOC_CTL = NOALLOC_UV + (size_t)KL_NOALLOC_MAX_RAYS * 8;   // count, head: CTR_STRIDE words apart
constexpr size_t NOALLOC_CN = NOALLOC_CTL + 2 * CTR_STRIDE * 4;                // Counters
constexpr size_t NOALLOC_OVF = NOALLOC_CN + 256;                               // stack entries beyond the LDS part
constexpr size_t NOALLOC_BYTES = NOALLOC_OVF + (size_t)NOALLOC_BLOCKS * ISECT_WAVES * ISECT_OVF_STACK * 64u * 4u;
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

constexpr int NOALLOC_MAX_DEVICES = 64;
static unsigned char *g_scratch_ptr[NOALLOC_MAX_DEVICES];      // device address of `scratch` on each GPU, set once per device
// Calls on one GPU share that GPU's array. Each call records an event behind its last launch; a call that arrives on another
// stream than the one before it first makes its stream wait for that event, so calls from different contexts or threads run
// one after the other instead of racing on the array (no allocation: the event is made once per device, at set-up).
static hipEvent_t g_scratch_done[NOALLOC_MAX_DEVICES];
static hipStream_t g_scratch_last_stream[NOALLOC_MAX_DEVICES];
static bool g_scratch_used[NOALLOC_MAX_DEVICES];
static std::mutex g_scratch_mutex;

size_t noalloc_scratch_bytes() { return NOALLOC_BYTES; }

// With `device` current: looks up the address of this GPU's copy of `scratch` (the array's name is a host-side
// stand-in, not a device address) and keeps it. Called from kl_create, i.e. once per context; cheap when already known.
hipError_t noalloc_init(int device) {
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return hipErrorInvalidDevice;
    if (__atomic_load_n(&g_scratch_ptr[device], __ATOMIC_ACQUIRE)) return hipSuccess;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return hipSuccess;
    unsigned char *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    if (!g_scratch_done[device]) {
        e = hipEventCreateWithFlags(&g_scratch_done[device], hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    __atomic_store_n(&g_scratch_ptr[device], ptr, __ATOMIC_RELEASE);
    return hipSuccess;
}

// caller rays (n x 3 floats each) -> queue entries that start as misses; also the segment's count / head and the
// counters block, so that nothing of the previous call is left where this one reads
__global__ __launch_bounds__(256) void k_pack_rays(const float *o, const float *d, uint32_t n, RayQueue q, uint32_t *ctl, Counters *cn,
                                                   float max_dist) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        ctl[0] = n; ctl[CTR_STRIDE] = 0u;
        Counters z; memset(&z, 0, sizeof z); *cn = z;
    }
    if (i >= n) return;
    q.od[i] = make_float4(o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i]);
    q.dt[i] = make_float4(d[3 * i + 1], d[3 * i + 2], max_dist, __uint_as_float(REF_MISS));
    q.uv[i] = make_float2(0.0f, 0.0f);
}

// o, d, t, prim, uv: device pointers. f: a frame description with nseg = 1, segcap = n, all_uv = 1.
hipError_t launch_trace_noalloc(hipStream_t s, int device, const LaunchCfg &lc, const SceneDev &sc, const FrameDev &f, const float *o,
                                const float *d, uint32_t n, float *t, uint32_t *prim, float2 *uv, bool bvh) {
    if (device < 0 || device >= NOALLOC_MAX_DEVICES || n == 0 || n > KL_NOALLOC_MAX_RAYS || f.nseg != 1u || f.segcap < n) return hipErrorInvalidValue;
    unsigned char *base = __atomic_load_n(&g_scratch_ptr[device], __ATOMIC_ACQUIRE);
    if (!base) return hipErrorNotInitialized;
    // one call at a time on this GPU's array: queue behind the previous call if it went to another stream
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_used[device] && g_scratch_last_stream[device] != s) {
        const hipError_t we = hipStreamWaitEvent(s, g_scratch_done[device], 0);
        if (we != hipSuccess) return we;
    }
    RayQueue q = { (float4 *)(base + NOALLOC_OD), (float4 *)(base + NOALLOC_DT), nullptr, (float2 *)(base + NOALLOC_UV) };
    uint32_t *ctl = (uint32_t *)(base + NOALLOC_CTL);
    Counters *cn = (Counters *)(base + NOALLOC_CN);
    int *ovf = (int *)(base + NOALLOC_OVF);
    hipLaunchKernelGGL(k_pack_rays, dim3((n + 255u) / 256u), dim3(256), 0, s, o, d, n, q, ctl, cn, f.max_dist);
    if (bvh && scene_has_flat(sc)) launch_intersect_flat(s, lc, sc, f, q, ctl, cn, false, n);
    LaunchCfg small = lc;                       // the overflow slab in `scratch` is sized for NOALLOC_BLOCKS workgroups
    small.cus = NOALLOC_BLOCKS; small.isect_blocks_per_cu = 1;
    launch_intersect(s, small, sc, f, q, ctl, ctl + CTR_STRIDE, cn, ovf, bvh, 0, n);
    launch_unpack_hits(s, sc, q, n, t, prim, uv);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return le;
    g_scratch_used[device] = true; g_scratch_last_stream[device] = s;
    return hipEventRecord(g_scratch_done[device], s);
}

}  // namespace kl
