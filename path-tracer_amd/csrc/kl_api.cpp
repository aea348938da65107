// C-ABI implementation (include/Math/kl_render.h): context, uploads, the frame
// loop that sequences the wavefront kernels on one HIP stream, read-back.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "Math/kl_render.h"
#include "kl_bvh.hpp"
#include "kl_internal.hpp"
#include "kl_kernels.hpp"
#include "kl_lbvh.hpp"
#include "kl_math.hpp"
#include "kl_scene.hpp"
#include "kl_types.hpp"

using namespace kl;

namespace {
constexpr uint32_t KL_MAX_BOUNCES = 4096;
constexpr uint32_t KL_MAX_DIM = 32768;
constexpr uint32_t KL_MAX_FRAMES_IN_FLIGHT = 64;
constexpr uint32_t KL_BRUTE_AUTO_LIMIT = 64;      // KL_ACCEL_AUTO uses the flat loop up to this many primitives
constexpr size_t KL_MAX_PENDING_EVENTS = 8192;
constexpr uint32_t KL_QUEUE_SEGMENTS = 64;        // independent sub-queues, each with its own tail/head counter

enum { EV_INTERSECT = 0, EV_RENDER = 1, EV_SHADE = 2, EV_GENERATE = 3, EV_ACCUMULATE = 4 };
struct EventPair { hipEvent_t a, b; int kind; };   // kind: which launch (or whole render call) the pair brackets

template <typename T> struct DevBuf {
    T *p = nullptr; size_t n = 0;
    hipError_t alloc(size_t count) {
        if (count == n && p) return hipSuccess;
        release();
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e == hipSuccess) n = count; else p = nullptr;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};
}  // namespace

constexpr uint32_t KL_MAX_BATCH_LANES = 4;
struct BatchLane {
    hipStream_t stream = nullptr;
    DevBuf<float4> d_od[2], d_dt[2], d_tp[2], d_raw;     // the two ray queues (layout: RayQueue in kl_types.hpp) and the raw frames
    DevBuf<float2> d_uv[2];
    DevBuf<float4> p_od, p_dt;        // one-frame queue of the camera rays and their hits, shared by the frames of a batch
    DevBuf<float2> p_uv;
    DevBuf<uint32_t> d_qctl;          // count[0..B], head[0..B]
    DevBuf<int> d_overflow;           // traversal stack entries beyond the LDS part
    hipEvent_t done = nullptr;        // recorded after the lane's latest accumulate
    void release() {
        for (int k = 0; k < 2; k++) { d_od[k].release(); d_dt[k].release(); d_tp[k].release(); d_uv[k].release(); }
        d_raw.release(); d_qctl.release(); d_overflow.release();
        p_od.release(); p_dt.release(); p_uv.release();
    }
};

struct kl_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    LaunchCfg lc{256, 5};
    uint32_t width = 0, height = 0;
    uint32_t rank = 0, world = 1, band_rows = 1, local_rows = 0, npl = 0;
    kl_camera cam;
    uint32_t mode = KL_MODE_PT;
    kl_params params;
    uint32_t accum_mode = KL_ACCUM_MEAN;
    uint32_t frame = 0;
    bool camera_change = true;
    bool flat_separate = false;       // KL_FLAT_SEPARATE=1: flat primitives by their own pass instead of inside generate / shade (A/B switch)
    // scene
    bool have_scene = false;
    uint32_t n_prims = 0;
    SceneDev sc;
    kl_bvh_stats bvh_stats;
    DevBuf<float4> d_rec, d_shrec, d_flatq;
    std::vector<float4> h_flatq;            // host copy of the flat-quad table, and which quads it holds (for kl_refit_scene)
    std::vector<uint32_t> flat_quad_index;
    int32_t tree_root = NODE_DONE;          // root reference as the builder gave it (a node index, or a leaf)
    DevBuf<uchar4> d_tex[2];
    DevBuf<float4> d_envf;
    // wavefront state
    uint32_t slots = 0;               // frames in flight the buffers are sized for
    uint32_t slots_npl = 0;           // ... at this many pixels
    uint32_t auto_slots = 0, auto_slots_npl = 0;   // automatic batch size, decided once per image size
    uint32_t segcap = 0;              // entries per queue segment
    uint32_t segcap1 = 0;             // ... of the one-frame camera-ray queue
    // Batches are independent until they accumulate, so consecutive batches run on different HIP streams
    // ("lanes"), each with its own queues: while one batch drains the long tail of a traversal launch, the
    // next one already fills the CUs that fell idle. Lane 0 is the context's own stream.
    BatchLane lanes[KL_MAX_BATCH_LANES];
    uint32_t n_lanes = 0;             // lanes whose buffers are allocated
    hipEvent_t ev_start = nullptr;    // everything queued on the context stream before a render call
    DevBuf<float4> d_acc, d_last_raw;
    DevBuf<Counters> d_counters;
    DevBuf<uchar4> d_resolve;
    // bookkeeping
    uint64_t frames_total = 0, paths_total = 0;
    std::vector<hipEvent_t> free_events;
    std::vector<EventPair> pending;
    kl_timing timing{};
    std::string err;
};

namespace {

kl_status fail(kl_ctx *c, kl_status s, const std::string &msg) { if (c) c->err = msg; return s; }

// No exception crosses the C ABI: host-side allocations (std::vector, std::string) of an entry point's body are
// turned into a status here.
template <typename F> kl_status guarded(kl_ctx *c, F body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc &) { try { if (c) c->err = "host memory exhausted"; } catch (...) {} return KL_E_NOMEM; }
    catch (const std::exception &e) { try { if (c) c->err = e.what(); } catch (...) {} return KL_E_INVALID; }
    catch (...) { return KL_E_INVALID; }
}

#define KL_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, e_ == hipErrorOutOfMemory ? KL_E_NOMEM : KL_E_DEVICE,                 \
                        std::string(#call) + ": " + hipGetErrorString(e_));                        \
    } while (0)

void compute_local(kl_ctx *c) {
    uint32_t rows = 0;
    uint32_t bands = (c->height + c->band_rows - 1) / c->band_rows;
    for (uint32_t b = c->rank; b < bands; b += c->world) {
        uint32_t y0 = b * c->band_rows, y1 = std::min(c->height, y0 + c->band_rows);
        rows += y1 - y0;
    }
    c->local_rows = rows;
    c->npl = rows * c->width;
}

// sub-samples per pixel and frame: SPP x SPP in the AO and path-traced modes (RawFrame.glsl:434-436,446-448), one in ZDepth (:458)
uint32_t sub_samples(const kl_ctx *c) {
    const uint32_t s = std::max(1u, c->params.spp_grid);
    return c->mode == KL_MODE_Z ? 1u : s * s;
}

uint32_t choose_slots(kl_ctx *c) {
    uint32_t f = c->params.frames_in_flight;
    if (f == 0 && c->auto_slots && c->auto_slots_npl == c->npl) return c->auto_slots;    // decided once per image size
    if (f == 0) {
        // Large batches: every launch ends in a tail of long rays during which the CUs drain, and the tail costs
        // the same whatever the launch holds, so a batch takes as many frames as memory comfortably allows
        // (128 bytes of queue, hit and raw-frame storage per path; at most a quarter of the free memory
        // for all lanes together, and at most 128 Mi paths).
        const uint64_t npl = std::max<uint32_t>(c->npl, 1);
        uint64_t paths = 128ull << 20;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) {
            uint64_t held = 0;                               // what the lanes already hold counts as available
            for (uint32_t l = 0; l < KL_MAX_BATCH_LANES; l++) held += (uint64_t)c->lanes[l].d_raw.n * 128u;
            paths = std::min<uint64_t>(paths, ((uint64_t)free_b + held) / 4u / (128u * 2u));
        }
        f = (uint32_t)std::min<uint64_t>(KL_MAX_FRAMES_IN_FLIGHT, std::max<uint64_t>(1, paths / npl));
        while (f > 1 && (uint64_t)f * npl >= (1ull << 31)) f--;
        c->auto_slots = f; c->auto_slots_npl = c->npl;
    }
    // a caller's frames_in_flight counts frames: sub_samples() slots each
    if (c->params.frames_in_flight != 0) f = f * sub_samples(c);
    return std::max(1u, std::min(f, KL_MAX_FRAMES_IN_FLIGHT));
}

uint32_t choose_lanes(kl_ctx *c) {
    uint32_t n = c->params.batch_lanes;
    // automatic: one. Two lanes used to pay (+2.8 %: while one batch drained the tail of a traversal launch or sat in shade's
    // half-empty second round of blocks, the other filled the CUs). Since shade's grid is made of many small blocks a single
    // lane is as fast or faster at every batch size (1080p, 4 / 8 / 16 / 32 / 64 frames: 1764 / 1934 / 2010 / 2044 / 2058 against
    // 1723 / 1912 / 1997 / 2045 / 2033 M samples/s) and holds one set of queues instead of two.
    if (n == 0) n = 1u;
    return std::min(n, KL_MAX_BATCH_LANES);
}
// frames per batch for a call that renders n frames on `lanes` lanes: as many as memory allows (choose_slots), but
// with automatic sizing and several lanes at most 48 (beyond that two concurrent batches were measured slower than
// one) and an equal share for every lane, so that no lane idles at the end
uint32_t frames_per_batch(kl_ctx *c, uint32_t n, uint32_t lanes) {
    uint32_t cap = std::max(1u, choose_slots(c) / sub_samples(c));      // a frame takes one slot per sub-sample
    if (c->params.frames_in_flight != 0 || lanes < 2 || n < 2) return std::max(1u, std::min(cap, n));
    cap = std::min(cap, 48u);
    uint32_t nb = (n + cap - 1) / cap;
    nb = (nb + lanes - 1) / lanes * lanes;
    return std::max(1u, (n + nb - 1) / nb);
}

// buffers of the first `lanes` batch lanes (and their streams and events), for batches of up to `frames` frames.
// Capacity only grows while the image size stays: a one-frame call does not shrink what a 64-frame call set up,
// and does not allocate for 64 frames either.
kl_status ensure_buffers(kl_ctx *c, uint32_t lanes = 1, uint32_t frames = 1) {
    KL_HIP(c, hipSetDevice(c->device));
    const uint32_t s2 = sub_samples(c);
    uint32_t slots = std::max(s2, std::min(choose_slots(c) / s2 * s2, frames * s2));      // whole frames: s2 slots each
    if (c->slots_npl == c->npl && c->slots > slots && c->slots <= std::max(s2, choose_slots(c))) slots = c->slots;
    const size_t cap = (size_t)slots * c->npl;
    if (cap >= (1ull << 31)) return fail(c, KL_E_LIMIT, "frames_in_flight x pixels exceeds 2^31 paths");
    c->slots = slots; c->slots_npl = c->npl;
    const size_t groups = (cap + 63) / 64;
    c->segcap = (uint32_t)((groups + KL_QUEUE_SEGMENTS - 1) / KL_QUEUE_SEGMENTS * 64);
    const size_t qcap = (size_t)c->segcap * KL_QUEUE_SEGMENTS;
    c->segcap1 = (uint32_t)((((size_t)c->npl * s2 + 63) / 64 + KL_QUEUE_SEGMENTS - 1) / KL_QUEUE_SEGMENTS * 64);   // one frame's camera rays
    const size_t qcap1 = std::max<size_t>((size_t)c->segcap1 * KL_QUEUE_SEGMENTS, 1);
    lanes = std::max(1u, std::min(lanes, KL_MAX_BATCH_LANES));
    for (uint32_t l = 0; l < lanes; l++) {
        BatchLane &L = c->lanes[l];
        if (!L.stream) {
            if (l == 0) L.stream = c->stream;
            else KL_HIP(c, hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        }
        if (!L.done) KL_HIP(c, hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
        for (int k = 0; k < 2; k++) {
            KL_HIP(c, L.d_od[k].alloc(qcap)); KL_HIP(c, L.d_dt[k].alloc(qcap)); KL_HIP(c, L.d_tp[k].alloc(qcap)); KL_HIP(c, L.d_uv[k].alloc(qcap));
        }
        KL_HIP(c, L.d_raw.alloc(cap));
        KL_HIP(c, L.d_qctl.alloc(2 * ((size_t)c->params.max_bounces + 3) * KL_QUEUE_SEGMENTS * CTR_STRIDE));
        KL_HIP(c, L.p_od.alloc(qcap1)); KL_HIP(c, L.p_dt.alloc(qcap1)); KL_HIP(c, L.p_uv.alloc(qcap1));
        KL_HIP(c, L.d_overflow.alloc(intersect_overflow_ints(c->lc)));
    }
    c->n_lanes = std::max(c->n_lanes, lanes);
    if (!c->ev_start) KL_HIP(c, hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming));
    bool fresh = c->d_acc.n != c->npl;
    KL_HIP(c, c->d_acc.alloc(c->npl));
    KL_HIP(c, c->d_last_raw.alloc(c->npl));
    if (fresh && c->npl) {
        KL_HIP(c, hipMemsetAsync(c->d_acc.p, 0, c->npl * sizeof(float4), c->stream));
        KL_HIP(c, hipMemsetAsync(c->d_last_raw.p, 0, c->npl * sizeof(float4), c->stream));
    }
    if (!c->d_counters.p) {
        KL_HIP(c, c->d_counters.alloc(1));
        KL_HIP(c, hipMemsetAsync(c->d_counters.p, 0, sizeof(Counters), c->stream));
    }
    return KL_OK;
}

hipEvent_t get_event(kl_ctx *c) {
    if (!c->free_events.empty()) { hipEvent_t e = c->free_events.back(); c->free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
void drain_events(kl_ctx *c) {
    if (c->pending.empty()) return;
    for (const EventPair &p : c->pending) {
        float ms = 0.0f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            switch (p.kind) {
                case EV_INTERSECT: c->timing.intersect_ms += ms; c->timing.intersect_launches++; break;
                case EV_RENDER: c->timing.render_ms += ms; c->timing.render_calls++; break;
                case EV_SHADE: c->timing.shade_ms += ms; c->timing.shade_launches++; break;
                case EV_GENERATE: c->timing.generate_ms += ms; c->timing.generate_launches++; break;
                default: c->timing.accumulate_ms += ms; c->timing.accumulate_launches++; break;
            }
        }
        c->free_events.push_back(p.a); c->free_events.push_back(p.b);
    }
    c->pending.clear();
}

void soft_reset(kl_ctx *c) { c->frame = 0; c->camera_change = true; }

// `nframes` frames from first_frame on: sub_samples() slots each
FrameDev make_frame(const kl_ctx *c, uint32_t first_frame, uint32_t nframes) {
    FrameDev f; std::memset(&f, 0, sizeof f);
    // projection frame of RawFrame.glsl:418-420, evaluated once on the host with the same fp32 operations
    v3 pos = V(c->cam.pos[0], c->cam.pos[1], c->cam.pos[2]);
    v3 front = V(c->cam.front[0], c->cam.front[1], c->cam.front[2]);
    v3 up = V(c->cam.up[0], c->cam.up[1], c->cam.up[2]);
    v3 pc = pos + front * c->cam.focal;
    v3 pu = normalize(cross(front, up)) * c->cam.sensor;
    v3 pv = normalize(cross(pu, front)) * (c->cam.sensor / 1.0f);
    f.pos[0] = pos.x; f.pos[1] = pos.y; f.pos[2] = pos.z;
    f.pc[0] = pc.x; f.pc[1] = pc.y; f.pc[2] = pc.z;
    f.pu[0] = pu.x; f.pu[1] = pu.y; f.pu[2] = pu.z;
    f.pv[0] = pv.x; f.pv[1] = pv.y; f.pv[2] = pv.z;
    f.width = c->width; f.height = c->height; f.npl = c->npl;
    f.rank = c->rank; f.world = c->world; f.band_rows = c->band_rows;
    f.mode = c->mode; f.first_frame = first_frame;
    f.spp2 = sub_samples(c); f.spp = f.spp2 == 1u ? 1u : std::max(1u, c->params.spp_grid); f.nslots = nframes * f.spp2;
    f.max_bounces = c->params.max_bounces; f.max_frames = c->params.max_frames;
    f.eps = c->params.eps; f.max_dist = c->params.max_dist;
    f.ao_length = c->params.ao_length; f.hdri_strength = c->params.hdri_strength;
    f.nseg = KL_QUEUE_SEGMENTS; f.segcap = c->segcap;
    fastdiv_make(std::max(c->npl, 1u), f.div_npl); fastdiv_make(c->width, f.div_width); fastdiv_make(c->band_rows, f.div_band);
    fastdiv_make(f.spp2, f.div_spp2); fastdiv_make(f.spp, f.div_spp);
    return f;
}

bool use_bvh(const kl_ctx *c) {
    if (c->sc.root == NODE_DONE) return false;
    if (c->params.accel == KL_ACCEL_BVH) return true;
    if (c->params.accel == KL_ACCEL_BRUTE) return false;
    return c->n_prims > KL_BRUTE_AUTO_LIMIT;
}

// `flat_done`: the kernel that produced the rays has already written their flat-primitive hits into the queue entries
kl_status timed_intersect(kl_ctx *c, BatchLane &L, const FrameDev &f, RayQueue q, uint32_t *count, uint32_t *head, uint32_t max_rays,
                          bool flat_done) {
    hipEvent_t a = nullptr, b = nullptr;
    const bool bvh = use_bvh(c);
    if (bvh && scene_has_flat(c->sc) && !flat_done)
        launch_intersect_flat(L.stream, c->lc, c->sc, f, q, count, c->d_counters.p, c->params.collect_stats == 1, max_rays);
    // the events bracket the traversal kernel only (k_intersect_bvh, or the flat loop for small scenes)
    if (c->params.profile_kernels) {
        if (c->pending.size() >= KL_MAX_PENDING_EVENTS) drain_events(c);
        a = get_event(c); b = get_event(c);
        if (a && b) KL_HIP(c, hipEventRecord(a, L.stream));
    }
    launch_intersect(L.stream, c->lc, c->sc, f, q, count, head, c->d_counters.p, L.d_overflow.p, bvh,
                     (int)std::min(c->params.collect_stats, 2u), max_rays, flat_done);
    if (a && b) { KL_HIP(c, hipEventRecord(b, L.stream)); c->pending.push_back({a, b, EV_INTERSECT}); }
    return KL_OK;
}

// profile_kernels: HIP events around one launch of the given kind on the lane's stream (begin before, end after)
struct Bracket {
    kl_ctx *c; hipStream_t s; int kind; hipEvent_t a = nullptr, b = nullptr;
    Bracket(kl_ctx *ctx, hipStream_t stream, int k) : c(ctx), s(stream), kind(k) {
        if (!c->params.profile_kernels) return;
        if (c->pending.size() >= KL_MAX_PENDING_EVENTS) drain_events(c);
        a = get_event(c); b = get_event(c);
        if (a && b) (void)hipEventRecord(a, s);
    }
    ~Bracket() { if (a && b && hipEventRecord(b, s) == hipSuccess) c->pending.push_back({a, b, kind}); }
};

// one wavefront batch on lane L: frames [first, first+nframes) traced together, then accumulated in order.
// `after` (may be null): the accumulate of the batch before this one, which this batch's accumulate follows.
kl_status render_batch(kl_ctx *c, BatchLane &L, hipEvent_t after, uint32_t first, uint32_t nframes, bool camera_change) {
    FrameDev f = make_frame(c, first, nframes);
    const uint32_t nslots = f.nslots;
    const uint32_t B = c->params.max_bounces;
    // count[b] / head[b]: one word per queue segment and bounce
    const size_t S = (size_t)KL_QUEUE_SEGMENTS * CTR_STRIDE;      // words per bounce level
    uint32_t *count = L.d_qctl.p, *head = L.d_qctl.p + (size_t)(B + 3) * S;
    const uint32_t max_rays = nslots * c->npl;
    RayQueue q[2] = { {L.d_od[0].p, L.d_dt[0].p, L.d_tp[0].p, L.d_uv[0].p}, {L.d_od[1].p, L.d_dt[1].p, L.d_tp[1].p, L.d_uv[1].p} };
    const bool fused = c->params.fused_compaction != 0;
    const uint32_t levels = c->mode == KL_MODE_PT ? B : (c->mode == KL_MODE_AO ? 2u : 1u);
    KL_HIP(c, hipMemsetAsync(count, 0, (size_t)(levels + 1) * S * sizeof(uint32_t), L.stream));
    KL_HIP(c, hipMemsetAsync(head, 0, (size_t)(levels + 1) * S * sizeof(uint32_t), L.stream));
    // Every frame of the batch starts with the same camera ray per pixel: unless told otherwise, that ray is traced
    // once (a one-frame queue of its own) and generate copies its hit to the paths of all frames.
    const bool shared = nframes > 1 && c->params.trace_primary_per_frame == 0;
    // The flat primitives (scene-spanning ones kept out of the tree) are tested by the kernel that makes a ray, generate
    // or shade, while the ray is in registers, and the hit travels in the ray's queue entry; only the stand-alone
    // compaction moves rays afterwards and needs the separate flat pass.
    const bool pflat = use_bvh(c) && scene_has_flat(c->sc) && fused && !c->flat_separate;
    if (shared) {
        FrameDev f1 = f; f1.nslots = f.spp2; f1.segcap = c->segcap1;       // one frame's worth of camera rays
        uint32_t *count1 = count + (size_t)(B + 2) * S, *head1 = head + (size_t)(B + 2) * S;
        KL_HIP(c, hipMemsetAsync(count1, 0, S * sizeof(uint32_t), L.stream));
        KL_HIP(c, hipMemsetAsync(head1, 0, S * sizeof(uint32_t), L.stream));
        RayQueue q1 = { L.p_od.p, L.p_dt.p, nullptr, L.p_uv.p };
        { Bracket t(c, L.stream, EV_GENERATE); launch_generate(L.stream, c->lc, c->sc, f1, q1, count1, pflat); }
        kl_status s1 = timed_intersect(c, L, f1, q1, count1, head1, c->npl * f.spp2, pflat);
        if (s1 != KL_OK) return s1;
        { Bracket t(c, L.stream, EV_GENERATE); launch_generate(L.stream, c->lc, c->sc, f, q[0], count, false, q1, c->segcap1); }
    } else {
        Bracket t(c, L.stream, EV_GENERATE);
        launch_generate(L.stream, c->lc, c->sc, f, q[0], count, pflat);
    }
    for (uint32_t b = 0; b < levels; b++) {
        RayQueue qi = q[b & 1u], qo = q[(b + 1u) & 1u];
        if (!(shared && b == 0)) {
            kl_status s = timed_intersect(c, L, f, qi, count + (size_t)b * S, head + (size_t)b * S, max_rays, pflat);
            if (s != KL_OK) return s;
        }
        {
            Bracket t(c, L.stream, EV_SHADE);
            launch_shade(L.stream, c->lc, c->sc, f, qi, count + (size_t)b * S, qo, count + (size_t)(b + 1) * S, L.d_raw.p, b, fused, max_rays,
                         pflat && b + 1 < levels);
        }
        if (!fused && b + 1 < levels) launch_compact(L.stream, c->lc, f, qi, count + (size_t)b * S, qo, count + (size_t)(b + 1) * S, max_rays);
    }
    if (after) KL_HIP(c, hipStreamWaitEvent(L.stream, after, 0));
    { Bracket t(c, L.stream, EV_ACCUMULATE); launch_accumulate(L.stream, c->lc, f, L.d_raw.p, c->d_acc.p, c->d_last_raw.p, camera_change, c->accum_mode); }
    KL_HIP(c, hipGetLastError());
    return KL_OK;
}

// splits [0, n) over up to `threads` host threads (record packing of million-primitive scenes)
template <typename F> void parallel_for(uint32_t n, unsigned threads, F body) {
    if (threads <= 1 || n < 65536u) { body(0u, n); return; }
    std::vector<std::thread> pool;
    const uint32_t per = (n + threads - 1) / threads;
    for (unsigned k = 0; k < threads; k++) {
        const uint32_t b = std::min<uint64_t>((uint64_t)k * per, n), e = std::min<uint64_t>((uint64_t)b + per, n);
        if (b < e) pool.emplace_back([=, &body]() { body(b, e); });
    }
    for (std::thread &t : pool) t.join();
}

template <typename T> kl_status upload(kl_ctx *c, DevBuf<T> &buf, const void *src, size_t count) {
    KL_HIP(c, buf.alloc(std::max<size_t>(count, 1)));
    if (count) KL_HIP(c, hipMemcpyAsync(buf.p, src, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return KL_OK;
}

// Ray-independent terms of the quad test (RawFrame.glsl:219-244) for a flat quad, in the test's own operations:
// F0..F3 of the flat-quad table (kl_types.hpp)
void flat_quad_terms(const kl_quad &x, float4 *f) {
    auto bits = [](uint32_t v) { float r; std::memcpy(&r, &v, 4); return r; };
    v3 v0 = V(x.v0[0], x.v0[1], x.v0[2]);
    v3 a = V(x.v1[0], x.v1[1], x.v1[2]) - v0, b = V(x.v3[0], x.v3[1], x.v3[2]) - v0, cc = V(x.v2[0], x.v2[1], x.v2[2]) - v0;
    v3 nor = cross(a, b);
    v3 mor = V(std::fabs(nor.x), std::fabs(nor.y), std::fabs(nor.z));
    int id = (mor.x > mor.y && mor.x > mor.z) ? 0 : ((mor.y > mor.z) ? 1 : 2);
    int idu = id == 0 ? 1 : (id == 1 ? 2 : 0), idv = id == 0 ? 2 : (id == 1 ? 0 : 1);
    float kax = comp(a, idu), kay = comp(a, idv), kbx = comp(b, idu), kby = comp(b, idv);
    float kcx = comp(cc, idu), kcy = comp(cc, idv);
    float kgx = kcx - kbx - kax, kgy = kcy - kby - kay;
    float k2 = cross2d(kcx - kbx, kcy - kby, kax, kay);
    f[0] = make_float4(v0.x, v0.y, v0.z, comp(nor, id));
    f[1] = make_float4(nor.x, nor.y, nor.z, k2);
    f[2] = make_float4(kax, kay, kbx, kby);
    f[3] = make_float4(kgx, kgy, bits((uint32_t)idu | ((uint32_t)idv << 2)), 0.0f);
}

}  // namespace

namespace kl {
kl_status ctx_view(kl_ctx *c, CtxView *out) {
    if (!c || !out) return KL_E_INVALID;
    kl_status s = ensure_buffers(c);
    if (s != KL_OK) return s;
    out->device = c->device; out->stream = c->stream; out->acc = c->d_acc.p;
    out->width = c->width; out->height = c->height; out->npl = c->npl; out->local_rows = c->local_rows;
    out->rank = c->rank; out->world = c->world; out->band_rows = c->band_rows;
    out->frame = c->frame; out->max_frames = c->params.max_frames; out->accum_mode = c->accum_mode; out->cus = c->lc.cus;
    return KL_OK;
}
}  // namespace kl

extern "C" {

uint32_t kl_abi_version(void) { return 2; }

const char *kl_status_string(kl_status s) {
    switch (s) {
        case KL_OK: return "ok";
        case KL_E_INVALID: return "invalid argument";
        case KL_E_DEVICE: return "HIP runtime error";
        case KL_E_NOMEM: return "out of memory";
        case KL_E_STATE: return "invalid state";
        case KL_E_LIMIT: return "size beyond a documented maximum";
        default: return "unknown status";
    }
}
const char *kl_last_error(kl_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

kl_status kl_default_params(kl_params *p) {
    if (!p) return KL_E_INVALID;
    std::memset(p, 0, sizeof *p);
    p->max_bounces = 32; p->max_frames = 30; p->eps = 0.001f; p->max_dist = 5000.0f;
    p->ao_length = 1.0f; p->hdri_strength = 1.0f; p->accel = KL_ACCEL_AUTO;
    p->spp_grid = 1; p->frames_in_flight = 0; p->fused_compaction = 1; p->collect_stats = 0; p->profile_kernels = 0; p->batch_lanes = 0; p->trace_primary_per_frame = 0;
    return KL_OK;
}
kl_status kl_default_camera(kl_camera *cam) {
    if (!cam) return KL_E_INVALID;
    const kl_camera d = { {0, 1, 5}, {0, 0, -1}, {0, 1, 0}, 0.05f, 0.036f };
    *cam = d;
    return KL_OK;
}

static kl_status kl_create_body(kl_ctx **out, int32_t device, uint32_t width, uint32_t height) {
    if (!out) return KL_E_INVALID;
    *out = nullptr;
    if (width == 0 || height == 0 || width > KL_MAX_DIM || height > KL_MAX_DIM) return KL_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return KL_E_DEVICE;
    kl_ctx *c = new (std::nothrow) kl_ctx();
    if (!c) return KL_E_NOMEM;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c; return KL_E_DEVICE;
    }
    if (noalloc_init(device) != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return KL_E_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->lc.cus = (uint32_t)prop.multiProcessorCount;
#ifndef KL_ISECT_BLOCKS_PER_CU
#define KL_ISECT_BLOCKS_PER_CU 6     // 4 waves x (2.5 KiB stack + 4.1 KiB fetch staging) -> six workgroups share a CU's 160 KiB LDS
#endif
    c->lc.isect_blocks_per_cu = KL_ISECT_BLOCKS_PER_CU;
    // Shade's grid: its blocks share a segment's rays equally, and a CU holds six of them (74 VGPRs). A grid of 8 per CU runs as
    // one full round and a second one a third full (21.1 ms per 64-spp step); many small blocks even that out (17.9 ms at 48
    // per CU and beyond). With two batch lanes the other lane's traversal used to fill that gap, so the step time is the same.
    c->lc.shade_blocks_per_cu = 48;
    if (const char *e = std::getenv("KL_SHADE_BLOCKS_PER_CU")) c->lc.shade_blocks_per_cu = (uint32_t)std::max(1, std::min(64, std::atoi(e)));
    if (const char *e = std::getenv("KL_FLAT_SEPARATE")) c->flat_separate = e[0] == '1';
    c->width = width; c->height = height;
    kl_default_camera(&c->cam);
    kl_default_params(&c->params);
    std::memset(&c->sc, 0, sizeof c->sc);
    c->sc.root = NODE_DONE;
    std::memset(&c->bvh_stats, 0, sizeof c->bvh_stats);
    compute_local(c);
    // 1x1 white textures until the caller uploads real ones
    const uint8_t white[4] = {255, 255, 255, 255};
    *out = c;
    kl_status s = kl_upload_texture(c, KL_TEX_HDRI, white, 1, 1, 4);
    if (s == KL_OK) s = kl_upload_texture(c, KL_TEX_ALBEDO, white, 1, 1, 4);
    if (s != KL_OK) { kl_destroy(c); *out = nullptr; }
    return s;
}
kl_status kl_create(kl_ctx **out, int32_t device, uint32_t width, uint32_t height) {
    return guarded(nullptr, [&]() { return kl_create_body(out, device, width, height); });
}

kl_status kl_destroy(kl_ctx *c) {
    if (!c) return KL_E_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (const EventPair &p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (hipEvent_t e : c->free_events) (void)hipEventDestroy(e);
    c->d_rec.release(); c->d_shrec.release(); c->d_flatq.release();
    for (int k = 0; k < 2; k++) c->d_tex[k].release();
    c->d_envf.release();
    for (uint32_t l = 0; l < KL_MAX_BATCH_LANES; l++) {
        BatchLane &L = c->lanes[l];
        if (l > 0 && L.stream) { (void)hipStreamSynchronize(L.stream); (void)hipStreamDestroy(L.stream); }
        if (L.done) (void)hipEventDestroy(L.done);
        L.release();
    }
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    c->d_acc.release(); c->d_last_raw.release();
    c->d_counters.release(); c->d_resolve.release();
    (void)hipStreamDestroy(c->stream);
    delete c;
    return KL_OK;
}

kl_status kl_resize(kl_ctx *c, uint32_t width, uint32_t height) {
    if (!c || width == 0 || height == 0 || width > KL_MAX_DIM || height > KL_MAX_DIM) return fail(c, KL_E_INVALID, "bad size");
    KL_HIP(c, hipSetDevice(c->device));
    KL_HIP(c, hipStreamSynchronize(c->stream));
    c->width = width; c->height = height;
    compute_local(c);
    c->d_acc.release();          // new attachments, as FBT::f_resize does
    c->frame = 0;                // :72 resets iFrame but not iCameraChange
    return KL_OK;
}
kl_status kl_recompile(kl_ctx *c) { if (!c) return KL_E_INVALID; soft_reset(c); return KL_OK; }
kl_status kl_reset(kl_ctx *c) {
    if (!c) return KL_E_INVALID;
    soft_reset(c);
    KL_HIP(c, hipSetDevice(c->device));
    // a fresh accumulation: frame 0 then yields exactly the raw frame even if a stale texel was non-finite
    if (c->d_acc.p) KL_HIP(c, hipMemsetAsync(c->d_acc.p, 0, c->d_acc.n * sizeof(float4), c->stream));
    return KL_OK;
}
kl_status kl_get_frame(kl_ctx *c, uint32_t *frame) { if (!c || !frame) return KL_E_INVALID; *frame = c->frame; return KL_OK; }

kl_status kl_set_camera(kl_ctx *c, const kl_camera *cam) {
    if (!c || !cam) return fail(c, KL_E_INVALID, "null camera");
    c->cam = *cam;
    return kl_reset(c);
}
kl_status kl_set_mode(kl_ctx *c, uint32_t mode) {
    if (!c || mode > KL_MODE_Z) return fail(c, KL_E_INVALID, "mode must be 0 (AO), 1 (path traced) or 2 (ZDepth)");
    c->mode = mode;
    return kl_reset(c);
}
kl_status kl_set_params(kl_ctx *c, const kl_params *p) {
    if (!c || !p) return fail(c, KL_E_INVALID, "null params");
    if (p->max_bounces == 0 || p->max_bounces > KL_MAX_BOUNCES) return fail(c, KL_E_LIMIT, "max_bounces must be 1..4096");
    if (p->frames_in_flight > KL_MAX_FRAMES_IN_FLIGHT) return fail(c, KL_E_LIMIT, "frames_in_flight must be <= 64");
    if (p->accel > KL_ACCEL_BVH) return fail(c, KL_E_INVALID, "bad accel");
    if (p->bvh_builder > KL_BUILD_GPU_LBVH) return fail(c, KL_E_INVALID, "bad bvh_builder");
    if (p->batch_lanes > KL_MAX_BATCH_LANES) return fail(c, KL_E_LIMIT, "batch_lanes must be <= 4");
    if (p->spp_grid > 8) return fail(c, KL_E_LIMIT, "spp_grid must be <= 8");
    if (!(p->max_dist > 0.0f) || !(p->eps >= 0.0f)) return fail(c, KL_E_INVALID, "bad eps/max_dist");
    KL_HIP(c, hipSetDevice(c->device));
    KL_HIP(c, hipStreamSynchronize(c->stream));
    c->params = *p;
    return KL_OK;
}
kl_status kl_get_params(kl_ctx *c, kl_params *p) { if (!c || !p) return KL_E_INVALID; *p = c->params; return KL_OK; }

static kl_status kl_upload_texture_body(kl_ctx *c, uint32_t which, const uint8_t *px, uint32_t w, uint32_t h, uint32_t ch) {
    if (!c || which > KL_TEX_ALBEDO || !px || !w || !h || w > KL_MAX_DIM || h > KL_MAX_DIM || (ch != 1 && ch != 3 && ch != 4))
        return fail(c, KL_E_INVALID, "texture must be 8-bit with 1, 3 or 4 channels");
    KL_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)w * h;
    std::vector<uchar4> tmp(n);
    for (size_t i = 0; i < n; i++) {
        if (ch == 4) tmp[i] = make_uchar4(px[4 * i], px[4 * i + 1], px[4 * i + 2], px[4 * i + 3]);
        else if (ch == 3) tmp[i] = make_uchar4(px[3 * i], px[3 * i + 1], px[3 * i + 2], 255);
        else tmp[i] = make_uchar4(px[i], 0, 0, 255);       // GL_RED upload: g = b = 0 (Texture.cpp:13-14)
    }
    KL_HIP(c, hipStreamSynchronize(c->stream));
    KL_HIP(c, c->d_tex[which].alloc(n));
    KL_HIP(c, hipMemcpy(c->d_tex[which].p, tmp.data(), n * sizeof(uchar4), hipMemcpyHostToDevice));
    c->sc.tex[which] = c->d_tex[which].p; c->sc.tex_w[which] = w; c->sc.tex_h[which] = h;
    return KL_OK;
}
kl_status kl_upload_texture(kl_ctx *c, uint32_t which, const uint8_t *px, uint32_t w, uint32_t h, uint32_t ch) {
    return guarded(c, [&]() { return kl_upload_texture_body(c, which, px, w, h, ch); });
}

static kl_status kl_upload_env_float_body(kl_ctx *c, const float *px, uint32_t w, uint32_t h, uint32_t ch) {
    if (!c) return KL_E_INVALID;
    KL_HIP(c, hipSetDevice(c->device));
    KL_HIP(c, hipStreamSynchronize(c->stream));
    if (!px) {                                   // back to the 8-bit environment
        c->d_envf.release(); c->sc.envf = nullptr; c->sc.envf_w = c->sc.envf_h = 0;
        return kl_reset(c);
    }
    if (!w || !h || w > KL_MAX_DIM || h > KL_MAX_DIM || (ch != 3 && ch != 4))
        return fail(c, KL_E_INVALID, "float environment must have 3 or 4 channels");
    const size_t n = (size_t)w * h;
    std::vector<float4> tmp(n);
    for (size_t i = 0; i < n; i++) tmp[i] = make_float4(px[ch * i], px[ch * i + 1], px[ch * i + 2], 1.0f);
    KL_HIP(c, c->d_envf.alloc(n));
    KL_HIP(c, hipMemcpy(c->d_envf.p, tmp.data(), n * sizeof(float4), hipMemcpyHostToDevice));
    c->sc.envf = c->d_envf.p; c->sc.envf_w = w; c->sc.envf_h = h;
    return kl_reset(c);
}
kl_status kl_upload_env_float(kl_ctx *c, const float *px, uint32_t w, uint32_t h, uint32_t ch) {
    return guarded(c, [&]() { return kl_upload_env_float_body(c, px, w, h, ch); });
}

static kl_status kl_upload_scene_body(kl_ctx *c, const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq,
                          const kl_tri *t, uint32_t nt, const kl_material *m, uint32_t nm) {
    if (!c || (ns && !s) || (nq && !q) || (nt && !t) || !m || nm == 0) return fail(c, KL_E_INVALID, "null scene array");
    if ((uint64_t)ns + nq + nt >= MAX_RECORDS / 2)
        return fail(c, KL_E_LIMIT, "at most 2^25 primitives (nodes + primitives share 2^26 records)");
    for (uint32_t i = 0; i < ns; i++) if (s[i].material >= nm) return fail(c, KL_E_INVALID, "sphere material index out of range");
    for (uint32_t i = 0; i < nq; i++) if (q[i].material >= nm) return fail(c, KL_E_INVALID, "quad material index out of range");
    for (uint32_t i = 0; i < nt; i++) if (t[i].material >= nm) return fail(c, KL_E_INVALID, "triangle material index out of range");
    KL_HIP(c, hipSetDevice(c->device));
    KL_HIP(c, hipStreamSynchronize(c->stream));
    HostBVH bvh;
    unsigned threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    // scenes small enough for the flat loop keep everything in one list; otherwise primitives whose box is
    // >= 2 % of the scene's (at most 32 of them: walls, floors, sky domes) are tested outside the tree
    const bool small = ns + nq + nt <= KL_BRUTE_AUTO_LIMIT;
    uint32_t builder = KL_BUILD_HOST_SAH;
    GpuScene packed;
    bool on_gpu = false;                // records already written on the device by the builder
    if (c->params.bvh_builder == KL_BUILD_GPU_LBVH && !small) {
        std::string why;
        // the GPU builder also writes the records there (KL_PACK_HOST=1: pack them here instead, from its tree)
        const char *ph = std::getenv("KL_PACK_HOST");
        const bool pack_here = ph && ph[0] == '1';
        const LbvhResult r = build_bvh_gpu(c->stream, s, ns, q, nq, t, nt, bvh, 0.02f, 32, why, m, nm, pack_here ? nullptr : &packed);
        if (r == LBVH_HIP_ERROR) return fail(c, KL_E_DEVICE, "GPU BVH build: " + why);
        if (r == LBVH_OK) { builder = KL_BUILD_GPU_LBVH; on_gpu = !pack_here; }      // LBVH_TOO_DEEP: the host builder below
    }
    if (builder == KL_BUILD_HOST_SAH) {
        build_bvh(s, ns, q, nq, t, nt, bvh, threads, small ? 0.0f : 0.02f, 32);
        collapse_bvh4(bvh);
    }
    // one array of 64-byte records: nodes, then spheres, quads, triangles in leaf order (layout: kl_types.hpp)
    auto bits = [](uint32_t v) { float f; std::memcpy(&f, &v, 4); return f; };
    const uint32_t n_nodes = on_gpu ? bvh.stats.nodes : (uint32_t)(bvh.nodes4.size() / 16);
    const uint32_t base[3] = { n_nodes, n_nodes + ns, n_nodes + ns + nq };
    const size_t n_rec = std::max<size_t>((size_t)n_nodes + ns + nq + nt, 1);   // record 0 always exists
    if (n_rec >= MAX_RECORDS) return fail(c, KL_E_LIMIT, "nodes + primitives exceed 2^26 records");
    // (plain arrays: every word is written by the packing loops below, a zero fill of a gigabyte first would cost as much as they do)
    const float4 zero4 = make_float4(0, 0, 0, 0);
    std::unique_ptr<float4[]> rec(on_gpu ? nullptr : new (std::nothrow) float4[n_rec * 4]);
    // shading records (layout: kl_types.hpp), one per primitive in record order
    const size_t n_shrec = std::max<size_t>((size_t)ns + nq + nt, 1);
    std::unique_ptr<float4[]> shrec(on_gpu ? nullptr : new (std::nothrow) float4[n_shrec * 4]);
    if (!on_gpu && (!rec || !shrec)) return fail(c, KL_E_NOMEM, "host memory for the scene records");
    if (!on_gpu) for (int k = 0; k < 4; k++) { rec[k] = zero4; shrec[k] = zero4; }      // record 0 exists even in an empty scene
    auto shade_rec = [&](size_t prim, v3 n, const kl_material &mt, float4 s2, float4 s3) {
        float4 *r = &shrec[4 * prim];
        r[0] = make_float4(n.x, n.y, n.z, bits((uint32_t)mt.type));
        r[1] = make_float4(mt.color[0], mt.color[1], mt.color[2], mt.emissive);
        r[2] = make_float4(mt.roughness, mt.ior, s2.z, s2.w);
        r[3] = s3;
    };
    auto patch_child = [&](int32_t child) -> int32_t {        // builder leaves count within their type
        if (child >= 0 || child == NODE_DONE) return child;
        uint32_t lv = (uint32_t)~child, type = (lv >> 29) & 3u, cnt = ((lv >> 26) & 7u) + 1u, start = lv & 0x3FFFFFFu;
        return make_leaf(type, base[type] + start, cnt);
    };
    if (!on_gpu) {
    parallel_for(n_nodes, threads, [&](uint32_t kb, uint32_t ke) {
        for (uint32_t k = kb; k < ke; k++) {
            uint32_t w[16]; std::memcpy(w, &bvh.nodes4[(size_t)k * 16], sizeof w);
            for (int j = 12; j < 16; j++) w[j] = (uint32_t)patch_child((int32_t)w[j]);
            std::memcpy(&rec[4 * (size_t)k], w, sizeof w);
        }
    });
    parallel_for(ns, threads, [&](uint32_t kb, uint32_t ke) {
        for (uint32_t k = kb; k < ke; k++) {
            const kl_sphere &x = s[bvh.sph_order[k]];
            float4 *r = &rec[4 * ((size_t)base[PRIM_SPHERE] + k)];
            r[0] = make_float4(x.center[0], x.center[1], x.center[2], x.radius);
            r[1] = make_float4(bits(bvh.sph_order[k]), bits(x.material), 0.0f, 0.0f);
            r[2] = zero4; r[3] = zero4;
            shade_rec(k, V(x.center[0], x.center[1], x.center[2]), m[x.material], make_float4(0, 0, x.radius, 0.0f), make_float4(0, 0, 0, 0));
        }
    });
    parallel_for(nq, threads, [&](uint32_t kb, uint32_t ke) {
        for (uint32_t k = kb; k < ke; k++) {
            const kl_quad &x = q[bvh.quad_order[k]];
            float4 *r = &rec[4 * ((size_t)base[PRIM_QUAD] + k)];
            r[0] = make_float4(x.v0[0], x.v0[1], x.v0[2], x.v1[0]);
            r[1] = make_float4(x.v1[1], x.v1[2], x.v2[0], x.v2[1]);
            r[2] = make_float4(x.v2[2], x.v3[0], x.v3[1], x.v3[2]);
            r[3] = make_float4(bits(ns + bvh.quad_order[k]), bits(x.material), bits(m[x.material].type == KL_TEXTURED ? 1u : 0u), 0.0f);
            v3 v1 = V(x.v1[0], x.v1[1], x.v1[2]), v2 = V(x.v2[0], x.v2[1], x.v2[2]), v3_ = V(x.v3[0], x.v3[1], x.v3[2]);
            shade_rec((size_t)ns + k, normalize(cross(v2 - v1, v3_ - v1)), m[x.material], make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0));   // RawFrame.glsl:319
        }
    });
    parallel_for(nt, threads, [&](uint32_t kb, uint32_t ke) {
        for (uint32_t k = kb; k < ke; k++) {
            const kl_tri &x = t[bvh.tri_order[k]];
            v3 v0 = V(x.v0[0], x.v0[1], x.v0[2]);
            v3 e1 = V(x.v1[0], x.v1[1], x.v1[2]) - v0, e2 = V(x.v2[0], x.v2[1], x.v2[2]) - v0;
            v3 nor = normalize(cross(e1, e2));       // the flat shading normal, in the arithmetic the shade stage would use
            float4 *r = &rec[4 * ((size_t)base[PRIM_TRI] + k)];
            r[0] = make_float4(v0.x, v0.y, v0.z, e1.x);
            r[1] = make_float4(e1.y, e1.z, e2.x, e2.y);
            r[2] = make_float4(e2.z, bits(ns + nq + bvh.tri_order[k]), bits(x.material), bits(m[x.material].type == KL_TEXTURED ? 1u : 0u));
            r[3] = zero4;
            shade_rec((size_t)ns + nq + k, nor, m[x.material], make_float4(0, 0, x.uv0[0], x.uv0[1]), make_float4(x.uv1[0], x.uv1[1], x.uv2[0], x.uv2[1]));
        }
    });
    }
    const int32_t tree_root = bvh.root4;
    bvh.stats.nodes = n_nodes; bvh.stats.max_depth = bvh.depth4; bvh.stats.node_bytes = (uint64_t)n_nodes * 64u;
    bvh.stats.builder = builder;
    // the flat quads' table
    const uint32_t nfq = bvh.flat_count[PRIM_QUAD];
    std::vector<float4> flatq((size_t)nfq * 5);
    c->flat_quad_index.resize(nfq);
    for (uint32_t k = 0; k < nfq; k++) {
        const uint32_t qi = on_gpu ? packed.flat_quads[k] : bvh.quad_order[bvh.flat_start[PRIM_QUAD] + k];
        c->flat_quad_index[k] = qi;
        flat_quad_terms(q[qi], &flatq[5 * (size_t)k]);
        flatq[5 * (size_t)k + 4] = make_float4(bits(ns + qi), bits(m[q[qi].material].type == KL_TEXTURED ? 1u : 0u), 0.0f, 0.0f);
    }
    c->h_flatq = flatq;
    kl_status st;
    if ((st = upload(c, c->d_flatq, flatq.data(), flatq.size())) != KL_OK) return st;
    if (on_gpu) {                                    // the builder's buffers become the scene's
        c->d_rec.release(); c->d_rec.p = (float4 *)packed.d_rec; c->d_rec.n = packed.n_rec * 4; packed.d_rec = nullptr;
        c->d_shrec.release(); c->d_shrec.p = (float4 *)packed.d_shrec; c->d_shrec.n = packed.n_shrec * 4; packed.d_shrec = nullptr;
    } else {
        if ((st = upload(c, c->d_rec, rec.get(), n_rec * 4)) != KL_OK) return st;
        if ((st = upload(c, c->d_shrec, shrec.get(), n_shrec * 4)) != KL_OK) return st;
    }
    KL_HIP(c, hipStreamSynchronize(c->stream));     // the host vectors go out of scope below
    c->sc.rec = c->d_rec.p; c->sc.root = patch_child(tree_root); c->sc.n_nodes = n_nodes; c->tree_root = tree_root;
    c->sc.shrec = c->d_shrec.p; c->sc.flatq = c->d_flatq.p;
    c->sc.count[PRIM_SPHERE] = ns; c->sc.count[PRIM_QUAD] = nq; c->sc.count[PRIM_TRI] = nt;
    for (int k = 0; k < 3; k++) {
        c->sc.base[k] = base[k];
        c->sc.flat_start[k] = base[k] + bvh.flat_start[k]; c->sc.flat_count[k] = bvh.flat_count[k];
    }
    c->n_prims = ns + nq + nt;
    c->bvh_stats = bvh.stats;
    c->bvh_stats.refit_seconds = 0.0;
    c->bvh_stats.prim_bytes = ((uint64_t)ns + nq + nt) * 64;
    c->bvh_stats.flat_spheres = bvh.flat_count[PRIM_SPHERE]; c->bvh_stats.flat_quads = bvh.flat_count[PRIM_QUAD];
    c->bvh_stats.flat_tris = bvh.flat_count[PRIM_TRI];
    c->have_scene = true;
    return kl_reset(c);
}
kl_status kl_upload_scene(kl_ctx *c, const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq,
                          const kl_tri *t, uint32_t nt, const kl_material *m, uint32_t nm) {
    return guarded(c, [&]() { return kl_upload_scene_body(c, s, ns, q, nq, t, nt, m, nm); });
}

static kl_status kl_load_builtin_scene_body(kl_ctx *c) {
    if (!c) return KL_E_INVALID;
    std::vector<kl_sphere> s; std::vector<kl_quad> q; std::vector<kl_material> m;
    builtin_scene(s, q, m);
    return kl_upload_scene(c, s.data(), (uint32_t)s.size(), q.data(), (uint32_t)q.size(), nullptr, 0, m.data(), (uint32_t)m.size());
}
kl_status kl_load_builtin_scene(kl_ctx *c) {
    return guarded(c, [&]() { return kl_load_builtin_scene_body(c); });
}
static kl_status kl_refit_scene_body(kl_ctx *c, const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq, const kl_tri *t, uint32_t nt) {
    if (!c || (ns && !s) || (nq && !q) || (nt && !t)) return fail(c, KL_E_INVALID, "null scene array");
    if (!c->have_scene) return fail(c, KL_E_STATE, "no scene uploaded");
    if (ns != c->sc.count[PRIM_SPHERE] || nq != c->sc.count[PRIM_QUAD] || nt != c->sc.count[PRIM_TRI])
        return fail(c, KL_E_INVALID, "refit needs the primitive counts of the uploaded scene");
    KL_HIP(c, hipSetDevice(c->device));
    KL_HIP(c, hipStreamSynchronize(c->stream));
    const auto t0 = std::chrono::steady_clock::now();
    std::string why;
    if (refit_scene_gpu(c->stream, c->d_rec.p, c->d_shrec.p, c->sc.n_nodes, c->tree_root, (uint32_t)BVH_MAX_DEPTH + 1u, s, ns, q, nq, t, nt, why) != LBVH_OK)
        return fail(c, KL_E_DEVICE, "refit: " + why);
    for (size_t k = 0; k < c->flat_quad_index.size(); k++) flat_quad_terms(q[c->flat_quad_index[k]], &c->h_flatq[5 * k]);
    if (!c->h_flatq.empty()) {
        KL_HIP(c, hipMemcpyAsync(c->d_flatq.p, c->h_flatq.data(), c->h_flatq.size() * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        KL_HIP(c, hipStreamSynchronize(c->stream));
    }
    c->bvh_stats.refit_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return kl_reset(c);
}
kl_status kl_refit_scene(kl_ctx *c, const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq, const kl_tri *t, uint32_t nt) {
    return guarded(c, [&]() { return kl_refit_scene_body(c, s, ns, q, nq, t, nt); });
}
// ---- the packed scene as one relocatable block (kl_scene_export / kl_scene_import): header, record array, shading records,
// flat-quad table, flat-quad indices. Record references are indices, so the block is valid on any device.
namespace {
struct SceneBlobHeader {
    uint32_t magic, version;
    uint64_t bytes;                     // of the whole block
    uint64_t n_rec_f4, n_shrec_f4, n_flatq_f4, n_flat_index;
    uint32_t n_nodes, n_prims;
    int32_t root, tree_root;
    uint32_t base[3], count[3], flat_start[3], flat_count[3];
    kl_bvh_stats stats;
};
constexpr uint32_t SCENE_BLOB_MAGIC = 0x43534C4Bu, SCENE_BLOB_VERSION = 1u;   // "KLSC"
constexpr size_t SCENE_BLOB_HEADER_BYTES = 512;
static_assert(sizeof(SceneBlobHeader) <= SCENE_BLOB_HEADER_BYTES, "header block too small");
size_t round256(size_t x) { return (x + 255u) & ~(size_t)255u; }
struct SceneBlobLayout { size_t rec, shrec, flatq, index, total; };
SceneBlobLayout blob_layout(uint64_t n_rec_f4, uint64_t n_shrec_f4, uint64_t n_flatq_f4, uint64_t n_index) {
    SceneBlobLayout l;
    l.rec = SCENE_BLOB_HEADER_BYTES;
    l.shrec = l.rec + round256(n_rec_f4 * sizeof(float4));
    l.flatq = l.shrec + round256(n_shrec_f4 * sizeof(float4));
    l.index = l.flatq + round256(n_flatq_f4 * sizeof(float4));
    l.total = l.index + round256(n_index * sizeof(uint32_t));
    return l;
}
}  // namespace

kl_status kl_scene_blob_bytes(kl_ctx *c, size_t *bytes) {
    if (!c || !bytes) return KL_E_INVALID;
    if (!c->have_scene) return fail(c, KL_E_STATE, "no scene uploaded");
    *bytes = blob_layout(c->d_rec.n, c->d_shrec.n, c->h_flatq.size(), c->flat_quad_index.size()).total;
    return KL_OK;
}
static kl_status kl_scene_export_body(kl_ctx *c, void *dst, size_t bytes) {
    if (!c || !dst) return KL_E_INVALID;
    if (!c->have_scene) return fail(c, KL_E_STATE, "no scene uploaded");
    const SceneBlobLayout l = blob_layout(c->d_rec.n, c->d_shrec.n, c->h_flatq.size(), c->flat_quad_index.size());
    if (bytes != l.total) return fail(c, KL_E_INVALID, "kl_scene_export: the block is kl_scene_blob_bytes() long");
    KL_HIP(c, hipSetDevice(c->device));
    std::vector<unsigned char> head(SCENE_BLOB_HEADER_BYTES, 0);
    SceneBlobHeader h; std::memset(&h, 0, sizeof h);
    h.magic = SCENE_BLOB_MAGIC; h.version = SCENE_BLOB_VERSION; h.bytes = l.total;
    h.n_rec_f4 = c->d_rec.n; h.n_shrec_f4 = c->d_shrec.n; h.n_flatq_f4 = c->h_flatq.size(); h.n_flat_index = c->flat_quad_index.size();
    h.n_nodes = c->sc.n_nodes; h.n_prims = c->n_prims; h.root = c->sc.root; h.tree_root = c->tree_root;
    for (int k = 0; k < 3; k++) { h.base[k] = c->sc.base[k]; h.count[k] = c->sc.count[k]; h.flat_start[k] = c->sc.flat_start[k]; h.flat_count[k] = c->sc.flat_count[k]; }
    h.stats = c->bvh_stats;
    std::memcpy(head.data(), &h, sizeof h);
    unsigned char *d = (unsigned char *)dst;
    KL_HIP(c, hipMemcpyAsync(d, head.data(), head.size(), hipMemcpyHostToDevice, c->stream));
    KL_HIP(c, hipMemcpyAsync(d + l.rec, c->d_rec.p, c->d_rec.n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    KL_HIP(c, hipMemcpyAsync(d + l.shrec, c->d_shrec.p, c->d_shrec.n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    if (!c->h_flatq.empty())
        KL_HIP(c, hipMemcpyAsync(d + l.flatq, c->h_flatq.data(), c->h_flatq.size() * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    if (!c->flat_quad_index.empty())
        KL_HIP(c, hipMemcpyAsync(d + l.index, c->flat_quad_index.data(), c->flat_quad_index.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    KL_HIP(c, hipStreamSynchronize(c->stream));         // `head` goes out of scope
    return KL_OK;
}
kl_status kl_scene_export(kl_ctx *c, void *dst_device, size_t bytes) {
    return guarded(c, [&]() { return kl_scene_export_body(c, dst_device, bytes); });
}
static kl_status kl_scene_import_body(kl_ctx *c, const void *src, size_t bytes) {
    if (!c || !src) return KL_E_INVALID;
    if (bytes < SCENE_BLOB_HEADER_BYTES) return fail(c, KL_E_INVALID, "kl_scene_import: block shorter than its header");
    KL_HIP(c, hipSetDevice(c->device));
    KL_HIP(c, hipStreamSynchronize(c->stream));
    SceneBlobHeader h;
    KL_HIP(c, hipMemcpy(&h, src, sizeof h, hipMemcpyDeviceToHost));
    if (h.magic != SCENE_BLOB_MAGIC || h.version != SCENE_BLOB_VERSION) return fail(c, KL_E_INVALID, "kl_scene_import: not a packed scene of this library version");
    // every size is checked against the block before anything is allocated or copied
    if (h.n_rec_f4 > (uint64_t)MAX_RECORDS * 4 || h.n_shrec_f4 > (uint64_t)MAX_RECORDS * 4 || h.n_flatq_f4 > 5u * 32u || h.n_flat_index > 32u ||
        h.n_rec_f4 % 4 || h.n_shrec_f4 % 4 || h.n_rec_f4 == 0 || h.n_shrec_f4 == 0 || h.n_flatq_f4 != 5 * h.n_flat_index)
        return fail(c, KL_E_INVALID, "kl_scene_import: sizes in the header are not those of a packed scene");
    const SceneBlobLayout l = blob_layout(h.n_rec_f4, h.n_shrec_f4, h.n_flatq_f4, h.n_flat_index);
    if (h.bytes != l.total || bytes != l.total) return fail(c, KL_E_INVALID, "kl_scene_import: block length does not match its header");
    const uint64_t n_rec = h.n_rec_f4 / 4, prims = (uint64_t)h.count[0] + h.count[1] + h.count[2];
    bool ok = h.n_nodes <= n_rec && prims == h.n_prims && (uint64_t)h.n_nodes + prims <= n_rec && h.n_shrec_f4 / 4 >= std::max<uint64_t>(prims, 1) &&
              h.base[0] == h.n_nodes && h.base[1] == h.base[0] + h.count[0] && h.base[2] == h.base[1] + h.count[1] && h.flat_count[PRIM_QUAD] == h.n_flat_index;
    for (int k = 0; k < 3 && ok; k++)
        ok = h.flat_count[k] <= h.count[k] && h.flat_start[k] >= h.base[k] && (uint64_t)h.flat_start[k] + h.flat_count[k] <= (uint64_t)h.base[k] + h.count[k];
    if (ok && h.root != NODE_DONE) {
        if (h.root >= 0) ok = (uint32_t)h.root < h.n_nodes;
        else { const uint32_t lv = (uint32_t)~h.root, start = lv & 0x3FFFFFFu, cnt = ((lv >> 26) & 7u) + 1u; ok = (uint64_t)start + cnt <= n_rec; }
    }
    if (!ok) return fail(c, KL_E_INVALID, "kl_scene_import: inconsistent counts in the header");
    const unsigned char *sp = (const unsigned char *)src;
    std::vector<float4> flatq(h.n_flatq_f4);
    std::vector<uint32_t> index(h.n_flat_index);
    if (!flatq.empty()) KL_HIP(c, hipMemcpy(flatq.data(), sp + l.flatq, flatq.size() * sizeof(float4), hipMemcpyDeviceToHost));
    if (!index.empty()) KL_HIP(c, hipMemcpy(index.data(), sp + l.index, index.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t qi : index) if (qi >= h.count[PRIM_QUAD]) return fail(c, KL_E_INVALID, "kl_scene_import: flat-quad index out of range");
    KL_HIP(c, c->d_rec.alloc(h.n_rec_f4)); KL_HIP(c, c->d_shrec.alloc(h.n_shrec_f4));
    KL_HIP(c, hipMemcpyAsync(c->d_rec.p, sp + l.rec, h.n_rec_f4 * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    KL_HIP(c, hipMemcpyAsync(c->d_shrec.p, sp + l.shrec, h.n_shrec_f4 * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    kl_status st;
    if ((st = upload(c, c->d_flatq, flatq.data(), flatq.size())) != KL_OK) return st;
    KL_HIP(c, hipStreamSynchronize(c->stream));
    c->h_flatq = flatq; c->flat_quad_index = index;
    c->sc.rec = c->d_rec.p; c->sc.shrec = c->d_shrec.p; c->sc.flatq = c->d_flatq.p;
    c->sc.root = h.root; c->sc.n_nodes = h.n_nodes; c->tree_root = h.tree_root;
    for (int k = 0; k < 3; k++) { c->sc.base[k] = h.base[k]; c->sc.count[k] = h.count[k]; c->sc.flat_start[k] = h.flat_start[k]; c->sc.flat_count[k] = h.flat_count[k]; }
    c->n_prims = h.n_prims;
    c->bvh_stats = h.stats;
    c->have_scene = true;
    return kl_reset(c);
}
kl_status kl_scene_import(kl_ctx *c, const void *src_device, size_t bytes) {
    return guarded(c, [&]() { return kl_scene_import_body(c, src_device, bytes); });
}

kl_status kl_get_bvh_stats(kl_ctx *c, kl_bvh_stats *out) {
    if (!c || !out) return KL_E_INVALID;
    if (!c->have_scene) return fail(c, KL_E_STATE, "no scene uploaded");
    *out = c->bvh_stats;
    return KL_OK;
}
kl_status kl_synth_scene_counts(const kl_synth_params *sp, uint32_t *ns, uint32_t *nq, uint32_t *nt, uint32_t *nm) {
    if (!sp || !ns || !nq || !nt || !nm || sp->tris_per_blob < 12) return KL_E_INVALID;
    if ((uint64_t)sp->blobs * sp->tris_per_blob + sp->analytic_spheres >= MAX_RECORDS / 2) return KL_E_LIMIT;
    synth_counts(*sp, *ns, *nq, *nt, *nm);
    return KL_OK;
}
static kl_status kl_synth_scene_fill_body(const kl_synth_params *sp, kl_sphere *s, kl_quad *q, kl_tri *t, kl_material *m) {
    uint32_t ns, nq, nt, nm;
    kl_status st = kl_synth_scene_counts(sp, &ns, &nq, &nt, &nm);
    if (st != KL_OK) return st;
    if ((ns && !s) || !q || (nt && !t) || !m) return KL_E_INVALID;
    synth_fill(*sp, s, q, t, m);
    return KL_OK;
}
kl_status kl_synth_scene_fill(const kl_synth_params *sp, kl_sphere *s, kl_quad *q, kl_tri *t, kl_material *m) {
    return guarded(nullptr, [&]() { return kl_synth_scene_fill_body(sp, s, q, t, m); });
}

static kl_status kl_render_frames_body(kl_ctx *c, uint32_t n, uint32_t first_frame) {
    if (!c) return KL_E_INVALID;
    if (!c->have_scene) return fail(c, KL_E_STATE, "no scene uploaded");
    if (c->npl == 0 || n == 0) return KL_OK;
    // as many lanes as there are batches to overlap
    const uint32_t per_batch = frames_per_batch(c, n, choose_lanes(c));
    const uint32_t batches = (n + per_batch - 1) / per_batch;
    const uint32_t lanes = std::min(choose_lanes(c), batches);
    kl_status s = ensure_buffers(c, lanes, per_batch);
    if (s != KL_OK) return s;
    uint32_t frame = first_frame == KL_CONTINUE ? c->frame : first_frame;
    if ((uint64_t)frame + n > 0xFFFFFFFEull) return fail(c, KL_E_LIMIT, "frame index overflow");
    hipEvent_t ea = nullptr, eb = nullptr;
    if (c->pending.size() >= KL_MAX_PENDING_EVENTS) drain_events(c);
    ea = get_event(c); eb = get_event(c);
    if (ea && eb) KL_HIP(c, hipEventRecord(ea, c->stream));
    if (lanes > 1) {
        // the other lanes start after everything already queued on the context stream
        KL_HIP(c, hipEventRecord(c->ev_start, c->stream));
        for (uint32_t l = 1; l < lanes; l++) KL_HIP(c, hipStreamWaitEvent(c->lanes[l].stream, c->ev_start, 0));
    }
    uint32_t left = n, batch = 0;
    hipEvent_t last_done = nullptr;            // accumulate of the previous batch (null: it ran on this lane or there is none)
    uint32_t last_lane = 0;
    bool lane_used[KL_MAX_BATCH_LANES] = { false, false, false, false };
    while (left) {
        const uint32_t mf = c->params.max_frames;
        const uint32_t l = batch % lanes;
        BatchLane &L = c->lanes[l];
        hipEvent_t after = (batch > 0 && last_lane != l) ? last_done : nullptr;
        uint32_t k;
        if (mf && frame >= mf) {
            // past SAMPLES the raw frame is the accumulation texel (RawFrame.glsl:461-463): nothing to trace
            k = std::min(left, 1024u);
            FrameDev f = make_frame(c, frame, k);
            if (after) KL_HIP(c, hipStreamWaitEvent(L.stream, after, 0));
            launch_accumulate(L.stream, c->lc, f, L.d_raw.p, c->d_acc.p, c->d_last_raw.p, false, c->accum_mode);
        } else {
            k = std::min(left, std::min(per_batch, std::max(1u, c->slots / sub_samples(c))));
            if (mf) k = std::min(k, mf - frame);
            if ((s = render_batch(c, L, after, frame, k, c->camera_change)) != KL_OK) {
                // the context stream still resumes after whatever the lanes were given, and the call's events go back
                // (best effort: the error that is reported is the first one)
                lane_used[l] = true;
                for (uint32_t j = 1; j < lanes; j++)
                    if (lane_used[j] && hipEventRecord(c->lanes[j].done, c->lanes[j].stream) == hipSuccess)
                        (void)hipStreamWaitEvent(c->stream, c->lanes[j].done, 0);
                if (ea) c->free_events.push_back(ea);
                if (eb) c->free_events.push_back(eb);
                return s;
            }
            c->paths_total += (uint64_t)k * c->npl * sub_samples(c);
        }
        if (lanes > 1) { KL_HIP(c, hipEventRecord(L.done, L.stream)); last_done = L.done; }
        last_lane = l; lane_used[l] = true;
        c->camera_change = false;       // GLSL_Renderer.cpp:348
        frame += k; left -= k; batch++;
        c->frames_total += k;
    }
    // the context stream resumes after every lane: later calls (reads, resolves, the next render) stay ordered
    for (uint32_t l = 1; l < lanes; l++)
        if (lane_used[l]) KL_HIP(c, hipStreamWaitEvent(c->stream, c->lanes[l].done, 0));
    c->frame = frame;                   // iFrame++ per frame, GLSL_Renderer.cpp:347
    if (ea && eb) { KL_HIP(c, hipEventRecord(eb, c->stream)); c->pending.push_back({ea, eb, EV_RENDER}); }
    return KL_OK;
}
kl_status kl_render_frames(kl_ctx *c, uint32_t n, uint32_t first_frame) {
    return guarded(c, [&]() { return kl_render_frames_body(c, n, first_frame); });
}

kl_status kl_synchronize(kl_ctx *c) {
    if (!c) return KL_E_INVALID;
    KL_HIP(c, hipSetDevice(c->device));
    KL_HIP(c, hipStreamSynchronize(c->stream));
    return KL_OK;
}

kl_status kl_get_stream(kl_ctx *c, void **hip_stream) {
    if (!c || !hip_stream) return KL_E_INVALID;
    *hip_stream = (void *)c->stream;
    return KL_OK;
}

static kl_status read_buf(kl_ctx *c, const DevBuf<float4> &b, float *rgba, size_t bytes) {
    if (!c || !rgba) return fail(c, KL_E_INVALID, "null output");
    if (!b.p || b.n != c->npl) return fail(c, KL_E_STATE, "nothing rendered yet");
    if (bytes != (size_t)c->npl * 16) return fail(c, KL_E_INVALID, "output must be local_rows x width x 16 bytes");
    KL_HIP(c, hipSetDevice(c->device));
    KL_HIP(c, hipMemcpyAsync(rgba, b.p, bytes, hipMemcpyDeviceToHost, c->stream));
    KL_HIP(c, hipStreamSynchronize(c->stream));
    return KL_OK;
}
kl_status kl_read_raw(kl_ctx *c, float *rgba, size_t bytes) { return c ? read_buf(c, c->d_last_raw, rgba, bytes) : KL_E_INVALID; }
kl_status kl_read_accum(kl_ctx *c, float *rgba, size_t bytes) { return c ? read_buf(c, c->d_acc, rgba, bytes) : KL_E_INVALID; }

kl_status kl_resolve(kl_ctx *c, uint32_t tonemap, uint8_t *rgba8, size_t bytes) {
    if (!c || !rgba8 || tonemap > KL_TONEMAP_ACES) return fail(c, KL_E_INVALID, "bad resolve arguments");
    if (!c->d_acc.p || c->d_acc.n != c->npl) return fail(c, KL_E_STATE, "nothing rendered yet");
    if (bytes != (size_t)c->npl * 4) return fail(c, KL_E_INVALID, "output must be local_rows x width x 4 bytes");
    KL_HIP(c, hipSetDevice(c->device));
    KL_HIP(c, c->d_resolve.alloc(c->npl));
    launch_resolve(c->stream, c->lc, c->d_acc.p, c->d_resolve.p, c->npl, tonemap);
    KL_HIP(c, hipMemcpyAsync(rgba8, c->d_resolve.p, bytes, hipMemcpyDeviceToHost, c->stream));
    KL_HIP(c, hipStreamSynchronize(c->stream));
    return KL_OK;
}

kl_status kl_get_counters(kl_ctx *c, kl_counters *out) {
    if (!c || !out) return KL_E_INVALID;
    std::memset(out, 0, sizeof *out);
    out->frames = c->frames_total; out->paths = c->paths_total;
    KL_HIP(c, hipSetDevice(c->device));
    if (c->d_counters.p) {
        Counters h;
        KL_HIP(c, hipMemcpyAsync(&h, c->d_counters.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
        KL_HIP(c, hipStreamSynchronize(c->stream));
        out->rays = h.rays; out->nodes = h.nodes; out->sphere_tests = h.sphere_tests; out->quad_tests = h.quad_tests;
        out->tri_tests = h.tri_tests; out->lane_steps_active = h.lane_active; out->lane_steps_total = h.lane_total;
    }
    return KL_OK;
}
static kl_status kl_reset_counters_body(kl_ctx *c) {
    if (!c) return KL_E_INVALID;
    KL_HIP(c, hipSetDevice(c->device));
    c->frames_total = 0; c->paths_total = 0;
    drain_events(c);
    std::memset(&c->timing, 0, sizeof c->timing);
    if (c->d_counters.p) KL_HIP(c, hipMemsetAsync(c->d_counters.p, 0, sizeof(Counters), c->stream));
    return KL_OK;
}
kl_status kl_reset_counters(kl_ctx *c) {
    return guarded(c, [&]() { return kl_reset_counters_body(c); });
}
static kl_status kl_get_timing_body(kl_ctx *c, kl_timing *out) {
    if (!c || !out) return KL_E_INVALID;
    KL_HIP(c, hipSetDevice(c->device));
    drain_events(c);
    *out = c->timing;
    return KL_OK;
}
kl_status kl_get_timing(kl_ctx *c, kl_timing *out) {
    return guarded(c, [&]() { return kl_get_timing_body(c, out); });
}

static kl_status kl_trace_rays_body(kl_ctx *c, const float *o, const float *d, uint32_t n, float *t, uint32_t *prim, float *uv) {
    if (!c || !o || !d || !t || !prim || !uv) return fail(c, KL_E_INVALID, "null ray arrays");
    if (!c->have_scene) return fail(c, KL_E_STATE, "no scene uploaded");
    if (n == 0) return KL_OK;
    if (n > (1u << 28)) return fail(c, KL_E_LIMIT, "at most 2^28 rays per call");
    KL_HIP(c, hipSetDevice(c->device));
    // the caller's rays as one queue segment; every entry starts as a miss
    const float miss_t = c->params.max_dist;
    float miss_ref; { const uint32_t r = REF_MISS; std::memcpy(&miss_ref, &r, 4); }
    std::vector<float4> od(n), dt(n);
    for (uint32_t i = 0; i < n; i++) {
        od[i] = make_float4(o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i]);
        dt[i] = make_float4(d[3 * i + 1], d[3 * i + 2], miss_t, miss_ref);
    }
    DevBuf<float4> b_od, b_dt; DevBuf<float2> b_quv, b_uv; DevBuf<float> b_t; DevBuf<uint32_t> b_prim, b_ctl;
    struct Cleanup { DevBuf<float4> &a, &b; DevBuf<float2> &c, &d; DevBuf<float> &e; DevBuf<uint32_t> &f, &g;
                     ~Cleanup() { a.release(); b.release(); c.release(); d.release(); e.release(); f.release(); g.release(); } }
        cleanup{b_od, b_dt, b_quv, b_uv, b_t, b_prim, b_ctl};
    KL_HIP(c, b_od.alloc(n)); KL_HIP(c, b_dt.alloc(n)); KL_HIP(c, b_quv.alloc(n)); KL_HIP(c, b_uv.alloc(n));
    KL_HIP(c, b_t.alloc(n)); KL_HIP(c, b_prim.alloc(n)); KL_HIP(c, b_ctl.alloc(2));   // one segment: count, head
    if (!c->d_counters.p) { KL_HIP(c, c->d_counters.alloc(1)); KL_HIP(c, hipMemsetAsync(c->d_counters.p, 0, sizeof(Counters), c->stream)); }
    KL_HIP(c, c->lanes[0].d_overflow.alloc(intersect_overflow_ints(c->lc)));
    const uint32_t ctl[2] = { n, 0 };
    KL_HIP(c, hipMemcpyAsync(b_od.p, od.data(), n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    KL_HIP(c, hipMemcpyAsync(b_dt.p, dt.data(), n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    KL_HIP(c, hipMemsetAsync(b_quv.p, 0, n * sizeof(float2), c->stream));
    KL_HIP(c, hipMemcpyAsync(b_ctl.p, ctl, sizeof ctl, hipMemcpyHostToDevice, c->stream));
    FrameDev f = make_frame(c, 0, 1);
    f.nseg = 1; f.segcap = n;                       // caller rays form one segment
    f.all_uv = 1;                                   // (u, v) of every hit, whatever its material
    RayQueue q = { b_od.p, b_dt.p, nullptr, b_quv.p };
    if (use_bvh(c) && scene_has_flat(c->sc))
        launch_intersect_flat(c->stream, c->lc, c->sc, f, q, &b_ctl.p[0], c->d_counters.p, c->params.collect_stats == 1, n);
    launch_intersect(c->stream, c->lc, c->sc, f, q, &b_ctl.p[0], &b_ctl.p[1], c->d_counters.p, c->lanes[0].d_overflow.p, use_bvh(c),
                     (int)std::min(c->params.collect_stats, 2u), n);
    launch_unpack_hits(c->stream, c->sc, q, n, b_t.p, b_prim.p, b_uv.p);
    KL_HIP(c, hipMemcpyAsync(t, b_t.p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    KL_HIP(c, hipMemcpyAsync(prim, b_prim.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    KL_HIP(c, hipMemcpyAsync(uv, b_uv.p, n * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
    KL_HIP(c, hipStreamSynchronize(c->stream));
    KL_HIP(c, hipGetLastError());
    return KL_OK;
}
kl_status kl_trace_rays(kl_ctx *c, const float *o, const float *d, uint32_t n, float *t, uint32_t *prim, float *uv) {
    return guarded(c, [&]() { return kl_trace_rays_body(c, o, d, n, t, prim, uv); });
}

// kl_trace_rays for callers that cannot allocate: device pointers in and out, intermediate storage in the library's
// fixed device array (kl_kernels.hip), no hipMalloc / hipFree and no host container on the way.
kl_status kl_trace_rays_noalloc(kl_ctx *c, const float *o, const float *d, uint32_t n, float *t, uint32_t *prim, float *uv) {
    if (!c) return KL_E_INVALID;
    if (!o || !d || !t || !prim || !uv) { c->err = "null ray arrays"; return KL_E_INVALID; }
    if (!c->have_scene) { c->err = "no scene uploaded"; return KL_E_STATE; }
    if (n == 0) return KL_OK;
    if (n > KL_NOALLOC_MAX_RAYS) { c->err = "kl_trace_rays_noalloc takes at most KL_NOALLOC_MAX_RAYS (262144) rays per call"; return KL_E_LIMIT; }
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice"; return KL_E_DEVICE; }
    FrameDev f = make_frame(c, 0, 1);
    f.nseg = 1; f.segcap = n;                       // caller rays form one segment
    f.all_uv = 1;                                   // (u, v) of every hit, whatever its material
    const hipError_t e = launch_trace_noalloc(c->stream, c->device, c->lc, c->sc, f, o, d, n, t, prim, (float2 *)uv, use_bvh(c));
    if (e != hipSuccess) { c->err = hipGetErrorString(e); return KL_E_DEVICE; }
    return KL_OK;
}

kl_status kl_set_shard(kl_ctx *c, uint32_t rank, uint32_t world, uint32_t band_rows) {
    if (!c || world == 0 || rank >= world || band_rows == 0) return fail(c, KL_E_INVALID, "need rank < world and band_rows >= 1");
    KL_HIP(c, hipSetDevice(c->device));
    KL_HIP(c, hipStreamSynchronize(c->stream));
    c->rank = rank; c->world = world; c->band_rows = band_rows;
    compute_local(c);
    c->d_acc.release();
    soft_reset(c);
    return KL_OK;
}
kl_status kl_get_local_rows(kl_ctx *c, uint32_t *rows) { if (!c || !rows) return KL_E_INVALID; *rows = c->local_rows; return KL_OK; }
kl_status kl_set_accum_mode(kl_ctx *c, uint32_t mode) {
    if (!c || mode > KL_ACCUM_MEAN_UNORM8) return fail(c, KL_E_INVALID, "bad accumulation mode");
    c->accum_mode = mode;
    return kl_reset(c);
}
kl_status kl_copy_accum_to_device(kl_ctx *c, void *dst, size_t bytes) {
    if (!c || !dst) return fail(c, KL_E_INVALID, "null destination");
    if (!c->d_acc.p || c->d_acc.n != c->npl) return fail(c, KL_E_STATE, "nothing rendered yet");
    if (bytes != (size_t)c->npl * 16) return fail(c, KL_E_INVALID, "size must be local_rows x width x 16 bytes");
    KL_HIP(c, hipSetDevice(c->device));
    KL_HIP(c, hipMemcpyAsync(dst, c->d_acc.p, bytes, hipMemcpyDeviceToDevice, c->stream));
    KL_HIP(c, hipStreamSynchronize(c->stream));
    return KL_OK;
}
kl_status kl_copy_accum_from_device(kl_ctx *c, const void *src, size_t bytes) {
    if (!c || !src) return fail(c, KL_E_INVALID, "null source");
    kl_status s = ensure_buffers(c);
    if (s != KL_OK) return s;
    if (bytes != (size_t)c->npl * 16) return fail(c, KL_E_INVALID, "size must be local_rows x width x 16 bytes");
    KL_HIP(c, hipMemcpyAsync(c->d_acc.p, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    KL_HIP(c, hipStreamSynchronize(c->stream));
    return KL_OK;
}

// ---- accumulation checkpoint: header + accumulation + last raw frame, little-endian as in memory ----
namespace {
struct CkHeader {
    char magic[8];                 // "KLCKPT1\0"
    uint32_t width, height, rank, world, band_rows, local_rows;
    uint32_t frame, mode, accum_mode, camera_change;
    kl_camera cam;
    kl_params params;
    uint64_t texels;               // local_rows * width
};
}  // namespace

static kl_status kl_checkpoint_save_body(kl_ctx *c, const char *path) {
    if (!c || !path) return fail(c, KL_E_INVALID, "null path");
    if (!c->d_acc.p || c->d_acc.n != c->npl || !c->npl) return fail(c, KL_E_STATE, "nothing rendered yet");
    KL_HIP(c, hipSetDevice(c->device));
    std::vector<float4> acc(c->npl), raw(c->npl);
    KL_HIP(c, hipMemcpyAsync(acc.data(), c->d_acc.p, c->npl * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    KL_HIP(c, hipMemcpyAsync(raw.data(), c->d_last_raw.p, c->npl * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    KL_HIP(c, hipStreamSynchronize(c->stream));
    CkHeader hd; std::memset(&hd, 0, sizeof hd);
    std::memcpy(hd.magic, "KLCKPT1", 8);
    hd.width = c->width; hd.height = c->height; hd.rank = c->rank; hd.world = c->world; hd.band_rows = c->band_rows;
    hd.local_rows = c->local_rows; hd.frame = c->frame; hd.mode = c->mode; hd.accum_mode = c->accum_mode;
    hd.camera_change = c->camera_change ? 1u : 0u; hd.cam = c->cam; hd.params = c->params; hd.texels = c->npl;
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return fail(c, KL_E_INVALID, std::string("cannot write ") + path);
    bool ok = std::fwrite(&hd, sizeof hd, 1, fp) == 1 && std::fwrite(acc.data(), sizeof(float4), acc.size(), fp) == acc.size() &&
              std::fwrite(raw.data(), sizeof(float4), raw.size(), fp) == raw.size();
    ok = (std::fclose(fp) == 0) && ok;
    return ok ? KL_OK : fail(c, KL_E_INVALID, std::string("short write to ") + path);
}
kl_status kl_checkpoint_save(kl_ctx *c, const char *path) {
    return guarded(c, [&]() { return kl_checkpoint_save_body(c, path); });
}

static kl_status kl_checkpoint_load_body(kl_ctx *c, const char *path) {
    if (!c || !path) return fail(c, KL_E_INVALID, "null path");
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return fail(c, KL_E_INVALID, std::string("cannot read ") + path);
    CkHeader hd;
    if (std::fread(&hd, sizeof hd, 1, fp) != 1 || std::memcmp(hd.magic, "KLCKPT1", 8) != 0) {
        std::fclose(fp); return fail(c, KL_E_INVALID, "not a checkpoint file");
    }
    // the image, its partition and the view mode must be the context's; camera and parameters are restored
    if (hd.width != c->width || hd.height != c->height || hd.rank != c->rank || hd.world != c->world ||
        hd.band_rows != c->band_rows || hd.local_rows != c->local_rows || hd.texels != c->npl) {
        std::fclose(fp); return fail(c, KL_E_STATE, "checkpoint was written for another image size or shard");
    }
    if (hd.mode > KL_MODE_Z || hd.accum_mode > KL_ACCUM_MEAN_UNORM8) { std::fclose(fp); return fail(c, KL_E_INVALID, "corrupt checkpoint header"); }
    std::vector<float4> acc(c->npl), raw(c->npl);
    bool ok = std::fread(acc.data(), sizeof(float4), acc.size(), fp) == acc.size() &&
              std::fread(raw.data(), sizeof(float4), raw.size(), fp) == raw.size();
    std::fclose(fp);
    if (!ok) return fail(c, KL_E_INVALID, "truncated checkpoint");
    kl_status s = kl_set_params(c, &hd.params);
    if (s != KL_OK) return s;
    s = ensure_buffers(c);
    if (s != KL_OK) return s;
    KL_HIP(c, hipMemcpyAsync(c->d_acc.p, acc.data(), c->npl * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    KL_HIP(c, hipMemcpyAsync(c->d_last_raw.p, raw.data(), c->npl * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    KL_HIP(c, hipStreamSynchronize(c->stream));
    c->cam = hd.cam; c->mode = hd.mode; c->accum_mode = hd.accum_mode;
    c->frame = hd.frame; c->camera_change = hd.camera_change != 0;
    return KL_OK;
}
kl_status kl_checkpoint_load(kl_ctx *c, const char *path) {
    return guarded(c, [&]() { return kl_checkpoint_load_body(c, path); });
}

}  // extern "C"
