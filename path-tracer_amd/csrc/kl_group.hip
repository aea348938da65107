// Multi-GPU group behind the C ABI (include/Math/kl_render.h, kl_group_*): one process, one context per GPU, and
// the one exchange step each partition has, run directly on the contexts' accumulation buffers:
//   image tiles    -> ncclAllGather of the per-GPU row bands + an un-tile kernel        (bit-identical to one GPU)
//   sample ranges  -> ncclAllReduce(sum) of the per-GPU sums + one divide               (fp32 re-association only)
// RCCL is loaded when the first group is created (dlopen: a process that never makes a group never loads it). The
// reference has no counterpart: it renders on the one GPU that owns its GL context.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "Math/kl_render.h"
#include "kl_internal.hpp"

using namespace kl;

namespace {

// image row y lives in block (y / band_rows) % world of the gathered buffer, at local row
// (y / band_rows / world) * band_rows + y % band_rows (the layout kl_set_shard gives a context's buffers)
__global__ __launch_bounds__(256) void k_untile(const float4 *gathered, float4 *full, uint32_t width, uint32_t height, uint32_t world,
                                                uint32_t band_rows, uint32_t pad_rows) {
    const uint32_t n = width * height;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t y = i / width, x = i - y * width;
        const uint32_t band = y / band_rows, rank = band % world, lrow = (band / world) * band_rows + (y - band * band_rows);
        full[i] = gathered[((size_t)rank * pad_rows + lrow) * width + x];
    }
}
// sum over all frames -> mean (alpha 1), in place
__global__ __launch_bounds__(256) void k_sum_to_mean(float4 *img, uint32_t n, float frames) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 v = img[i];
        img[i] = make_float4(v.x / frames, v.y / frames, v.z / frames, 1.0f);
    }
}
// the copy exchange's reduction: blocks of `n` texels added in member order
__global__ __launch_bounds__(256) void k_sum_blocks(const float4 *blocks, float4 *out, uint32_t n, uint32_t world) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 a = blocks[i];
        for (uint32_t r = 1; r < world; r++) { const float4 b = blocks[(size_t)r * n + i]; a.x += b.x; a.y += b.y; a.z += b.z; }
        out[i] = a;
    }
}

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string &err) {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
#define KL_SYM(field, sym) field = (decltype(field))dlsym(lib, sym); if (!field) { err = std::string("librccl lacks ") + sym; return false; }
        KL_SYM(CommInitAll, "ncclCommInitAll") KL_SYM(CommDestroy, "ncclCommDestroy") KL_SYM(AllGather, "ncclAllGather")
        KL_SYM(AllReduce, "ncclAllReduce") KL_SYM(GroupStart, "ncclGroupStart") KL_SYM(GroupEnd, "ncclGroupEnd")
        KL_SYM(GetErrorString, "ncclGetErrorString")
#undef KL_SYM
        return true;
    }
};
Rccl g_rccl;

struct Member {
    kl_ctx *ctx = nullptr;
    int device = 0;
    ncclComm_t comm = nullptr;
    float4 *gathered = nullptr;   // tiles: world blocks of pad_rows x width; copy exchange of sample ranges: world full images
    float4 *stage = nullptr;      // tiles with a ragged last band: this member's rows padded to pad_rows
    float4 *full = nullptr;       // the whole image (mean) after kl_group_gather
};

}  // namespace

struct kl_group {
    uint32_t n = 0, split = 0, exchange = 0, width = 0, height = 0, band_rows = 1, pad_rows = 0;
    std::vector<Member> m;
    uint32_t next_frame = 0;          // sample ranges: the frame after the last one handed out
    uint64_t frames_given = 0;        // ... and how many were handed out since the last reset (the divisor of the mean)
    bool broken = false;              // a render call failed part-way: some members hold frames the others do not, until kl_group_reset
    std::string err;
};

namespace {

// why the last kl_group_create on this thread failed (there is no group yet to hold the text); kl_group_last_error(NULL) returns it
thread_local std::string t_create_err;

kl_status gfail(kl_group *g, kl_status s, const std::string &msg) { if (g) { try { g->err = msg; } catch (...) {} } return s; }
// No exception crosses the C ABI: the std::vector / std::string allocations of an entry point's body (the error
// macros below build strings too) become a status here.
template <typename F> kl_status gguarded(kl_group *g, F body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc &) { if (g) { try { g->err = "host memory exhausted"; } catch (...) {} } return KL_E_NOMEM; }
    catch (const std::exception &e) { if (g) { try { g->err = e.what(); } catch (...) {} } return KL_E_INVALID; }
    catch (...) { return KL_E_INVALID; }
}
#define KLG_HIP(g, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return gfail(g, e_ == hipErrorOutOfMemory ? KL_E_NOMEM : KL_E_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
#define KLG_NCCL(g, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return gfail(g, KL_E_DEVICE, std::string(#call) + ": " + g_rccl.GetErrorString(r_)); } while (0)
#define KLG_CTX(g, k, call) do { kl_status s_ = (call); if (s_ != KL_OK) return gfail(g, s_, std::string(#call) + " on member " + std::to_string(k) + ": " + kl_last_error((g)->m[k].ctx)); } while (0)

uint32_t rows_of(uint32_t height, uint32_t rank, uint32_t world, uint32_t band_rows) {
    uint32_t rows = 0;
    for (uint32_t b = rank; b * band_rows < height; b += world) rows += std::min(height, (b + 1) * band_rows) - b * band_rows;
    return rows;
}
uint32_t grid_for(uint32_t n) { return std::max(1u, std::min((n + 255u) / 256u, 2048u)); }

void release(kl_group *g) {
    for (Member &mb : g->m) {
        (void)hipSetDevice(mb.device);
        if (mb.comm) (void)g_rccl.CommDestroy(mb.comm);
        if (mb.ctx) (void)kl_destroy(mb.ctx);
        if (mb.gathered) (void)hipFree(mb.gathered);
        if (mb.stage) (void)hipFree(mb.stage);
        if (mb.full) (void)hipFree(mb.full);
    }
    delete g;
}

}  // namespace

extern "C" {

kl_status kl_group_create(kl_group **out, const int32_t *devices, uint32_t n, uint32_t split, uint32_t width, uint32_t height,
                          uint32_t band_rows) {
    if (!out) return KL_E_INVALID;
    *out = nullptr;
    const uint32_t exchange = split & KL_EXCHANGE_COPY, how = split & ~(uint32_t)KL_EXCHANGE_COPY;
    auto why_failed = [](const std::string &msg) { try { t_create_err = msg; } catch (...) {} };
    if (!devices || n == 0 || n > 64 || how > KL_SPLIT_SAMPLES || band_rows == 0) { why_failed("bad argument"); return KL_E_INVALID; }
    kl_group *g = new (std::nothrow) kl_group();
    if (!g) { why_failed("host memory exhausted"); return KL_E_NOMEM; }
    try {
        g->n = n; g->split = how; g->exchange = exchange; g->width = width; g->height = height; g->band_rows = band_rows;
        g->m.resize(n);
        for (uint32_t r = 0; r < n; r++) g->pad_rows = std::max(g->pad_rows, how == KL_SPLIT_TILES ? rows_of(height, r, n, band_rows) : height);
        std::vector<int> devs(n);
        for (uint32_t r = 0; r < n; r++) {
            Member &mb = g->m[r];
            mb.device = devs[r] = devices[r];
            kl_status s = kl_create(&mb.ctx, devices[r], width, height);
            if (s == KL_OK && how == KL_SPLIT_TILES) s = kl_set_shard(mb.ctx, r, n, band_rows);
            if (s == KL_OK && how == KL_SPLIT_SAMPLES) s = kl_set_accum_mode(mb.ctx, KL_ACCUM_SUM);
            if (s != KL_OK) {
                why_failed("member " + std::to_string(r) + " (device " + std::to_string(devices[r]) + "): " + (mb.ctx ? kl_last_error(mb.ctx) : kl_status_string(s)));
                release(g); return s;
            }
            const size_t full = (size_t)width * height, block = (size_t)g->pad_rows * width;
            hipError_t e = hipSetDevice(mb.device);
            if (e == hipSuccess) e = hipMalloc((void **)&mb.full, full * sizeof(float4));
            if (e == hipSuccess && (how == KL_SPLIT_TILES || exchange)) e = hipMalloc((void **)&mb.gathered, block * n * sizeof(float4));
            if (e == hipSuccess && how == KL_SPLIT_TILES && rows_of(height, r, n, band_rows) != g->pad_rows) {
                e = hipMalloc((void **)&mb.stage, block * sizeof(float4));
                if (e == hipSuccess) e = hipMemset(mb.stage, 0, block * sizeof(float4));
            }
            if (e != hipSuccess) {
                why_failed("exchange buffers of member " + std::to_string(r) + ": " + hipGetErrorString(e));
                release(g); return e == hipErrorOutOfMemory ? KL_E_NOMEM : KL_E_DEVICE;
            }
        }
        if (!exchange) {
            std::string why;
            if (!g_rccl.load(why)) { why_failed(why); release(g); return KL_E_DEVICE; }
            std::vector<ncclComm_t> comms(n, nullptr);
            const ncclResult_t cr = g_rccl.CommInitAll(comms.data(), (int)n, devs.data());
            if (cr != ncclSuccess) { why_failed(std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(cr)); release(g); return KL_E_DEVICE; }
            for (uint32_t r = 0; r < n; r++) g->m[r].comm = comms[r];
        }
    } catch (...) { why_failed("host memory exhausted"); release(g); return KL_E_NOMEM; }
    *out = g;
    return KL_OK;
}

kl_status kl_group_destroy(kl_group *g) {
    if (!g) return KL_E_INVALID;
    release(g);
    return KL_OK;
}
kl_status kl_group_size(kl_group *g, uint32_t *n) { if (!g || !n) return KL_E_INVALID; *n = g->n; return KL_OK; }
kl_status kl_group_member(kl_group *g, uint32_t rank, kl_ctx **ctx) {
    if (!g || !ctx || rank >= g->n) return KL_E_INVALID;
    *ctx = g->m[rank].ctx;
    return KL_OK;
}
// g = NULL: why the last kl_group_create on the calling thread failed
const char *kl_group_last_error(kl_group *g) { return g ? g->err.c_str() : t_create_err.c_str(); }

// ---- the same input to every member
static kl_status kl_group_upload_scene_body(kl_group *g, const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq, const kl_tri *t, uint32_t nt,
                                const kl_material *mt, uint32_t nm) {
    if (!g) return KL_E_INVALID;
    // One tree build for the whole group: member 0 builds, the packed scene (kl_scene_export: records, tree, shading records,
    // flat-quad table as one relocatable block) goes to every other member's GPU by a peer copy, and they import it.
    KLG_CTX(g, 0, kl_upload_scene(g->m[0].ctx, s, ns, q, nq, t, nt, mt, nm));
    if (g->n > 1) {
        size_t bytes = 0;
        KLG_CTX(g, 0, kl_scene_blob_bytes(g->m[0].ctx, &bytes));
        struct Block {          // freed on every way out
            int dev; void *p = nullptr;
            explicit Block(int d) : dev(d) {}
            ~Block() { if (p) { (void)hipSetDevice(dev); (void)hipFree(p); } }
        };
        Block src(g->m[0].device);
        KLG_HIP(g, hipSetDevice(src.dev));
        KLG_HIP(g, hipMalloc(&src.p, bytes));
        KLG_CTX(g, 0, kl_scene_export(g->m[0].ctx, src.p, bytes));
        for (uint32_t k = 1; k < g->n; k++) {
            Block dst(g->m[k].device);
            KLG_HIP(g, hipSetDevice(dst.dev));
            KLG_HIP(g, hipMalloc(&dst.p, bytes));
            KLG_HIP(g, hipMemcpyPeer(dst.p, dst.dev, src.p, src.dev, bytes));
            KLG_CTX(g, k, kl_scene_import(g->m[k].ctx, dst.p, bytes));
        }
    }
    return kl_group_reset(g);
}
kl_status kl_group_upload_scene(kl_group *g, const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq, const kl_tri *t, uint32_t nt,
                                const kl_material *mt, uint32_t nm) {
    return gguarded(g, [&]() { return kl_group_upload_scene_body(g, s, ns, q, nq, t, nt, mt, nm); });
}
static kl_status kl_group_load_builtin_scene_body(kl_group *g) {
    if (!g) return KL_E_INVALID;
    for (uint32_t k = 0; k < g->n; k++) KLG_CTX(g, k, kl_load_builtin_scene(g->m[k].ctx));
    return kl_group_reset(g);
}
kl_status kl_group_load_builtin_scene(kl_group *g) {
    return gguarded(g, [&]() { return kl_group_load_builtin_scene_body(g); });
}
static kl_status kl_group_upload_texture_body(kl_group *g, uint32_t which, const uint8_t *px, uint32_t w, uint32_t h, uint32_t ch) {
    if (!g) return KL_E_INVALID;
    for (uint32_t k = 0; k < g->n; k++) KLG_CTX(g, k, kl_upload_texture(g->m[k].ctx, which, px, w, h, ch));
    return KL_OK;
}
kl_status kl_group_upload_texture(kl_group *g, uint32_t which, const uint8_t *px, uint32_t w, uint32_t h, uint32_t ch) {
    return gguarded(g, [&]() { return kl_group_upload_texture_body(g, which, px, w, h, ch); });
}
static kl_status kl_group_set_params_body(kl_group *g, const kl_params *p) {
    if (!g) return KL_E_INVALID;
    for (uint32_t k = 0; k < g->n; k++) KLG_CTX(g, k, kl_set_params(g->m[k].ctx, p));
    return KL_OK;
}
kl_status kl_group_set_params(kl_group *g, const kl_params *p) {
    return gguarded(g, [&]() { return kl_group_set_params_body(g, p); });
}
static kl_status kl_group_set_camera_body(kl_group *g, const kl_camera *cam) {
    if (!g) return KL_E_INVALID;
    for (uint32_t k = 0; k < g->n; k++) KLG_CTX(g, k, kl_set_camera(g->m[k].ctx, cam));
    return kl_group_reset(g);
}
kl_status kl_group_set_camera(kl_group *g, const kl_camera *cam) {
    return gguarded(g, [&]() { return kl_group_set_camera_body(g, cam); });
}
static kl_status kl_group_set_mode_body(kl_group *g, uint32_t mode) {
    if (!g) return KL_E_INVALID;
    for (uint32_t k = 0; k < g->n; k++) KLG_CTX(g, k, kl_set_mode(g->m[k].ctx, mode));
    return kl_group_reset(g);
}
kl_status kl_group_set_mode(kl_group *g, uint32_t mode) {
    return gguarded(g, [&]() { return kl_group_set_mode_body(g, mode); });
}
static kl_status kl_group_reset_body(kl_group *g) {
    if (!g) return KL_E_INVALID;
    for (uint32_t k = 0; k < g->n; k++) KLG_CTX(g, k, kl_reset(g->m[k].ctx));
    g->next_frame = 0; g->frames_given = 0; g->broken = false;
    return KL_OK;
}
kl_status kl_group_reset(kl_group *g) {
    return gguarded(g, [&]() { return kl_group_reset_body(g); });
}

// Image tiles: every member renders the n frames over its row bands. Sample ranges: member r renders frames
// [first + n*r/N, first + n*(r+1)/N) over the whole image as a sum. Work is queued on every member before any is waited for.
static kl_status kl_group_render_frames_body(kl_group *g, uint32_t n, uint32_t first_frame) {
    if (!g) return KL_E_INVALID;
    if (n == 0) return KL_OK;
    if (g->broken) return gfail(g, KL_E_STATE, "an earlier kl_group_render_frames failed part-way; kl_group_reset first");
    const uint32_t first = first_frame == KL_CONTINUE ? g->next_frame : first_frame;
    if ((uint64_t)first + n > 0xFFFFFFFEull) return gfail(g, KL_E_LIMIT, "frame index overflow");
    // A member that fails leaves the members before it with frames the others do not have (their sums and frame counters have
    // moved on): the group then refuses to render or gather until kl_group_reset brings every member back to frame 0.
    g->broken = true;
    for (uint32_t k = 0; k < g->n; k++) {
        if (g->split == KL_SPLIT_TILES) {
            KLG_CTX(g, k, kl_render_frames(g->m[k].ctx, n, first));
        } else {
            const uint32_t lo = first + (uint32_t)(((uint64_t)n * k) / g->n), hi = first + (uint32_t)(((uint64_t)n * (k + 1)) / g->n);
            if (hi > lo) KLG_CTX(g, k, kl_render_frames(g->m[k].ctx, hi - lo, lo));
        }
    }
    g->broken = false;
    // frames that add to the sum: those before the SAMPLES freeze (RawFrame.glsl:426), which every member applies alike
    CtxView v;
    KLG_CTX(g, 0, ctx_view(g->m[0].ctx, &v));
    const uint64_t end = (uint64_t)first + n, live_end = v.max_frames ? std::min<uint64_t>(end, v.max_frames) : end;
    g->frames_given += live_end > first ? live_end - first : 0;
    g->next_frame = first + n;
    return KL_OK;
}
kl_status kl_group_render_frames(kl_group *g, uint32_t n, uint32_t first_frame) {
    return gguarded(g, [&]() { return kl_group_render_frames_body(g, n, first_frame); });
}

// The exchange step: afterwards every member holds the whole image (for sample ranges: the mean over all frames given out
// since the last reset) in a buffer of its own; member 0's is copied to rgba if that is not null. Waits for completion.
static kl_status kl_group_gather_body(kl_group *g, float *rgba, size_t bytes) {
    if (!g) return KL_E_INVALID;
    const size_t full = (size_t)g->width * g->height, block = (size_t)g->pad_rows * g->width;
    if (rgba && bytes != full * sizeof(float4)) return gfail(g, KL_E_INVALID, "output must be height x width x 16 bytes");
    if (g->broken) return gfail(g, KL_E_STATE, "an earlier kl_group_render_frames failed part-way; kl_group_reset first");
    std::vector<CtxView> v(g->n);
    for (uint32_t k = 0; k < g->n; k++) KLG_CTX(g, k, ctx_view(g->m[k].ctx, &v[k]));
    // what each member contributes: its accumulation buffer as it stands, or (tiles, ragged last band) padded to the block size
    std::vector<const float4 *> send(g->n);
    for (uint32_t k = 0; k < g->n; k++) {
        Member &mb = g->m[k];
        send[k] = v[k].acc;
        if (g->split == KL_SPLIT_TILES && mb.stage) {
            KLG_HIP(g, hipSetDevice(mb.device));
            if (v[k].npl) KLG_HIP(g, hipMemcpyAsync(mb.stage, v[k].acc, (size_t)v[k].npl * sizeof(float4), hipMemcpyDeviceToDevice, v[k].stream));
            send[k] = mb.stage;
        }
    }
    const size_t count = (g->split == KL_SPLIT_TILES ? block : full);
    if (!g->exchange) {
        KLG_NCCL(g, g_rccl.GroupStart());
        for (uint32_t k = 0; k < g->n; k++) {
            Member &mb = g->m[k];
            KLG_HIP(g, hipSetDevice(mb.device));
            ncclResult_t r = g->split == KL_SPLIT_TILES
                ? g_rccl.AllGather(send[k], mb.gathered, count * 4, ncclFloat, mb.comm, v[k].stream)
                : g_rccl.AllReduce(send[k], mb.full, count * 4, ncclFloat, ncclSum, mb.comm, v[k].stream);
            if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return gfail(g, KL_E_DEVICE, std::string("RCCL: ") + g_rccl.GetErrorString(r)); }
        }
        KLG_NCCL(g, g_rccl.GroupEnd());
    } else {
        // rehearsal exchange without RCCL (members may share a GPU, which RCCL refuses): every member copies every block
        for (uint32_t k = 0; k < g->n; k++) KLG_HIP(g, hipStreamSynchronize(v[k].stream));       // sources complete
        for (uint32_t k = 0; k < g->n; k++) {
            Member &mb = g->m[k];
            KLG_HIP(g, hipSetDevice(mb.device));
            for (uint32_t s = 0; s < g->n; s++)
                KLG_HIP(g, hipMemcpyPeerAsync(mb.gathered + (size_t)s * count, mb.device, send[s], g->m[s].device, count * sizeof(float4), v[k].stream));
            if (g->split == KL_SPLIT_SAMPLES)
                hipLaunchKernelGGL(k_sum_blocks, dim3(grid_for((uint32_t)full)), dim3(256), 0, v[k].stream, mb.gathered, mb.full, (uint32_t)full, g->n);
        }
    }
    for (uint32_t k = 0; k < g->n; k++) {
        Member &mb = g->m[k];
        KLG_HIP(g, hipSetDevice(mb.device));
        if (g->split == KL_SPLIT_TILES)
            hipLaunchKernelGGL(k_untile, dim3(grid_for((uint32_t)full)), dim3(256), 0, v[k].stream, mb.gathered, mb.full, g->width, g->height, g->n,
                               g->band_rows, g->pad_rows);
        else
            hipLaunchKernelGGL(k_sum_to_mean, dim3(grid_for((uint32_t)full)), dim3(256), 0, v[k].stream, mb.full, (uint32_t)full,
                               (float)std::max<uint64_t>(g->frames_given, 1));
        KLG_HIP(g, hipGetLastError());
    }
    for (uint32_t k = 0; k < g->n; k++) {
        KLG_HIP(g, hipSetDevice(g->m[k].device));
        if (k == 0 && rgba) KLG_HIP(g, hipMemcpyAsync(rgba, g->m[0].full, bytes, hipMemcpyDeviceToHost, v[0].stream));
        KLG_HIP(g, hipStreamSynchronize(v[k].stream));
    }
    return KL_OK;
}
kl_status kl_group_gather(kl_group *g, float *rgba, size_t bytes) {
    return gguarded(g, [&]() { return kl_group_gather_body(g, rgba, bytes); });
}

kl_status kl_group_image_device(kl_group *g, uint32_t rank, const float **rgba_device) {
    if (!g || !rgba_device || rank >= g->n) return KL_E_INVALID;
    *rgba_device = (const float *)g->m[rank].full;
    return KL_OK;
}

}  // extern "C"
