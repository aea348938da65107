/*
 * kl_render.h -- C ABI of the MI355X path-tracing backend.
 *
 * This is the surface a host application binds in place of the OpenGL render
 * dispatch of the reference (source/Rendering/Opengl/GLSL_Renderer.cpp:201-240
 * set-up, :281-348 per-frame passes). Each entry point cites the reference
 * interface it replaces. Plain C types only; every function returns a
 * kl_status (0 = KL_OK, negative = error) and never throws.
 *
 * Ownership and threading: a kl_ctx is created and destroyed by the caller,
 * owns one HIP device + stream, and must be used from one thread at a time
 * (the reference drives everything from the thread that owns the GL context,
 * GLSL_Renderer.cpp:243-359). Input pointers are borrowed for the duration of
 * the call. Work is queued on the context's stream; the read/resolve/copy
 * entry points and kl_synchronize wait for it.
 */
#ifndef KL_RENDER_H
#define KL_RENDER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define KL_API __attribute__((visibility("default")))
#else
#define KL_API
#endif

typedef int32_t kl_status;
enum {
    KL_OK = 0,
    KL_E_INVALID = -1,      /* bad argument */
    KL_E_DEVICE = -2,       /* HIP runtime error (kl_last_error has the text) */
    KL_E_NOMEM = -3,
    KL_E_STATE = -4,        /* call not valid in the current state (e.g. no scene) */
    KL_E_LIMIT = -5         /* size beyond a documented maximum */
};

/* view modes: Render_Mode::Enum, include/Rendering/Opengl/GLSL_Renderer.h:15-21;
 * uploaded as iRenderMode, GLSL_Renderer.cpp:290 */
enum { KL_MODE_AO = 0, KL_MODE_PT = 1, KL_MODE_Z = 2 };
/* material types: resources/Shaders/RawFrame.glsl:126-129 */
enum { KL_DIFFUSE = 0, KL_GLASS = 1, KL_EMISSIVE = 2, KL_TEXTURED = 3 };
/* sampler slots: iHdri (unit 2) and iAlbedo (unit 3), GLSL_Renderer.cpp:301-304 */
enum { KL_TEX_HDRI = 0, KL_TEX_ALBEDO = 1 };
/* present pass: PostProcess.glsl:38 (CLAMP) or its overwritten lines 19-37 (ACES) */
enum { KL_TONEMAP_CLAMP = 0, KL_TONEMAP_ACES = 1 };
/* accumulation: running mean of Accumulation.glsl:16-24, or a plain sum for sample-range sharding */
enum { KL_ACCUM_MEAN = 0, KL_ACCUM_SUM = 1,
       /* the running mean exactly as the shipped application stores it: its colour attachments are unsized GL_RGBA
        * (source/Rendering/Opengl/FBT.cpp:6,21), i.e. RGBA8, so the raw frame and the mean are clamped to [0,1] and rounded
        * to 8 bits every frame. KL_ACCUM_MEAN keeps the shader's float values (the parity target); this mode reproduces what
        * the application shows on screen. */
       KL_ACCUM_MEAN_UNORM8 = 2 };
/* intersection back end: brute force in declaration order (what RawFrame.glsl:294-328 does) or the SAH BVH */
enum { KL_ACCEL_AUTO = 0, KL_ACCEL_BRUTE = 1, KL_ACCEL_BVH = 2 };
/* who builds the tree at kl_upload_scene (no reference counterpart: the reference has no acceleration structure):
 * HOST_SAH = binned surface-area heuristic on the host cores (the better tree, 0.1-0.2 s per million primitives);
 * GPU_LBVH = Morton-ordered radix tree built, refitted and collapsed on the GPU (milliseconds; rays cost more in it).
 * Both give the same images: the closest hit does not depend on the tree. */
enum { KL_BUILD_HOST_SAH = 0, KL_BUILD_GPU_LBVH = 1 };
#define KL_CONTINUE 0xFFFFFFFFu

/* the camera uniforms iCameraPos/Front/Up/FocalLength/SensorWidth,
 * RawFrame.glsl:9-13, uploaded at GLSL_Renderer.cpp:293-297 (already narrowed to float) */
typedef struct kl_camera {
    float pos[3], front[3], up[3];
    float focal, sensor;
} kl_camera;

/* the shader's five-field material, RawFrame.glsl:131-137 (32 bytes) */
typedef struct kl_material {
    int32_t type;
    float color[3];
    float emissive;
    float roughness;
    float ior;
    float pad;
} kl_material;

/* scene upload records shaped after include/Object/Sphere.h:6-12 and
 * include/Object/Mesh.h:5-18 (Tri, Quad), carrying a material index */
typedef struct kl_sphere { float center[3]; float radius; uint32_t material; } kl_sphere;
typedef struct kl_quad { float v0[3], v1[3], v2[3], v3[3]; uint32_t material; } kl_quad;
typedef struct kl_tri { float v0[3], v1[3], v2[3]; float uv0[2], uv1[2], uv2[2]; uint32_t material; } kl_tri;

/* the shader's compile-time defines (RawFrame.glsl:32-40) as run-time parameters */
typedef struct kl_params {
    uint32_t max_bounces;      /* RAY_BOUNCES, 32 */
    uint32_t max_frames;       /* SAMPLES, 30; 0 = never freeze */
    float eps;                 /* EPSILON, 0.001 */
    float max_dist;            /* MAX_DIST, 5000 */
    float ao_length;           /* AO_LENGTH, 1 */
    float hdri_strength;       /* HDRI_STRENGTH, 1 */
    uint32_t accel;            /* KL_ACCEL_* */
    uint32_t frames_in_flight; /* frames traced together as one wavefront batch (0 = choose: as many as a quarter of the free memory holds, up to 64,
                                  with several lanes up to 48 and the call's frames shared equally among the lanes) */
    uint32_t fused_compaction; /* 1: shade pushes survivors itself; 0: separate compact kernel */
    uint32_t collect_stats;    /* 1: an instrumented traversal kernel counts nodes / primitive tests / active lanes per lane (slower);
                                  2: the kernel the scene runs anyway tallies the same per wave (ballot + popcount: about 1 % slower) */
    uint32_t profile_kernels;  /* 1: HIP events around every generate / intersect / shade / accumulate launch, see kl_get_timing */
    uint32_t batch_lanes;      /* batches traced concurrently on separate HIP streams, each with its own queues: 0 = choose
                                  (one), at most 4. The image does not depend on it (batches accumulate in frame order). */
    uint32_t trace_primary_per_frame; /* 0: the camera ray of a pixel, which is the same in every frame (RawFrame.glsl:436 adds no
                                  jitter at SPP = 1), is traced once per batch and its hit shared by the batch's frames;
                                  1: traced again for every frame, as the shader does. The image is the same either way. */
    uint32_t bvh_builder;      /* KL_BUILD_*, read by the next kl_upload_scene / kl_load_mesh; a GPU build whose tree turns out
                                  deeper than the traversal stack allows is redone on the host (kl_bvh_stats.builder says which ran) */
    uint32_t spp_grid;         /* SPP, 1: spp_grid x spp_grid sub-samples per pixel and frame in the AO and path-traced modes, each offset
                                  by (x / SPP, y / SPP) of a pixel and all sharing the frame's random numbers (RawFrame.glsl:37,434-440,
                                  446-453); the raw frame is their sum divided by SPP * SPP. 0 = 1; at most 8 */
    uint32_t reserved[1];
} kl_params;

typedef struct kl_counters {
    uint64_t frames;           /* frames rendered since kl_reset_counters */
    uint64_t paths;            /* path samples (pixels x frames) */
    uint64_t rays;             /* rays that entered an intersect kernel */
    uint64_t nodes;            /* BVH nodes visited          (collect_stats only) */
    uint64_t sphere_tests, quad_tests, tri_tests; /*          (collect_stats only) */
    uint64_t lane_steps_active, lane_steps_total; /* traversal-loop lane occupancy (collect_stats only) */
} kl_counters;

typedef struct kl_timing {
    double intersect_ms;       /* sum of HIP-event durations of intersect launches (profile_kernels only; with several batch lanes
                                  launches of different lanes overlap, so sums over kinds can exceed the wall time) */
    uint64_t intersect_launches;
    double render_ms;          /* sum of HIP-event durations of whole kl_render_frames calls */
    uint64_t render_calls;
    double shade_ms;           /* the same per shade / generate / accumulate launch (profile_kernels only) */
    uint64_t shade_launches;
    double generate_ms;
    uint64_t generate_launches;
    double accumulate_ms;
    uint64_t accumulate_launches;
} kl_timing;

typedef struct kl_bvh_stats {
    uint32_t nodes, leaves, max_depth, max_leaf_size;
    uint32_t spheres, quads, tris;
    float sah_cost;
    double build_seconds;
    uint64_t node_bytes, prim_bytes;
    uint32_t flat_spheres, flat_quads, flat_tris; /* scene-spanning primitives kept out of the tree and tested for every ray */
    uint32_t builder;          /* KL_BUILD_* that produced the tree in use */
    double refit_seconds;      /* duration of the last kl_refit_scene, 0 if none since the upload */
} kl_bvh_stats;

/* parameters of the seed-deterministic synthetic benchmark scene */
typedef struct kl_synth_params {
    uint32_t blobs;            /* tessellated blobs inside the room */
    uint32_t tris_per_blob;    /* rounded down to 2*k*(k-1) for a k x k lat-long tessellation */
    uint32_t analytic_spheres; /* extra analytic spheres (mixed sphere/mesh scenes) */
    uint32_t seed;
} kl_synth_params;

typedef struct kl_ctx kl_ctx;

/* -- life cycle: GLSL_Renderer::GLSL_Renderer + f_init set-up, GLSL_Renderer.cpp:3-46,164-240 -- */
KL_API kl_status kl_create(kl_ctx **out, int32_t device, uint32_t width, uint32_t height);
KL_API kl_status kl_destroy(kl_ctx *ctx);
/* framebuffer_size_callback, GLSL_Renderer.cpp:67-82: new size, frame counter back to 0 */
KL_API kl_status kl_resize(kl_ctx *ctx, uint32_t width, uint32_t height);
/* f_recompile, GLSL_Renderer.cpp:48-65: kernels are compiled ahead of time, so only the reset part remains */
KL_API kl_status kl_recompile(kl_ctx *ctx);

/* -- uniforms: GLSL_Renderer.cpp:287-298 -- */
KL_API kl_status kl_set_camera(kl_ctx *ctx, const kl_camera *cam);      /* :293-297; resets like :247-273 */
KL_API kl_status kl_set_mode(kl_ctx *ctx, uint32_t mode);               /* :290; resets like :140-144 */
KL_API kl_status kl_default_params(kl_params *p);                       /* RawFrame.glsl:32-40 values */
KL_API kl_status kl_default_camera(kl_camera *cam);                     /* source/Object/Camera.cpp:3-14 */
KL_API kl_status kl_set_params(kl_ctx *ctx, const kl_params *p);
KL_API kl_status kl_get_params(kl_ctx *ctx, kl_params *p);
/* iFrame = 0, iCameraChange = true: the tail of every input handler, e.g. GLSL_Renderer.cpp:141-143 */
KL_API kl_status kl_reset(kl_ctx *ctx);
KL_API kl_status kl_get_frame(kl_ctx *ctx, uint32_t *frame);            /* iFrame, GLSL_Renderer.h:34 */

/* -- textures: Texture::f_init, source/Rendering/Opengl/Texture.cpp:3-16 over Image::f_load,
 *    source/I-O/File.cpp:13-29: 8-bit, 1/3/4 channels, rows as decoded (no flip), NEAREST, clamp -- */
KL_API kl_status kl_upload_texture(kl_ctx *ctx, uint32_t which, const uint8_t *pixels,
                                   uint32_t width, uint32_t height, uint32_t channels);

/* Float equirectangular environment, rgb or rgba rows as given, used in place of the 8-bit iHdri with the same
 * lookup (RawFrame.glsl:330-339: NEAREST, clamp) and no 1/255 scaling. The role north_star's Background.exr was
 * meant to play: the reference never loads it (GLSL_Renderer.cpp:237 loads Background.png). pixels = NULL goes
 * back to the 8-bit texture. Resets the frame counter like every other input change. */
KL_API kl_status kl_upload_env_float(kl_ctx *ctx, const float *pixels, uint32_t width, uint32_t height, uint32_t channels);

/* -- scene: replaces the const arrays of RawFrame.glsl:171-192 -- */
KL_API kl_status kl_upload_scene(kl_ctx *ctx, const kl_sphere *spheres, uint32_t n_spheres,
                                 const kl_quad *quads, uint32_t n_quads,
                                 const kl_tri *tris, uint32_t n_tris,
                                 const kl_material *materials, uint32_t n_materials);
KL_API kl_status kl_load_builtin_scene(kl_ctx *ctx);                    /* RawFrame.glsl:171-192 */
/* Scene ingestion (no loader in the reference; shaped after the CPU-side containers of include/Object/Mesh.h:5-26):
 * reads a Wavefront .obj or a Stanford .ply (ascii or binary_little_endian) into a malloc'ed kl_tri array that the
 * caller hands to kl_upload_scene and releases with kl_free. Polygons become triangle fans; per-corner texture
 * coordinates are kept; in an .obj the k-th distinct `usemtl` name selects material index `material + k` (clamped
 * to n_materials - 1 when n_materials != 0). Needs no context and no GPU. kl_load_mesh_error() = what was wrong. */
KL_API kl_status kl_load_mesh(const char *path, uint32_t material, uint32_t n_materials, kl_tri **tris, uint32_t *n_tris);
KL_API const char *kl_load_mesh_error(void);
KL_API void kl_free(void *p);
/* Moved geometry (no reference counterpart: the shader's scene is compiled in, RawFrame.glsl:171-192). The arrays hold
 * the primitives of the last kl_upload_scene in the same order, with new positions (and triangle uvs); their material
 * fields are ignored, materials stay as uploaded. The tree keeps its shape and every box is recomputed on the GPU, so
 * the result is exact; a tree refitted far from the geometry it was built for is slower, not wrong. The frame counter
 * goes back to 0 as after any scene change. KL_E_INVALID if the counts differ from the uploaded scene's. */
KL_API kl_status kl_refit_scene(kl_ctx *ctx, const kl_sphere *spheres, uint32_t n_spheres, const kl_quad *quads, uint32_t n_quads,
                                const kl_tri *tris, uint32_t n_tris);
KL_API kl_status kl_get_bvh_stats(kl_ctx *ctx, kl_bvh_stats *out);
/* The uploaded scene, packed and with its tree, as ONE relocatable block in device memory (no reference counterpart). One
 * process builds the tree (kl_upload_scene), exports the block, and hands it to the other GPUs' contexts -- e.g. one
 * RCCL broadcast -- which import it instead of building the same tree again: kl_scene_blob_bytes gives the block's size,
 * kl_scene_export writes it to dst_device (a device pointer on the context's device, exactly that many bytes),
 * kl_scene_import reads such a block from src_device and makes it the context's scene, as kl_upload_scene would have
 * (frame counter back to 0; kl_refit_scene works on it; textures are not part of the block). Record references are
 * indices, so the block is valid on any device. The header's sizes are checked against `bytes` before anything is copied:
 * KL_E_INVALID for a block that is not one of this library version's. */
KL_API kl_status kl_scene_blob_bytes(kl_ctx *ctx, size_t *bytes);
KL_API kl_status kl_scene_export(kl_ctx *ctx, void *dst_device, size_t bytes);
KL_API kl_status kl_scene_import(kl_ctx *ctx, const void *src_device, size_t bytes);
/* synthetic benchmark scene (no reference counterpart): sizes first, then fill caller-owned arrays */
KL_API kl_status kl_synth_scene_counts(const kl_synth_params *sp, uint32_t *n_spheres, uint32_t *n_quads,
                                       uint32_t *n_tris, uint32_t *n_materials);
KL_API kl_status kl_synth_scene_fill(const kl_synth_params *sp, kl_sphere *spheres, kl_quad *quads,
                                     kl_tri *tris, kl_material *materials);

/* -- the frame loop body: passes 1 and 2 of GLSL_Renderer.cpp:283-328 plus iFrame++ (:347), n times.
 *    first_frame = KL_CONTINUE continues from the context's frame counter -- */
KL_API kl_status kl_render_frames(kl_ctx *ctx, uint32_t n, uint32_t first_frame);
KL_API kl_status kl_synchronize(kl_ctx *ctx);
/* the HIP stream (a hipStream_t, handed over as void*) on which the context queues its work, so that a caller can order its
 * own streams against it with events. No reference counterpart (the reference has one GL context and no streams). */
KL_API kl_status kl_get_stream(kl_ctx *ctx, void **hip_stream);

/* -- accumulation checkpoint (no reference counterpart: the reference restarts from iFrame = 0): writes / reads
 *    {image size and shard, frame counter, view mode, camera, parameters, accumulation, last raw frame}. After a
 *    load into a context of the same size and shard, kl_render_frames(KL_CONTINUE) carries on bit for bit as the
 *    saved session would have. Scene and textures are not part of the file. -- */
KL_API kl_status kl_checkpoint_save(kl_ctx *ctx, const char *path);
KL_API kl_status kl_checkpoint_load(kl_ctx *ctx, const char *path);

/* -- results: what raw_frame_tex / accumulation_tex hold (GLSL_Renderer.cpp:223-234), float RGBA,
 *    row 0 = bottom row (gl_FragCoord.y = 0.5); and pass 3, PostProcess.glsl:34-39 -- */
KL_API kl_status kl_read_raw(kl_ctx *ctx, float *rgba, size_t bytes);
KL_API kl_status kl_read_accum(kl_ctx *ctx, float *rgba, size_t bytes);
KL_API kl_status kl_resolve(kl_ctx *ctx, uint32_t tonemap, uint8_t *rgba8, size_t bytes);

/* -- measurement (no reference counterpart; the reference only shows 1/frame_time in the
 *    window title, GLSL_Renderer.cpp:350-355) -- */
KL_API kl_status kl_get_counters(kl_ctx *ctx, kl_counters *out);
KL_API kl_status kl_reset_counters(kl_ctx *ctx);
KL_API kl_status kl_get_timing(kl_ctx *ctx, kl_timing *out);
/* closest hit for caller-supplied rays through the same intersect kernel the renderer uses:
 * o, d are n x 3 floats; t (n), prim (n, declaration-order id or 0xFFFFFFFF), uv (n x 2) are outputs */
KL_API kl_status kl_trace_rays(kl_ctx *ctx, const float *o, const float *d, uint32_t n,
                               float *t, uint32_t *prim, float *uv);

/* One entry point that allocates nothing. It has no counterpart in the reference; it is kl_trace_rays for callers that
 * cannot allocate device memory: tight serving loops, harnesses that forbid hipMalloc, code whose surrounding region
 * must stay free of allocation calls.
 * All pointers are DEVICE pointers on the context's device (o, d: n x 3 floats; t: n; prim: n; uv: n x 2 floats).
 * The queue entries, counters and traversal stack spill that kl_trace_rays allocates per call live in a fixed
 * device array of the library instead (about 33 MB per GPU, held until the process ends), hence the limit:
 * n <= KL_NOALLOC_MAX_RAYS, else KL_E_LIMIT.
 * Ordering: the kernels are queued on the CONTEXT's stream (kl_get_stream), which is a non-blocking stream: it is ordered
 * neither with the null stream nor with the caller's streams. The inputs must therefore be complete before the call (the
 * caller synchronises its own stream or device, or makes the context's stream wait on an event of its own), and the outputs
 * are valid after kl_synchronize (or after an event the caller records on the context's stream). The call returns without
 * waiting. Every call on a GPU uses that GPU's one array: the library orders calls that reach one GPU through different
 * contexts or threads one after the other (an event wait between them, no allocation), so they do not race, but they do not
 * overlap either. Same hits as kl_trace_rays, bit for bit; rays traced here are not tallied by kl_get_counters. */
#define KL_NOALLOC_MAX_RAYS 262144u
KL_API kl_status kl_trace_rays_noalloc(kl_ctx *ctx, const float *o_dev, const float *d_dev, uint32_t n,
                                       float *t_dev, uint32_t *prim_dev, float *uv_dev);

/* -- partitioning over GPUs (no reference counterpart). Image tiles: this context renders the
 *    row bands b with b % world == rank, band_rows rows each; its buffers then hold only those
 *    rows, in increasing row order. Sample ranges: use KL_ACCUM_SUM and disjoint first_frame ranges. -- */
KL_API kl_status kl_set_shard(kl_ctx *ctx, uint32_t rank, uint32_t world, uint32_t band_rows);
KL_API kl_status kl_get_local_rows(kl_ctx *ctx, uint32_t *rows);
KL_API kl_status kl_set_accum_mode(kl_ctx *ctx, uint32_t mode);
/* device-to-device copies of the accumulation buffer for collectives run by the caller
 * (e.g. torch.distributed over RCCL); dst/src are device pointers on the context's device */
KL_API kl_status kl_copy_accum_to_device(kl_ctx *ctx, void *dst_device, size_t bytes);
KL_API kl_status kl_copy_accum_from_device(kl_ctx *ctx, const void *src_device, size_t bytes);

/* -- several GPUs from one process (no reference counterpart: the reference renders on the one GPU that owns its GL
 *    context, GLSL_Renderer.cpp:164-188). A group holds one context per device and runs the one exchange step of its
 *    partition over RCCL (xGMI), directly on the contexts' accumulation buffers:
 *      KL_SPLIT_TILES    member r renders the row bands b with b % n == r; kl_group_gather = all-gather + un-tile:
 *                        the image is bit-identical to one GPU's;
 *      KL_SPLIT_SAMPLES  member r sums its share of every frame range; kl_group_gather = all-reduce(sum) + one divide:
 *                        differs from the sequential mean by fp32 re-association only.
 *    OR KL_EXCHANGE_COPY into `split` to exchange by device-to-device copies instead of RCCL (a rehearsal mode: members
 *    may then share a GPU, which RCCL refuses). Scene, textures, camera and parameters go to every member through the
 *    kl_group_* setters (or to one member through kl_group_member and the per-context calls). A group is used from one
 *    thread at a time; librccl is loaded when the first RCCL group is created. -- */
typedef struct kl_group kl_group;
enum { KL_SPLIT_TILES = 0, KL_SPLIT_SAMPLES = 1, KL_EXCHANGE_COPY = 0x100 };
KL_API kl_status kl_group_create(kl_group **out, const int32_t *devices, uint32_t n, uint32_t split,
                                 uint32_t width, uint32_t height, uint32_t band_rows);
KL_API kl_status kl_group_destroy(kl_group *g);
KL_API kl_status kl_group_size(kl_group *g, uint32_t *n);
KL_API kl_status kl_group_member(kl_group *g, uint32_t rank, kl_ctx **ctx);
KL_API kl_status kl_group_upload_scene(kl_group *g, const kl_sphere *spheres, uint32_t n_spheres, const kl_quad *quads, uint32_t n_quads,
                                       const kl_tri *tris, uint32_t n_tris, const kl_material *materials, uint32_t n_materials);
KL_API kl_status kl_group_load_builtin_scene(kl_group *g);
KL_API kl_status kl_group_upload_texture(kl_group *g, uint32_t which, const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels);
KL_API kl_status kl_group_set_params(kl_group *g, const kl_params *p);
KL_API kl_status kl_group_set_camera(kl_group *g, const kl_camera *cam);
KL_API kl_status kl_group_set_mode(kl_group *g, uint32_t mode);
KL_API kl_status kl_group_reset(kl_group *g);
/* n frames from first_frame (KL_CONTINUE: after the last ones), queued on every member; returns without waiting. If a member
 * fails, the members before it already hold frames the others lack: the group then answers KL_E_STATE to further render and
 * gather calls until kl_group_reset. */
KL_API kl_status kl_group_render_frames(kl_group *g, uint32_t n, uint32_t first_frame);
/* the exchange: afterwards every member holds the whole image (sample ranges: the mean over the frames rendered since the
 * last reset) on its device; member 0's copy goes to rgba (height x width x 4 floats, row 0 = bottom) unless that is NULL */
KL_API kl_status kl_group_gather(kl_group *g, float *rgba, size_t bytes);
KL_API kl_status kl_group_image_device(kl_group *g, uint32_t rank, const float **rgba_device);
/* text of the group's last error; g = NULL: why the last kl_group_create on the calling thread failed */
KL_API const char *kl_group_last_error(kl_group *g);

KL_API const char *kl_status_string(kl_status s);
KL_API const char *kl_last_error(kl_ctx *ctx);
KL_API uint32_t kl_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KL_RENDER_H */
