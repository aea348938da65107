/*
 * kl_oracle.h -- TEST INFRASTRUCTURE. CPU restatement of the reference's
 * per-pixel path-tracing hot path (resources/Shaders/RawFrame.glsl,
 * Accumulation.glsl, PostProcess.glsl of the reference checkout), in strict
 * fp32 C. Only tests, the smoke check and the CPU-baseline leg of the
 * benchmark may link or load this; the product library never does.
 *
 * Parity status: pinned by golden vectors produced by executing the
 * reference shaders themselves on Mesa llvmpipe (oracle/ref_harness.c,
 * tests/golden/), because the reference owns no tests of its own.
 */
#ifndef KL_ORACLE_H
#define KL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* material types, RawFrame.glsl:126-129 */
enum { KO_DIFFUSE = 0, KO_GLASS = 1, KO_EMISSIVE = 2, KO_TEXTURED = 3 };
/* view modes, include/Rendering/Opengl/GLSL_Renderer.h:15-21 */
enum { KO_MODE_AO = 0, KO_MODE_PT = 1, KO_MODE_Z = 2 };
enum { KO_TEX_HDRI = 0, KO_TEX_ALBEDO = 1 };
enum { KO_TONEMAP_CLAMP = 0, KO_TONEMAP_ACES = 1 };

/* RawFrame.glsl:131-137 */
typedef struct { int32_t type; float color[3]; float emissive; float roughness; float ior; float pad; } ko_material;
typedef struct { float center[3]; float radius; uint32_t material; } ko_sphere;             /* :151-155 */
typedef struct { float v0[3], v1[3], v2[3], v3[3]; uint32_t material; } ko_quad;            /* :156-162 */
typedef struct { float v0[3], v1[3], v2[3]; float uv0[2], uv1[2], uv2[2]; uint32_t material; } ko_tri; /* :163-168 + uv */
typedef struct { float pos[3], front[3], up[3]; float focal, sensor; } ko_camera;           /* :9-13 */

typedef struct {
    uint32_t width, height;
    uint32_t mode;          /* KO_MODE_* */
    uint32_t max_bounces;   /* RAY_BOUNCES, RawFrame.glsl:36 */
    uint32_t max_frames;    /* SAMPLES, :38; 0 = never freeze */
    float eps;              /* EPSILON :40 */
    float max_dist;         /* MAX_DIST :35 */
    float ao_length;        /* AO_LENGTH :33 */
    float hdri_strength;    /* HDRI_STRENGTH :32 */
    int32_t use_bvh;        /* 0 = brute force in declaration order, 1 = oracle's own BVH */
    uint32_t spp;           /* SPP, :37: spp x spp sub-samples per pixel and frame in the AO and path-traced modes; 0 = 1 */
} ko_params;

typedef struct {
    uint64_t paths, rays, nodes, sphere_tests, quad_tests, tri_tests;
} ko_counters;

typedef struct ko_scene ko_scene;

void ko_default_params(ko_params *p);
void ko_default_camera(ko_camera *c);    /* source/Object/Camera.cpp:3-14 */

ko_scene *ko_scene_create(const ko_sphere *s, uint32_t ns, const ko_quad *q, uint32_t nq,
                          const ko_tri *t, uint32_t nt, const ko_material *m, uint32_t nm);
ko_scene *ko_scene_builtin(void);        /* RawFrame.glsl:171-192 */
void ko_scene_destroy(ko_scene *s);
int ko_scene_set_texture(ko_scene *s, int which, const uint8_t *pixels, uint32_t w, uint32_t h, uint32_t channels);
/* float RGB(A) equirectangular environment in place of the 8-bit one (NULL: back to 8-bit) */
int ko_scene_set_env_float(ko_scene *s, const float *pixels, uint32_t w, uint32_t h, uint32_t channels);

/* Renders rows [y0,y1) of one raw frame into rgba (full-image layout,
 * row 0 = bottom row = gl_FragCoord.y 0.5). last_acc supplies the
 * `iFrame >= SAMPLES` freeze branch and may be NULL when max_frames == 0.
 * Returns 0 on success. */
int ko_render_raw(const ko_scene *s, const ko_camera *cam, const ko_params *p, uint32_t frame,
                  uint32_t y0, uint32_t y1, float *rgba, const float *last_acc,
                  ko_counters *counters, int nthreads);
/* Accumulation.glsl:16-24 */
void ko_accumulate(float *acc, const float *raw, size_t npix, uint32_t frame, int camera_change);
/* the same through RGBA8 attachments, as the application ships (FBT.cpp:6,21): raw and mean rounded to 8 bits every frame */
void ko_accumulate_unorm8(float *acc, const float *raw, size_t npix, uint32_t frame, int camera_change);
/* PostProcess.glsl:34-39 (CLAMP = what reaches the screen; ACES = the dead-code curve :19-37) */
void ko_resolve(const float *acc, uint8_t *rgba8, size_t npix, int tonemap);

/* unit-level entry points for known-answer tests */
void ko_hash(const uint32_t seed[4], uint32_t out[4]);                       /* :62-68 */
void ko_rand4(uint32_t x, uint32_t y, uint32_t frame, float out[4]);         /* :69-73,82-85 */
void ko_camera_ray(const ko_camera *c, uint32_t w, uint32_t h, uint32_t x, uint32_t y, int subpixel, float o[3], float d[3]);
int ko_sphere_intersect(const float o[3], const float d[3], const float c[3], float r, float eps, float *t);
int ko_quad_intersect(const float o[3], const float d[3], const float v0[3], const float v1[3],
                      const float v2[3], const float v3[3], float *t, float uv[2]);
int ko_tri_intersect(const float o[3], const float d[3], const float v0[3], const float v1[3],
                     const float v2[3], float *t, float bary[2]);
void ko_env_uv(const float d[3], float uv[2]);                               /* :330-336 */
float ko_sinf(float x); float ko_cosf(float x); float ko_atan2f(float y, float x); float ko_acosf(float x);
/* closest hit for one ray: returns primitive id (spheres, then quads, then tris) or -1 */
int64_t ko_trace(const ko_scene *s, const ko_params *p, const float o[3], const float d[3], float *t, float uv[2]);

#ifdef __cplusplus
}
#endif
#endif
