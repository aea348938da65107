/*
 * kl_oracle.c -- TEST INFRASTRUCTURE (see kl_oracle.h). Strict-fp32 CPU
 * restatement of the reference path-tracing shader. Every function cites the
 * lines of resources/Shaders/RawFrame.glsl (abbreviated RF) it follows.
 *
 * Build with -ffp-contract=off -fno-fast-math: reference arithmetic is
 * restated as separate IEEE fp32 operations in source order; the functions
 * that have no reference counterpart (triangle test, BVH slabs) use explicit
 * fmaf() where they fuse, so that a second implementation can match bit for
 * bit.
 *
 * Choices the GLSL text leaves to the implementation, fixed here:
 *   normalize(v) = v * (1 / sqrt(dot(v,v)))      (KO_NORMALIZE_DIV: v / sqrt)
 *   dot(a,b)     = (a.z*b.z + a.y*b.y) + a.x*b.x (KO_DOT_FORWARD: x first)
 *   sin/cos/atan/acos = the Cephes single-precision algorithms below
 *   vec3 col is zero-initialised (RF:433,445 leave it undefined)
 *   Hit.Ray_Inside = false for quads/triangles, Hit_UV = (0,0) for spheres
 *   texture(): NEAREST, clamp-to-edge, texel = floor(coord * size), value = byte / 255
 *   acos() input is clamped to [-1,1]; NaN texture coordinates select texel 0
 */
#include "kl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) {
#ifdef KO_DOT_FORWARD
    return (a.x * b.x + a.y * b.y) + a.z * b.z;
#else
    return (a.z * b.z + a.y * b.y) + a.x * b.x;
#endif
}
static inline v3 vcross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float vlength(v3 a) { return sqrtf(vdot(a, a)); }
static inline v3 vnormalize(v3 a) {
#ifdef KO_NORMALIZE_DIV
    float l = sqrtf(vdot(a, a));
    return V(a.x / l, a.y / l, a.z / l);
#else
    float il = 1.0f / sqrtf(vdot(a, a));
    return vscale(a, il);
#endif
}
/* GLSL reflect(I,N) = I - 2*dot(N,I)*N */
static inline v3 vreflect(v3 i, v3 n) { return vsub(i, vscale(n, 2.0f * vdot(n, i))); }
/* GLSL mix(x,y,a) = x*(1-a) + y*a */
static inline v3 vmix(v3 x, v3 y, float a) {
#ifdef KO_MIX_FAST
    return vadd(x, vscale(vsub(y, x), a));
#else
    return vadd(vscale(x, 1.0f - a), vscale(y, a));
#endif
}
static inline float cross2d(float ax, float ay, float bx, float by) { return ax * by - ay * bx; } /* RF:54 */

/* ------------------------------------------------------------------ math */
/* Single-precision sin/cos/atan/acos after the Cephes library (sinf.c, atanf.c,
 * asinf.c), evaluated with separate multiplies and adds. */
static const float KO_FOPI = 1.27323954473516f;
static const float KO_DP1 = 0.78515625f, KO_DP2 = 2.4187564849853515625e-4f, KO_DP3 = 3.77489497744594108e-8f;
static const float KO_PI = 3.14159265358979323846f, KO_PIO2 = 1.5707963267948966192f, KO_PIO4 = 0.7853981633974483096f;

static inline float ko_sin_poly(float x, float z) {
    return ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x + x;
}
static inline float ko_cos_poly(float z) {
    return ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z - 0.5f * z + 1.0f;
}
float ko_sinf(float x) {
    int sign = 1;
    if (x < 0.0f) { sign = -1; x = -x; }
    if (!(x <= 8192.0f)) return x - x;
    int j = (int)(KO_FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { sign = -sign; j -= 4; }
    x = ((x - y * KO_DP1) - y * KO_DP2) - y * KO_DP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? ko_cos_poly(z) : ko_sin_poly(x, z);
    return sign < 0 ? -r : r;
}
float ko_cosf(float x) {
    int sign = 1;
    if (x < 0.0f) x = -x;
    if (!(x <= 8192.0f)) return x - x;
    int j = (int)(KO_FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { j -= 4; sign = -sign; }
    if (j > 1) sign = -sign;
    x = ((x - y * KO_DP1) - y * KO_DP2) - y * KO_DP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? ko_sin_poly(x, z) : ko_cos_poly(z);
    return sign < 0 ? -r : r;
}
static float ko_atanf(float x) {
    int sign = 1;
    if (x < 0.0f) { sign = -1; x = -x; }
    float y;
    if (x > 2.414213562373095f) { y = KO_PIO2; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = KO_PIO4; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z - 3.33329491539E-1f) * z * x + x;
    return sign < 0 ? -y : y;
}
float ko_atan2f(float y, float x) {
    int code = 0;
    if (x < 0.0f) code = 2;
    if (y < 0.0f) code |= 1;
    if (x == 0.0f) {
        if (code & 1) return -KO_PIO2;
        if (y == 0.0f) return 0.0f;
        return KO_PIO2;
    }
    if (y == 0.0f) return (code & 2) ? KO_PI : 0.0f;
    float w = code == 2 ? KO_PI : code == 3 ? -KO_PI : 0.0f;
    return w + ko_atanf(y / x);
}
static float ko_asinf(float x) {
    int sign = 1, flag = 0;
    float a = x, z;
    if (x < 0.0f) { sign = -1; a = -x; }
    if (a < 1.0e-4f) { z = a; }
    else {
        float s;
        if (a > 0.5f) { z = 0.5f * (1.0f - a); s = sqrtf(z); flag = 1; }
        else { s = a; z = s * s; }
        z = ((((4.2163199048E-2f * z + 2.4181311049E-2f) * z + 4.5470025998E-2f) * z + 7.4953002686E-2f) * z + 1.6666752422E-1f) * z * s + s;
        if (flag) { z = z + z; z = KO_PIO2 - z; }
    }
    return sign < 0 ? -z : z;
}
float ko_acosf(float x) {
    if (x < -1.0f) x = -1.0f;
    if (x > 1.0f) x = 1.0f;
    if (x < -0.5f) return KO_PI - 2.0f * ko_asinf(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * ko_asinf(sqrtf(0.5f * (1.0f - x)));
    return KO_PIO2 - ko_asinf(x);
}

/* ------------------------------------------------------------------- RNG */
/* RF:62-68 */
void ko_hash(const uint32_t seed[4], uint32_t out[4]) {
    uint32_t x = seed[0] * 1664525u + 1013904223u;
    uint32_t y = seed[1] * 1664525u + 1013904223u;
    uint32_t z = seed[2] * 1664525u + 1013904223u;
    uint32_t w = seed[3] * 1664525u + 1013904223u;
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x += y * w; y += z * x; z += x * y; w += y * z;
    out[0] = x; out[1] = y; out[2] = z; out[3] = w;
}
/* RF:69-73 with the seed of RF:82-85. The seed is never advanced (hash takes
 * it by value), so every rand*() of one (pixel, frame) returns these values. */
void ko_rand4(uint32_t px, uint32_t py, uint32_t frame, float out[4]) {
    uint32_t seed[4] = { px, py, frame, px + py }, h[4];
    ko_hash(seed, h);
    for (int i = 0; i < 4; i++) out[i] = (float)h[i] / 4294967296.0f; /* float(0xffffffffu) == 2^32 */
}

/* ----------------------------------------------------------------- scene */
typedef struct { float lo[3], hi[3]; int32_t left, right; uint32_t start, count; } ko_node;

struct ko_scene {
    ko_sphere *spheres; uint32_t ns;
    ko_quad *quads; uint32_t nq;
    ko_tri *tris; uint32_t nt;
    ko_material *mats; uint32_t nm;
    uint8_t *tex[2]; uint32_t tw[2], th[2];   /* RGBA8 */
    float *envf; uint32_t envf_w, envf_h;     /* optional float RGB environment (no reference counterpart), replaces tex[HDRI] */
    ko_node *nodes; uint32_t nnodes; uint32_t *order;  /* BVH over global primitive ids */
};

void ko_default_params(ko_params *p) {
    p->width = 1920; p->height = 1080; p->mode = KO_MODE_PT;
    p->max_bounces = 32; p->max_frames = 30;
    p->eps = 0.001f; p->max_dist = 5000.0f; p->ao_length = 1.0f; p->hdri_strength = 1.0f;
    p->use_bvh = 0; p->spp = 1;
}
void ko_default_camera(ko_camera *c) {
    c->pos[0] = 0; c->pos[1] = 1; c->pos[2] = 5;
    c->front[0] = 0; c->front[1] = 0; c->front[2] = -1;
    c->up[0] = 0; c->up[1] = 1; c->up[2] = 0;
    c->focal = 0.05f; c->sensor = 0.036f;
}

static void prim_bounds(const ko_scene *s, uint32_t id, float lo[3], float hi[3]) {
    if (id < s->ns) {
        const ko_sphere *sp = &s->spheres[id];
        for (int k = 0; k < 3; k++) { lo[k] = sp->center[k] - sp->radius; hi[k] = sp->center[k] + sp->radius; }
    } else if (id < s->ns + s->nq) {
        const ko_quad *q = &s->quads[id - s->ns];
        for (int k = 0; k < 3; k++) {
            lo[k] = fminf(fminf(q->v0[k], q->v1[k]), fminf(q->v2[k], q->v3[k]));
            hi[k] = fmaxf(fmaxf(q->v0[k], q->v1[k]), fmaxf(q->v2[k], q->v3[k]));
        }
    } else {
        const ko_tri *t = &s->tris[id - s->ns - s->nq];
        for (int k = 0; k < 3; k++) {
            lo[k] = fminf(t->v0[k], fminf(t->v1[k], t->v2[k]));
            hi[k] = fmaxf(t->v0[k], fmaxf(t->v1[k], t->v2[k]));
        }
    }
    /* widen so that rounding in the slab test can never cull a true hit */
    for (int k = 0; k < 3; k++) {
        float pad = 1e-5f + 1e-5f * fmaxf(fabsf(lo[k]), fabsf(hi[k]));
        lo[k] -= pad; hi[k] += pad;
    }
}

typedef struct { float lo[3], hi[3], c[3]; } ko_pb;
static ko_pb *g_pb; static int g_axis;
static int cmp_centroid(const void *a, const void *b) {
    float ca = g_pb[*(const uint32_t *)a].c[g_axis], cb = g_pb[*(const uint32_t *)b].c[g_axis];
    return ca < cb ? -1 : ca > cb ? 1 : 0;
}
/* median split on the widest centroid axis; the oracle's tree only has to be
 * correct, its shape never shows in results */
static uint32_t build_rec(ko_scene *s, ko_pb *pb, uint32_t start, uint32_t count) {
    uint32_t idx = s->nnodes++;
    ko_node *n = &s->nodes[idx];
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < 3; k++) { n->lo[k] = INFINITY; n->hi[k] = -INFINITY; }
    for (uint32_t i = start; i < start + count; i++) {
        const ko_pb *b = &pb[s->order[i]];
        for (int k = 0; k < 3; k++) {
            n->lo[k] = fminf(n->lo[k], b->lo[k]); n->hi[k] = fmaxf(n->hi[k], b->hi[k]);
            clo[k] = fminf(clo[k], b->c[k]); chi[k] = fmaxf(chi[k], b->c[k]);
        }
    }
    n->start = start; n->count = count; n->left = n->right = -1;
    if (count <= 4) return idx;
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    g_pb = pb; g_axis = axis;
    qsort(s->order + start, count, sizeof(uint32_t), cmp_centroid);
    uint32_t half = count / 2;
    uint32_t l = build_rec(s, pb, start, half);
    uint32_t r = build_rec(s, pb, start + half, count - half);
    n = &s->nodes[idx];
    n->left = (int32_t)l; n->right = (int32_t)r; n->count = 0;
    return idx;
}
static void build_bvh(ko_scene *s) {
    uint32_t n = s->ns + s->nq + s->nt;
    if (n == 0) return;
    ko_pb *pb = (ko_pb *)malloc(sizeof(ko_pb) * n);
    s->order = (uint32_t *)malloc(sizeof(uint32_t) * n);
    s->nodes = (ko_node *)malloc(sizeof(ko_node) * (2 * (size_t)n));
    for (uint32_t i = 0; i < n; i++) {
        prim_bounds(s, i, pb[i].lo, pb[i].hi);
        for (int k = 0; k < 3; k++) pb[i].c[k] = 0.5f * (pb[i].lo[k] + pb[i].hi[k]);
        s->order[i] = i;
    }
    s->nnodes = 0;
    build_rec(s, pb, 0, n);
    free(pb);
}

ko_scene *ko_scene_create(const ko_sphere *sp, uint32_t ns, const ko_quad *q, uint32_t nq,
                          const ko_tri *t, uint32_t nt, const ko_material *m, uint32_t nm) {
    ko_scene *s = (ko_scene *)calloc(1, sizeof *s);
    s->ns = ns; s->nq = nq; s->nt = nt; s->nm = nm;
    s->spheres = (ko_sphere *)malloc(sizeof(ko_sphere) * (ns ? ns : 1)); if (ns) memcpy(s->spheres, sp, sizeof(ko_sphere) * ns);
    s->quads = (ko_quad *)malloc(sizeof(ko_quad) * (nq ? nq : 1)); if (nq) memcpy(s->quads, q, sizeof(ko_quad) * nq);
    s->tris = (ko_tri *)malloc(sizeof(ko_tri) * (nt ? nt : 1)); if (nt) memcpy(s->tris, t, sizeof(ko_tri) * nt);
    s->mats = (ko_material *)malloc(sizeof(ko_material) * (nm ? nm : 1)); if (nm) memcpy(s->mats, m, sizeof(ko_material) * nm);
    static const uint8_t white[4] = {255, 255, 255, 255};
    for (int k = 0; k < 2; k++) { s->tex[k] = (uint8_t *)malloc(4); memcpy(s->tex[k], white, 4); s->tw[k] = s->th[k] = 1; }
    build_bvh(s);
    return s;
}
void ko_scene_destroy(ko_scene *s) {
    if (!s) return;
    free(s->spheres); free(s->quads); free(s->tris); free(s->mats);
    free(s->tex[0]); free(s->tex[1]); free(s->envf); free(s->nodes); free(s->order); free(s);
}
/* source/I-O/File.cpp:13-29 + source/Rendering/Opengl/Texture.cpp:8-12:
 * 8-bit, 1/3/4 channels, no flip; stored here expanded to RGBA8 */
int ko_scene_set_texture(ko_scene *s, int which, const uint8_t *px, uint32_t w, uint32_t h, uint32_t ch) {
    if (which < 0 || which > 1 || !px || !w || !h || (ch != 1 && ch != 3 && ch != 4)) return -1;
    uint8_t *d = (uint8_t *)malloc((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        if (ch == 4) { memcpy(d + 4 * i, px + 4 * i, 4); }
        else if (ch == 3) { d[4*i] = px[3*i]; d[4*i+1] = px[3*i+1]; d[4*i+2] = px[3*i+2]; d[4*i+3] = 255; }
        else { d[4*i] = px[i]; d[4*i+1] = 0; d[4*i+2] = 0; d[4*i+3] = 255; } /* GL_RED upload: g=b=0 */
    }
    free(s->tex[which]); s->tex[which] = d; s->tw[which] = w; s->th[which] = h;
    return 0;
}

/* Float equirectangular environment (the role Background.exr was meant to play; the reference never loads
 * it). Same lookup as the 8-bit one: NEAREST, clamp, rows as given; the texel is used as it is, times
 * HDRI_STRENGTH. channels = 3 or 4 (alpha ignored); px = NULL goes back to the 8-bit texture. */
int ko_scene_set_env_float(ko_scene *s, const float *px, uint32_t w, uint32_t h, uint32_t ch) {
    if (!s) return -1;
    if (!px) { free(s->envf); s->envf = NULL; s->envf_w = s->envf_h = 0; return 0; }
    if (!w || !h || (ch != 3 && ch != 4)) return -1;
    float *d = (float *)malloc((size_t)w * h * 3 * sizeof(float));
    for (size_t i = 0; i < (size_t)w * h; i++) { d[3*i] = px[ch*i]; d[3*i+1] = px[ch*i+1]; d[3*i+2] = px[ch*i+2]; }
    free(s->envf); s->envf = d; s->envf_w = w; s->envf_h = h;
    return 0;
}

/* RF:171-192 as data: six spheres, seven quads, one material each */
ko_scene *ko_scene_builtin(void) {
    static const float S[6][11] = {
        /* cx cy cz r | type r g b emissive roughness ior */
        {-1, 0.3f, -1, 0.2f, KO_DIFFUSE, 1, 1, 1, 0, 1, 1.2f},
        {-1, 0.9f, -1, 0.2f, KO_DIFFUSE, 1, 1, 1, 0, 0, 1.2f},
        {-1, 1.5f, -1, 0.2f, KO_GLASS, 1, 1, 1, 0, 0, 1.2f},
        { 1, 0.3f, -1, 0.2f, KO_DIFFUSE, 1, 0, 0, 0, 0, 1.2f},
        { 1, 0.9f, -1, 0.2f, KO_DIFFUSE, 0, 1, 0, 0, 0.1f, 1.2f},
        { 1, 1.5f, -1, 0.2f, KO_DIFFUSE, 0, 0, 1, 0, 0, 1.2f},
    };
    static const float Q[7][19] = {
        {-2.66f, 0, -15,  2.66f, 0, -15,  2.66f, 0, 5,  -2.66f, 0, 5,   KO_DIFFUSE, 1, 1, 1, 0, 1, 1},
        { 2.66f, 0, -15,  2.66f, 0, 5,    2.66f, 3, 5,   2.66f, 3, -15, KO_DIFFUSE, 1, 1, 1, 0, 0.01f, 1},
        {-2.66f, 0, -15, -2.66f, 0, 5,   -2.66f, 3, 5,  -2.66f, 3, -15, KO_DIFFUSE, 1, 1, 1, 0, 0.01f, 1},
        {-2.66f, 0, -15,  2.66f, 0, -15,  2.66f, 3, -15, -2.66f, 3, -15, KO_TEXTURED, 1, 1, 1, 0, 1, 1},
        {-2.66f, 3, -15,  2.66f, 3, -15,  2.66f, 3, 5,  -2.66f, 3, 5,   KO_DIFFUSE, 1, 1, 1, 0, 1, 1},
        {-1.8f, 2.9f, -10, -1, 2.9f, -10, -1, 2.9f, 3,  -1.8f, 2.9f, 3, KO_EMISSIVE, 1, 1, 1, 2.5f, 1, 1},
        { 1.8f, 2.9f, -10,  1, 2.9f, -10,  1, 2.9f, 3,   1.8f, 2.9f, 3, KO_EMISSIVE, 1, 1, 1, 2.5f, 1, 1},
    };
    ko_sphere sp[6]; ko_quad q[7]; ko_material m[13];
    for (int i = 0; i < 6; i++) {
        memcpy(sp[i].center, S[i], 12); sp[i].radius = S[i][3]; sp[i].material = (uint32_t)i;
        m[i].type = (int32_t)S[i][4]; memcpy(m[i].color, &S[i][5], 12);
        m[i].emissive = S[i][8]; m[i].roughness = S[i][9]; m[i].ior = S[i][10]; m[i].pad = 0;
    }
    for (int i = 0; i < 7; i++) {
        memcpy(q[i].v0, &Q[i][0], 12); memcpy(q[i].v1, &Q[i][3], 12);
        memcpy(q[i].v2, &Q[i][6], 12); memcpy(q[i].v3, &Q[i][9], 12);
        q[i].material = (uint32_t)(6 + i);
        ko_material *mm = &m[6 + i];
        mm->type = (int32_t)Q[i][12]; memcpy(mm->color, &Q[i][13], 12);
        mm->emissive = Q[i][16]; mm->roughness = Q[i][17]; mm->ior = Q[i][18]; mm->pad = 0;
    }
    return ko_scene_create(sp, 6, q, 7, NULL, 0, m, 13);
}

/* --------------------------------------------------------- intersections */
/* RF:196-216. `radius` is the field the shader calls Diameter. */
int ko_sphere_intersect(const float o[3], const float d[3], const float c[3], float r, float eps, float *t) {
    v3 oc = vsub(V(o[0], o[1], o[2]), V(c[0], c[1], c[2]));
    v3 dir = V(d[0], d[1], d[2]);
    float b = vdot(oc, dir);
    float delta = b * b - vdot(oc, oc) + r * r;
    if (delta < 0.0f) return 0;
    float sq = sqrtf(delta);
    if (-b - sq > eps) { *t = -b - sq; return 1; }
    else if (-b + sq > eps) { *t = -b + sq; return 1; }
    return 0;
}
/* RF:218-269, inverse bilinear patch test; returns UV = (v, 1-u). A NaN t can
 * be returned with "true": the caller's `t < best` rejects it (RF:316). */
int ko_quad_intersect(const float o[3], const float d[3], const float q0[3], const float q1[3],
                      const float q2[3], const float q3[3], float *t_out, float uv[2]) {
    static const int Quad_Face[4] = {1, 2, 0, 1}; /* RF:44 */
    v3 v0 = V(q0[0], q0[1], q0[2]);
    v3 a = vsub(V(q1[0], q1[1], q1[2]), v0);
    v3 b = vsub(V(q3[0], q3[1], q3[2]), v0);
    v3 c = vsub(V(q2[0], q2[1], q2[2]), v0);
    v3 p = vsub(V(o[0], o[1], o[2]), v0);
    v3 dir = V(d[0], d[1], d[2]);
    v3 nor = vcross(a, b);
    float t = -vdot(p, nor) / vdot(dir, nor);
    if (t < 0.0f) return 0;
    v3 pos = vadd(p, vscale(dir, t));
    v3 mor = V(fabsf(nor.x), fabsf(nor.y), fabsf(nor.z));
    int id = (mor.x > mor.y && mor.x > mor.z) ? 0 : (mor.y > mor.z) ? 1 : 2;
    int idu = Quad_Face[id], idv = Quad_Face[id + 1];
    const float *pa = &pos.x, *aa = &a.x, *ba = &b.x, *ca = &c.x, *na = &nor.x;
    float kpx = pa[idu], kpy = pa[idv];
    float kax = aa[idu], kay = aa[idv];
    float kbx = ba[idu], kby = ba[idv];
    float kcx = ca[idu], kcy = ca[idv];
    float kgx = kcx - kbx - kax, kgy = kcy - kby - kay;
    float k0 = cross2d(kpx, kpy, kbx, kby);
    float k2 = cross2d(kcx - kbx, kcy - kby, kax, kay);
    float k1 = cross2d(kpx, kpy, kgx, kgy) - na[id];
    float u, v;
    if (fabsf(k2) < 0.00001f) {
        v = -k0 / k1;
        u = cross2d(kpx, kpy, kax, kay) / k1;
    } else {
        float w = k1 * k1 - 4.0f * k0 * k2;
        if (w < 0.0f) return 0;
        w = sqrtf(w);
        float ik2 = 1.0f / (2.0f * k2);
        v = (-k1 - w) * ik2;
        if (v < 0.0f || v > 1.0f) v = (-k1 + w) * ik2;
        u = (kpx - kax * v) / (kbx + kgx * v);
    }
    if (u < 0.0f || u > 1.0f || v < 0.0f || v > 1.0f) return 0;
    *t_out = t;
    uv[0] = v; uv[1] = 1.0f - u;
    return 1;
}
/* No reference counterpart (RF:163-168 only declares Tri). Two-sided
 * Moeller-Trumbore on (v0, e1 = v1-v0, e2 = v2-v0) with explicit fused
 * multiply-adds; bary = (u, v) weights of v1 and v2. */
static inline float fdot(v3 a, v3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
static inline v3 fcross(v3 a, v3 b) {
    return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline int tri_test(v3 o, v3 d, v3 v0, v3 e1, v3 e2, float *t, float *bu, float *bv) {
    v3 p = fcross(d, e2);
    float det = fdot(e1, p);
    float inv = 1.0f / det;
    v3 s = vsub(o, v0);
    float u = fdot(s, p) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return 0;
    v3 q = fcross(s, e1);
    float v = fdot(d, q) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return 0;
    *t = fdot(e2, q) * inv; *bu = u; *bv = v;
    return 1;
}
int ko_tri_intersect(const float o[3], const float d[3], const float a[3], const float b[3],
                     const float c[3], float *t, float bary[2]) {
    v3 v0 = V(a[0], a[1], a[2]);
    return tri_test(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), v0,
                    vsub(V(b[0], b[1], b[2]), v0), vsub(V(c[0], c[1], c[2]), v0), t, &bary[0], &bary[1]);
}

typedef struct {
    float t; int64_t prim; float u, v;
} ko_hit;

/* one candidate primitive against the running closest hit. Acceptance is
 * RF:302,316 (`t < best && t > 0.001`); on exactly equal t the lower primitive
 * id wins, which is what declaration-order brute force with a strict `<` does. */
static inline void test_prim(const ko_scene *s, const ko_params *p, uint32_t id, v3 o, v3 d, ko_hit *h, ko_counters *cn) {
    float t, uv[2] = {0, 0};
    int ok;
    if (id < s->ns) {
        const ko_sphere *sp = &s->spheres[id];
        if (cn) cn->sphere_tests++;
        ok = ko_sphere_intersect(&o.x, &d.x, sp->center, sp->radius, p->eps, &t);
    } else if (id < s->ns + s->nq) {
        const ko_quad *q = &s->quads[id - s->ns];
        if (cn) cn->quad_tests++;
        ok = ko_quad_intersect(&o.x, &d.x, q->v0, q->v1, q->v2, q->v3, &t, uv);
    } else {
        const ko_tri *tr = &s->tris[id - s->ns - s->nq];
        if (cn) cn->tri_tests++;
        ok = ko_tri_intersect(&o.x, &d.x, tr->v0, tr->v1, tr->v2, &t, uv);
    }
    if (!ok) return;
    if (t > 0.001f && (t < h->t || (t == h->t && (int64_t)id < h->prim))) {
        h->t = t; h->prim = id; h->u = uv[0]; h->v = uv[1];
    }
}

static ko_hit closest_hit(const ko_scene *s, const ko_params *p, v3 o, v3 d, ko_counters *cn) {
    ko_hit h; h.t = p->max_dist; h.prim = -1; h.u = h.v = 0.0f;
    uint32_t n = s->ns + s->nq + s->nt;
    if (cn) cn->rays++;
    if (!p->use_bvh || !s->nodes) {
        for (uint32_t i = 0; i < n; i++) test_prim(s, p, i, o, d, &h, cn);
        /* a hit exactly at max_dist never registers: RF:302 is a strict `<` */
        return h;
    }
    v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    uint32_t stack[128]; int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const ko_node *nd = &s->nodes[stack[--sp]];
        if (cn) cn->nodes++;
        float t0 = 0.0f, t1 = h.t;
        const float *oo = &o.x, *ii = &inv.x;
        int miss = 0;
        for (int k = 0; k < 3; k++) {
            float ta = (nd->lo[k] - oo[k]) * ii[k], tb = (nd->hi[k] - oo[k]) * ii[k];
            float tn = fminf(ta, tb), tf = fmaxf(ta, tb); /* fmin/fmax drop NaN operands: conservative */
            t0 = fmaxf(t0, tn); t1 = fminf(t1, tf);
        }
        if (t0 > t1) miss = 1;
        /* a NaN ray (RF:366 can produce one) hits nothing in brute force either */
        if (d.x != d.x || d.y != d.y || d.z != d.z || o.x != o.x || o.y != o.y || o.z != o.z) miss = 1;
        if (miss) continue;
        if (nd->left < 0) {
            for (uint32_t i = nd->start; i < nd->start + nd->count; i++) test_prim(s, p, s->order[i], o, d, &h, cn);
        } else {
            if (sp + 2 > 128) break;
            stack[sp++] = (uint32_t)nd->left; stack[sp++] = (uint32_t)nd->right;
        }
    }
    return h;
}

int64_t ko_trace(const ko_scene *s, const ko_params *p, const float o[3], const float d[3], float *t, float uv[2]) {
    ko_hit h = closest_hit(s, p, V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), NULL);
    *t = h.t; uv[0] = h.u; uv[1] = h.v;
    return h.prim;
}

/* the fields of RF:142-150 that shading reads, rebuilt from (ray, t, prim) */
typedef struct { v3 pos, n; int inside; const ko_material *mat; float u, v; } ko_surf;

static ko_surf surface(const ko_scene *s, v3 o, v3 d, const ko_hit *h) {
    ko_surf sf; memset(&sf, 0, sizeof sf);
    uint32_t id = (uint32_t)h->prim;
    sf.pos = vadd(o, vscale(d, h->t));                                   /* RF:304,318 */
    if (id < s->ns) {
        const ko_sphere *sp = &s->spheres[id];
        v3 c = V(sp->center[0], sp->center[1], sp->center[2]);
        sf.n = vnormalize(vsub(sf.pos, c));                              /* RF:305 */
        sf.inside = vlength(vsub(o, c)) <= sp->radius;                   /* RF:308 */
        if (sf.inside) sf.n = vscale(sf.n, -1.0f);                       /* RF:309 */
        sf.mat = &s->mats[sp->material];
        sf.u = 0.0f; sf.v = 0.0f;
    } else if (id < s->ns + s->nq) {
        const ko_quad *q = &s->quads[id - s->ns];
        v3 v1 = V(q->v1[0], q->v1[1], q->v1[2]);
        v3 nor = vnormalize(vcross(vsub(V(q->v2[0], q->v2[1], q->v2[2]), v1), vsub(V(q->v3[0], q->v3[1], q->v3[2]), v1))); /* RF:319 */
        sf.n = vdot(nor, d) < 0.0f ? nor : vneg(nor);                    /* faceforward, RF:320 */
        sf.mat = &s->mats[q->material];
        sf.u = h->u; sf.v = h->v;
    } else {
        const ko_tri *t = &s->tris[id - s->ns - s->nq];
        v3 v0 = V(t->v0[0], t->v0[1], t->v0[2]);
        v3 e1 = vsub(V(t->v1[0], t->v1[1], t->v1[2]), v0), e2 = vsub(V(t->v2[0], t->v2[1], t->v2[2]), v0);
        v3 nor = vnormalize(vcross(e1, e2));
        sf.n = vdot(nor, d) < 0.0f ? nor : vneg(nor);
        sf.mat = &s->mats[t->material];
        float w = (1.0f - h->u) - h->v;
        sf.u = (t->uv0[0] * w + t->uv1[0] * h->u) + t->uv2[0] * h->v;
        sf.v = (t->uv0[1] * w + t->uv1[1] * h->u) + t->uv2[1] * h->v;
    }
    return sf;
}

/* texture(sampler, uv).rgb with the sampler state of Texture.cpp:8-12 */
static v3 tex_fetch(const ko_scene *s, int which, float u, float v) {
    uint32_t w = s->tw[which], h = s->th[which];
    float fx = floorf(u * (float)w), fy = floorf(v * (float)h);
    uint32_t ix = !(fx >= 0.0f) ? 0u : (fx >= (float)w ? w - 1 : (uint32_t)fx);
    uint32_t iy = !(fy >= 0.0f) ? 0u : (fy >= (float)h ? h - 1 : (uint32_t)fy);
    const uint8_t *px = s->tex[which] + 4 * ((size_t)iy * w + ix);
#ifdef KO_UNORM_DIV
    return V((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
#else
    const float k = 1.0f / 255.0f;
    return V((float)px[0] * k, (float)px[1] * k, (float)px[2] * k);
#endif
}

/* RF:330-336 with eulerToRot(0,-90,0) (RF:98-122) folded to its constant
 * matrix: d' = (d.x, c*d.y + s*d.z, -s*d.y + c*d.z), c = cosf(-pi/2), s = -1 */
void ko_env_uv(const float d[3], float uv[2]) {
    const float c = -4.371138828673793e-08f, sn = -1.0f;
    float rx = (1.0f * d[0] + 0.0f * d[1]) + 0.0f * d[2];
    float ry = (0.0f * d[0] + c * d[1]) + sn * d[2];
    float rz = (0.0f * d[0] + (-sn) * d[1]) + c * d[2];
    float phi = ko_atan2f(ry, rx);
    float theta = ko_acosf(rz);
    uv[0] = phi / 6.28318530718f + 0.5f;
    uv[1] = theta / 3.14159265359f;
}
static v3 environment(const ko_scene *s, const ko_params *p, v3 d) {
    float uv[2];
    ko_env_uv(&d.x, uv);
    if (s->envf) {
        uint32_t w = s->envf_w, h = s->envf_h;
        float fx = floorf(uv[0] * (float)w), fy = floorf(uv[1] * (float)h);
        uint32_t ix = !(fx >= 0.0f) ? 0u : (fx >= (float)w ? w - 1 : (uint32_t)fx);
        uint32_t iy = !(fy >= 0.0f) ? 0u : (fy >= (float)h ? h - 1 : (uint32_t)fy);
        const float *px = s->envf + 3 * ((size_t)iy * w + ix);
        return vscale(V(px[0], px[1], px[2]), p->hdri_strength);
    }
    return vscale(tex_fetch(s, KO_TEX_HDRI, uv[0], uv[1]), p->hdri_strength); /* RF:338 */
}

/* RF:273-277 */
static v3 hemisphere(const float r[4]) {
    float px = 2.0f * 3.14159265359f * r[0];
    float py = sqrtf(r[1]);
    return vnormalize(V(ko_sinf(px) * py, ko_cosf(px) * py, sqrtf(1.0f - py * py)));
}
/* RF:279-292 */
static v3 cone_roughness(v3 dir, float theta, const float r[4]) {
    v3 left = vcross(dir, V(0.0f, 1.0f, 0.0f));
    left = vlength(left) > 0.1f ? vnormalize(left) : vnormalize(vcross(dir, V(0.0f, 0.0f, 1.0f)));
    v3 up = vnormalize(vcross(dir, left));
    float cos_theta = (1.0f - r[0]) + r[0] * ko_cosf(theta);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi = r[1] * 2.0f * 3.14159265359f;
    v3 a = vscale(vscale(left, ko_cosf(phi)), sin_theta);
    v3 b = vscale(vscale(up, ko_sinf(phi)), sin_theta);
    v3 c = vscale(dir, cos_theta);
    return vnormalize(vadd(vadd(a, b), c));
}

/* RF:409-415 */
static v3 zdepth(const ko_scene *s, const ko_params *p, v3 o, v3 d, ko_counters *cn) {
    ko_hit h = closest_hit(s, p, o, d, cn);
    if (h.t >= p->max_dist) return V(0, 0, 0);
    float z = 1.0f - fminf(h.t / 50.0f, 0.9f);
    return V(z, z, z);
}
/* RF:341-353 */
static v3 ambient_occlusion(const ko_scene *s, const ko_params *p, v3 o, v3 d, const float r[4], ko_counters *cn) {
    float dist = 0.0f;
    for (int b = 0; b < 2; b++) {
        ko_hit h = closest_hit(s, p, o, d, cn);
        if (h.t >= p->max_dist) return V(0, 0, 0);
        ko_surf sf = surface(s, o, d, &h);
        d = cone_roughness(sf.n, 10.0f, r);
        o = vadd(sf.pos, vscale(d, p->eps));
        dist = fminf(h.t / p->ao_length, 1.0f);
    }
    return V(dist, dist, dist);
}
/* RF:87-96 */
static inline float snell(float sin_theta, float iori, float iort) { return iori / iort * sin_theta; }
static inline float fresnel(float iori, float iort, float cosi, float cost) {
    float rpar = (iort * cosi - iori * cost) / (iort * cosi + iori * cost);
    float rper = (iori * cosi - iort * cost) / (iori * cosi + iort * cost);
    rpar *= rpar; rper *= rper;
    return (rpar + rper) / 2.0f;
}
/* RF:355-407 */
static v3 radiance(const ko_scene *s, const ko_params *p, v3 o, v3 d, const float r[4], ko_counters *cn) {
    v3 rad = V(0, 0, 0), brdf = V(1, 1, 1);
    for (uint32_t b = 0; b < p->max_bounces; b++) {
        ko_hit h = closest_hit(s, p, o, d, cn);
        if (h.t >= p->max_dist) return vadd(rad, vmul(brdf, environment(s, p, d)));
        ko_surf sf = surface(s, o, d, &h);
        const ko_material *m = sf.mat;
        if (m->type == KO_DIFFUSE || m->type == KO_TEXTURED) {
            v3 tangent = vnormalize(vcross(d, sf.n));
            v3 bitangent = vnormalize(vcross(sf.n, tangent));
            v3 nl = hemisphere(r);
            v3 local = vnormalize(vadd(vadd(vscale(tangent, nl.x), vscale(bitangent, nl.y)), vscale(sf.n, nl.z)));
            d = vnormalize(vmix(vreflect(d, sf.n), local, m->roughness));
            if (m->type == KO_TEXTURED) return vadd(rad, vmul(brdf, tex_fetch(s, KO_TEX_ALBEDO, sf.u, sf.v)));
            brdf = vmul(brdf, V(m->color[0], m->color[1], m->color[2]));
        } else if (m->type == KO_GLASS) {
            float cosi = fabsf(vdot(sf.n, d));
            float sini = sqrtf(1.0f - cosi * cosi);
            float iort = m->ior, iori = 1.0f;
            if (sf.inside) { iori = iort; iort = 1.0f; }
            float sint = snell(sini, iori, iort);
            float cost = sqrtf(1.0f - sint * sint);   /* NaN past the critical angle: RF has no TIR branch */
            float frsn = fresnel(iori, iort, cosi, cost);
            if (r[0] > frsn) {                        /* false for NaN -> reflect */
                v3 bit = vnormalize(vsub(d, vscale(sf.n, vdot(sf.n, d))));
                d = vnormalize(vsub(vscale(bit, sint), vscale(sf.n, cost)));
                brdf = vmul(brdf, V(m->color[0], m->color[1], m->color[2]));
            } else {
                d = vreflect(d, sf.n);
            }
        } else if (m->type == KO_EMISSIVE) {
            return vadd(rad, vscale(vmul(brdf, V(m->color[0], m->color[1], m->color[2])), m->emissive));
        }
        o = vadd(sf.pos, vscale(d, p->eps));          /* RF:404 */
    }
    return rad;
}

/* RF:417-422 and the uv arithmetic of RF:428-429,436 */
/* sub-sample (sx, sy) of the SPP x SPP grid of RF:434-436,446-448 */
static void camera_ray_sub(const ko_camera *c, uint32_t w, uint32_t h, uint32_t x, uint32_t y, int subpixel, uint32_t sx, uint32_t sy,
                           uint32_t spp, float o[3], float d[3]) {
    float rx = (float)w, ry = (float)h;
    float m = fmaxf(rx, ry);
    float ux = (((float)x + 0.5f) - 1.0f - rx / 2.0f) / m;
    float uy = (((float)y + 0.5f) - 1.0f - ry / 2.0f) / m;
    if (subpixel) {                                   /* RF:436: uv - 0.5 * pixel_size + vec2(x / SPP, y / SPP) * pixel_size */
        float psx = 1.0f / rx, psy = 1.0f / ry;
        ux = (ux - 0.5f * psx) + ((float)sx / (float)spp) * psx;
        uy = (uy - 0.5f * psy) + ((float)sy / (float)spp) * psy;
    }
    v3 pos = V(c->pos[0], c->pos[1], c->pos[2]), front = V(c->front[0], c->front[1], c->front[2]);
    v3 up = V(c->up[0], c->up[1], c->up[2]);
    v3 pc = vadd(pos, vscale(front, c->focal));
    v3 pu = vscale(vnormalize(vcross(front, up)), c->sensor);
    v3 pv = vscale(vnormalize(vcross(pu, front)), c->sensor / 1.0f);
    v3 dir = vnormalize(vsub(vadd(vadd(pc, vscale(pu, ux)), vscale(pv, uy)), pos));
    o[0] = pos.x; o[1] = pos.y; o[2] = pos.z;
    d[0] = dir.x; d[1] = dir.y; d[2] = dir.z;
}
void ko_camera_ray(const ko_camera *c, uint32_t w, uint32_t h, uint32_t x, uint32_t y, int subpixel, float o[3], float d[3]) {
    camera_ray_sub(c, w, h, x, y, subpixel, 0, 0, 1, o, d);      /* SPP = 1 as shipped */
}

/* RF:425-464 for rows [y0,y1) */
int ko_render_raw(const ko_scene *s, const ko_camera *cam, const ko_params *p, uint32_t frame,
                  uint32_t y0, uint32_t y1, float *rgba, const float *last_acc,
                  ko_counters *counters, int nthreads) {
    if (!s || !cam || !p || !rgba || y1 > p->height || y0 > y1) return -1;
    if (p->max_frames && frame >= p->max_frames && !last_acc) return -2;
    ko_counters total; memset(&total, 0, sizeof total);
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        ko_counters cn; memset(&cn, 0, sizeof cn);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (uint32_t y = y0; y < y1; y++) {
            for (uint32_t x = 0; x < p->width; x++) {
                float *out = rgba + 4 * ((size_t)y * p->width + x);
                if (p->max_frames && frame >= p->max_frames) {          /* RF:461-463 */
                    memcpy(out, last_acc + 4 * ((size_t)y * p->width + x), 16);
                    continue;
                }
                float r[4], o[3], d[3];
                ko_rand4(x, y, frame, r);
                v3 col = V(0, 0, 0);
                cn.paths++;
                const uint32_t spp = p->spp ? p->spp : 1u;            /* #define SPP, RF:37 */
                const float nsub = (float)(spp * spp);
                if (p->mode == KO_MODE_AO) {                          /* RF:432-442; every sub-sample sees the same random numbers */
                    for (uint32_t sx = 0; sx < spp; sx++)
                        for (uint32_t sy = 0; sy < spp; sy++) {
                            camera_ray_sub(cam, p->width, p->height, x, y, 1, sx, sy, spp, o, d);
                            col = vadd(col, ambient_occlusion(s, p, V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), r, &cn));
                        }
                    col = V(col.x / nsub, col.y / nsub, col.z / nsub);
                } else if (p->mode == KO_MODE_PT) {                   /* RF:444-455 */
                    for (uint32_t sx = 0; sx < spp; sx++)
                        for (uint32_t sy = 0; sy < spp; sy++) {
                            camera_ray_sub(cam, p->width, p->height, x, y, 1, sx, sy, spp, o, d);
                            col = vadd(col, radiance(s, p, V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), r, &cn));
                        }
                    col = V(col.x / nsub, col.y / nsub, col.z / nsub);
                } else {
                    ko_camera_ray(cam, p->width, p->height, x, y, 0, o, d);
                    col = zdepth(s, p, V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), &cn);
                }
                out[0] = col.x; out[1] = col.y; out[2] = col.z; out[3] = 1.0f;
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            total.paths += cn.paths; total.rays += cn.rays; total.nodes += cn.nodes;
            total.sphere_tests += cn.sphere_tests; total.quad_tests += cn.quad_tests; total.tri_tests += cn.tri_tests;
        }
    }
    if (counters) {
        counters->paths += total.paths; counters->rays += total.rays; counters->nodes += total.nodes;
        counters->sphere_tests += total.sphere_tests; counters->quad_tests += total.quad_tests; counters->tri_tests += total.tri_tests;
    }
    return 0;
}

/* Accumulation.glsl:16-24 */
void ko_accumulate(float *acc, const float *raw, size_t npix, uint32_t frame, int camera_change) {
    float weight = (float)frame;
    for (size_t i = 0; i < npix; i++) {
        if (frame <= 1 || !camera_change) {
            for (int k = 0; k < 3; k++) acc[4 * i + k] = (acc[4 * i + k] * weight + raw[4 * i + k]) / (weight + 1.0f);
            acc[4 * i + 3] = 1.0f;
        } else {
            memcpy(acc + 4 * i, raw + 4 * i, 16);
        }
    }
}

/* Accumulation.glsl:16-24 through the colour attachments the application really creates: unsized GL_RGBA
 * (source/Rendering/Opengl/FBT.cpp:6,21), which llvmpipe makes RGBA8. Every frame the raw colour is clamped to [0,1] and
 * rounded to 8 bits when it is written, read back as byte * (1/255), and the running mean is rounded the same way when it is
 * written. Rounding found by matching the reference run bit for bit: rint(x * 255) in fp32 (ties to even); NaN writes 0. */
static float unorm8(float x) {
    x = !(x >= 0.0f) ? 0.0f : (x > 1.0f ? 1.0f : x);
    return rintf(x * 255.0f) * (1.0f / 255.0f);
}
void ko_accumulate_unorm8(float *acc, const float *raw, size_t npix, uint32_t frame, int camera_change) {
    float weight = (float)frame;
    for (size_t i = 0; i < npix; i++) {
        for (int k = 0; k < 3; k++) {
            float r = unorm8(raw[4 * i + k]);
            if (frame <= 1 || !camera_change) acc[4 * i + k] = unorm8((acc[4 * i + k] * weight + r) / (weight + 1.0f));
            else acc[4 * i + k] = r;
        }
        acc[4 * i + 3] = 1.0f;
    }
}

/* PostProcess.glsl:34-39. CLAMP is the value that line 38 hands to an 8-bit
 * UNORM backbuffer; ACES evaluates the overwritten lines 35-37. The backbuffer keeps a value the way the RGBA8
 * attachments do (unorm8 above, the rule pinned by the reference run): clamp, rint(x * 255), ties to even. */
void ko_resolve(const float *acc, uint8_t *rgba8, size_t npix, int tonemap) {
    for (size_t i = 0; i < npix; i++) {
        float c[4] = { acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3] };
        if (tonemap == KO_TONEMAP_ACES) {
            float x = fmaxf(c[0], 0.0f), y = fmaxf(c[1], 0.0f), z = fmaxf(c[2], 0.0f);
            /* column-major mat3 * vec3, PostProcess.glsl:28-32 */
            float r = (3.2404542f * x + -1.5371385f * y) + -0.4985314f * z;
            float g = (-0.9692660f * x + 1.8760108f * y) + 0.0415560f * z;
            float b = (0.0556434f * x + -0.2040259f * y) + 1.0572252f * z;
            float v[3] = { fmaxf(r, 0.0f), fmaxf(g, 0.0f), fmaxf(b, 0.0f) };
            for (int k = 0; k < 3; k++) c[k] = (v[k] * (2.51f * v[k] + 0.03f)) / (v[k] * (2.43f * v[k] + 0.59f) + 0.14f);
            c[3] = 1.0f;
        }
        for (int k = 0; k < 4; k++) rgba8[4 * i + k] = (uint8_t)rintf(unorm8(c[k]) * 255.0f);
    }
}
