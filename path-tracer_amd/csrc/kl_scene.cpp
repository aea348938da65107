// Scene sources: the reference's built-in scene as upload records, and the
// seed-deterministic synthetic benchmark scene.
#include "kl_scene.hpp"

#include <cmath>
#include <cstring>

#include "kl_math.hpp"

namespace kl {

// The constant arrays of resources/Shaders/RawFrame.glsl:171-192 as data:
// six spheres, seven quads (floor, walls, textured back wall, ceiling, two lights).
void builtin_scene(std::vector<kl_sphere> &sph, std::vector<kl_quad> &quads, std::vector<kl_material> &mats) {
    static const float S[6][11] = {
        // centre, radius | type, colour, emissive, roughness, ior
        {-1, 0.3f, -1, 0.2f, KL_DIFFUSE, 1, 1, 1, 0, 1, 1.2f},
        {-1, 0.9f, -1, 0.2f, KL_DIFFUSE, 1, 1, 1, 0, 0, 1.2f},
        {-1, 1.5f, -1, 0.2f, KL_GLASS, 1, 1, 1, 0, 0, 1.2f},
        { 1, 0.3f, -1, 0.2f, KL_DIFFUSE, 1, 0, 0, 0, 0, 1.2f},
        { 1, 0.9f, -1, 0.2f, KL_DIFFUSE, 0, 1, 0, 0, 0.1f, 1.2f},
        { 1, 1.5f, -1, 0.2f, KL_DIFFUSE, 0, 0, 1, 0, 0, 1.2f},
    };
    static const float Q[7][19] = {
        {-2.66f, 0, -15,  2.66f, 0, -15,  2.66f, 0, 5,  -2.66f, 0, 5,   KL_DIFFUSE, 1, 1, 1, 0, 1, 1},
        { 2.66f, 0, -15,  2.66f, 0, 5,    2.66f, 3, 5,   2.66f, 3, -15, KL_DIFFUSE, 1, 1, 1, 0, 0.01f, 1},
        {-2.66f, 0, -15, -2.66f, 0, 5,   -2.66f, 3, 5,  -2.66f, 3, -15, KL_DIFFUSE, 1, 1, 1, 0, 0.01f, 1},
        {-2.66f, 0, -15,  2.66f, 0, -15,  2.66f, 3, -15, -2.66f, 3, -15, KL_TEXTURED, 1, 1, 1, 0, 1, 1},
        {-2.66f, 3, -15,  2.66f, 3, -15,  2.66f, 3, 5,  -2.66f, 3, 5,   KL_DIFFUSE, 1, 1, 1, 0, 1, 1},
        {-1.8f, 2.9f, -10, -1, 2.9f, -10, -1, 2.9f, 3,  -1.8f, 2.9f, 3, KL_EMISSIVE, 1, 1, 1, 2.5f, 1, 1},
        { 1.8f, 2.9f, -10,  1, 2.9f, -10,  1, 2.9f, 3,   1.8f, 2.9f, 3, KL_EMISSIVE, 1, 1, 1, 2.5f, 1, 1},
    };
    sph.resize(6); quads.resize(7); mats.resize(13);
    for (int i = 0; i < 6; i++) {
        std::memcpy(sph[i].center, S[i], 12); sph[i].radius = S[i][3]; sph[i].material = (uint32_t)i;
        kl_material &m = mats[i];
        m.type = (int32_t)S[i][4]; std::memcpy(m.color, &S[i][5], 12);
        m.emissive = S[i][8]; m.roughness = S[i][9]; m.ior = S[i][10]; m.pad = 0;
    }
    for (int i = 0; i < 7; i++) {
        std::memcpy(quads[i].v0, &Q[i][0], 12); std::memcpy(quads[i].v1, &Q[i][3], 12);
        std::memcpy(quads[i].v2, &Q[i][6], 12); std::memcpy(quads[i].v3, &Q[i][9], 12);
        quads[i].material = (uint32_t)(6 + i);
        kl_material &m = mats[6 + i];
        m.type = (int32_t)Q[i][12]; std::memcpy(m.color, &Q[i][13], 12);
        m.emissive = Q[i][16]; m.roughness = Q[i][17]; m.ior = Q[i][18]; m.pad = 0;
    }
}

// ---- synthetic scene --------------------------------------------------------------
// Room and lights = the seven quads above. Inside x in [-2.4,2.4], y in [0.1,2.8], z in [-14,4]:
// `blobs` lat-long tessellated spheres with per-vertex radial jitter, then `analytic_spheres`
// analytic spheres. Every random number is the reference hash (RawFrame.glsl:62-68) of a
// counter, so the scene depends on nothing but kl_synth_params.
namespace {

f4 rnd(uint32_t a, uint32_t b, uint32_t c) { return rand4(a, b, c); }

void tessellation(uint32_t tris_per_blob, uint32_t &stacks, uint32_t &slices) {
    // largest 2*slices*(stacks-1) <= tris_per_blob with slices/stacks close to 1.2
    uint32_t best = 0; float best_aspect = 1e9f; stacks = 3; slices = 3;
    for (uint32_t st = 3; st <= 256; st++) {
        uint32_t sl = tris_per_blob / (2u * (st - 1u));
        if (sl < 3) break;
        uint32_t n = 2u * sl * (st - 1u);
        float aspect = std::fabs((float)sl / (float)st - 1.2f);
        if (n > best || (n == best && aspect < best_aspect)) { best = n; best_aspect = aspect; stacks = st; slices = sl; }
    }
}

kl_material blob_material(uint32_t i, uint32_t seed) {
    kl_material m; std::memset(&m, 0, sizeof m);
    f4 c = rnd(i, seed ^ 0xC0105u, 2);
    m.color[0] = 0.2f + 0.8f * c.x; m.color[1] = 0.2f + 0.8f * c.y; m.color[2] = 0.2f + 0.8f * c.z;
    m.ior = 1.2f; m.roughness = 1.0f; m.emissive = 0.0f; m.type = KL_DIFFUSE;
    switch (i % 8u) {
        case 3: m.roughness = 0.1f; break;
        case 4: m.roughness = 0.0f; break;
        case 5: m.type = KL_GLASS; m.roughness = 0.0f; break;
        case 6: m.type = KL_TEXTURED; break;
        case 7: if (i % 64u == 7u) { m.type = KL_EMISSIVE; m.emissive = 2.5f; } break;
        default: break;
    }
    return m;
}

void placement(uint32_t i, uint32_t seed, float c[3], float &radius) {
    f4 r = rnd(i, seed ^ 0x5EEDu, 1234);
    c[0] = -2.4f + 4.8f * r.x; c[1] = 0.1f + 2.7f * r.y; c[2] = -14.0f + 18.0f * r.z;
    radius = 0.02f * std::pow(6.0f, r.w);            // log-uniform in [0.02, 0.12]
}

}  // namespace

void synth_counts(const kl_synth_params &sp, uint32_t &ns, uint32_t &nq, uint32_t &nt, uint32_t &nm) {
    uint32_t st, sl;
    tessellation(sp.tris_per_blob, st, sl);
    ns = sp.analytic_spheres; nq = 7;
    nt = sp.blobs * 2u * sl * (st - 1u);
    nm = 7u + sp.blobs + sp.analytic_spheres;
}

void synth_fill(const kl_synth_params &sp, kl_sphere *sph, kl_quad *quads, kl_tri *tris, kl_material *mats) {
    std::vector<kl_sphere> bs; std::vector<kl_quad> bq; std::vector<kl_material> bm;
    builtin_scene(bs, bq, bm);
    for (uint32_t i = 0; i < 7; i++) { quads[i] = bq[i]; quads[i].material = i; mats[i] = bm[6 + i]; }
    uint32_t st, sl;
    tessellation(sp.tris_per_blob, st, sl);
    const uint32_t tpb = 2u * sl * (st - 1u);
    const double PI = 3.14159265358979323846;
    std::vector<float> vx((size_t)(st + 1) * sl * 3);
    for (uint32_t i = 0; i < sp.blobs; i++) {
        float c[3], radius;
        placement(i, sp.seed, c, radius);
        mats[7 + i] = blob_material(i, sp.seed);
        // vertices: ring a = 0..st (poles included as degenerate rings), b = 0..sl-1
        for (uint32_t a = 0; a <= st; a++) {
            for (uint32_t b = 0; b < sl; b++) {
                uint32_t vid = (a == 0 || a == st) ? a * sl : a * sl + b;      // one jitter per pole
                f4 j = rnd(i, sp.seed ^ 0x7177u, 100000u + vid);
                double rr = (double)radius * (1.0 + 0.05 * (2.0 * (double)j.x - 1.0));
                double th = PI * (double)a / (double)st, ph = 2.0 * PI * (double)b / (double)sl;
                float *v = &vx[((size_t)a * sl + b) * 3];
                v[0] = c[0] + (float)(rr * std::sin(th) * std::cos(ph));
                v[1] = c[1] + (float)(rr * std::cos(th));
                v[2] = c[2] + (float)(rr * std::sin(th) * std::sin(ph));
            }
        }
        kl_tri *out = tris + (size_t)i * tpb;
        uint32_t k = 0;
        auto emit = [&](uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2) {
            kl_tri &t = out[k++];
            const uint32_t aa[3] = {a0, a1, a2}, bb[3] = {b0, b1, b2};
            float *vv[3] = {t.v0, t.v1, t.v2}; float *uu[3] = {t.uv0, t.uv1, t.uv2};
            for (int q = 0; q < 3; q++) {
                std::memcpy(vv[q], &vx[((size_t)aa[q] * sl + (bb[q] % sl)) * 3], 12);
                uu[q][0] = (float)bb[q] / (float)sl; uu[q][1] = (float)aa[q] / (float)st;
            }
            t.material = 7u + i;
        };
        for (uint32_t a = 0; a < st; a++) {
            for (uint32_t b = 0; b < sl; b++) {
                if (a != 0) emit(a, b, a + 1, b, a, b + 1);            // upper triangle of the band cell
                if (a != st - 1) emit(a, b + 1, a + 1, b, a + 1, b + 1);
            }
        }
    }
    for (uint32_t i = 0; i < sp.analytic_spheres; i++) {
        uint32_t id = sp.blobs + i;
        placement(id, sp.seed, sph[i].center, sph[i].radius);
        sph[i].material = 7u + id;
        mats[7 + id] = blob_material(id, sp.seed);
    }
}

}  // namespace kl
