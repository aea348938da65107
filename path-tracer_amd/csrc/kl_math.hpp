// fp32 vector helpers and the four transcendental functions used by the render
// kernels. Compiled for host and device with -ffp-contract=off: every operation
// below is one IEEE fp32 operation in the written order, fused only where fmaf
// is spelled out, so results do not depend on the compiler's contraction
// choices.
//
// dot() sums z-first, (a.z*b.z + a.y*b.y) + a.x*b.x, and normalize() multiplies
// by 1/sqrt: these are the evaluation orders under which the reference shader's
// dot()/normalize() calls (resources/Shaders/RawFrame.glsl) reproduce bit for
// bit on Mesa llvmpipe.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define KL_HD __host__ __device__ __forceinline__

namespace kl {

struct v3 { float x, y, z; };

KL_HD v3 V(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
KL_HD v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
KL_HD v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
KL_HD v3 operator*(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
KL_HD v3 operator*(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
KL_HD v3 operator-(v3 a) { return V(-a.x, -a.y, -a.z); }
KL_HD float dot(v3 a, v3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
KL_HD v3 cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
KL_HD float length(v3 a) { return sqrtf(dot(a, a)); }
KL_HD v3 normalize(v3 a) { float il = 1.0f / sqrtf(dot(a, a)); return a * il; }
KL_HD v3 reflect(v3 i, v3 n) { return i - n * (2.0f * dot(n, i)); }        // GLSL reflect
KL_HD v3 mix(v3 x, v3 y, float a) { return x * (1.0f - a) + y * a; }       // GLSL mix
KL_HD float cross2d(float ax, float ay, float bx, float by) { return ax * by - ay * bx; } // RawFrame.glsl:54
KL_HD float comp(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// fused forms for code that has no counterpart in the reference shader
KL_HD float fdot(v3 a, v3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
KL_HD v3 fcross(v3 a, v3 b) {
    return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}

// ---- sin / cos / atan2 / acos: single-precision Cephes algorithms, unfused ----
constexpr float FOPI = 1.27323954473516f;
constexpr float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
constexpr float PI_F = 3.14159265358979323846f, PIO2_F = 1.5707963267948966192f, PIO4_F = 0.7853981633974483096f;

KL_HD float sin_poly(float x, float z) {
    return ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x + x;
}
KL_HD float cos_poly(float z) {
    return ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z - 0.5f * z + 1.0f;
}
KL_HD float sinf_(float x) {
    int sign = 1;
    if (x < 0.0f) { sign = -1; x = -x; }
    if (!(x <= 8192.0f)) return x - x;
    int j = (int)(FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { sign = -sign; j -= 4; }
    x = ((x - y * DP1) - y * DP2) - y * DP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? cos_poly(z) : sin_poly(x, z);
    return sign < 0 ? -r : r;
}
KL_HD float cosf_(float x) {
    int sign = 1;
    if (x < 0.0f) x = -x;
    if (!(x <= 8192.0f)) return x - x;
    int j = (int)(FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { j -= 4; sign = -sign; }
    if (j > 1) sign = -sign;
    x = ((x - y * DP1) - y * DP2) - y * DP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? sin_poly(x, z) : cos_poly(z);
    return sign < 0 ? -r : r;
}
KL_HD float atanf_(float x) {
    int sign = 1;
    if (x < 0.0f) { sign = -1; x = -x; }
    float y;
    if (x > 2.414213562373095f) { y = PIO2_F; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = PIO4_F; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z - 3.33329491539E-1f) * z * x + x;
    return sign < 0 ? -y : y;
}
KL_HD float atan2f_(float y, float x) {
    int code = 0;
    if (x < 0.0f) code = 2;
    if (y < 0.0f) code |= 1;
    if (x == 0.0f) {
        if (code & 1) return -PIO2_F;
        if (y == 0.0f) return 0.0f;
        return PIO2_F;
    }
    if (y == 0.0f) return (code & 2) ? PI_F : 0.0f;
    float w = code == 2 ? PI_F : (code == 3 ? -PI_F : 0.0f);
    return w + atanf_(y / x);
}
KL_HD float asinf_(float x) {
    int sign = 1, flag = 0;
    float a = x, z;
    if (x < 0.0f) { sign = -1; a = -x; }
    if (a < 1.0e-4f) { z = a; }
    else {
        float s;
        if (a > 0.5f) { z = 0.5f * (1.0f - a); s = sqrtf(z); flag = 1; }
        else { s = a; z = s * s; }
        z = ((((4.2163199048E-2f * z + 2.4181311049E-2f) * z + 4.5470025998E-2f) * z + 7.4953002686E-2f) * z + 1.6666752422E-1f) * z * s + s;
        if (flag) { z = z + z; z = PIO2_F - z; }
    }
    return sign < 0 ? -z : z;
}
KL_HD float acosf_(float x) {
    if (x < -1.0f) x = -1.0f;     // out-of-domain inputs (rounding in normalize) are clamped
    if (x > 1.0f) x = 1.0f;
    if (x < -0.5f) return PI_F - 2.0f * asinf_(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asinf_(sqrtf(0.5f * (1.0f - x)));
    return PIO2_F - asinf_(x);
}

// ---- RNG: resources/Shaders/RawFrame.glsl:62-73,82-85 ----
struct u4 { uint32_t x, y, z, w; };
KL_HD u4 hash(u4 s) {
    s.x = s.x * 1664525u + 1013904223u; s.y = s.y * 1664525u + 1013904223u;
    s.z = s.z * 1664525u + 1013904223u; s.w = s.w * 1664525u + 1013904223u;
    s.x += s.y * s.w; s.y += s.z * s.x; s.z += s.x * s.y; s.w += s.y * s.z;
    s.x ^= s.x >> 16; s.y ^= s.y >> 16; s.z ^= s.z >> 16; s.w ^= s.w >> 16;
    s.x += s.y * s.w; s.y += s.z * s.x; s.z += s.x * s.y; s.w += s.y * s.z;
    return s;
}
struct f4 { float x, y, z, w; };
// rand4() of one (pixel, frame). The shader never writes its seed back, so these
// four numbers are what every rand*() call of that invocation returns.
KL_HD f4 rand4(uint32_t px, uint32_t py, uint32_t frame) {
    u4 s; s.x = px; s.y = py; s.z = frame; s.w = px + py;
    u4 h = hash(s);
    f4 r;
    r.x = (float)h.x / 4294967296.0f; r.y = (float)h.y / 4294967296.0f;   // float(0xffffffffu) is 2^32
    r.z = (float)h.z / 4294967296.0f; r.w = (float)h.w / 4294967296.0f;
    return r;
}

}  // namespace kl
