// Micro-benchmark: issue rate of common VALU operations on one SIMD (cycles per wave64 instruction),
// measured with s_memtime around unrolled instruction streams. One workgroup of W waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define N 4096
template <int OP> __global__ void k(float *out, unsigned long long *cyc, float a, float b) {
    float x0 = a + threadIdx.x, x1 = b + threadIdx.x, x2 = a * 3 + threadIdx.x, x3 = b * 5 + threadIdx.x;
    unsigned u0 = threadIdx.x * 77u + 1u, u1 = threadIdx.x * 13u + 5u;
    unsigned long long t0 = __builtin_readcyclecounter();
    #pragma unroll 64
    for (int i = 0; i < N; i++) {
        if (OP == 0) { x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b); }          // 4 independent fma
        if (OP == 1) { x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); }          // 4 dependent fma
        if (OP == 2) { x0 = fmaxf(fmaxf(x0, x1), x2); x1 = fminf(fminf(x1, x2), x3); x2 = fmaxf(fmaxf(x2, x3), x0); x3 = fminf(fminf(x3, x0), x1); }  // max3/min3
        if (OP == 3) { x0 += (float)(u0 & 0xFFu); x1 += (float)((u0 >> 8) & 0xFFu); x2 += (float)((u0 >> 16) & 0xFFu); x3 += (float)(u0 >> 24); u0 += 0x01010101u; } // cvt ubyte + add (9 instr)
        if (OP == 4) { u0 = x0 < x1 ? u0 : u1; u1 = x2 < x3 ? u1 : u0; x0 += 1.0f; x2 += 1.0f; }                       // cmp+cndmask
        if (OP == 5) { x0 = 1.0f / x0 + a; x1 = 1.0f / x1 + a; }                                                      // IEEE division
        if (OP == 6) { x0 = __builtin_amdgcn_rcpf(x0) + a; x1 = __builtin_amdgcn_rcpf(x1) + a; }                      // v_rcp
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + (float)(u0 + u1);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char *name, int waves_per_cu, float instr_per_iter) {
    int blocks = 256; float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 64 * waves_per_cu>>>(out, cyc, 1.0001f, 0.5f);
    hipEventRecord(e0); k<OP><<<blocks, 64 * waves_per_cu>>>(out, cyc, 1.0001f, 0.5f); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // each SIMD holds waves_per_cu/4 waves; wave-instructions per SIMD = that x N x instr_per_iter
    double per_simd = (waves_per_cu / 4.0) * N * instr_per_iter;
    printf("%-28s waves/CU=%2d  %.3f ms  -> %.2f ns per wave-instr per SIMD (x2.4 GHz = %.2f clk)\n", name, waves_per_cu, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {4, 8, 16}) {
        run<0>("4 indep fma", w, 4); run<1>("4 dep fma", w, 4); run<2>("max3/min3 x4", w, 4);
        run<3>("cvt_ubyte+add (8)", w, 9); run<4>("2x(cmp+cndmask)+2 add", w, 6); run<5>("2 IEEE div + 2 add", w, 2); run<6>("2 v_rcp + 2 add", w, 4);
    }
    return 0;
}
