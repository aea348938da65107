// Micro-benchmark: VALU issue rate per SIMD (wave64) on the device, long enough that launch overhead vanishes.
// Every wave runs `iters` x 64 dependent v_fma_f32; W waves per SIMD share the pipe. Reports ns per
// wave-instruction per SIMD from HIP events, and cycles per instruction from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND> __global__ void k(float *out, unsigned long long *cyc, float a, float b, int iters) {
    float x0 = a + threadIdx.x, x1 = b + threadIdx.x;
    unsigned u0 = threadIdx.x * 77u + 1u;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        #pragma unroll
        for (int j = 0; j < 64; j++) {
            if (KIND == 0) x0 = fmaf(x0, a, b);                                   // dependent fma
            if (KIND == 1) { x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); }          // two chains (may pack)
            if (KIND == 2) x0 = fmaxf(x0 * a, b);                                 // mul + max
            if (KIND == 3) { u0 = u0 * 3u + 1u; }                                 // integer mad
            if (KIND == 4) { x0 += (float)(u0 & 0xFFu); u0 = (u0 >> 1) | (u0 << 31); }   // cvt_ubyte + add + alignbit
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + (float)u0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void run(const char *name, int waves_per_simd, double instr_per_j) {
    int blocks = 256, iters = 4096; float *out; unsigned long long *cyc, hc = 0;
    (void)hipMalloc(&out, blocks * 2048 * 4); (void)hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<KIND><<<blocks, 256 * waves_per_simd>>>(out, cyc, 1.0001f, 0.5f, 16);
    (void)hipEventRecord(e0); k<KIND><<<blocks, 256 * waves_per_simd>>>(out, cyc, 1.0001f, 0.5f, iters); (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    double per_simd = (double)waves_per_simd * iters * 64 * instr_per_j;
    printf("%-22s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD  (memtime: %.2f ticks per instr of one wave)\n", name, waves_per_simd, ms,
           ms * 1e6 / per_simd, (double)hc / (iters * 64.0 * instr_per_j));
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("dep fma", w, 1); run<1>("2 chains fma", w, 2); run<2>("mul+max", w, 2); run<3>("int mad", w, 1); run<4>("cvt_ubyte+add+alignbit", w, 3);
    }
    return 0;
}
