// Micro-benchmark: does a VALU instruction cost its SIMD less when part of the wave is masked off (EXEC)? Streams of one
// instruction as in valu_issue.hip, six waves per SIMD, with EXEC = all 64 lanes, the low 32, the low 16, one lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define I8(OP, TAIL) OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" \
                     OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n"
#define KERNEL(NAME, OP, TAIL)                                                                          \
    __global__ void NAME(float *out, unsigned long long *clk, int iters, unsigned long long mask) {      \
        float r[8]; float a = 1.0001f + threadIdx.x * 1e-7f, b = 0.5f, c = 0.25f;                        \
        for (int k = 0; k < 8; k++) r[k] = threadIdx.x + k;                                             \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();    \
        if ((mask >> (threadIdx.x & 63)) & 1ull)                                                        \
        for (int i = 0; i < iters; i++) {                                                               \
            asm volatile(I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) \
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])  \
                         : "v"(a), "v"(b), "v"(c));                                                     \
        }                                                                                               \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();    \
        float s = 0; for (int k = 0; k < 8; k++) s += r[k];                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                 \
        if ((threadIdx.x & 63) == 0) { unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); clk[2 * w] = t1 - t0; clk[2 * w + 1] = w1 - w0; } \
    }
KERNEL(k_fma,   "v_fma_f32", ", %8, %9")
KERNEL(k_max3,  "v_max3_f32", ", %8, %9")
KERNEL(k_cvtub, "v_cvt_f32_ubyte1", "")
KERNEL(k_rcp,   "v_rcp_f32", "")
template <typename K> void run(const char *name, K kern, int cus) {
    const int iters = 4096, w = 6;
    for (unsigned long long mask : {~0ull, 0xFFFFFFFFull, 0xFFFFull, 0x1ull, 0x0000FFFF0000FFFFull, 0x5555555555555555ull}) {
        const int per_block = 128 * w, blocks = 2 * cus, waves = blocks * per_block / 64;
        float *out; unsigned long long *clk;
        (void)hipMalloc(&out, (size_t)blocks * per_block * 4); (void)hipMalloc(&clk, (size_t)waves * 16);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        kern<<<blocks, per_block>>>(out, clk, 64, mask);
        (void)hipEventRecord(e0); kern<<<blocks, per_block>>>(out, clk, iters, mask); (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)waves * 2);
        (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> mhz;
        for (int i = 0; i < waves; i++) if (h[2 * i + 1]) mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
        std::sort(mhz.begin(), mhz.end());
        const double clock_mhz = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
        const double ns = ms * 1e6 / (w * (double)iters * 64.0);
        printf("%-18s EXEC=%016llx  %.3f ns/wave-instr/SIMD  clock %.0f MHz -> %.2f clk/wave-instr/SIMD\n", name, mask, ns, clock_mhz, ns * clock_mhz * 1e-3);
        (void)hipFree(out); (void)hipFree(clk);
    }
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    run("v_fma_f32", k_fma, p.multiProcessorCount); run("v_max3_f32", k_max3, p.multiProcessorCount);
    run("v_cvt_f32_ubyte1", k_cvtub, p.multiProcessorCount); run("v_rcp_f32", k_rcp, p.multiProcessorCount);
    return 0;
}
