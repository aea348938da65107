// Micro-benchmark (second set): what a wave64 VALU instruction costs a SIMD on gfx950, as a function of how many waves share the SIMD.
// Every wave runs an unrolled stream of ONE instruction over eight independent registers (consecutive instructions never
// depend on each other); W waves per SIMD on every SIMD of the chip. Reported per (instruction, W):
//   wall ns per wave-instruction per SIMD, the in-kernel clock (s_memtime ticks per s_memrealtime tick x 100 MHz), and
//   their product: shader clocks the SIMD spends per wave-instruction. 2.0 = a SIMD-32 pipe kept full (32 lanes a clock),
//   4.0 = one wave64 instruction every four clocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define I8(OP, TAIL) OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" \
                     OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n"
#define KERNEL(NAME, OP, TAIL)                                                                          \
    __global__ void NAME(float *out, unsigned long long *clk, int iters) {                              \
        float r[8]; float a = 1.0001f + threadIdx.x * 1e-7f, b = 0.5f, c = 0.25f;                        \
        for (int k = 0; k < 8; k++) r[k] = threadIdx.x + k;                                             \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();    \
        for (int i = 0; i < iters; i++) {                                                               \
            asm volatile(I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) \
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])  \
                         : "v"(a), "v"(b), "v"(c) : "vcc");                                             \
        }                                                                                               \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();    \
        float s = 0; for (int k = 0; k < 8; k++) s += r[k];                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                 \
        if ((threadIdx.x & 63) == 0) { unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); clk[2 * w] = t1 - t0; clk[2 * w + 1] = w1 - w0; } \
    }
// Issue cost of a few more opcodes (six waves per SIMD): candidates for a cheaper byte -> float path in the node test.
KERNEL(k_fma,     "v_fma_f32", ", %8, %9")
KERNEL(k_cvtub1,  "v_cvt_f32_ubyte1", "")
KERNEL(k_fmamix,  "v_fma_mix_f32", ", %8, %9 op_sel_hi:[1,0,0]")
KERNEL(k_fmamixh, "v_fma_mix_f32", ", %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
KERNEL(k_perm,    "v_perm_b32", ", %8, %9")
KERNEL(k_andor,   "v_and_or_b32", ", %8, %9")
KERNEL(k_lshlor,  "v_lshl_or_b32", ", 8, %9")
KERNEL(k_alignbit,"v_alignbit_b32", ", %8, 8")
KERNEL(k_cvtf16,  "v_cvt_f32_f16", "")
KERNEL(k_pkfmaf16,"v_pk_fma_f16", ", %8, %9")
KERNEL(k_mulsdwa, "v_mul_f32_sdwa", ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL(k_lshr,    "v_lshrrev_b32", ", %8")
KERNEL(k_min3,    "v_min3_f32", ", %8, %9")
KERNEL(k_pkmin,   "v_pk_min_f16", ", %8")
static int g_only = 0;
template <typename K> void run(const char *name, K kern, int cus, bool short_list = false) {
    const int iters = 4096;
    for (int w : (short_list ? std::vector<int>{4, 6, 8} : std::vector<int>{1, 2, 3, 4, 6, 8})) {
        if (g_only && w != g_only) continue;
        // W waves per SIMD = 4W waves per CU: one workgroup of 256*W threads (W <= 4), else two of 128*W
        const int per_block = w <= 4 ? 256 * w : 128 * w, blocks = (w <= 4 ? 1 : 2) * cus;
        const int waves = blocks * per_block / 64;
        float *out; unsigned long long *clk;
        (void)hipMalloc(&out, (size_t)blocks * per_block * 4); (void)hipMalloc(&clk, (size_t)waves * 16);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        kern<<<blocks, per_block>>>(out, clk, 64);
        (void)hipEventRecord(e0); kern<<<blocks, per_block>>>(out, clk, iters); (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)waves * 2);
        (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> mhz, ticks;
        for (int i = 0; i < waves; i++) if (h[2 * i + 1]) { mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0); ticks.push_back((double)h[2 * i]); }
        std::sort(mhz.begin(), mhz.end()); std::sort(ticks.begin(), ticks.end());
        const double clock_mhz = mhz.empty() ? 0.0 : mhz[mhz.size() / 2], wave_ticks = ticks.empty() ? 0.0 : ticks[ticks.size() / 2];
        const double instr_per_wave = (double)iters * 64.0;
        const double ns = ms * 1e6 / (w * instr_per_wave);                              // wall per wave-instruction per SIMD
        printf("%-18s waves/SIMD=%d  wall %.3f ms  %.3f ns/wave-instr/SIMD  clock %.0f MHz  -> %.2f clk/wave-instr/SIMD (wall x clock)   "
               "%.2f clk by one wave's own s_memtime / (W x instr)\n",
               name, w, ms, ns, clock_mhz, ns * clock_mhz * 1e-3, wave_ticks / (w * instr_per_wave));
        (void)hipFree(out); (void)hipFree(clk); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    g_only = 6;
    const int cu = p.multiProcessorCount;
    run("v_fma_f32", k_fma, cu); run("v_cvt_f32_ubyte1", k_cvtub1, cu); run("v_fma_mix_f32 lo", k_fmamix, cu); run("v_fma_mix_f32 hi", k_fmamixh, cu);
    run("v_perm_b32", k_perm, cu); run("v_and_or_b32", k_andor, cu); run("v_lshl_or_b32", k_lshlor, cu); run("v_alignbit_b32", k_alignbit, cu);
    run("v_cvt_f32_f16", k_cvtf16, cu); run("v_pk_fma_f16", k_pkfmaf16, cu); run("v_mul_f32_sdwa b1", k_mulsdwa, cu); run("v_lshrrev_b32", k_lshr, cu);
    run("v_min3_f32", k_min3, cu); run("v_pk_min_f16", k_pkmin, cu);
    return 0;
}
