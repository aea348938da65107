// Micro-benchmark: what a wave64 VALU instruction costs a SIMD on gfx950, as a function of how many waves share the SIMD.
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
// two-register packed forms need register pairs: written out separately
__global__ void k_pkfma(float *out, unsigned long long *clk, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 r[8]; f2 a = {1.0001f + threadIdx.x * 1e-7f, 1.0002f}, b = {0.5f, 0.25f};
    for (int k = 0; k < 8; k++) r[k] = f2{(float)threadIdx.x + k, (float)k};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        asm volatile(I8("v_pk_fma_f32", ", %8, %9") I8("v_pk_fma_f32", ", %8, %9") I8("v_pk_fma_f32", ", %8, %9") I8("v_pk_fma_f32", ", %8, %9")
                     I8("v_pk_fma_f32", ", %8, %9") I8("v_pk_fma_f32", ", %8, %9") I8("v_pk_fma_f32", ", %8, %9") I8("v_pk_fma_f32", ", %8, %9")
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    float s = 0; for (int k = 0; k < 8; k++) s += r[k].x + r[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); clk[2 * w] = t1 - t0; clk[2 * w + 1] = w1 - w0; }
}
KERNEL(k_mov,   "v_mov_b32", "")
KERNEL(k_fma,   "v_fma_f32", ", %8, %9")
KERNEL(k_add,   "v_add_f32", ", %8")
KERNEL(k_max3,  "v_max3_f32", ", %8, %9")
KERNEL(k_cnd,   "v_cndmask_b32", ", %8, vcc")
KERNEL(k_cvtub, "v_cvt_f32_ubyte1", "")
KERNEL(k_bfe,   "v_bfe_u32", ", 8, 8")
KERNEL(k_mulhi, "v_mul_hi_u32", ", %8")
KERNEL(k_rcp,   "v_rcp_f32", "")
KERNEL(k_sqrt,  "v_sqrt_f32", "")

// v_cndmask_b32 in its forms: lane mask in vcc or in an SGPR pair, mask all-zero / mixed, result fed back or not
#define KERNELM(NAME, BODY8, MASKINIT)                                                                  \
    __global__ void NAME(float *out, unsigned long long *clk, int iters) {                              \
        float r[8]; float a = 1.0001f + threadIdx.x * 1e-7f, b = 0.5f, c = 0.25f;                        \
        for (int k = 0; k < 8; k++) r[k] = threadIdx.x + k;                                             \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();    \
        for (int i = 0; i < iters; i++) {                                                               \
            asm volatile(MASKINIT "\n" BODY8 BODY8 BODY8 BODY8 BODY8 BODY8 BODY8 BODY8                  \
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])  \
                         : "v"(a), "v"(b), "v"(c) : "vcc", "s20", "s21");                               \
        }                                                                                               \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();    \
        float s = 0; for (int k = 0; k < 8; k++) s += r[k];                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                 \
        if ((threadIdx.x & 63) == 0) { unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); clk[2 * w] = t1 - t0; clk[2 * w + 1] = w1 - w0; } \
    }
KERNELM(k_cnd_vcc0,  I8("v_cndmask_b32", ", %8, vcc"), "s_mov_b64 vcc, 0")
KERNELM(k_cnd_vccm,  I8("v_cndmask_b32", ", %8, vcc"), "s_mov_b64 vcc, 0x0f0f0f0f")
KERNELM(k_cnd_sgpr,  I8("v_cndmask_b32_e64", ", %8, s[20:21]"), "s_mov_b64 s[20:21], 0x0f0f0f0f")
KERNELM(k_cnd_indep, "v_cndmask_b32 %0, %8, %9, vcc\nv_cndmask_b32 %1, %8, %9, vcc\nv_cndmask_b32 %2, %8, %9, vcc\nv_cndmask_b32 %3, %8, %9, vcc\n"
                     "v_cndmask_b32 %4, %8, %9, vcc\nv_cndmask_b32 %5, %8, %9, vcc\nv_cndmask_b32 %6, %8, %9, vcc\nv_cndmask_b32 %7, %8, %9, vcc\n", "s_mov_b64 vcc, 0x0f0f0f0f")
KERNELM(k_cmp_vcc,   "v_cmp_lt_f32 vcc, %0, %8\nv_cmp_lt_f32 vcc, %1, %8\nv_cmp_lt_f32 vcc, %2, %8\nv_cmp_lt_f32 vcc, %3, %8\n"
                     "v_cmp_lt_f32 vcc, %4, %8\nv_cmp_lt_f32 vcc, %5, %8\nv_cmp_lt_f32 vcc, %6, %8\nv_cmp_lt_f32 vcc, %7, %8\n", "")
KERNELM(k_cmp_sgpr,  "v_cmp_lt_f32_e64 s[20:21], %0, %8\nv_cmp_lt_f32_e64 s[20:21], %1, %8\nv_cmp_lt_f32_e64 s[20:21], %2, %8\nv_cmp_lt_f32_e64 s[20:21], %3, %8\n"
                     "v_cmp_lt_f32_e64 s[20:21], %4, %8\nv_cmp_lt_f32_e64 s[20:21], %5, %8\nv_cmp_lt_f32_e64 s[20:21], %6, %8\nv_cmp_lt_f32_e64 s[20:21], %7, %8\n", "")
KERNELM(k_cmp_cnd,   "v_cmp_lt_f32 vcc, %0, %8\nv_cndmask_b32 %0, %0, %9, vcc\nv_cmp_lt_f32 vcc, %1, %8\nv_cndmask_b32 %1, %1, %9, vcc\n"
                     "v_cmp_lt_f32 vcc, %2, %8\nv_cndmask_b32 %2, %2, %9, vcc\nv_cmp_lt_f32 vcc, %3, %8\nv_cndmask_b32 %3, %3, %9, vcc\n", "")
// how a v_cndmask_b32 (VOP2, lane mask implicitly in vcc) depends on where vcc came from
KERNELM(k_cnd_1w7r,  "v_cmp_lt_f32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %9, vcc\nv_cndmask_b32 %2, %2, %9, vcc\nv_cndmask_b32 %3, %3, %9, vcc\n"
                     "v_cndmask_b32 %4, %4, %9, vcc\nv_cndmask_b32 %5, %5, %9, vcc\nv_cndmask_b32 %6, %6, %9, vcc\nv_cndmask_b32 %7, %7, %9, vcc\n", "")
KERNELM(k_cnd_late,  "v_cmp_lt_f32 vcc, %0, %8\nv_add_f32 %1, %1, %8\nv_add_f32 %2, %2, %8\nv_add_f32 %3, %3, %8\n"
                     "v_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\nv_add_f32 %6, %6, %8\nv_cndmask_b32 %7, %7, %9, vcc\n", "")
KERNELM(k_cnd_e64vcc, I8("v_cndmask_b32_e64", ", %8, vcc"), "s_mov_b64 vcc, 0x0f0f0f0f")
KERNELM(k_cnd_salu,  "s_and_b64 vcc, exec, s[20:21]\nv_cndmask_b32 %0, %0, %9, vcc\ns_and_b64 vcc, exec, s[20:21]\nv_cndmask_b32 %1, %1, %9, vcc\n"
                     "s_and_b64 vcc, exec, s[20:21]\nv_cndmask_b32 %2, %2, %9, vcc\ns_and_b64 vcc, exec, s[20:21]\nv_cndmask_b32 %3, %3, %9, vcc\n"
                     "s_and_b64 vcc, exec, s[20:21]\nv_cndmask_b32 %4, %4, %9, vcc\ns_and_b64 vcc, exec, s[20:21]\nv_cndmask_b32 %5, %5, %9, vcc\n"
                     "s_and_b64 vcc, exec, s[20:21]\nv_cndmask_b32 %6, %6, %9, vcc\ns_and_b64 vcc, exec, s[20:21]\nv_cndmask_b32 %7, %7, %9, vcc\n", "s_mov_b64 s[20:21], 0x0f0f0f0f")
KERNEL(k_min,    "v_min_f32", ", %8")
KERNEL(k_mul,    "v_mul_f32", ", %8")
KERNEL(k_fmac,   "v_fmac_f32", ", %8")
KERNEL(k_min3,   "v_min3_f32", ", %8, %9")
KERNEL(k_med3,   "v_med3_f32", ", %8, %9")
KERNEL(k_and,    "v_and_b32", ", %8")
KERNEL(k_or,     "v_or_b32", ", %8")
KERNEL(k_lshl,   "v_lshlrev_b32", ", %8")
KERNEL(k_lshladd,"v_lshl_add_u32", ", 2, %8")
KERNEL(k_add3,   "v_add3_u32", ", %8, %9")
KERNEL(k_addu,   "v_add_u32", ", %8")
KERNEL(k_bfi,    "v_bfi_b32", ", %8, %9")
KERNEL(k_perm,   "v_perm_b32", ", %8, %9")
KERNEL(k_cvtub0, "v_cvt_f32_ubyte0", "")
KERNEL(k_cvtu2f, "v_cvt_f32_u32", "")
KERNEL(k_cvtf2u, "v_cvt_u32_f32", "")
KERNEL(k_mad24,  "v_mad_u32_u24", ", %8, %9")
KERNEL(k_mullo,  "v_mul_lo_u32", ", %8")
KERNEL(k_rsq,    "v_rsq_f32", "")
KERNEL(k_floor,  "v_floor_f32", "")
KERNEL(k_dpp,    "v_mov_b32_dpp", " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_divfix, "v_div_fixup_f32", ", %8, %9")
KERNEL(k_sdwa,   "v_cvt_f32_ubyte0_sdwa", " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1")
KERNEL(k_subf,   "v_sub_f32", ", %8")
KERNEL(k_xor,    "v_xor_b32", ", %8")
KERNEL(k_subu,   "v_sub_u32", ", %8")
KERNEL(k_maxf,   "v_max_f32", ", %8")
KERNEL(k_mulneg, "v_mul_f32_e64", ", -%8")
KERNEL(k_divfmas,"v_div_fmas_f32", ", %8, %9")
KERNEL(k_mbcnt,  "v_mbcnt_lo_u32_b32", ", %8")
KERNEL(k_lshr,   "v_lshrrev_b32", ", %8")
// 64-bit and packed forms work on register pairs
#define KERNEL2(NAME, OP, TAIL)                                                                         \
    __global__ void NAME(float *out, unsigned long long *clk, int iters) {                              \
        typedef float f2 __attribute__((ext_vector_type(2)));                                           \
        f2 r[8]; f2 a = {1.0001f + threadIdx.x * 1e-7f, 1.0002f}, b = {0.5f, 0.25f};                     \
        for (int k = 0; k < 8; k++) r[k] = f2{(float)threadIdx.x + k, (float)k};                        \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();    \
        for (int i = 0; i < iters; i++) {                                                               \
            asm volatile(I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) \
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b) : "vcc"); \
        }                                                                                               \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();    \
        float s = 0; for (int k = 0; k < 8; k++) s += r[k].x + r[k].y;                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                 \
        if ((threadIdx.x & 63) == 0) { unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); clk[2 * w] = t1 - t0; clk[2 * w + 1] = w1 - w0; } \
    }
KERNEL2(k_pkmul,  "v_pk_mul_f32", ", %8")
KERNEL2(k_pkadd,  "v_pk_add_f32", ", %8")
KERNEL2(k_mov64,  "v_mov_b64", "")
KERNEL2(k_lshladd64, "v_lshl_add_u64", ", 2, %8")
KERNEL2(k_pkmov,  "v_pk_mov_b32", ", %8")

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
int main(int argc, char **argv) {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    if (argc > 1) {     // calibration of the SQ counters under rocprofv3 --pmc: one instruction of each issue class, six waves per SIMD
        g_only = 6;
        run("v_add_f32", k_add, p.multiProcessorCount); run("v_max3_f32", k_max3, p.multiProcessorCount);
        run("v_cvt_f32_ubyte1", k_cvtub, p.multiProcessorCount); run("v_rcp_f32", k_rcp, p.multiProcessorCount);
        if (argv[1][0] == 'c' && argv[1][1] == '2') {      // which SQ_INSTS_VALU_* class counts which instruction
            const int cu = p.multiProcessorCount;
            run("v_mov_b32", k_mov, cu); run("v_fma_f32", k_fma, cu); run("v_mul_f32", k_mul, cu); run("v_fmac_f32", k_fmac, cu); run("v_min_f32", k_min, cu);
            run("v_min3_f32", k_min3, cu); run("v_and_b32", k_and, cu); run("v_add_u32", k_addu, cu); run("v_lshlrev_b32", k_lshl, cu);
            run("cndmask e64 sgpr", k_cnd_sgpr, cu); run("v_cmp_lt_f32 sgpr", k_cmp_sgpr, cu); run("v_bfe_u32", k_bfe, cu); run("v_mov_b32_dpp", k_dpp, cu);
            run("v_div_fixup_f32", k_divfix, cu); run("v_floor_f32", k_floor, cu); run("v_mul_hi_u32", k_mulhi, cu); run("v_cvt_f32_u32", k_cvtu2f, cu);
        }
        return 0;
    }
    printf("device %s, %d CUs, clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run("v_mov_b32", k_mov, p.multiProcessorCount);
    run("v_fma_f32", k_fma, p.multiProcessorCount);
    run("v_add_f32", k_add, p.multiProcessorCount);
    run("v_max3_f32", k_max3, p.multiProcessorCount);
    run("v_cndmask_b32", k_cnd, p.multiProcessorCount);
    run("v_cvt_f32_ubyte1", k_cvtub, p.multiProcessorCount);
    run("v_bfe_u32", k_bfe, p.multiProcessorCount);
    run("v_mul_hi_u32", k_mulhi, p.multiProcessorCount);
    run("v_pk_fma_f32", k_pkfma, p.multiProcessorCount);
    run("v_rcp_f32", k_rcp, p.multiProcessorCount);
    run("v_sqrt_f32", k_sqrt, p.multiProcessorCount);
    const int cu = p.multiProcessorCount;
    run("cndmask vcc=0", k_cnd_vcc0, cu, true); run("cndmask vcc mixed", k_cnd_vccm, cu, true); run("cndmask e64 sgpr", k_cnd_sgpr, cu, true);
    run("cndmask no feedback", k_cnd_indep, cu, true); run("v_cmp_lt_f32 vcc", k_cmp_vcc, cu, true); run("v_cmp_lt_f32 sgpr", k_cmp_sgpr, cu, true);
    run("cmp+cndmask pairs", k_cmp_cnd, cu, true);
    run("1 cmp + 7 cndmask vcc", k_cnd_1w7r, cu, true); run("cmp, 6 add, cndmask", k_cnd_late, cu, true); run("cndmask e64 mask=vcc", k_cnd_e64vcc, cu, true);
    run("v_min_f32", k_min, cu, true); run("v_mul_f32", k_mul, cu, true); run("v_fmac_f32", k_fmac, cu, true); run("v_min3_f32", k_min3, cu, true);
    run("v_med3_f32", k_med3, cu, true); run("v_and_b32", k_and, cu, true); run("v_or_b32", k_or, cu, true); run("v_lshlrev_b32", k_lshl, cu, true);
    run("v_lshl_add_u32", k_lshladd, cu, true); run("v_add3_u32", k_add3, cu, true); run("v_add_u32", k_addu, cu, true); run("v_bfi_b32", k_bfi, cu, true);
    run("v_perm_b32", k_perm, cu, true); run("v_cvt_f32_ubyte0", k_cvtub0, cu, true); run("v_cvt_f32_u32", k_cvtu2f, cu, true); run("v_cvt_u32_f32", k_cvtf2u, cu, true);
    run("v_mad_u32_u24", k_mad24, cu, true); run("v_mul_lo_u32", k_mullo, cu, true); run("v_rsq_f32", k_rsq, cu, true); run("v_floor_f32", k_floor, cu, true);
    run("v_sub_f32", k_subf, cu, true); run("v_xor_b32", k_xor, cu, true); run("v_sub_u32", k_subu, cu, true); run("v_max_f32", k_maxf, cu, true);
    run("v_mbcnt_lo_u32_b32", k_mbcnt, cu, true); run("v_lshrrev_b32", k_lshr, cu, true);
    run("v_pk_mul_f32", k_pkmul, cu, true); run("v_pk_add_f32", k_pkadd, cu, true); run("v_mov_b64", k_mov64, cu, true);
    run("v_mov_b32_dpp", k_dpp, cu, true); run("v_div_fixup_f32", k_divfix, cu, true); run("v_cvt_f32_ubyte sdwa", k_sdwa, cu, true);
    return 0;
}
