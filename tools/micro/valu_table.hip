// Micro-benchmark: cost of individual VALU instructions on gfx950 (wave64), in clocks per instruction per SIMD.
// Each kernel runs an unrolled stream of ONE instruction on eight independent registers (no dependency stalls),
// 4 waves per SIMD, long enough that launch overhead vanishes. Clocks are s_memtime ticks of one wave divided by
// the instructions that wave issued, times the waves sharing the SIMD... reported instead as wall ns per
// instruction per SIMD, which needs no clock assumption, next to v_mov_b32 as the unit.
#include <hip/hip_runtime.h>
#include <cstdio>
// eight different registers inside one statement: consecutive instructions never depend on each other
#define I8(OP, TAIL) OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" \
                     OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n"
#define KERNEL(NAME, OP, TAIL)                                                                          \
    __global__ void NAME(float *out, int iters) {                                                       \
        float r[8]; float a = 1.0001f + threadIdx.x * 1e-7f, b = 0.5f, c = 0.25f;                        \
        for (int k = 0; k < 8; k++) r[k] = threadIdx.x + k;                                             \
        for (int i = 0; i < iters; i++) {                                                               \
            asm volatile(I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL) \
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])  \
                         : "v"(a), "v"(b), "v"(c) : "vcc");                                             \
        }                                                                                               \
        float s = 0; for (int k = 0; k < 8; k++) s += r[k];                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                 \
    }
KERNEL(k_mov,     "v_mov_b32", "")
KERNEL(k_add,     "v_add_f32", ", %8")
KERNEL(k_mul,     "v_mul_f32", ", %8")
KERNEL(k_max,     "v_max_f32", ", %8")
KERNEL(k_fmac,    "v_fmac_f32", ", %8")
KERNEL(k_fma,     "v_fma_f32", ", %8, %9")
KERNEL(k_max3,    "v_max3_f32", ", %8, %9")
KERNEL(k_min3,    "v_min3_f32", ", %8, %9")
KERNEL(k_cnd32,   "v_cndmask_b32", ", %8, vcc")
KERNEL(k_cvtub,   "v_cvt_f32_ubyte1", "")
KERNEL(k_and,     "v_and_b32", ", %8")
KERNEL(k_lshl,    "v_lshlrev_b32", ", %8")
KERNEL(k_addu,    "v_add_u32", ", %8")
KERNEL(k_mulhi,   "v_mul_hi_u32", ", %8")
KERNEL(k_rcp,     "v_rcp_f32", "")
KERNEL(k_sqrt,    "v_sqrt_f32", "")
KERNEL(k_bfe,     "v_bfe_u32", ", 8, 8")
KERNEL(k_cvtf2u,  "v_cvt_u32_f32", "")
KERNEL(k_sub,     "v_sub_f32", ", %8")
KERNEL(k_min,     "v_min_f32", ", %8")
KERNEL(k_xor,     "v_xor_b32", ", %8")
KERNEL(k_lshladd, "v_lshl_add_u32", ", 2, %8")
KERNEL(k_perm,    "v_perm_b32", ", %8, %9")
KERNEL(k_cvtu2f,  "v_cvt_f32_u32", "")
KERNEL(k_med3,    "v_med3_f32", ", %8, %9")
KERNEL(k_fmamix,  "v_fma_mix_f32", ", %8, %9")
template <typename K> void run(const char *name, K kern, double unit_ns, double *store) {
    int blocks = 256, iters = 2048, waves_per_simd = 4; float *out;
    (void)hipMalloc(&out, blocks * 1024 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<blocks, 256 * waves_per_simd>>>(out, 8);
    (void)hipEventRecord(e0); kern<<<blocks, 256 * waves_per_simd>>>(out, iters); (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double per = ms * 1e6 / ((double)waves_per_simd * iters * 64);   // 8 x 8 instructions per loop turn
    if (store) *store = per;
    printf("%-22s %.3f ns per instruction per SIMD   (%.2f x v_mov_b32)\n", name, per, unit_ns > 0 ? per / unit_ns : 1.0);
    (void)hipFree(out);
}
int main() {
    double mov = 0; run("v_mov_b32", k_mov, 0, &mov);
#define R(N, K) run(N, K, mov, nullptr);
    R("v_add_f32", k_add) R("v_sub_f32", k_sub) R("v_mul_f32", k_mul) R("v_fmac_f32", k_fmac) R("v_fma_f32", k_fma) R("v_fma_mix_f32", k_fmamix)
    R("v_max_f32", k_max) R("v_min_f32", k_min) R("v_max3_f32", k_max3) R("v_min3_f32", k_min3) R("v_med3_f32", k_med3)
    R("v_cndmask_b32 (vcc)", k_cnd32) R("v_cvt_f32_ubyte1", k_cvtub) R("v_cvt_f32_u32", k_cvtu2f) R("v_cvt_u32_f32", k_cvtf2u)
    R("v_and_b32", k_and) R("v_xor_b32", k_xor) R("v_lshlrev_b32", k_lshl) R("v_lshl_add_u32", k_lshladd) R("v_bfe_u32", k_bfe) R("v_perm_b32", k_perm)
    R("v_add_u32", k_addu) R("v_mul_hi_u32", k_mulhi) R("v_rcp_f32", k_rcp) R("v_sqrt_f32", k_sqrt)
    return 0;
}
