// Calibration of SQ instruction counters: kernels with a known number of wave-level instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_fma(float *out, float a, float b) {          // 1024 dependent v_fma per wave
    float x = a + threadIdx.x;
    #pragma unroll 64
    for (int i = 0; i < 1024; i++) x = fmaf(x, a, b);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void k_ld4(const float4 *in, float *out) {           // 256 global_load_dwordx4 per wave
    float s = 0; const float4 *p = in + blockIdx.x * blockDim.x + threadIdx.x;
    #pragma unroll 8
    for (int i = 0; i < 256; i++) { float4 v = p[(size_t)i * 65536]; s += v.x + v.w; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float *out; float4 *in;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&in, (size_t)256 * 65536 * 16 + 65536 * 16); (void)hipMemset(in, 0, (size_t)256 * 65536 * 16 + 65536 * 16);
    k_fma<<<256, 256>>>(out, 1.0001f, 0.5f);      // 1024 waves x 1024 fma = 1,048,576 wave-level fma
    k_ld4<<<256, 256>>>(in, out);                 // 1024 waves x 256 loads = 262,144 wave-level loads
    (void)hipDeviceSynchronize();
    printf("expected: k_fma 1048576 wave-level v_fma (+ few); k_ld4 262144 wave-level global_load_dwordx4\n");
    return 0;
}
