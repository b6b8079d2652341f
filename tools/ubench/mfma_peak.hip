// Micro-benchmark: sustained fp32 MFMA rate on gfx950 (v_mfma_f32_32x32x2_f32 / 16x16x4), register operands only,
// for runs of ~1 ms and ~5 ms (clock/power behaviour of a long MFMA-bound kernel such as pconv_mfma_kernel).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, const float *rnd) {
    // rnd == nullptr: near-constant operands (the chip holds ~2.4 GHz); else random operands (DVFS lowers the clock)
    float a = rnd ? rnd[threadIdx.x + 256 * (blockIdx.x & 7)] : (float)threadIdx.x * 1e-3f;
    float b = rnd ? rnd[4096 + threadIdx.x + 256 * (blockIdx.x & 7)] : 1.0f + (float)blockIdx.x * 1e-6f;
    float r = 0;
    if (MODE == 0) {
        f16v acc[4];
        for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            asm volatile("" : "+v"(a));
        }
        for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) r += acc[i][e];
    } else {
        f4v acc[8];
        for (int i = 0; i < 8; i++) for (int e = 0; e < 4; e++) acc[i][e] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            asm volatile("" : "+v"(a));
        }
        for (int i = 0; i < 8; i++) for (int e = 0; e < 4; e++) r += acc[i][e];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    float *out, *rnd;
    hipMalloc(&out, 4 * 256 * 4096);
    hipMalloc(&rnd, 4 * 8192);
    { float h[8192]; unsigned x = 12345u; for (int i = 0; i < 8192; i++) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) * (1.0f / 16777216.0f) - 0.5f) * 1e-2f; } hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice); }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char *names[2] = {"v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x4_f32"};
    for (int data = 0; data < 2; data++)
    for (int iters = 2048; iters <= 16384; iters *= 8)
        for (int wps = 1; wps <= 2; wps++)
            for (int m = 0; m < 2; m++) {
                const int blocks = 256 * wps;
                float ms = 0;
                for (int rep = 0; rep < 3; rep++) {
                    hipEventRecord(a);
                    if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, data ? rnd : nullptr);
                    else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, data ? rnd : nullptr);
                    hipEventRecord(b); hipEventSynchronize(b);
                    hipEventElapsedTime(&ms, a, b);
                }
                // per wave per iteration: MODE 0: 16 MFMA x 4096 flop; MODE 1: 32 MFMA x 2048 flop
                const double flops = 65536.0 * iters * 4.0 * blocks;
                printf("%s %-24s waves/SIMD %d iters %5d : %7.3f ms %7.1f TFLOP/s\n", data ? "random  " : "constant", names[m], wps, iters, ms, flops / ms / 1e9);
            }
    return 0;
}
