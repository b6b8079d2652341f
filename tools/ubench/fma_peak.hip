// Micro-benchmark: fp32 VALU FMA issue rate on gfx950 for the instruction forms the conv kernels use.
//   hipcc --offload-arch=gfx950 -O3 -o fma_peak fma_peak.hip && ./fma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, const float *w, float s0, float s1) {
    f2 acc[16];
    for (int i = 0; i < 16; i++) acc[i] = (f2){(float)threadIdx.x, 1.f};
    f2 x = {out[threadIdx.x & 7], out[(threadIdx.x + 1) & 7]};
    float xs = x[0];
    f2 sw = {s0, s1};
    for (int it = 0; it < ITERS; it++) {
        if (MODE == 0) { // v_pk_fma_f32, all VGPR
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = __builtin_elementwise_fma(acc[i], x, acc[i]);
        } else if (MODE == 1) { // v_pk_fma_f32, SGPR pair x broadcast VGPR (the dconv form)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = __builtin_elementwise_fma(sw, (f2){xs, xs}, acc[i]);
        } else { // v_fma_f32 scalar form, 32 independent chains
#pragma unroll
            for (int i = 0; i < 16; i++) { acc[i][0] = fmaf(acc[i][0], xs, acc[i][0]); acc[i][1] = fmaf(acc[i][1], xs, acc[i][1]); }
        }
        asm volatile("" : "+v"(xs));
    }
    float r = 0;
    for (int i = 0; i < 16; i++) r += acc[i][0] + acc[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    float *out, *w;
    hipMalloc(&out, 4 * 256 * 8192); hipMalloc(&w, 4096);
    hipMemset(out, 0, 4 * 256 * 8192);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char *names[3] = {"v_pk_fma_f32 VGPR operands", "v_pk_fma_f32 SGPR-pair x VGPR-broadcast", "v_fma_f32"};
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2)
        for (int m = 0; m < 3; m++) {
            const int blocks = 256 * waves_per_simd; // 256-thread blocks: 1 wave per SIMD per block
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(a);
                if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, w, 1.0f, 0.5f);
                if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, w, 1.0f, 0.5f);
                if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, w, 1.0f, 0.5f);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            const double flops = 2.0 * 32 * ITERS * 256.0 * blocks;
            printf("%-44s waves/SIMD %d : %7.1f TFLOP/s\n", names[m], waves_per_simd, flops / ms / 1e9);
        }
    return 0;
}
