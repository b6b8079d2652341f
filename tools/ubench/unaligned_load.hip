// Does gfx950 under ROCm serve 16-byte / 8-byte global loads whose address is only 2-byte aligned (needed by the bf16
// convolution's tap-shifted operand loads)?  Checks values and times aligned vs misaligned streaming.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 __attribute__((aligned(2))) u32x4_u;
typedef u32x2 __attribute__((aligned(2))) u32x2_u;
__global__ void k16(const unsigned short *in, u32x4 *out, size_t n, int shift) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = *(const u32x4_u *)(in + 8 * i + shift);
}
__global__ void k8(const unsigned short *in, u32x2 *out, size_t n, int shift) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = *(const u32x2_u *)(in + 4 * i + shift);
}
int main() {
    const size_t n = 1 << 24; // 16M vectors of 16 B = 256 MB
    std::vector<unsigned short> h(8 * n + 64);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned short)(i * 2654435761u >> 7);
    unsigned short *d; u32x4 *o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, n * 16);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    std::vector<unsigned short> r(8 * n);
    for (int shift = 0; shift <= 3; shift++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        k16<<<n / 256, 256>>>(d, o, n, shift);
        hipEventRecord(a);
        for (int it = 0; it < 5; it++) k16<<<n / 256, 256>>>(d, o, n, shift);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < 8 * n; i++) bad += r[i] != h[i + shift];
        printf("16-B loads, shift %d elements: %zu wrong, %.1f GB/s (read+write)\n", shift, bad, 5 * 2.0 * n * 16 / ms / 1e6);
    }
    for (int shift = 0; shift <= 3; shift++) {
        k8<<<2 * n / 256, 256>>>(d, (u32x2 *)o, 2 * n, shift);
        hipDeviceSynchronize();
        hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < 8 * n; i++) bad += r[i] != h[i + shift];
        printf("8-B loads, shift %d elements: %zu wrong\n", shift, bad);
    }
    hipError_t e = hipGetLastError();
    printf("last error: %s\n", hipGetErrorString(e));
    return 0;
}
