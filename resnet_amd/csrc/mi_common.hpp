// mi_common.hpp -- helpers shared by the kernels_*.hip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

void mi_record_error(const char *what, const char *detail);

#define MI_LAUNCH_CHECK(name)                                              \
    do {                                                                   \
        hipError_t e_ = hipGetLastError();                                 \
        if (e_ != hipSuccess) { mi_record_error(name, hipGetErrorString(e_)); return -1; } \
    } while (0)

// per-kernel-family timing with HIP events on the launch stream (bench.py roofline): begin/end bracket ONE launch
enum { MI_FAM_DCONV = 0, MI_FAM_WGRAD = 1, MI_FAM_GEMM = 2, MI_FAM_BN = 3, MI_FAM_OTHER = 4, MI_FAM_PCONV = 5, MI_FAM_COUNT = 6 };
void mi_prof_begin(hipStream_t st, int fam, double flops, double bytes);
void mi_prof_end(hipStream_t st);

static inline int mi_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// wave64 sum by DPP-free shuffles (6 steps); result valid in every lane
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact unsigned division by a runtime divisor through a precomputed 32-bit magic (n < 2^31, d < 2^31)
struct FastDiv {
    uint32_t d, magic, shift;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d ? d : 1;
    if (f.d == 1) { f.magic = 0; f.shift = 0; return f; }
    uint32_t l = 0;
    while ((1ull << l) < f.d) l++;
    f.shift = l - 1;
    f.magic = (uint32_t)(((1ull << 32) * ((1ull << l) - f.d)) / f.d + 1);
    return f;
}
// branch-free form for divisors known to be >= 2 (inside software-pipelined loops, where a branch splits the schedule)
__host__ __device__ __forceinline__ uint32_t fd_div_ge2(uint32_t n, const FastDiv f) {
    uint32_t t = (uint32_t)(((uint64_t)n * f.magic) >> 32);
    return (t + ((n - t) >> 1)) >> f.shift;
}
__host__ __device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv f) {
    if (f.d == 1) return n;
    uint32_t t = (uint32_t)(((uint64_t)n * f.magic) >> 32);
    return (t + ((n - t) >> 1)) >> f.shift;
}
