// mi_common.hpp -- helpers shared by the kernels_*.hip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

void mi_record_error(const char *what, const char *detail);

// diagnostic (RESNET_MI_TRACE=1): the names of the last launches, printed when the process is aborted (a GPU memory fault reaches
// the host as SIGABRT from the runtime's event thread, with nothing that names the kernel)
void mi_trace_launch(const char *name);
#define MI_LAUNCH_CHECK(name)                                              \
    do {                                                                   \
        mi_trace_launch(name);                                             \
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

// ---- storage types of activation tensors: fp32, or bf16 (configs[4]: bf16 activations, fp32 arithmetic) ----
typedef unsigned short bf16_t;
__device__ __forceinline__ float mi_bf2f(bf16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint32_t mi_pack_bf2(float a, float b) { // RNE, NaN stays NaN: v_cvt_pk_bf16_f32
    typedef __bf16 bf2_ __attribute__((ext_vector_type(2)));
    typedef float f2_ __attribute__((ext_vector_type(2)));
    f2_ v = {a, b};
    bf2_ r = __builtin_convertvector(v, bf2_);
    return *(uint32_t *)&r;
}
__device__ __forceinline__ bf16_t mi_f2bf(float a) { return (bf16_t)(mi_pack_bf2(a, 0.f) & 0xffffu); }

// V consecutive elements <-> float registers; V = 1, 4 or 8 (the pointer is V-element aligned)
template <typename T, int V> struct VecIO;
template <int V> struct VecIO<float, V> {
    static __device__ __forceinline__ void load(const float *p, float *o) {
        if (V == 1) o[0] = p[0];
        else {
#pragma unroll
            for (int q = 0; q < V / 4; q++) { const float4 t = *(const float4 *)(p + 4 * q); o[4 * q] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w; }
        }
    }
    static __device__ __forceinline__ void store(float *p, const float *o) {
        if (V == 1) p[0] = o[0];
        else {
#pragma unroll
            for (int q = 0; q < V / 4; q++) *(float4 *)(p + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        }
    }
};
template <int V> struct VecIO<bf16_t, V> {
    static __device__ __forceinline__ void load(const bf16_t *p, float *o) {
        if (V == 1) o[0] = mi_bf2f(p[0]);
        else if (V == 4) {
            const uint2 t = *(const uint2 *)p;
            o[0] = __uint_as_float(t.x << 16); o[1] = __uint_as_float(t.x & 0xffff0000u);
            o[2] = __uint_as_float(t.y << 16); o[3] = __uint_as_float(t.y & 0xffff0000u);
        } else {
            const uint4 t = *(const uint4 *)p;
            o[0] = __uint_as_float(t.x << 16); o[1] = __uint_as_float(t.x & 0xffff0000u);
            o[2] = __uint_as_float(t.y << 16); o[3] = __uint_as_float(t.y & 0xffff0000u);
            o[4] = __uint_as_float(t.z << 16); o[5] = __uint_as_float(t.z & 0xffff0000u);
            o[6] = __uint_as_float(t.w << 16); o[7] = __uint_as_float(t.w & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(bf16_t *p, const float *o) {
        if (V == 1) p[0] = mi_f2bf(o[0]);
        else if (V == 4) *(uint2 *)p = make_uint2(mi_pack_bf2(o[0], o[1]), mi_pack_bf2(o[2], o[3]));
        else *(uint4 *)p = make_uint4(mi_pack_bf2(o[0], o[1]), mi_pack_bf2(o[2], o[3]), mi_pack_bf2(o[4], o[5]), mi_pack_bf2(o[6], o[7]));
    }
};
