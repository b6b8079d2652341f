// kernels_misc.hip -- the remaining HBM-bound kernels of the training step, NCHW:
// max-pool (resnet.cu:433-494), global average pool (:500-542), ReLU' / add+ReLU (:59-65, :545-564),
// soft-max + cross-entropy derivative (:569-602, stable form of resnet_cudnn.cu:572-583), fused Adam
// (:605-662), layout conversion for NHWC shards/buffers, and the seeded synthetic batch generators.
#include "mi_common.hpp"
#include "mi_device.h"

template <typename T> __device__ __forceinline__ float ldf(const T *p);
template <> __device__ __forceinline__ float ldf<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t *p) { return mi_bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T *p, float v);
template <> __device__ __forceinline__ void stf<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t *p, float v) { *p = mi_f2bf(v); }
template <typename T> __device__ __forceinline__ void stf2(T *p, float a, float b); // two consecutive elements, 2-element aligned
template <> __device__ __forceinline__ void stf2<float>(float *p, float a, float b) { *(float2 *)p = make_float2(a, b); }
template <> __device__ __forceinline__ void stf2<bf16_t>(bf16_t *p, float a, float b) { *(uint32_t *)p = mi_pack_bf2(a, b); }

static int ew_blocks(size_t n, int per = 256) {
    size_t b = (n + per - 1) / per;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- max pool: window centred at stride*o, OOB skipped, strict '>' (first max wins), init -1024 ----
// (element counts < 2^32: 32-bit indices and precomputed fast division; the 64-bit % and / of a size_t index cost more
// than the memory traffic of these kernels)
template <typename T>
__global__ void __launch_bounds__(256)
maxpool_fwd_kernel(const T *__restrict__ x, T *__restrict__ y, int *__restrict__ idx, uint32_t total, int H, int Ho,
                   int k, int stride, FastDiv fdHo) {
    const int half = k / 2;
    for (uint32_t o = blockIdx.x * 256u + threadIdx.x; o < total; o += gridDim.x * 256u) {
        const uint32_t t = fd_div(o, fdHo);
        const int ow = (int)(o - t * Ho);
        const uint32_t nc = fd_div(t, fdHo);
        const int oh = (int)(t - nc * Ho);
        const uint32_t pbase = nc * (uint32_t)(H * H);
        const T *xp = x + pbase;
        float mv = -1024.f;
        int mi = -1024;
        for (int r = -half; r <= half; r++) {
            const int ih = stride * oh + r;
            if (ih < 0 || ih >= H) continue;
            for (int c = -half; c <= half; c++) {
                const int iw = stride * ow + c;
                if (iw < 0 || iw >= H) continue;
                const float v = ldf<T>(xp + ih * H + iw);
                if (v > mv) { mv = v; mi = (int)pbase + ih * H + iw; }
            }
        }
        stf<T>(y + o, mv);
        idx[o] = mi;
    }
}
// The 3x3 / stride-2 forward on even planes whose rows are a multiple of 8 elements (the network's only max-pool): one thread = FOUR
// consecutive outputs of a row, i.e. input columns 2ow-1 .. 2ow+7 of three rows: one 16-/32-byte vector load per row plus the element to
// its left, against 27 scalar strided loads (the generic kernel ran at 1.8 TB/s: bound by its instruction count, not by memory).  Same comparisons in the
// same (r, c) scan order with strict '>', so values AND arg-max indices are those of the generic kernel bit for bit.
template <typename T>
__global__ void __launch_bounds__(256)
maxpool_fwd_3x3s2_kernel(const T *__restrict__ x, T *__restrict__ y, int *__restrict__ idx, uint32_t cells, int H, int Ho, FastDiv fdQ, FastDiv fdHo) {
    const int Q = Ho >> 2;                          // cells per output row
    for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < cells; q += gridDim.x * 256u) {
        const uint32_t t = fd_div(q, fdQ);
        const int ow = (int)(q - t * Q) * 4;
        const uint32_t nc = fd_div(t, fdHo);
        const int oh = (int)(t - nc * Ho);
        const uint32_t pbase = nc * (uint32_t)(H * H);
        float mv[4] = {-1024.f, -1024.f, -1024.f, -1024.f};
        int mi[4] = {-1024, -1024, -1024, -1024};
#pragma unroll
        for (int r = -1; r <= 1; r++) {
            const int ih = 2 * oh + r;
            if (ih < 0) continue;                   // (ih <= H - 1 always: H = 2 Ho)
            const uint32_t rb = pbase + (uint32_t)(ih * H + 2 * ow);
            float v[9];
            VecIO<T, 8>::load(x + rb, v + 1);
            const bool has_left = ow > 0;
            v[0] = has_left ? ldf<T>(x + rb - 1) : 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    if (j == 0 && c == 0 && !has_left) continue;
                    const float e = v[2 * j + c];
                    if (e > mv[j]) { mv[j] = e; mi[j] = (int)rb + 2 * j + c - 1; }
                }
            }
        }
        const uint32_t o = (nc * (uint32_t)Ho + (uint32_t)oh) * (uint32_t)Ho + (uint32_t)ow;
        VecIO<T, 4>::store(y + o, mv);
        *(int4 *)(idx + o) = make_int4(mi[0], mi[1], mi[2], mi[3]);
    }
}
// Backward in gather form: each input element looks at the windows that contain it, in the reference's
// (oh, ow) scan order, and keeps the LAST one whose arg-max it is -- the deterministic execution of the
// reference's racy plain-store scatter (resnet.cu:493; memset 0 at :2186).
template <typename T>
__global__ void __launch_bounds__(256)
maxpool_bwd_kernel(const int *__restrict__ idx, const T *__restrict__ dy, T *__restrict__ dx, uint32_t total, int H,
                   int Ho, int k, int stride, FastDiv fdH) {
    const int half = k / 2;
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
        const uint32_t t = fd_div(e, fdH);
        const int iw = (int)(e - t * H);
        const uint32_t nc = fd_div(t, fdH);
        const int ih = (int)(t - nc * H);
        // outputs whose window covers (ih, iw): stride*oh - half <= ih <= stride*oh + half
        int oh_lo = (ih - half + stride - 1) / stride; if (ih - half < 0) oh_lo = 0;
        int oh_hi = (ih + half) / stride; if (oh_hi > Ho - 1) oh_hi = Ho - 1;
        int ow_lo = (iw - half + stride - 1) / stride; if (iw - half < 0) ow_lo = 0;
        int ow_hi = (iw + half) / stride; if (ow_hi > Ho - 1) ow_hi = Ho - 1;
        float v = 0.f;
        for (int oh = oh_lo; oh <= oh_hi; oh++)
            for (int ow = ow_lo; ow <= ow_hi; ow++) {
                const uint32_t o = (nc * Ho + oh) * Ho + ow;
                if (idx[o] == (int)e) v = ldf<T>(dy + o);
            }
        stf<T>(dx + e, v);
    }
}

// The reference's pooling (3x3 window, stride 2, pad 1, even H): one thread per 2x2 input cell (2a..2a+1, 2b..2b+1).  Only
// the windows (a..a+1, b..b+1) reach the cell -- four (index, dy) pairs serve four outputs (the general kernel above looks up
// 2.25 windows per element) and the result leaves as two 8-byte stores.  Same "last writer in (oh, ow) scan order" rule.
template <typename T>
__global__ void __launch_bounds__(256)
maxpool_bwd_3x3s2_kernel(const int *__restrict__ idx, const T *__restrict__ dy, T *__restrict__ dx, uint32_t cells, int H,
                         int Ho, FastDiv fdHo) {
    for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < cells; q += gridDim.x * 256u) {
        const uint32_t t = fd_div(q, fdHo);
        const int b = (int)(q - t * Ho);
        const uint32_t nc = fd_div(t, fdHo);
        const int a = (int)(t - nc * Ho);
        const uint32_t obase = nc * (uint32_t)(Ho * Ho), ibase = nc * (uint32_t)(H * H);
        const bool a1 = a + 1 < Ho, b1 = b + 1 < Ho;
        const uint32_t o00 = obase + a * Ho + b;
        const int i00 = idx[o00], i01 = b1 ? idx[o00 + 1] : -1, i10 = a1 ? idx[o00 + Ho] : -1, i11 = (a1 && b1) ? idx[o00 + Ho + 1] : -1;
        const float d00 = ldf<T>(dy + o00), d01 = b1 ? ldf<T>(dy + o00 + 1) : 0.f, d10 = a1 ? ldf<T>(dy + o00 + Ho) : 0.f, d11 = (a1 && b1) ? ldf<T>(dy + o00 + Ho + 1) : 0.f;
        const int e00 = (int)ibase + (2 * a) * H + 2 * b, e01 = e00 + 1, e10 = e00 + H, e11 = e10 + 1;
        float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
        // windows in scan order (a,b), (a,b+1), (a+1,b), (a+1,b+1): a later match overwrites an earlier one
        if (i00 == e00) v00 = d00;
        if (i00 == e01) v01 = d00;
        if (i01 == e01) v01 = d01;
        if (i00 == e10) v10 = d00;
        if (i10 == e10) v10 = d10;
        if (i00 == e11) v11 = d00;
        if (i01 == e11) v11 = d01;
        if (i10 == e11) v11 = d10;
        if (i11 == e11) v11 = d11;
        stf2<T>(dx + e00, v00, v01);
        stf2<T>(dx + e10, v10, v11);
    }
}

// ---- global average pool: one wave per (n,c) plane ----
template <typename T>
__global__ void __launch_bounds__(256) avgpool_fwd_kernel(const T *__restrict__ x, float *__restrict__ y, int NC, int P) {
    const int lane = threadIdx.x & 63;
    const int plane = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (plane >= NC) return;
    const T *xp = x + (size_t)plane * P;
    float s = 0.f;
    for (int i = lane; i < P; i += 64) s += ldf<T>(xp + i);
    s = wave_sum(s);
    if (lane == 0) y[plane] = s / (float)P;
}
template <typename T>
__global__ void avgpool_bwd_kernel(const float *__restrict__ dy, T *__restrict__ dx, size_t total, int P) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
        stf<T>(dx + e, dy[e / P] / (float)P);
}

// ---- elementwise ----
__global__ void relu_deriv_kernel(const float *__restrict__ x, const float *__restrict__ up, float *__restrict__ out, size_t n) {
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = ((const float4 *)x)[i], u = ((const float4 *)up)[i];
        ((float4 *)out)[i] = make_float4(a.x > 0.f ? u.x : 0.f, a.y > 0.f ? u.y : 0.f, a.z > 0.f ? u.z : 0.f, a.w > 0.f ? u.w : 0.f);
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = x[i] > 0.f ? up[i] : 0.f;
}
__global__ void add_relu_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ sum_out,
                                float *__restrict__ act_out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float s = a[i] + b[i];
        if (sum_out) sum_out[i] = s;
        act_out[i] = fmaxf(0.f, s);
    }
}

// ---- soft-max (one wave per row) and cross-entropy derivative (no 1/N: resnet.cu:1806-1811) ----
__global__ void __launch_bounds__(64) softmax_kernel(const float *__restrict__ x, float *__restrict__ out, int L) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const float *xr = x + (size_t)row * L;
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) mx = fmaxf(mx, xr[j]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < L; j += 64) s += expf(xr[j] - mx);
    s = wave_sum(s);
    for (int j = lane; j < L; j += 64) out[(size_t)row * L + j] = expf(xr[j] - mx) / s;
}
__global__ void ce_deriv_kernel(const float *__restrict__ pred, const int *__restrict__ labels, float *__restrict__ d, int N, int L) {
    const size_t total = (size_t)N * L;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / L), col = (int)(i % L);
        d[i] = pred[i] - (labels[row] == col ? 1.f : 0.f);
    }
}

// ---- Adam, the three reference kernels fused; same guards (NaN/Inf gradient keeps m,v; NaN/Inf result keeps p) ----
// zero_grads: the gradient is cleared where it was finite (the memset of resnet.cu:2972-2978 folded in); a NaN / Inf
// gradient STAYS in the arena, so that the diagnostic dump of check_errors (resnet.cu:2879-2907) still holds it even
// though the flag is read only at the next host synchronisation point
// nan_flag: 0 = clean, else (highest offending locations[] index + 1) -- the reference's check_errors walks locations[] from
// the last to the first and names the first it finds (resnet.cu:2879-2907, :2952).  loc_off (optional): the n_loc + 1 float
// offsets of the tensors in the arena; `base` = arena offset of p[0] (a bucket launch starts mid-arena).  Without a table the
// flag is 1.  The search runs only for an offending element.
__device__ __forceinline__ int adam_loc_of(const size_t *__restrict__ loc_off, int n_loc, size_t pos) {
    int lo = 0, hi = n_loc - 1;
    while (lo < hi) { // last location whose offset is <= pos
        const int mid = (lo + hi + 1) >> 1;
        if (loc_off[mid] <= pos) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ void adam_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                            size_t n, float lr, float wd, float b1, float b2, float cur_b1, float cur_b2, float eps,
                            int *__restrict__ nan_flag, int zero_grads, const size_t *__restrict__ loc_off, int n_loc, size_t base) {
    int bad = 0; // highest offending location + 1
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i], old = p[i];
        float mi = m[i], vi = v[i];
        bool b = false;
        if (isnan(gi) || isinf(gi)) b = true;
        else {
            const float gd = gi + wd * old;
            mi = b1 * mi + (1.f - b1) * gd;
            vi = b2 * vi + (1.f - b2) * gd * gd;
            m[i] = mi; v[i] = vi;
            if (zero_grads) g[i] = 0.f;
        }
        const float ma = mi / (1.f - cur_b1), va = vi / (1.f - cur_b2);
        float np = old - (lr * (ma / (sqrtf(va) + eps)) + wd * old);
        if (isnan(np) || isinf(np)) { np = old; b = true; }
        if (isnan(mi) || isinf(mi) || isnan(vi) || isinf(vi)) b = true;
        p[i] = np;
        if (b) {
            const int l = loc_off ? adam_loc_of(loc_off, n_loc, base + i) + 1 : 1;
            bad = l > bad ? l : bad;
        }
    }
    if (bad && nan_flag) atomicMax(nan_flag, bad);
}

// ---- layout ----
__global__ void nhwc_to_nchw_kernel(const float *__restrict__ in, float *__restrict__ out, int N, int HW, int C) {
    const size_t total = (size_t)N * HW * C;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(o % HW);
        const size_t t = o / HW;
        const int c = (int)(t % C);
        const size_t n = t / C;
        out[o] = in[(n * HW + p) * C + c];
    }
}
__global__ void nchw_to_nhwc_kernel(const float *__restrict__ in, float *__restrict__ out, int N, int C, int HW) {
    const size_t total = (size_t)N * HW * C;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(o % C);
        const size_t t = o / C;
        const int p = (int)(t % HW);
        const size_t n = t / HW;
        out[o] = in[(n * C + c) * HW + p];
    }
}

// ---- splitmix64 counter streams (same definition as tests/synth.py and csrc/synth.c) ----
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void fill_uniform_kernel(float *__restrict__ out, size_t n, uint64_t seed, uint64_t offset, float lo, float hi) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double u = (double)(splitmix64_at(seed, offset + i) >> 11) * (1.0 / 9007199254740992.0);
        out[i] = (float)((double)lo + ((double)hi - (double)lo) * u);
    }
}
__global__ void fill_labels_kernel(int *__restrict__ out, size_t n, uint64_t seed, uint64_t offset, int n_classes) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (int)(splitmix64_at(seed, offset + i) % (uint64_t)n_classes);
}

// ---- test aid: leave NaNs in every CU's LDS so kernels that read LDS they never wrote are caught ----
__global__ void __launch_bounds__(1024) lds_poison_kernel(float *sink) {
    extern __shared__ float p[];
    for (int i = threadIdx.x; i < 16000; i += blockDim.x) p[i] = __int_as_float(0x7fc00000);
    __syncthreads();
    if (sink && threadIdx.x == 0 && p[blockIdx.x & 1023] == 0.f) sink[0] = 1.f;
}

extern "C" {
int mid_lds_poison(mid_stream s) {
    hipLaunchKernelGGL(lds_poison_kernel, dim3(2048), dim3(1024), 64000, (hipStream_t)s, (float *)nullptr);
    MI_LAUNCH_CHECK("lds_poison_kernel");
    return 0;
}
int mid_maxpool_fwd_t(mid_stream s, const void *x, void *y, int dt, int *max_inds, int N, int C, int H, int k, int stride) {
    const int Ho = H / stride;
    const size_t total = (size_t)N * C * Ho * Ho;
    if ((double)N * C * H * H >= 2147483648.0) { mi_record_error("mid_maxpool_fwd", "tensor too large for 32-bit indices"); return -2; }
    if (k == 3 && stride == 2 && H % 8 == 0 && Ho % 4 == 0) { /* rows of 8-element vectors (16-byte aligned in both storage types), 4 outputs per thread */
        const size_t cells = total / 4;
        if (dt == MID_BF16)
            hipLaunchKernelGGL(maxpool_fwd_3x3s2_kernel<bf16_t>, dim3(ew_blocks(cells)), dim3(256), 0, (hipStream_t)s, (const bf16_t *)x, (bf16_t *)y, max_inds,
                               (uint32_t)cells, H, Ho, make_fastdiv(Ho / 4), make_fastdiv(Ho));
        else
            hipLaunchKernelGGL(maxpool_fwd_3x3s2_kernel<float>, dim3(ew_blocks(cells)), dim3(256), 0, (hipStream_t)s, (const float *)x, (float *)y, max_inds,
                               (uint32_t)cells, H, Ho, make_fastdiv(Ho / 4), make_fastdiv(Ho));
        MI_LAUNCH_CHECK("maxpool_fwd_3x3s2_kernel");
        return 0;
    }
    if (dt == MID_BF16)
        hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)s, (const bf16_t *)x, (bf16_t *)y, max_inds,
                           (uint32_t)total, H, Ho, k, stride, make_fastdiv(Ho));
    else
        hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)s, (const float *)x, (float *)y, max_inds,
                           (uint32_t)total, H, Ho, k, stride, make_fastdiv(Ho));
    MI_LAUNCH_CHECK("maxpool_fwd_kernel");
    return 0;
}
int mid_maxpool_fwd(mid_stream s, const float *x, float *y, int *max_inds, int N, int C, int H, int k, int stride) {
    return mid_maxpool_fwd_t(s, x, y, MID_F32, max_inds, N, C, H, k, stride);
}
} // extern "C"
template <typename T>
static int maxpool_bwd_launch(hipStream_t st, const int *max_inds, const T *dy, T *dx, int N, int C, int H, int k, int stride) {
    const int Ho = H / stride;
    const size_t total = (size_t)N * C * H * H;
    if ((double)total >= 2147483648.0) { mi_record_error("mid_maxpool_bwd", "tensor too large for 32-bit indices"); return -2; }
    if (k == 3 && stride == 2 && (H & 1) == 0) {
        const size_t cells = total / 4;
        hipLaunchKernelGGL(maxpool_bwd_3x3s2_kernel<T>, dim3(ew_blocks(cells)), dim3(256), 0, st, max_inds, dy, dx, (uint32_t)cells, H, Ho,
                           make_fastdiv(Ho));
        MI_LAUNCH_CHECK("maxpool_bwd_3x3s2_kernel");
        return 0;
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(ew_blocks(total)), dim3(256), 0, st, max_inds, dy, dx, (uint32_t)total, H, Ho, k,
                       stride, make_fastdiv(H));
    MI_LAUNCH_CHECK("maxpool_bwd_kernel");
    return 0;
}
extern "C" {
int mid_maxpool_bwd_t(mid_stream s, const int *max_inds, const void *dy, void *dx, int dt, int N, int C, int H, int k, int stride) {
    if (dt == MID_BF16) return maxpool_bwd_launch<bf16_t>((hipStream_t)s, max_inds, (const bf16_t *)dy, (bf16_t *)dx, N, C, H, k, stride);
    return maxpool_bwd_launch<float>((hipStream_t)s, max_inds, (const float *)dy, (float *)dx, N, C, H, k, stride);
}
int mid_maxpool_bwd(mid_stream s, const int *max_inds, const float *dy, float *dx, int N, int C, int H, int k, int stride) {
    return mid_maxpool_bwd_t(s, max_inds, dy, dx, MID_F32, N, C, H, k, stride);
}
int mid_avgpool_fwd_t(mid_stream s, const void *x, int dt, float *y, int N, int C, int P) {
    if (dt == MID_BF16) hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, dim3(mi_cdiv((long)N * C, 4)), dim3(256), 0, (hipStream_t)s, (const bf16_t *)x, y, N * C, P);
    else hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(mi_cdiv((long)N * C, 4)), dim3(256), 0, (hipStream_t)s, (const float *)x, y, N * C, P);
    MI_LAUNCH_CHECK("avgpool_fwd_kernel");
    return 0;
}
int mid_avgpool_fwd(mid_stream s, const float *x, float *y, int N, int C, int P) { return mid_avgpool_fwd_t(s, x, MID_F32, y, N, C, P); }
int mid_avgpool_bwd_t(mid_stream s, const float *dy, void *dx, int dt, int N, int C, int P) {
    const size_t total = (size_t)N * C * P;
    if (dt == MID_BF16) hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)s, dy, (bf16_t *)dx, total, P);
    else hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)s, dy, (float *)dx, total, P);
    MI_LAUNCH_CHECK("avgpool_bwd_kernel");
    return 0;
}
int mid_avgpool_bwd(mid_stream s, const float *dy, float *dx, int N, int C, int P) { return mid_avgpool_bwd_t(s, dy, dx, MID_F32, N, C, P); }
int mid_relu_deriv(mid_stream s, const float *x, const float *up, float *out, size_t n) {
    hipLaunchKernelGGL(relu_deriv_kernel, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)s, x, up, out, n);
    MI_LAUNCH_CHECK("relu_deriv_kernel");
    return 0;
}
int mid_add_relu(mid_stream s, const float *a, const float *b, float *sum_out, float *act_out, size_t n) {
    hipLaunchKernelGGL(add_relu_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)s, a, b, sum_out, act_out, n);
    MI_LAUNCH_CHECK("add_relu_kernel");
    return 0;
}
int mid_softmax(mid_stream s, const float *x, float *out, int N, int L) {
    hipLaunchKernelGGL(softmax_kernel, dim3(N), dim3(64), 0, (hipStream_t)s, x, out, L);
    MI_LAUNCH_CHECK("softmax_kernel");
    return 0;
}
int mid_ce_deriv(mid_stream s, const float *pred, const int *labels, float *d, int N, int L) {
    hipLaunchKernelGGL(ce_deriv_kernel, dim3(ew_blocks((size_t)N * L)), dim3(256), 0, (hipStream_t)s, pred, labels, d, N, L);
    MI_LAUNCH_CHECK("ce_deriv_kernel");
    return 0;
}
int mid_adam(mid_stream s, float *p, float *g, float *m, float *v, size_t n, float lr, float wd, float b1, float b2,
             float cur_b1, float cur_b2, float eps, int *nan_flag, int zero_grads, const size_t *loc_off_dev, int n_loc, size_t base) {
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)s, p, g, m, v, n, lr, wd, b1, b2, cur_b1, cur_b2, eps, nan_flag, zero_grads,
                       loc_off_dev, n_loc, base);
    MI_LAUNCH_CHECK("adam_kernel");
    return 0;
}
int mid_nhwc_to_nchw(mid_stream s, const float *in, float *out, int N, int H, int W, int C) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_blocks((size_t)N * H * W * C)), dim3(256), 0, (hipStream_t)s, in, out, N, H * W, C);
    MI_LAUNCH_CHECK("nhwc_to_nchw_kernel");
    return 0;
}
int mid_nchw_to_nhwc(mid_stream s, const float *in, float *out, int N, int C, int H, int W) {
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_blocks((size_t)N * H * W * C)), dim3(256), 0, (hipStream_t)s, in, out, N, C, H * W);
    MI_LAUNCH_CHECK("nchw_to_nhwc_kernel");
    return 0;
}
int mid_fill_uniform(mid_stream s, float *out, size_t n, uint64_t seed, uint64_t offset, float lo, float hi) {
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)s, out, n, seed, offset, lo, hi);
    MI_LAUNCH_CHECK("fill_uniform_kernel");
    return 0;
}
int mid_fill_labels(mid_stream s, int *out, size_t n, uint64_t seed, uint64_t offset, int n_classes) {
    hipLaunchKernelGGL(fill_labels_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)s, out, n, seed, offset, n_classes);
    MI_LAUNCH_CHECK("fill_labels_kernel");
    return 0;
}
}
