// kernels_bn.hip -- training-mode batch norm, forward and backward, NCHW (channel planes contiguous).
// Replaces doBatchNormAndActivate / activationAndBatchNormDeriv (resnet.cu:289-426), which run one
// thread per channel over N*H*W elements three times each.
//
// forward : (1) bn_stats   one HBM read of x: per-thread two-step (sum -> centred squares) on register
//               batches, merged with Chan's parallel formula through wave shuffles, LDS and a per-channel
//               partial table -- the result is the two-pass mean / biased variance of the reference,
//               without its sequential-fp32 drift;
//           (2) bn_apply   second read of x, one write: x_hat = (x-mean)/sqrt(var+eps), y = fma(gamma,
//               x_hat, beta), optional ReLU, optional fused residual add + ReLU (addVec + doActivation,
//               resnet.cu:1717-1723).  Algorithmic traffic 3 passes (2 reads + 1 write), +1 read if fused add.
// backward: (1) bn_bwd_reduce  reads x, dy (+ mask source): s1 = sum g, s2 = sum g*x_hat (g = dy gated by
//               ReLU');  (2) bn_bwd_apply  reads x, dy again, writes dx = gamma/sd * (g - s1/M - x_hat*s2/M),
//               which is the reference's textbook dVar/dMean form (resnet.cu:394-422) with the exact-zero
//               term sum(x-mean) dropped.  5 passes.
#include "mi_common.hpp"
#include "mi_device.h"

#define BN_SPLIT_MAX 64

struct Wel { float n, mean, m2; };
__device__ __forceinline__ Wel wel_merge(Wel a, Wel b) {
    Wel r;
    r.n = a.n + b.n;
    if (r.n == 0.f) { r.mean = 0.f; r.m2 = 0.f; return r; }
    const float d = b.mean - a.mean, f = b.n / r.n;
    r.mean = a.mean + d * f;
    r.m2 = a.m2 + b.m2 + d * d * a.n * f;
    return r;
}
__device__ __forceinline__ Wel wel_wave(Wel w) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Wel b;
        b.n = __shfl_xor(w.n, o, 64); b.mean = __shfl_xor(w.mean, o, 64); b.m2 = __shfl_xor(w.m2, o, 64);
        w = wel_merge(w, b);
    }
    return w;
}
__device__ __forceinline__ void wel_add4(Wel &w, float a, float b, float c, float d) {
    const float bm = ((a + b) + (c + d)) * 0.25f;
    const float da = a - bm, db = b - bm, dc = c - bm, dd = d - bm;
    Wel t; t.n = 4.f; t.mean = bm; t.m2 = (da * da + db * db) + (dc * dc + dd * dd);
    w = wel_merge(w, t);
}
__device__ __forceinline__ void wel_add1(Wel &w, float a) {
    Wel t; t.n = 1.f; t.mean = a; t.m2 = 0.f;
    w = wel_merge(w, t);
}

// Merge a batch of NB values (already summed: bs = sum, and centred squares bm2 about its own mean bmean) into w.
__device__ __forceinline__ void wel_add_batch(Wel &w, float nb, float bmean, float bm2) {
    Wel t; t.n = nb; t.mean = bmean; t.m2 = bm2;
    w = wel_merge(w, t);
}

// grid (C, nsplit): block handles channel c, images n = split, split+nsplit, ...  The (image, position) pairs of the
// block are ONE flattened index space walked by all 256 threads with four 16-byte loads in flight per thread: a 14x14 or
// 7x7 plane (49 float4 / 49 floats) no longer leaves 80% of the lanes idle, and a thread merges once per 16 values.
template <typename TX, int V>
__global__ void __launch_bounds__(256)
bn_stats_kernel(const TX *__restrict__ x, float *__restrict__ partial, int N, int C, int P, FastDiv fdPV) {
    const int c = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
    const int cnt = (N - split + nsplit - 1) / nsplit;            // images of this block
    const uint32_t PV = (uint32_t)P / V;                          // units per plane
    const uint32_t total = (uint32_t)cnt * PV;
    const size_t img_stride = (size_t)nsplit * C * P;
    const TX *base = x + ((size_t)split * C + c) * P;
    Wel w = {0.f, 0.f, 0.f};
    auto addr = [&](uint32_t idx) -> const TX * {
        const uint32_t j = fd_div(idx, fdPV), i = idx - j * PV;
        return base + (size_t)j * img_stride + (size_t)i * V;
    };
    uint32_t idx = threadIdx.x;
    for (; idx + 3 * 256 < total; idx += 4 * 256) {
        float v[4][V];
#pragma unroll
        for (int u = 0; u < 4; u++) VecIO<TX, V>::load(addr(idx + u * 256), v[u]);
        if (V == 1) {
            const float bm = ((v[0][0] + v[1][0]) + (v[2][0] + v[3][0])) * 0.25f;
            const float a = v[0][0] - bm, b = v[1][0] - bm, cc = v[2][0] - bm, d = v[3][0] - bm;
            wel_add_batch(w, 4.f, bm, (a * a + b * b) + (cc * cc + d * d));
        } else {
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int q = 0; q < V; q += 4) s += (v[u][q] + v[u][q + 1]) + (v[u][q + 2] + v[u][q + 3]);
            const float bm = s * (1.0f / (4.0f * V));
            float m2 = 0.f;
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int q = 0; q < V; q += 4) {
                    const float a = v[u][q] - bm, b = v[u][q + 1] - bm, cc = v[u][q + 2] - bm, d = v[u][q + 3] - bm;
                    m2 += (a * a + b * b) + (cc * cc + d * d);
                }
            wel_add_batch(w, 4.f * V, bm, m2);
        }
    }
    for (; idx < total; idx += 256) {
        float v[V];
        VecIO<TX, V>::load(addr(idx), v);
        if (V == 1) wel_add1(w, v[0]);
        else {
#pragma unroll
            for (int q = 0; q < V; q += 4) wel_add4(w, v[q], v[q + 1], v[q + 2], v[q + 3]);
        }
    }
    w = wel_wave(w);
    __shared__ Wel sh[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sh[wv] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        Wel r = sh[0];
        for (int i = 1; i < (int)(blockDim.x >> 6); i++) r = wel_merge(r, sh[i]);
        float *o = partial + ((size_t)c * BN_SPLIT_MAX + split) * 3;
        o[0] = r.n; o[1] = r.mean; o[2] = r.m2;
    }
}

// Statistics partials left by a convolution (kernels_igemm.hip: three planes [np][C] of count, mean, M2; channel
// contiguous) -> the (channel, split) partial table bn_finalize_kernel reads.  grid (ceil(C / 64), G): a workgroup
// merges one chunk of the np partials for 64 channels, four partials in flight per channel, in a fixed order.
__global__ void __launch_bounds__(256)
bn_parts_merge_kernel(const float *__restrict__ parts, int np, int C, float *__restrict__ partial) {
    const int cx = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int chunk = (np + gridDim.y - 1) / gridDim.y;
    const int p0 = blockIdx.y * chunk, p1 = min(np, p0 + chunk);
    const size_t plane = (size_t)np * C;
    Wel w = {0.f, 0.f, 0.f};
    if (c < C) {
        int p = p0 + sg;
        for (; p + 7 * 4 < p1; p += 8 * 4) { // eight partials in flight per thread (a load per merge was latency-bound)
            Wel b[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const size_t o = (size_t)(p + 4 * u) * C + c;
                b[u].n = parts[o]; b[u].mean = parts[plane + o]; b[u].m2 = parts[2 * plane + o];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) w = wel_merge(w, b[u]);
        }
        for (; p < p1; p += 4) {
            const size_t o = (size_t)p * C + c;
            Wel b = {parts[o], parts[plane + o], parts[2 * plane + o]};
            w = wel_merge(w, b);
        }
    }
    __shared__ Wel sh[4][64];
    sh[sg][cx] = w;
    __syncthreads();
    if (sg == 0 && c < C) {
        Wel r = sh[0][cx];
        for (int i = 1; i < 4; i++) r = wel_merge(r, sh[i][cx]);
        float *o = partial + ((size_t)c * BN_SPLIT_MAX + blockIdx.y) * 3;
        o[0] = r.n; o[1] = r.mean; o[2] = r.m2;
    }
}

// per-channel finalize: one wave per channel, lane i holds split partial i (nsplit <= 64), butterfly Chan merge (a fixed
// tree: deterministic); writes mean / biased var.  (One thread per channel merging 64 partials in sequence took 10 us per
// launch on the critical path of every BN.)
__global__ void __launch_bounds__(256)
bn_finalize_kernel(const float *__restrict__ partial, int nsplit, int C, float *__restrict__ means, float *__restrict__ vars) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    const float *p = partial + (size_t)c * BN_SPLIT_MAX * 3;
    Wel w = {0.f, 0.f, 0.f};
    if (lane < nsplit) { w.n = p[lane * 3]; w.mean = p[lane * 3 + 1]; w.m2 = p[lane * 3 + 2]; }
    w = wel_wave(w);
    if (lane == 0) { means[c] = w.mean; vars[c] = w.m2 / w.n; }
}

// x_hat and y exactly as the reference writes them (resnet.cu:329-331); shared by forward and the backward
// mask recomputation so the ReLU gate is bit-identical in both
__device__ __forceinline__ float bn_xhat(float x, float mean, float sd) { return (x - mean) / sd; }
__device__ __forceinline__ float bn_y(float xh, float g, float b) { return fmaf(g, xh, b); }

// TX: storage type of the convolution output x; TA: of the activation-side tensors (y, residual).  The full-store extras
// (x_hat, BN output) exist in fp32 only.
// STR: the plane size is not a multiple of V (7x7 planes: 49), so a vector of V consecutive elements may run over into the NEXT
// channel's plane (V <= P: at most one boundary): elements from `nfirst` on take that channel's scalars.  The scalar kernels this
// replaces ran the 7x7 stage at a quarter of the bandwidth (2.5 ms of the fp32 step).
template <typename TX, typename TA, int V, bool STR>
__global__ void __launch_bounds__(256)
bn_apply_kernel(const TX *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                const float *__restrict__ means, const float *__restrict__ vars, const TA *__restrict__ residual,
                TA *__restrict__ y, float *__restrict__ xhat_out, float *__restrict__ norm_out, int C, int P,
                FastDiv fdP, FastDiv fdC, size_t total, float eps, int relu) {
    const size_t nvec = total / V;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * V;
        const uint32_t plane = fd_div((uint32_t)e, fdP);
        const uint32_t c = plane - fd_div(plane, fdC) * C;
        const float mean0 = means[c], sd0 = sqrtf(vars[c] + eps), g0 = gamma[c], b0 = beta[c];
        int nfirst = V;
        float mean1 = mean0, sd1 = sd0, g1 = g0, b1 = b0;
        if (STR) {
            nfirst = min(V, P - (int)((uint32_t)e - plane * (uint32_t)P));
            const uint32_t c1 = c + 1 == (uint32_t)C ? 0u : c + 1;
            mean1 = means[c1]; sd1 = sqrtf(vars[c1] + eps); g1 = gamma[c1]; b1 = beta[c1];
        }
        float v[V], r[V], xh[V], nv[V];
        VecIO<TX, V>::load(x + e, v);
        if (residual) VecIO<TA, V>::load(residual + e, r);
#pragma unroll
        for (int q = 0; q < V; q++) {
            const bool nx = STR && q >= nfirst;
            const float mean = nx ? mean1 : mean0, sd = nx ? sd1 : sd0, g = nx ? g1 : g0, b = nx ? b1 : b0;
            xh[q] = bn_xhat(v[q], mean, sd);
            nv[q] = bn_y(xh[q], g, b);
            float o = nv[q];
            if (residual) o = fmaxf(o + r[q], 0.f);
            else if (relu) o = fmaxf(o, 0.f);
            v[q] = o;
        }
        VecIO<TA, V>::store(y + e, v);
        if (xhat_out) VecIO<float, V>::store(xhat_out + e, xh);
        if (norm_out) VecIO<float, V>::store(norm_out + e, nv);
    }
}

// BN + ReLU of a bf16 NCHW tensor written TWICE: as NCHW (what every other consumer reads) and as the zero-padded channel-last plane
// [N][H+2][W+2][C] the channel-last 3x3 kernels take (kernels_cl_bf16.hip) -- the re-layout pass of its own (read T, write T) becomes
// one extra write.  Tile = 64 channels x 64 pixels of one image, transposed through LDS; the arithmetic is bn_apply_kernel's.
typedef uint32_t bn_u32x4 __attribute__((ext_vector_type(4)));
typedef bn_u32x4 __attribute__((aligned(2))) bn_u32x4_u2;
__global__ void __launch_bounds__(256)
bn_apply_cl_kernel(const bf16_t *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ means,
                   const float *__restrict__ vars, const bf16_t *__restrict__ residual, bf16_t *__restrict__ y, bf16_t *__restrict__ ycl, int C, int P, int W,
                   float eps, FastDiv fdW, int par) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 72];  // [pixel][64 channels], pitch 72
    const int c0 = blockIdx.x * 64, p0 = blockIdx.y * 64, n = blockIdx.z;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int idx = threadIdx.x + 256 * u, cc = idx >> 3, part = idx & 7;
        const int c = c0 + cc, pp = p0 + part * 8;
        const float mean = means[c], sd = sqrtf(vars[c] + eps), g = gamma[c], b = beta[c];
        const size_t e = ((size_t)n * C + c) * P + pp;
        const int nv = min(8, P - pp);             // elements of this piece that lie in the plane (<= 0: none)
        float v[8], r[8];
        if (nv >= 8) {
            const bn_u32x4 t = *(const bn_u32x4_u2 *)(x + e);
#pragma unroll
            for (int q = 0; q < 4; q++) { v[2 * q] = __uint_as_float(t[q] << 16); v[2 * q + 1] = __uint_as_float(t[q] & 0xffff0000u); }
            if (residual) {
                const bn_u32x4 t2 = *(const bn_u32x4_u2 *)(residual + e);
#pragma unroll
                for (int q = 0; q < 4; q++) { r[2 * q] = __uint_as_float(t2[q] << 16); r[2 * q + 1] = __uint_as_float(t2[q] & 0xffff0000u); }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; q++) { v[q] = q < nv ? mi_bf2f(x[e + q]) : 0.f; r[q] = (residual && q < nv) ? mi_bf2f(residual[e + q]) : 0.f; }
        }
#pragma unroll
        for (int q = 0; q < 8; q++) { // bn_apply_kernel's arithmetic: BN, (+ residual), ReLU
            const float nvq = bn_y(bn_xhat(v[q], mean, sd), g, b);
            v[q] = residual ? fmaxf(nvq + r[q], 0.f) : fmaxf(nvq, 0.f);
        }
        const bn_u32x4 o = {mi_pack_bf2(v[0], v[1]), mi_pack_bf2(v[2], v[3]), mi_pack_bf2(v[4], v[5]), mi_pack_bf2(v[6], v[7])};
        if (nv >= 8) *(bn_u32x4_u2 *)(y + e) = o;
        else {
#pragma unroll
            for (int q = 0; q < 8; q++) if (q < nv) y[e + q] = (bf16_t)((q & 1) ? o[q >> 1] >> 16 : o[q >> 1] & 0xffffu);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            tile[(part * 8 + 2 * q) * 72 + cc] = (bf16_t)(o[q] & 0xffffu);
            tile[(part * 8 + 2 * q + 1) * 72 + cc] = (bf16_t)(o[q] >> 16);
        }
    }
    __syncthreads();
    const int H = W;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int idx = threadIdx.x + 256 * u, px = idx >> 3, piece = idx & 7;
        const int p = p0 + px;
        if (p < P) {
            const uint32_t yy = fd_div((uint32_t)p, fdW), xx = (uint32_t)p - yy * W;
            // par: the four parity planes a stride-2 3x3 reads (plane 2 (y & 1) + (x & 1), row (y >> 1) + 1, column (x >> 1) + 1, Hp = H / 2 + 1)
            const int Hp = H / 2 + 1;
            const size_t o = par ? ((((size_t)n * 4 + 2 * (yy & 1) + (xx & 1)) * Hp + (yy >> 1) + 1) * Hp + (xx >> 1) + 1) * C
                                 : (((size_t)n * (H + 2) + yy + 1) * (W + 2) + xx + 1) * C;
            *(bn_u32x4 *)(ycl + o + c0 + piece * 8) = *(const bn_u32x4 *)(tile + px * 72 + piece * 8);
        }
    }
}

// grid (C, nsplit): s1 = sum g, s2 = sum g * x_hat.  Same flattened (image, position) walk as bn_stats_kernel.
// MASK 3 = MASK 2 (gate dy by mask_src > 0) that also WRITES the gated dy: the identity blocks need relu'(out) * upstream
// twice more (BN' apply, shortcut addend of the 1x1 dgrad), and producing it here saves the separate ReLU' pass.
template <int MASK, typename TX, typename TA, int V>
__global__ void __launch_bounds__(256)
bn_bwd_reduce_kernel(const TX *__restrict__ x, const TA *__restrict__ dy, const TA *__restrict__ mask_src,
                     const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ means,
                     const float *__restrict__ vars, float *__restrict__ partial, TA *__restrict__ gated, int N, int C, int P,
                     float eps, FastDiv fdPV) {
    const int c = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
    const float mean = means[c], sd = sqrtf(vars[c] + eps), g = gamma[c], b = beta[c];
    const int cnt = (N - split + nsplit - 1) / nsplit;
    constexpr bool EXT = MASK == 2 || MASK == 3;
    const uint32_t PV = (uint32_t)P / V;
    const uint32_t total = (uint32_t)cnt * PV;
    const size_t img_stride = (size_t)nsplit * C * P;
    const size_t base = ((size_t)split * C + c) * P;
    float s1 = 0.f, s2 = 0.f;
    auto one = [&](float xv, float d, float m) -> float {
        const float xh = bn_xhat(xv, mean, sd);
        bool on = true;
        if (MASK == 1) on = bn_y(xh, g, b) > 0.f;
        if (EXT) on = m > 0.f;
        if (on) { s1 += d; s2 = fmaf(d, xh, s2); }
        return on ? d : 0.f;
    };
    auto off = [&](uint32_t idx) -> size_t {
        const uint32_t j = fd_div(idx, fdPV), i = idx - j * PV;
        return base + (size_t)j * img_stride + (size_t)i * V;
    };
    constexpr int U = (EXT ? 2 : 4) * (V == 8 ? 1 : 1); // units in flight per thread
    uint32_t idx = threadIdx.x;
    for (; idx + (U - 1) * 256 < total; idx += U * 256) {
        float xv[U][V], dv[U][V], mv[U][V];
        size_t o[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            o[u] = off(idx + u * 256);
            VecIO<TX, V>::load(x + o[u], xv[u]);
            VecIO<TA, V>::load(dy + o[u], dv[u]);
            if (EXT) VecIO<TA, V>::load(mask_src + o[u], mv[u]);
            else {
#pragma unroll
                for (int q = 0; q < V; q++) mv[u][q] = 1.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            float gq[V];
#pragma unroll
            for (int q = 0; q < V; q++) gq[q] = one(xv[u][q], dv[u][q], mv[u][q]);
            if (MASK == 3) VecIO<TA, V>::store(gated + o[u], gq);
        }
    }
    for (; idx < total; idx += 256) {
        const size_t o = off(idx);
        float xv[V], dv[V], mv[V], gq[V];
        VecIO<TX, V>::load(x + o, xv);
        VecIO<TA, V>::load(dy + o, dv);
        if (EXT) VecIO<TA, V>::load(mask_src + o, mv);
        else {
#pragma unroll
            for (int q = 0; q < V; q++) mv[q] = 1.f;
        }
#pragma unroll
        for (int q = 0; q < V; q++) gq[q] = one(xv[q], dv[q], mv[q]);
        if (MASK == 3) VecIO<TA, V>::store(gated + o, gq);
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    __shared__ float sh[8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sh[wv * 2] = s1; sh[wv * 2 + 1] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, bsum = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) { a += sh[i * 2]; bsum += sh[i * 2 + 1]; }
        float *o = partial + ((size_t)c * BN_SPLIT_MAX + split) * 3;
        o[0] = a; o[1] = bsum;
    }
}

// The reduction pass done by the dgrad that produced dy (kernels_igemm_bf16.hip: two planes [np][C] of sum g and sum g (x - mean)
// per column tile): -> the (channel, split) table bn_bwd_finalize_kernel reads, s2 divided by sd = sqrt(var + eps).
// grid (ceil(C / 64), G): a workgroup adds one chunk of the np partials for 64 channels, eight in flight per thread, fixed order.
__global__ void __launch_bounds__(256)
bn_bwd_parts_merge_kernel(const float *__restrict__ parts, int np, int C, const float *__restrict__ vars, float eps, float *__restrict__ partial) {
    const int cx = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int chunk = (np + gridDim.y - 1) / gridDim.y;
    const int p0 = blockIdx.y * chunk, p1 = min(np, p0 + chunk);
    const size_t plane = (size_t)np * C;
    float a1 = 0.f, a2 = 0.f;
    if (c < C) {
        int p = p0 + sg;
        for (; p + 7 * 4 < p1; p += 8 * 4) {
            float u1[8], u2[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const size_t o = (size_t)(p + 4 * u) * C + c;
                u1[u] = parts[o]; u2[u] = parts[plane + o];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) { a1 += u1[u]; a2 += u2[u]; }
        }
        for (; p < p1; p += 4) {
            const size_t o = (size_t)p * C + c;
            a1 += parts[o]; a2 += parts[plane + o];
        }
    }
    __shared__ float sh[2][4][64];
    sh[0][sg][cx] = a1; sh[1][sg][cx] = a2;
    __syncthreads();
    if (sg == 0 && c < C) {
        const float s1 = (sh[0][0][cx] + sh[0][1][cx]) + (sh[0][2][cx] + sh[0][3][cx]);
        const float s2 = (sh[1][0][cx] + sh[1][1][cx]) + (sh[1][2][cx] + sh[1][3][cx]);
        float *o = partial + ((size_t)c * BN_SPLIT_MAX + blockIdx.y) * 3;
        o[0] = s1; o[1] = s2 / sqrtf(vars[c] + eps);
    }
}

__global__ void __launch_bounds__(256)
bn_bwd_finalize_kernel(float *__restrict__ partial, int nsplit, int C, float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63; // one wave per channel, lane = split
    if (c >= C) return;
    const float *p = partial + (size_t)c * BN_SPLIT_MAX * 3;
    float s1 = 0.f, s2 = 0.f;
    if (lane < nsplit) { s1 = p[lane * 3]; s2 = p[lane * 3 + 1]; }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { dbeta[c] = s1; dgamma[c] = s2; }
}

template <int MASK, typename TX, typename TA, int V, bool STR>
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(const TX *__restrict__ x, const TA *__restrict__ dy, const TA *__restrict__ mask_src,
                    const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ means,
                    const float *__restrict__ vars, const float *__restrict__ dgamma, const float *__restrict__ dbeta,
                    TX *__restrict__ dx, int C, int P, FastDiv fdP, FastDiv fdC, size_t total, float inv_m, float eps) {
    const size_t nvec = total / V;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * V;
        const uint32_t plane = fd_div((uint32_t)e, fdP);
        const uint32_t c = plane - fd_div(plane, fdC) * C;
        const float mean0 = means[c], sd0 = sqrtf(vars[c] + eps), g0 = gamma[c], b0 = beta[c];
        const float k10 = dbeta[c] * inv_m, k20 = dgamma[c] * inv_m, scale0 = g0 / sd0;
        int nfirst = V;
        float mean1 = mean0, sd1 = sd0, g1 = g0, b1 = b0, k11 = k10, k21 = k20, scale1 = scale0;
        if (STR) { // (see bn_apply_kernel)
            nfirst = min(V, P - (int)((uint32_t)e - plane * (uint32_t)P));
            const uint32_t c1 = c + 1 == (uint32_t)C ? 0u : c + 1;
            mean1 = means[c1]; sd1 = sqrtf(vars[c1] + eps); g1 = gamma[c1]; b1 = beta[c1];
            k11 = dbeta[c1] * inv_m; k21 = dgamma[c1] * inv_m; scale1 = g1 / sd1;
        }
        float xv[V], dv[V], mv[V];
        VecIO<TX, V>::load(x + e, xv);
        VecIO<TA, V>::load(dy + e, dv);
        if (MASK == 2) VecIO<TA, V>::load(mask_src + e, mv);
#pragma unroll
        for (int q = 0; q < V; q++) {
            const bool nx = STR && q >= nfirst;
            const float mean = nx ? mean1 : mean0, sd = nx ? sd1 : sd0, g = nx ? g1 : g0, b = nx ? b1 : b0;
            const float k1 = nx ? k11 : k10, k2 = nx ? k21 : k20, scale = nx ? scale1 : scale0;
            const float xh = bn_xhat(xv[q], mean, sd);
            bool on = true;
            if (MASK == 1) on = bn_y(xh, g, b) > 0.f;
            if (MASK == 2) on = mv[q] > 0.f;
            const float gq = on ? dv[q] : 0.f;
            xv[q] = scale * (gq - k1 - xh * k2);
        }
        VecIO<TX, V>::store(dx + e, xv);
    }
}

static int bn_nsplit(int N, int C) {
    int ns = mi_cdiv(2048, C);
    if (ns > N) ns = N;
    if (ns > BN_SPLIT_MAX) ns = BN_SPLIT_MAX;
    if (ns < 1) ns = 1;
    return ns;
}
static int ew_blocks(size_t nvec) {
    size_t b = (nvec + 255) / 256; // one 16-byte vector per thread: measured +0.7 % img/s over a grid-stride loop of 8192 blocks
    if (b > (1u << 22)) b = 1u << 22;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- cross-replica ("sync") batch norm, an option the reference does not have (default off) ----
// Every replica holds the same number of samples per channel, so the merge of the replicas' (mean, biased var) is
//   mean_g = avg_r mean_r,   var_g = avg_r (var_r + (mean_r - mean_g)^2)
// -- two all-reduce SUMs of [C] floats per BN layer in forward; backward all-reduces the [2C] sums (dbeta, dgamma) once.
__global__ void bn_sync_k2(const float *__restrict__ sum_means, float *__restrict__ means, const float *__restrict__ vars,
                           float *__restrict__ tmp, int C, float inv_world) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mg = sum_means[c] * inv_world, d = means[c] - mg;
    tmp[c] = vars[c] + d * d;
    means[c] = mg;
}
__global__ void bn_sync_k3(const float *__restrict__ tmp, float *__restrict__ vars, int C, float inv_world) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) vars[c] = tmp[c] * inv_world;
}
__global__ void bn_sync_pack(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { out[c] = a[c]; out[C + c] = b[c]; }
}
// the replicas' summed (dbeta, dgamma) are what the dx formula needs; the gradient arena gets sum / world, so that the
// arena's own all-reduce SUM restores the global value on every replica
__global__ void bn_sync_unpack(const float *__restrict__ in, float *__restrict__ a, float *__restrict__ b, int C, float inv_world) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { a[c] = in[c] * inv_world; b[c] = in[C + c] * inv_world; }
}

// test aid (mid_bn_debug_merge): what an all-reduce SUM over R replicas delivers to each of them, for R buffers of one process
__global__ void bn_sync_emul_allreduce(float *__restrict__ bufs, int R, size_t stride, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int r = 0; r < R; r++) s += bufs[(size_t)r * stride + i];
    for (int r = 0; r < R; r++) bufs[(size_t)r * stride + i] = s;
}

// dtype codes of the storage types: MID_F32 / MID_BF16 (mi_device.h).  Supported (x, activation) pairs: (f32, f32) the
// reference's path, (bf16, bf16) the bf16-activation path, (f32, bf16) its stem (the 7x7 convolution keeps fp32 tensors).
static int bn_vec(int x_dt, int a_dt, int P) {
    if (x_dt == MID_F32 && a_dt == MID_F32) return (P & 3) == 0 ? 4 : 1;
    return (P & 7) == 0 ? 8 : (P & 3) == 0 ? 4 : 1;
}
static bool bn_pair_ok(int x_dt, int a_dt) {
    return (x_dt == MID_F32 && a_dt == MID_F32) || (x_dt == MID_BF16 && a_dt == MID_BF16) || (x_dt == MID_F32 && a_dt == MID_BF16);
}
// expands BODY(TX, TA, V) for the runtime (x_dt, a_dt, vec)
#define BN_DISPATCH(BODY)                                                                            \
    do {                                                                                             \
        if (x_dt == MID_F32 && a_dt == MID_F32) { if (vec == 4) { BODY(float, float, 4); } else { BODY(float, float, 1); } } \
        else if (x_dt == MID_BF16) { if (vec == 8) { BODY(bf16_t, bf16_t, 8); } else if (vec == 4) { BODY(bf16_t, bf16_t, 4); } else { BODY(bf16_t, bf16_t, 1); } } \
        else { if (vec == 8) { BODY(float, bf16_t, 8); } else if (vec == 4) { BODY(float, bf16_t, 4); } else { BODY(float, bf16_t, 1); } } \
    } while (0)

// elementwise kernels: the widest vector also where the plane size is not a multiple of it (STR), as long as the tensor is and a
// plane holds at least one vector.  BODY(TX, TA, V, STR)
static int bn_vec_ew(int x_dt, int a_dt, int P, size_t total, bool *str) {
    int vec = bn_vec(x_dt, a_dt, P);
    *str = false;
    const int vmax = (x_dt == MID_F32 && a_dt == MID_F32) ? 4 : 8;
    static int on = -1;
    if (on < 0) { const char *e = getenv("RESNET_MI_BN_STRADDLE"); on = e ? atoi(e) : 1; }
    if (on && vec == 1 && P >= vmax && total % vmax == 0) { vec = vmax; *str = true; }
    return vec;
}
#define BN_DISPATCH_EW(BODY)                                                                         \
    do {                                                                                             \
        if (x_dt == MID_F32 && a_dt == MID_F32) { if (str) { BODY(float, float, 4, true); } else if (vec == 4) { BODY(float, float, 4, false); } else { BODY(float, float, 1, false); } } \
        else if (x_dt == MID_BF16) { if (str) { BODY(bf16_t, bf16_t, 8, true); } else if (vec == 8) { BODY(bf16_t, bf16_t, 8, false); } else if (vec == 4) { BODY(bf16_t, bf16_t, 4, false); } else { BODY(bf16_t, bf16_t, 1, false); } } \
        else { if (str) { BODY(float, bf16_t, 8, true); } else if (vec == 8) { BODY(float, bf16_t, 8, false); } else if (vec == 4) { BODY(float, bf16_t, 4, false); } else { BODY(float, bf16_t, 1, false); } } \
    } while (0)

static size_t dt_bytes(int dt) { return dt == MID_BF16 ? 2 : 4; }
static struct { void *comm; int world, force; float *tmp; size_t tmp_floats; } g_bn_sync;

extern "C" {
/* comm == NULL turns it off.  force: also run the collectives with a one-rank communicator (self-test). */
void mid_bn_set_sync(void *comm, int world, float *tmp, size_t tmp_floats, int force) {
    g_bn_sync.comm = comm; g_bn_sync.world = world; g_bn_sync.tmp = tmp; g_bn_sync.tmp_floats = tmp_floats; g_bn_sync.force = force ? 1 : 0;
}
size_t mid_bn_ws_floats(int C) { return (size_t)C * BN_SPLIT_MAX * 3; }

/* Test aid: the device-side merge of cross-replica batch norm run on R supplied replicas of ONE process.  The launches are the
 * ones mid_bn_fwd_t / bn_bwd_impl make around their collectives (bn_sync_k2, bn_sync_k3, bn_sync_pack, bn_sync_unpack); only the
 * all-reduce itself is replaced by a kernel that sums the R buffers.  In / out, all [R][C] on the device:
 *   means, vars     per-replica statistics in, the merged statistics out (every replica ends with the same row)
 *   dgamma, dbeta   per-replica sums in (taken with the merged statistics), arena values out (sum / R: the gradient arena's own
 *                   all-reduce SUM then restores the global value)
 *   sums_out        [R][2C]: what the dx formula of each replica sees (sums over every replica's samples)
 * tmp: R * 2C floats of scratch. */
int mid_bn_debug_merge(mid_stream s, int R, int C, float *means, float *vars, float *dgamma, float *dbeta, float *sums_out, float *tmp) {
    hipStream_t st = (hipStream_t)s;
    if (R < 1 || C < 1) return -2;
    const float iw = 1.0f / (float)R;
    const size_t T = (size_t)2 * C;
    const dim3 gc(mi_cdiv(C, 256)), g2c(mi_cdiv(2 * C, 256)), b(256);
    if (means && vars) {
        for (int r = 0; r < R; r++) (void)hipMemcpyAsync(tmp + r * T, means + (size_t)r * C, sizeof(float) * C, hipMemcpyDeviceToDevice, st);
        hipLaunchKernelGGL(bn_sync_emul_allreduce, gc, b, 0, st, tmp, R, T, C);
        for (int r = 0; r < R; r++) hipLaunchKernelGGL(bn_sync_k2, gc, b, 0, st, tmp + r * T, means + (size_t)r * C, vars + (size_t)r * C, tmp + r * T + C, C, iw);
        hipLaunchKernelGGL(bn_sync_emul_allreduce, gc, b, 0, st, tmp + C, R, T, C);
        for (int r = 0; r < R; r++) hipLaunchKernelGGL(bn_sync_k3, gc, b, 0, st, tmp + r * T + C, vars + (size_t)r * C, C, iw);
    }
    if (dgamma && dbeta) {
        for (int r = 0; r < R; r++) hipLaunchKernelGGL(bn_sync_pack, gc, b, 0, st, dgamma + (size_t)r * C, dbeta + (size_t)r * C, tmp + r * T, C);
        hipLaunchKernelGGL(bn_sync_emul_allreduce, g2c, b, 0, st, tmp, R, T, 2 * C);
        if (sums_out) (void)hipMemcpyAsync(sums_out, tmp, sizeof(float) * R * T, hipMemcpyDeviceToDevice, st);
        for (int r = 0; r < R; r++) hipLaunchKernelGGL(bn_sync_unpack, gc, b, 0, st, tmp + r * T, dgamma + (size_t)r * C, dbeta + (size_t)r * C, C, iw);
    }
    MI_LAUNCH_CHECK("mid_bn_debug_merge");
    return 0;
}

/* (the bf16 kernels pad every image's columns to a multiple of 8) */
size_t mid_bn_parts_floats(int N, int K, int Ho) { return (size_t)3 * (size_t)mi_cdiv((long)N * ((Ho * Ho + 7) / 8 * 8), 128) * 4 * K; }

/* side output of the NEXT forward apply (one-shot: the caller sets it right before the call, the launcher consumes it): the activation
 * also as a zero-padded channel-last plane of H x H pixels (bn_apply_cl_kernel) */
static struct { void *out; int H; } g_bn_cl = {nullptr, 0};
/* H > 0: one plane with a halo of 1; H < 0: the four parity planes of a stride-2 3x3 over an |H| x |H| image */
extern "C" void mid_bn_set_cl_out(void *ycl, int H) { g_bn_cl.out = ycl; g_bn_cl.H = H; }
/* taken (and cleared) at the ENTRY of the public launchers, so that an early error return cannot leave it for some later layer */
static void *bn_take_cl_out(int *H) { void *p = g_bn_cl.out; *H = g_bn_cl.H; g_bn_cl.out = nullptr; return p; }
static int bn_fwd_apply(hipStream_t st, const void *x, int x_dt, const float *gamma, const float *beta, const void *residual,
                        const float *means, const float *vars, void *y, int a_dt, float *xhat_out, float *norm_out, int N, int C,
                        int P, float eps, int relu, void *ycl = nullptr, int Hcl = 0) {
    const int Hab = Hcl < 0 ? -Hcl : Hcl;
    if (ycl && x_dt == MID_BF16 && a_dt == MID_BF16 && (residual || relu) && !xhat_out && !norm_out && C % 64 == 0 && Hab * Hab == P && !(Hcl < 0 && (Hab & 1))) {
        hipLaunchKernelGGL(bn_apply_cl_kernel, dim3(C / 64, mi_cdiv(P, 64), N), dim3(256), 0, st, (const bf16_t *)x, gamma, beta, means, vars,
                           (const bf16_t *)residual, (bf16_t *)y, (bf16_t *)ycl, C, P, Hab, eps, make_fastdiv(Hab), Hcl < 0);
        MI_LAUNCH_CHECK("bn_apply_cl_kernel");
        return 0;
    }
    if (ycl) { mi_record_error("bn_fwd_apply", "channel-last side output requested for a form that has none"); return -2; }
    const size_t total = (size_t)N * C * P;
    const FastDiv fdP = make_fastdiv(P), fdC = make_fastdiv(C);
    bool str;
    const int vec = bn_vec_ew(x_dt, a_dt, P, total, &str);
#define APPLY(TX, TA, V, STR)                                                                                              \
    hipLaunchKernelGGL((bn_apply_kernel<TX, TA, V, STR>), dim3(ew_blocks(total / V)), dim3(256), 0, st, (const TX *)x, gamma, beta, \
                       means, vars, (const TA *)residual, (TA *)y, xhat_out, norm_out, C, P, fdP, fdC, total, eps, relu)
    BN_DISPATCH_EW(APPLY);
#undef APPLY
    MI_LAUNCH_CHECK("bn_apply_kernel");
    return 0;
}

static int bn_stats_launch(hipStream_t st, float *ws, const void *x, int x_dt, int N, int C, int P, int ns) {
    if (x_dt == MID_F32) {
        if ((P & 3) == 0) hipLaunchKernelGGL((bn_stats_kernel<float, 4>), dim3(C, ns), dim3(256), 0, st, (const float *)x, ws, N, C, P, make_fastdiv(P / 4));
        else hipLaunchKernelGGL((bn_stats_kernel<float, 1>), dim3(C, ns), dim3(256), 0, st, (const float *)x, ws, N, C, P, make_fastdiv(P));
    } else {
        if ((P & 7) == 0) hipLaunchKernelGGL((bn_stats_kernel<bf16_t, 8>), dim3(C, ns), dim3(256), 0, st, (const bf16_t *)x, ws, N, C, P, make_fastdiv(P / 8));
        else if ((P & 3) == 0) hipLaunchKernelGGL((bn_stats_kernel<bf16_t, 4>), dim3(C, ns), dim3(256), 0, st, (const bf16_t *)x, ws, N, C, P, make_fastdiv(P / 4));
        else hipLaunchKernelGGL((bn_stats_kernel<bf16_t, 1>), dim3(C, ns), dim3(256), 0, st, (const bf16_t *)x, ws, N, C, P, make_fastdiv(P));
    }
    MI_LAUNCH_CHECK("bn_stats_kernel");
    return 0;
}

/* statistics only (means / biased vars of x): the recompute policy re-derives activations from them in backward */
int mid_bn_stats_t(mid_stream s, float *ws, const void *x, int x_dt, float *means, float *vars, int N, int C, int P) {
    hipStream_t st = (hipStream_t)s;
    const int ns = bn_nsplit(N, C);
    if (bn_stats_launch(st, ws, x, x_dt, N, C, P, ns)) return -1;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(mi_cdiv(C, 4)), dim3(256), 0, st, ws, ns, C, means, vars);
    MI_LAUNCH_CHECK("bn_finalize_kernel");
    return 0;
}

/* y from x and GIVEN statistics (no reduction): forward's second half, and the backward-time recomputation of an activation */
int mid_bn_apply_t(mid_stream s, const void *x, int x_dt, const float *gamma, const float *beta, const void *residual,
                   const float *means, const float *vars, void *y, int a_dt, int N, int C, int P, float eps, int relu) {
    int Hcl = 0;
    void *ycl = bn_take_cl_out(&Hcl);
    if (!bn_pair_ok(x_dt, a_dt)) { mi_record_error("mid_bn_apply_t", "unsupported storage types"); return -2; }
    hipStream_t st = (hipStream_t)s;
    mi_prof_begin(st, MI_FAM_BN, 0.0, (double)N * C * P * (dt_bytes(x_dt) + dt_bytes(a_dt) * (residual ? 2 : 1)));
    const int rc = bn_fwd_apply(st, x, x_dt, gamma, beta, residual, means, vars, y, a_dt, nullptr, nullptr, N, C, P, eps, relu, ycl, Hcl);
    mi_prof_end(st);
    return rc;
}

int mid_bn_fwd_t(mid_stream s, float *ws, const mid_bn_parts *parts, const void *x, int x_dt, const float *gamma, const float *beta,
                 const void *residual, float *means, float *vars, void *y, int a_dt, float *xhat_out, float *norm_out, int N, int C,
                 int P, float eps, int relu) {
    int Hcl = 0;
    void *ycl = bn_take_cl_out(&Hcl);
    if (!bn_pair_ok(x_dt, a_dt)) { mi_record_error("mid_bn_fwd_t", "unsupported storage types"); return -2; }
    if ((xhat_out || norm_out) && !(x_dt == MID_F32 && a_dt == MID_F32)) { mi_record_error("mid_bn_fwd_t", "full-store tensors exist in fp32 only"); return -2; }
    hipStream_t st = (hipStream_t)s;
    const double xb = (double)N * C * P * dt_bytes(x_dt), ab = (double)N * C * P * dt_bytes(a_dt);
    if (!parts || parts->nparts <= 0) {
        const int ns = bn_nsplit(N, C);
        mi_prof_begin(st, MI_FAM_BN, 0.0, 2 * xb + ab * (residual ? 2 : 1));
        if (bn_stats_launch(st, ws, x, x_dt, N, C, P, ns)) return -1;
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(mi_cdiv(C, 4)), dim3(256), 0, st, ws, ns, C, means, vars);
        MI_LAUNCH_CHECK("bn_finalize_kernel");
    } else {
        int G = parts->nparts / 64;
        if (G < 1) G = 1;
        if (G > BN_SPLIT_MAX) G = BN_SPLIT_MAX;
        // the statistics pass over x is gone: 2 passes (+1 fused add) instead of 3
        mi_prof_begin(st, MI_FAM_BN, 0.0, xb + ab * (residual ? 2 : 1));
        hipLaunchKernelGGL(bn_parts_merge_kernel, dim3(mi_cdiv(C, 64), G), dim3(256), 0, st, parts->buf, parts->nparts, C, ws);
        MI_LAUNCH_CHECK("bn_parts_merge_kernel");
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(mi_cdiv(C, 4)), dim3(256), 0, st, ws, G, C, means, vars);
        MI_LAUNCH_CHECK("bn_finalize_kernel");
    }
    if (g_bn_sync.comm && g_bn_sync.world > 1 - g_bn_sync.force) {
        // statistics over ALL replicas (equal counts): two small all-reduces on the launch stream, through the sync-BN
        // communicator (its own: the gradient buckets' collectives run concurrently on the comm stream)
        const float iw = 1.0f / (float)g_bn_sync.world;
        float *tmp = g_bn_sync.tmp;
        if (C > g_bn_sync.tmp_floats / 2) { mi_record_error("sync BN", "scratch too small"); return -3; }
        (void)hipMemcpyAsync(tmp, means, sizeof(float) * C, hipMemcpyDeviceToDevice, st);
        if (mid_rccl_allreduce_sum(g_bn_sync.comm, tmp, (size_t)C, (mid_stream)st)) return -1;
        hipLaunchKernelGGL(bn_sync_k2, dim3(mi_cdiv(C, 256)), dim3(256), 0, st, tmp, means, vars, tmp + C, C, iw);
        if (mid_rccl_allreduce_sum(g_bn_sync.comm, tmp + C, (size_t)C, (mid_stream)st)) return -1;
        hipLaunchKernelGGL(bn_sync_k3, dim3(mi_cdiv(C, 256)), dim3(256), 0, st, tmp + C, vars, C, iw);
        MI_LAUNCH_CHECK("bn_sync");
    }
    const int rc = bn_fwd_apply(st, x, x_dt, gamma, beta, residual, means, vars, y, a_dt, xhat_out, norm_out, N, C, P, eps, relu, ycl, Hcl);
    mi_prof_end(st);
    return rc;
}

int mid_bn_fwd(mid_stream s, float *ws, const float *x, const float *gamma, const float *beta, const float *residual,
               float *means, float *vars, float *y, float *xhat_out, float *norm_out, int N, int C, int P, float eps,
               int relu) {
    return mid_bn_fwd_t(s, ws, nullptr, x, MID_F32, gamma, beta, residual, means, vars, y, MID_F32, xhat_out, norm_out, N, C, P, eps, relu);
}

int mid_bn_fwd_parts(mid_stream s, float *ws, const mid_bn_parts *parts, const float *x, const float *gamma, const float *beta,
                     const float *residual, float *means, float *vars, float *y, float *xhat_out, float *norm_out, int N,
                     int C, int P, float eps, int relu) {
    return mid_bn_fwd_t(s, ws, parts, x, MID_F32, gamma, beta, residual, means, vars, y, MID_F32, xhat_out, norm_out, N, C, P, eps, relu);
}

// bparts != NULL (nparts > 0): the reduction pass was done by the dgrad that produced dy (which is already gated: mask_mode 0)
static int bn_bwd_impl(hipStream_t st, float *ws, const void *x, int x_dt, const float *gamma, const float *beta, const float *means,
                       const float *vars, const void *dy, const void *mask_src, void *gated_out, int a_dt, void *dx, float *dgamma,
                       float *dbeta, int N, int C, int P, float eps, int mask_mode, const mid_bn_bwd_parts *bparts = nullptr) {
    if (!bn_pair_ok(x_dt, a_dt)) { mi_record_error("mid_bn_bwd", "unsupported storage types"); return -2; }
    int ns = bn_nsplit(N, C);
    dim3 grid(C, ns), block(256);
    if (mask_mode >= 2 && !mask_src) { mi_record_error("mid_bn_bwd", "mask_src missing"); return -2; }
    if (mask_mode == 3 && !gated_out) { mi_record_error("mid_bn_bwd_gate", "gated_out missing"); return -2; }
    // passes over N*C*P elements: reduce reads x, dy (+mask) (+writes gated); apply reads x, dy (+mask) | x, gated; writes dx
    const double xb = (double)N * C * P * dt_bytes(x_dt), ab = (double)N * C * P * dt_bytes(a_dt);
    mi_prof_begin(st, MI_FAM_BN, 0.0, bparts ? 2 * xb + ab : 3 * xb + ab * (mask_mode >= 2 ? 4 : 2));
    const int vec = bn_vec(x_dt, a_dt, P);
    const FastDiv fdPV = make_fastdiv(P / vec);
    if (bparts) {
        ns = bparts->nparts >= 64 * 8 ? 64 : bparts->nparts >= 64 ? 8 : 1;
        hipLaunchKernelGGL(bn_bwd_parts_merge_kernel, dim3(mi_cdiv(C, 64), ns), dim3(256), 0, st, bparts->buf, bparts->nparts, C, vars, eps, ws);
        MI_LAUNCH_CHECK("bn_bwd_parts_merge_kernel");
    } else {
#define BWD_REDUCE_M(M_, TX, TA, V)                                                                                         \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<M_, TX, TA, V>), grid, block, 0, st, (const TX *)x, (const TA *)dy, (const TA *)mask_src, \
                       gamma, beta, means, vars, ws, (TA *)gated_out, N, C, P, eps, fdPV)
#define BWD_R0(TX, TA, V) BWD_REDUCE_M(0, TX, TA, V)
#define BWD_R1(TX, TA, V) BWD_REDUCE_M(1, TX, TA, V)
#define BWD_R2(TX, TA, V) BWD_REDUCE_M(2, TX, TA, V)
#define BWD_R3(TX, TA, V) BWD_REDUCE_M(3, TX, TA, V)
    if (mask_mode == 0) BN_DISPATCH(BWD_R0);
    else if (mask_mode == 1) BN_DISPATCH(BWD_R1);
    else if (mask_mode == 2) BN_DISPATCH(BWD_R2);
    else BN_DISPATCH(BWD_R3);
#undef BWD_R0
#undef BWD_R1
#undef BWD_R2
#undef BWD_R3
#undef BWD_REDUCE_M
    MI_LAUNCH_CHECK("bn_bwd_reduce_kernel");
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(mi_cdiv(C, 4)), dim3(256), 0, st, ws, ns, C, dgamma, dbeta);
    MI_LAUNCH_CHECK("bn_bwd_finalize_kernel");
    const size_t total = (size_t)N * C * P;
    const FastDiv fdP = make_fastdiv(P), fdC = make_fastdiv(C);
    float inv_m = 1.0f / (float)((size_t)N * P);
    const float *dg_apply = dgamma, *db_apply = dbeta;
    const bool sync = g_bn_sync.comm && g_bn_sync.world > 1 - g_bn_sync.force;
    if (sync) {
        if (2 * C > g_bn_sync.tmp_floats) { mi_record_error("sync BN", "scratch too small"); return -3; }
        float *tmp = g_bn_sync.tmp;
        hipLaunchKernelGGL(bn_sync_pack, dim3(mi_cdiv(C, 256)), dim3(256), 0, st, dgamma, dbeta, tmp, C);
        if (mid_rccl_allreduce_sum(g_bn_sync.comm, tmp, (size_t)2 * C, (mid_stream)st)) return -1;
        dg_apply = tmp; db_apply = tmp + C;              // sums over every replica's samples
        inv_m = 1.0f / ((float)g_bn_sync.world * (float)((size_t)N * P));
    }
    const void *dy_apply = mask_mode == 3 ? gated_out : dy; // mode 3: the gated dy is already there, no mask needed
    {
    bool str;
    const int vec = bn_vec_ew(x_dt, a_dt, P, total, &str); // (shadows the reduction pass's vector width)
    dim3 g2(ew_blocks(total / vec));
#define BWD_APPLY_M(M_, TX, TA, V, STR)                                                                                        \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<M_, TX, TA, V, STR>), g2, block, 0, st, (const TX *)x, (const TA *)dy_apply, (const TA *)mask_src, \
                       gamma, beta, means, vars, dg_apply, db_apply, (TX *)dx, C, P, fdP, fdC, total, inv_m, eps)
#define BWD_A0(TX, TA, V, STR) BWD_APPLY_M(0, TX, TA, V, STR)
#define BWD_A1(TX, TA, V, STR) BWD_APPLY_M(1, TX, TA, V, STR)
#define BWD_A2(TX, TA, V, STR) BWD_APPLY_M(2, TX, TA, V, STR)
    if (mask_mode == 0 || mask_mode == 3) BN_DISPATCH_EW(BWD_A0);
    else if (mask_mode == 1) BN_DISPATCH_EW(BWD_A1);
    else BN_DISPATCH_EW(BWD_A2);
    }
#undef BWD_A0
#undef BWD_A1
#undef BWD_A2
#undef BWD_APPLY_M
    if (sync) hipLaunchKernelGGL(bn_sync_unpack, dim3(mi_cdiv(C, 256)), dim3(256), 0, st, g_bn_sync.tmp, dgamma, dbeta, C, 1.0f / (float)g_bn_sync.world);
    mi_prof_end(st);
    MI_LAUNCH_CHECK("bn_bwd_apply_kernel");
    return 0;
}

int mid_bn_bwd_t(mid_stream s, float *ws, const void *x, int x_dt, const float *gamma, const float *beta, const float *means,
                 const float *vars, const void *dy, const void *mask_src, void *gated_out, int a_dt, void *dx, float *dgamma,
                 float *dbeta, int N, int C, int P, float eps, int mask_mode) {
    if (mask_mode < 0 || mask_mode > 3) { mi_record_error("mid_bn_bwd", "mask_mode"); return -2; }
    return bn_bwd_impl((hipStream_t)s, ws, x, x_dt, gamma, beta, means, vars, dy, mask_src, gated_out, a_dt, dx, dgamma, dbeta, N, C, P, eps, mask_mode);
}
int mid_bn_bwd_parts_t(mid_stream s, float *ws, const mid_bn_bwd_parts *parts, const void *x, int x_dt, const float *gamma, const float *beta,
                       const float *means, const float *vars, const void *dy_gated, int a_dt, void *dx, float *dgamma, float *dbeta, int N,
                       int C, int P, float eps) {
    if (!parts || parts->nparts <= 0) { mi_record_error("mid_bn_bwd_parts_t", "no partials"); return -2; }
    return bn_bwd_impl((hipStream_t)s, ws, x, x_dt, gamma, beta, means, vars, dy_gated, nullptr, nullptr, a_dt, dx, dgamma, dbeta, N, C, P, eps, 0, parts);
}
int mid_bn_bwd(mid_stream s, float *ws, const float *x, const float *gamma, const float *beta, const float *means,
               const float *vars, const float *dy, const float *mask_src, float *dx, float *dgamma, float *dbeta, int N,
               int C, int P, float eps, int mask_mode) {
    if (mask_mode < 0 || mask_mode > 2) { mi_record_error("mid_bn_bwd", "mask_mode"); return -2; }
    return bn_bwd_impl((hipStream_t)s, ws, x, MID_F32, gamma, beta, means, vars, dy, mask_src, nullptr, MID_F32, dx, dgamma, dbeta, N, C, P, eps, mask_mode);
}
int mid_bn_bwd_gate(mid_stream s, float *ws, const float *x, const float *gamma, const float *beta, const float *means,
                    const float *vars, const float *dy, const float *mask_src, float *gated_out, float *dx, float *dgamma,
                    float *dbeta, int N, int C, int P, float eps) {
    return bn_bwd_impl((hipStream_t)s, ws, x, MID_F32, gamma, beta, means, vars, dy, mask_src, gated_out, MID_F32, dx, dgamma, dbeta, N, C, P, eps, 3);
}
}
