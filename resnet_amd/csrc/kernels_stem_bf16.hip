// kernels_stem_bf16.hip -- the 7x7 stride-2 stem convolution (3 -> 64 channels; doConvolution / convolutionDerivWeights,
// resnet.cu:109-156, 227-281, at the shape of resnet.cu:3245-3250) on v_mfma_f32_32x32x16_bf16, used when the trainer stores
// activations as bf16 (mi_trainer_set_dtype).  The image and the weights are rounded to bf16 on the way in, products are
// accumulated in fp32, the convolution output / its gradient stay fp32 tensors.
//
// Three input channels do not tile a 64-deep reduction, so this is not the implicit GEMM of kernels_igemm_bf16.hip.  Instead:
//   * st_pad_kernel re-lays the image once per batch as ZERO-PADDED PARITY PLANES in bf16,
//         xp[n][c][ph][pw][i][j] = x[n][c][2 i + ph - 4][2 j + pw - 4]   (0 outside the image),   i < Ho + 3, j < pitch
//     so that tap (r, s) of output pixel (ho, wo) is the element (ph, pw, i, j) = ((r+1)&1, (s+1)&1, ho + (r+1)/2, wo + (s+1)/2):
//     every tap is a unit-stride read along the output pixels and NO tap needs a mask.
//   * forward: pixels are the MFMA rows.  For one (c, r) the four taps s = 0, 2, 4, 6 are four CONSECUTIVE elements of plane
//     pw = 1 starting at j = wo, and s = 1, 3, 5 are elements 1..3 of the same four of plane pw = 0: a lane's 8 reduction values
//     are two 8-byte loads (the unused element meets a zero weight).  Reduction = 21 (c, r) groups of 8 (+1 zero group) = 176.
//     The 64 x 176 weights live in registers for the whole kernel; a wave walks 32-pixel tiles.  Output through a wave-private
//     LDS image so that a wave instruction writes whole 128-byte lines.
//   * weight gradient: dW[64][147] = sum over pixels dY[64][px] * patch[147][px]: both operands are contiguous along the pixels
//     (dY as it lies, the patch as 8 consecutive elements of a parity plane), loaded straight into the MFMA operand registers --
//     no LDS at all.  Every wave owns a fixed set of 16-pixel chunks and writes its own 64 x 160 partial; a second kernel adds
//     the partials in wave order (deterministic) into KCRS.
// Bounds: both kernels are bound by the fp32 tensor they stream (822 MB at N = 256: 0.16 ms at 5 TB/s); MFMA time 26 us.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "mi_common.hpp"
#include "mi_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float pf4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef u32x4 __attribute__((aligned(2))) u32x4_u;
typedef u32x2 __attribute__((aligned(2))) u32x2_u;

#ifndef ST_ABL
#define ST_ABL 0 /* experiments: 1 no output stores, 2 aligned patch loads (wrong results), 3 no patch loads */
#endif
// compile-time loop: f(integral_constant<int, 0>{}) ... f(integral_constant<int, N - 1>{})
template <class F, int... I> __device__ __forceinline__ void st_static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void st_static_for(F &&f) { st_static_for_impl(f, std::make_integer_sequence<int, N>{}); }
#define ST_K 64        /* output channels */
#define ST_KRED 176    /* forward reduction: 22 groups of 8 */
#define ST_COLS 160    /* weight-gradient columns: 147 (c, r, s) padded to 5 x 32 */
#define ST_WAVES 2048  /* waves of the forward / weight-gradient grids (2 per SIMD) */

struct StArgs {
    int N, H, Ho, Wo, P;   // P = Ho * Wo
    int R, pitch;          // rows and row pitch (elements) of one parity plane: Ho + 3, (Wo + 3) rounded up to 8
    size_t img;            // elements of one image in xp: 3 * 4 * R * pitch
    FastDiv fdWo, fdTpi;   // fdTpi: tiles (forward, 32 px) or chunks (wgrad, 16 px) per image
};

static inline int st_pitch(int Wo) { return (Wo + 3 + 7) & ~7; }

// ---- image -> zero-padded parity planes (bf16) ----
__global__ void __launch_bounds__(256)
st_pad_kernel(const float *__restrict__ x, u16 *__restrict__ xp, const StArgs g, size_t total_pairs) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_pairs; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 2;                       // two consecutive j of one row
        const uint32_t hp = (uint32_t)g.pitch;
        const size_t row = e / hp;
        const int j = (int)(e - row * hp);
        const int ii = (int)(row % (uint32_t)g.R);
        size_t q = row / (uint32_t)g.R;               // ((n * 3 + c) * 2 + ph) * 2 + pw
        const int pw = (int)(q & 1); q >>= 1;
        const int ph = (int)(q & 1); q >>= 1;         // q = n * 3 + c
        const int y = 2 * ii + ph - 4;
        float v0 = 0.f, v1 = 0.f;
        if (y >= 0 && y < g.H) {
            const float *src = x + (q * g.H + y) * (size_t)g.H;
            const int x0 = 2 * j + pw - 4, x1 = x0 + 2;
            if (x0 >= 0 && x0 < g.H) v0 = src[x0];
            if (x1 >= 0 && x1 < g.H) v1 = src[x1];
        }
        *(uint32_t *)(xp + e) = mi_pack_bf2(v0, v1);
    }
}

// ---- weights KCRS fp32 -> forward operand [64][176] bf16: slot 8 g + e, g = 7 c + r; e 0..3: s = 2 e (plane pw = 1);
//      e 4..7: s = 2 (e - 4) - 1 (plane pw = 0; e = 4 is the unused element: 0); g = 21: 0 ----
__global__ void __launch_bounds__(256)
st_wt_kernel(const float *__restrict__ w, u16 *__restrict__ wf) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ST_K * ST_KRED) return;
    const int k = i / ST_KRED, q = i - k * ST_KRED, gq = q >> 3, e = q & 7;
    float v = 0.f;
    if (gq < 21) {
        const int s = e < 4 ? 2 * e : 2 * (e - 4) - 1;
        if (s >= 0) v = w[(size_t)k * 147 + gq * 7 + s]; // KCRS: ((k * 3 + c) * 7 + r) * 7 + s
    }
    wf[i] = mi_f2bf(v);
}

// Batch-norm statistics of the stem's output, taken from the forward kernels' accumulators (what kernels_igemm*.hip do for every
// other convolution): a lane holds 16 pixels of ONE channel per accumulator, so it keeps a running (count, sum of d, sum of d^2)
// with d = value - its own first value; at the end the two half-waves are Chan-merged and the wave writes one partial row
// (count, mean, M2) per channel -- three planes [waves][64], the format bn_parts_merge_kernel reads.
struct StStat { float s0, sd, sq; int n; };
__device__ __forceinline__ void st_stat_add(StStat &a, const f32x16 &v) {
    if (a.n == 0) a.s0 = v[0];
#pragma unroll
    for (int r = 0; r < 16; r++) { const float d = v[r] - a.s0; a.sd += d; a.sq = fmaf(d, d, a.sq); }
    a.n += 16;
}
__device__ __forceinline__ void st_stat_write(const StStat &a, float *__restrict__ part, int np, int row, int ch, int kh) {
    // this half-wave's (n, mean, M2), merged with the other half's (same channel, other pixels)
    const float n = (float)a.n, inv = a.n ? 1.0f / n : 0.f;
    const float mean = a.n ? a.s0 + a.sd * inv : 0.f, m2 = fmaxf(a.sq - a.sd * a.sd * inv, 0.f);
    const float n2 = __shfl_xor(n, 32, 64), mean2 = __shfl_xor(mean, 32, 64), m22 = __shfl_xor(m2, 32, 64);
    const float nt = n + n2, dl = mean2 - mean;
    const float mt = nt > 0.f ? mean + dl * (n2 / nt) : 0.f, m2t = nt > 0.f ? m2 + m22 + dl * dl * (n * n2 / nt) : 0.f;
    if (kh == 0) {
        const size_t o = (size_t)row * ST_K + ch, plane = (size_t)np * ST_K;
        part[o] = nt; part[plane + o] = mt; part[2 * plane + o] = m2t;
    }
}

// ---- forward ----
// YB: the output tensor is bf16 (the bf16 trainer: the stem's output is an activation tensor like any other convolution's; the statistics
// still come from the fp32 accumulators), else fp32
template <bool YB>
__global__ void __launch_bounds__(256)
st_fwd_kernel(const u16 *__restrict__ xp, const u16 *__restrict__ wf, void *__restrict__ yv, const StArgs g, int ntiles, float *__restrict__ bn_part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char st_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    constexpr int PITCH = 32 * 4 + 16;                // LDS image [64 channels][32 pixels] fp32
    unsigned char *img = st_smem + wave * (64 * PITCH);
    // weights: B operand (columns = output channels), lane = (channel l31 + 32 t, reduction half kh)
    bf16x8 wfr[2][ST_KRED / 16];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int s = 0; s < ST_KRED / 16; s++) {
            const u32x4 v = *(const u32x4 *)(wf + (size_t)(t * 32 + l31) * ST_KRED + s * 16 + kh * 8);
            wfr[t][s] = *(const bf16x8 *)&v;
        }
    // element offset of this lane's (c, r) group of sub-step s inside one image of xp (plane pw = 1; pw = 0 is R * pitch behind... in front)
    uint32_t goff[ST_KRED / 16];
#pragma unroll
    for (int s = 0; s < ST_KRED / 16; s++) {
        int gq = 2 * s + kh;
        if (gq > 20) gq = 20;                        // the zero group: any valid address
        const int c = gq / 7, r = gq - 7 * c;
        const int ph = (r + 1) & 1, di = (r + 1) >> 1;
        goff[s] = (uint32_t)((((c * 2 + ph) * 2 + 1) * g.R + di) * g.pitch);
    }
    const uint32_t pw0 = (uint32_t)(g.R * g.pitch);  // plane pw = 0 lies R * pitch elements BEFORE plane pw = 1
    // a wave walks a CONTIGUOUS range of tiles: its reads of the planes and its 64 output rows are sequential streams
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int per = (ntiles + nw - 1) / nw, t_end = min(ntiles, (gw + 1) * per);
    StStat stat[2] = {{0.f, 0.f, 0.f, 0}, {0.f, 0.f, 0.f, 0}};
    for (int tile = gw * per; tile < t_end; tile++) {
        const uint32_t n = fd_div((uint32_t)tile, g.fdTpi);
        const uint32_t p0 = ((uint32_t)tile - n * g.fdTpi.d) * 32u;
        const uint32_t p = p0 + (uint32_t)l31;
        const uint32_t ho = fd_div(p, g.fdWo), wo = p - ho * (uint32_t)g.Wo;
        const u16 *base = xp + (size_t)n * g.img + (size_t)ho * g.pitch + (ST_ABL == 2 ? (wo & ~3u) : wo);
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
        // all 22 loads of the tile first (the scheduler would otherwise keep two sub-steps in flight and pay a memory latency per
        // sub-step), then the MFMAs as the data arrives
        u32x2 a1[ST_KRED / 16], a0[ST_KRED / 16];
#pragma unroll
        for (int s = 0; s < ST_KRED / 16; s++) {
            if (ST_ABL == 3 && tile != gw * per) continue;
            a1[s] = *(const u32x2_u *)(base + goff[s]);        // plane pw = 1: s = 0, 2, 4, 6
            a0[s] = *(const u32x2_u *)(base + goff[s] - pw0);  // plane pw = 0: (unused), s = 1, 3, 5
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < ST_KRED / 16; s++) {
            const u32x4 av = {a1[s][0], a1[s][1], a0[s][0], a0[s][1]};
#pragma unroll
            for (int t = 0; t < 2; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8 *)&av, wfr[t][s], acc[t], 0, 0, 0);
        }
        // accumulator t: rows = pixels (r & 3) + 8 (r >> 2) + 4 kh, column = channel l31 + 32 t
        if (bn_part) { st_stat_add(stat[0], acc[0]); st_stat_add(stat[1], acc[1]); }
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                pf4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
                *(pf4 *)(img + (t * 32 + l31) * PITCH + (8 * q + 4 * kh) * 4) = v;
            }
        // read back with 8 lanes along a channel row (32 pixels = 128 bytes): 8 channels per wave instruction
        const int c4 = lane & 7, r0 = lane >> 3;
        const size_t doff = ((size_t)n * ST_K) * g.P + p0 + 4 * c4;
#pragma unroll
        for (int ps = 0; ps < 8; ps++) {
            const int ch = ps * 8 + r0;
            const pf4 v = *(const pf4 *)(img + ch * PITCH + c4 * 16);
            if (ST_ABL != 1 || v[0] == 1.2345f) {
                if (YB) *(u32x2 *)((u16 *)yv + doff + (size_t)ch * g.P) = u32x2{mi_pack_bf2(v[0], v[1]), mi_pack_bf2(v[2], v[3])};
                else *(pf4 *)((float *)yv + doff + (size_t)ch * g.P) = v;
            }
        }
    }
    if (bn_part) { st_stat_write(stat[0], bn_part, nw, gw, l31, kh); st_stat_write(stat[1], bn_part, nw, gw, 32 + l31, kh); }
}

// ---- weight gradient: per-wave partials [wave][64][160] ----
// NS register sets of operands in a ring (see st32_wgrad_kernel below for why the loads are unconditional): a chunk's 10 MFMAs are
// 320 cycles, so seven chunks of loads are kept in flight; one wave per SIMD.
// DB: dY is a bf16 tensor (the bf16 trainer), else fp32 rounded to bf16 on the way in -- the same operand bits either way
template <bool DB>
__global__ void __launch_bounds__(256)
st_wgrad_kernel(const u16 *__restrict__ xp, const void *__restrict__ dyv, float *__restrict__ part, const StArgs g, int nchunks) {
    constexpr int NS = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    // B operand: column j = l31 + 32 t = (c * 7 + r) * 7 + s; 8 consecutive pixels of its parity plane
    uint32_t coff[5];
#pragma unroll
    for (int t = 0; t < 5; t++) {
        int j = l31 + 32 * t;
        if (j > 146) j = 146;                        // padding columns: any valid address (their results are never read)
        const int c = j / 49, rs = j - 49 * c, r = rs / 7, s = rs - 7 * r;
        const int ph = (r + 1) & 1, di = (r + 1) >> 1, pw = (s + 1) & 1, dj = (s + 1) >> 1;
        coff[t] = (uint32_t)((((c * 2 + ph) * 2 + pw) * g.R + di) * g.pitch + dj + 8 * kh);
    }
    f32x16 acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int t = 0; t < 5; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][t][r] = 0.f;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    // a wave owns a CONTIGUOUS range of chunks: its 64 dY rows and its plane rows are sequential streams
    const int per = (nchunks + nw - 1) / nw, c_beg = gw * per, c_end = min(nchunks, (gw + 1) * per);
    pf4 a[NS][2][DB ? 1 : 2];       // DB: 8 bf16 values = 16 bytes, kept as the raw bits in one pf4
    u32x4 b[NS][5];
    if (c_beg < c_end) {
        auto load = [&](int chunk, auto set_tag) {
            constexpr int S = decltype(set_tag)::value;
            chunk = min(chunk, c_end - 1);
            const uint32_t n = fd_div((uint32_t)chunk, g.fdTpi);
            const uint32_t p0 = ((uint32_t)chunk - n * g.fdTpi.d) * 16u;   // 16 consecutive pixels of one output row (Wo % 16 == 0)
            const uint32_t ho = fd_div(p0, g.fdWo), wo0 = p0 - ho * (uint32_t)g.Wo;
            const size_t da = ((size_t)n * ST_K + l31) * g.P + p0 + 8 * kh;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                if constexpr (DB) a[S][i][0] = *(const pf4 *)((const u16 *)dyv + da + (size_t)(32 * i) * g.P);
                else {
                    a[S][i][0] = *(const pf4 *)((const float *)dyv + da + (size_t)(32 * i) * g.P);
                    a[S][i][DB ? 0 : 1] = *(const pf4 *)((const float *)dyv + da + (size_t)(32 * i) * g.P + 4);
                }
            }
            const u16 *xb = xp + (size_t)n * g.img + (size_t)ho * g.pitch + wo0;
#pragma unroll
            for (int t = 0; t < 5; t++) b[S][t] = *(const u32x4_u *)(xb + coff[t]);
        };
        auto mul = [&](auto set_tag) {
            constexpr int S = decltype(set_tag)::value;
            bf16x8 av[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                if constexpr (DB) av[i] = *(const bf16x8 *)&a[S][i][0];
                else {
                    const u32x4 pk = {mi_pack_bf2(a[S][i][0][0], a[S][i][0][1]), mi_pack_bf2(a[S][i][0][2], a[S][i][0][3]),
                                      mi_pack_bf2(a[S][i][DB ? 0 : 1][0], a[S][i][DB ? 0 : 1][1]), mi_pack_bf2(a[S][i][DB ? 0 : 1][2], a[S][i][DB ? 0 : 1][3])};
                    av[i] = *(const bf16x8 *)&pk;
                }
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int t = 0; t < 5; t++) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], *(const bf16x8 *)&b[S][t], acc[i][t], 0, 0, 0);
        };
        st_static_for<NS - 1>([&](auto J) { load(c_beg + decltype(J)::value, J); });
        for (int chunk = c_beg; chunk < c_end; chunk += NS)
            st_static_for<NS>([&](auto J) {
                constexpr int j = decltype(J)::value;
                load(chunk + j + NS - 1, std::integral_constant<int, (j + NS - 1) % NS>{});
                if (chunk + j < c_end) mul(J);
            });
    }
    // accumulator (i, t): rows = channels 32 i + (r & 3) + 8 (r >> 2) + 4 kh, column = l31 + 32 t
    float *o = part + (size_t)gw * (ST_K * ST_COLS);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int t = 0; t < 5; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) o[(32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh) * ST_COLS + l31 + 32 * t] = acc[i][t][r];
}

// dW[k][c][r][s] (KCRS: k * 147 + j) = sum over waves of part[w][k][j] in a fixed order: a workgroup owns 64 outputs, its four
// waves take every fourth partial (eight loads in flight per thread), the four sums are added in wave order
__global__ void __launch_bounds__(256)
st_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dw, int nwaves) {
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), sg = threadIdx.x >> 6;
    const bool ok = i < ST_K * 147;
    const int k = ok ? i / 147 : 0, j = ok ? i - 147 * k : 0;
    const float *p = part + (size_t)k * ST_COLS + j;
    constexpr size_t WS = (size_t)ST_K * ST_COLS;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int w = sg;
    for (; w + 28 < nwaves; w += 32) {
#pragma unroll
        for (int u = 0; u < 8; u++) s[u] += p[(size_t)(w + 4 * u) * WS];
    }
    for (; w < nwaves; w += 4) s[0] += p[(size_t)w * WS];
    __shared__ float sh[4][64];
    sh[sg][threadIdx.x & 63] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (sg == 0 && ok) dw[i] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// =====================================================================================================================
// The same two kernels in exact fp32 (v_mfma_f32_32x32x2_f32: a lane supplies ONE value per operand, k = lane >> 5), for the
// fp32 storage mode.  The planes are fp32; a lane loads 4 consecutive elements (16 bytes) and feeds them to 4 MFMA steps:
//   forward:  for one (c, r), half-wave 0 holds the taps s = 0, 2, 4, 6 (plane pw = 1), half-wave 1 the taps (-), 1, 3, 5 (plane
//             pw = 0): step i multiplies k = {s = 2 i, s = 2 i - 1}; the weights of a step come from LDS (one ds_read_b32)
//   wgrad:    half-wave h holds pixels 4 h .. 4 h + 3 of an 8-pixel chunk of dY / of the patch: step i multiplies the pixel
//             pair (i, 4 + i)
template <int DUMMY>
__global__ void __launch_bounds__(256)
st32_pad_kernel(const float *__restrict__ x, float *__restrict__ xp, const StArgs g, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const uint32_t hp = (uint32_t)g.pitch;
        const size_t row = e / hp;
        const int j = (int)(e - row * hp);
        const int ii = (int)(row % (uint32_t)g.R);
        size_t q = row / (uint32_t)g.R;
        const int pw = (int)(q & 1); q >>= 1;
        const int ph = (int)(q & 1); q >>= 1;
        const int y = 2 * ii + ph - 4, x0 = 2 * j + pw - 4;
        float v = 0.f;
        if (y >= 0 && y < g.H && x0 >= 0 && x0 < g.H) v = x[(q * g.H + y) * (size_t)g.H + x0];
        xp[e] = v;
    }
}
// weights KCRS -> LDS order [g = 7 c + r][step i][k half][64 channels]: half 0: s = 2 i, half 1: s = 2 i - 1 (i = 0: 0)
__global__ void __launch_bounds__(256)
st32_wt_kernel(const float *__restrict__ w, float *__restrict__ wl) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 21 * 4 * 2 * ST_K) return;
    const int ch = idx & 63, kh = (idx >> 6) & 1, i = (idx >> 7) & 3, gq = idx >> 9;
    const int s = kh ? 2 * i - 1 : 2 * i;
    wl[idx] = s >= 0 ? w[(size_t)ch * 147 + gq * 7 + s] : 0.f;
}
__global__ void __launch_bounds__(256)
st32_fwd_kernel(const float *__restrict__ xp, const float *__restrict__ wl, float *__restrict__ y, const StArgs g, int ntiles, float *__restrict__ bn_part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char st_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    constexpr int PITCH = 32 * 4 + 16;
    constexpr int WL = 21 * 4 * 2 * ST_K;            // floats
    float *ws = (float *)st_smem;                     // [21][4][2][64]
    unsigned char *img = st_smem + WL * 4 + wave * (64 * PITCH);
    for (int i = threadIdx.x; i < WL; i += 256) ws[i] = wl[i];
    __syncthreads();
    uint32_t goff[21];
#pragma unroll
    for (int gq = 0; gq < 21; gq++) {
        const int c = gq / 7, r = gq - 7 * c;
        const int ph = (r + 1) & 1, di = (r + 1) >> 1;
        goff[gq] = (uint32_t)((((c * 2 + ph) * 2 + (kh ? 0 : 1)) * g.R + di) * g.pitch);   // half 0 reads plane pw = 1, half 1 plane pw = 0
    }
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int per = (ntiles + nw - 1) / nw, t_end = min(ntiles, (gw + 1) * per);
    StStat stat[2] = {{0.f, 0.f, 0.f, 0}, {0.f, 0.f, 0.f, 0}};
    for (int tile = gw * per; tile < t_end; tile++) {
        const uint32_t n = fd_div((uint32_t)tile, g.fdTpi);
        const uint32_t p0 = ((uint32_t)tile - n * g.fdTpi.d) * 32u;
        const uint32_t p = p0 + (uint32_t)l31;
        const uint32_t ho = fd_div(p, g.fdWo), wo = p - ho * (uint32_t)g.Wo;
        const float *base = xp + (size_t)n * g.img + (size_t)ho * g.pitch + wo;
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
        typedef pf4 __attribute__((aligned(4))) pf4_u;
        pf4 a[21];
#pragma unroll
        for (int gq = 0; gq < 21; gq++) a[gq] = *(const pf4_u *)(base + goff[gq]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gq = 0; gq < 21; gq++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float *wp = ws + ((gq * 4 + i) * 2 + kh) * ST_K + l31;
#pragma unroll
                for (int t = 0; t < 2; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gq][i], wp[32 * t], acc[t], 0, 0, 0);
            }
        if (bn_part) { st_stat_add(stat[0], acc[0]); st_stat_add(stat[1], acc[1]); }
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                pf4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
                *(pf4 *)(img + (t * 32 + l31) * PITCH + (8 * q + 4 * kh) * 4) = v;
            }
        const int c4 = lane & 7, r0 = lane >> 3;
        float *dst = y + ((size_t)n * ST_K) * g.P + p0 + 4 * c4;
#pragma unroll
        for (int ps = 0; ps < 8; ps++) {
            const int ch = ps * 8 + r0;
            const pf4 v = *(const pf4 *)(img + ch * PITCH + c4 * 16);
            *(pf4 *)(dst + (size_t)ch * g.P) = v;
        }
    }
    if (bn_part) { st_stat_write(stat[0], bn_part, nw, gw, l31, kh); st_stat_write(stat[1], bn_part, nw, gw, 32 + l31, kh); }
}
// NS register sets of operands in a ring: the loads of chunks j + 1 .. j + NS - 1 are in flight while chunk j is multiplied (one
// wave per SIMD; a chunk's 40 MFMAs are 640 cycles, HBM latency several thousand).  Every load is unconditional -- past the
// wave's range it re-reads the last chunk -- so that the compiler's s_waitcnt before a multiply counts exactly the younger sets.
__global__ void __launch_bounds__(256)
st32_wgrad_kernel(const float *__restrict__ xp, const float *__restrict__ dy, float *__restrict__ part, const StArgs g, int nchunks) {
    constexpr int NS = 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    uint32_t coff[5];
#pragma unroll
    for (int t = 0; t < 5; t++) {
        int j = l31 + 32 * t;
        if (j > 146) j = 146;
        const int c = j / 49, rs = j - 49 * c, r = rs / 7, s = rs - 7 * r;
        const int ph = (r + 1) & 1, di = (r + 1) >> 1, pw = (s + 1) & 1, dj = (s + 1) >> 1;
        coff[t] = (uint32_t)((((c * 2 + ph) * 2 + pw) * g.R + di) * g.pitch + dj + 4 * kh);
    }
    f32x16 acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int t = 0; t < 5; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][t][r] = 0.f;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int per = (nchunks + nw - 1) / nw, c_beg = gw * per, c_end = min(nchunks, (gw + 1) * per);
    typedef pf4 __attribute__((aligned(4))) pf4_u;
    pf4 a[NS][2], b[NS][5];
    if (c_beg < c_end) {
        auto load = [&](int chunk, auto set_tag) {
            constexpr int S = decltype(set_tag)::value;
            chunk = min(chunk, c_end - 1);
            const uint32_t n = fd_div((uint32_t)chunk, g.fdTpi);
            const uint32_t p0 = ((uint32_t)chunk - n * g.fdTpi.d) * 8u;    // 8 consecutive pixels of one output row (Wo % 8 == 0)
            const uint32_t ho = fd_div(p0, g.fdWo), wo0 = p0 - ho * (uint32_t)g.Wo;
            const float *da = dy + ((size_t)n * ST_K + l31) * g.P + p0 + 4 * kh;
#pragma unroll
            for (int i = 0; i < 2; i++) a[S][i] = *(const pf4 *)(da + (size_t)(32 * i) * g.P);
            const float *xb = xp + (size_t)n * g.img + (size_t)ho * g.pitch + wo0;
#pragma unroll
            for (int t = 0; t < 5; t++) b[S][t] = *(const pf4_u *)(xb + coff[t]);
        };
        auto mul = [&](auto set_tag) {
            constexpr int S = decltype(set_tag)::value;
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int t = 0; t < 5; t++) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S][i][e], b[S][t][e], acc[i][t], 0, 0, 0);
        };
        st_static_for<NS - 1>([&](auto J) { load(c_beg + decltype(J)::value, J); });
        for (int chunk = c_beg; chunk < c_end; chunk += NS)
            st_static_for<NS>([&](auto J) {
                constexpr int j = decltype(J)::value;
                load(chunk + j + NS - 1, std::integral_constant<int, (j + NS - 1) % NS>{});
                if (chunk + j < c_end) mul(J);
            });
    }
    float *o = part + (size_t)gw * (ST_K * ST_COLS);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int t = 0; t < 5; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) o[(32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh) * ST_COLS + l31 + 32 * t] = acc[i][t][r];
}

// ---- host ----
static int st_geometry(StArgs &g, int N, int H) {
    g.N = N; g.H = H; g.Ho = H / 2; g.Wo = H / 2; g.P = g.Ho * g.Wo;
    g.R = g.Ho + 3; g.pitch = st_pitch(g.Wo);
    g.img = (size_t)3 * 4 * g.R * g.pitch;
    g.fdWo = make_fastdiv(g.Wo);
    return 0;
}
// where the forward leaves its statistics partials (one row per wave), or NULL
static float *st_parts(mid_bn_parts *parts, int waves) {
    if (!parts) return nullptr;
    parts->nparts = 0;
    if (!parts->buf || parts->floats < (size_t)3 * waves * ST_K) return nullptr;
    parts->nparts = waves;
    return parts->buf;
}
static int st_waves(long units, int cap = ST_WAVES) { // waves that share the work: at most cap, a multiple of 4
    long w = units < cap ? units : cap;
    w = (w + 3) / 4 * 4;
    return (int)w;
}

extern "C" {
/* the shapes these kernels cover: 3 input channels, 64 filters, 7x7, stride 2, output rows a multiple of 16 pixels */
int mid_stem_bf16_supported(int C, int H, int K, int k, int stride) {
    return C == 3 && K == ST_K && k == 7 && stride == 2 && H >= 32 && (H / 2) % 16 == 0 && H % 2 == 0;
}
size_t mid_stem_bf16_xp_bytes(int N, int H) {
    StArgs g; st_geometry(g, N, H);
    return (size_t)N * g.img * 2 + 64; /* (+ slack: a 16-byte load may start 3 elements before the end of the last row) */
}
size_t mid_stem_bf16_part_floats(int N, int H) {
    (void)N; (void)H;
    return (size_t)ST_WAVES * ST_K * ST_COLS + ST_K * ST_KRED; /* wave partials; the forward's bf16 weights sit behind them */
}
/* y (fp32, or bf16 with y_dt = MID_BF16) = conv7x7s2(bf16(x), bf16(w)); leaves the padded parity planes of x in xp for the weight gradient */
int mid_stem_fwd_bf16(mid_stream s, const float *x, const float *w, void *y, int y_dt, void *xp, size_t xp_bytes, float *scratch, size_t scratch_floats,
                      int N, int H, mid_bn_parts *parts) {
    hipStream_t st = (hipStream_t)s;
    StArgs g; st_geometry(g, N, H);
    if (xp_bytes < mid_stem_bf16_xp_bytes(N, H) || scratch_floats < mid_stem_bf16_part_floats(N, H)) { mi_record_error("mid_stem_fwd_bf16", "workspace too small"); return -3; }
    const size_t pairs = (size_t)N * g.img / 2;
    size_t pb = (pairs + 255) / 256; if (pb > (1u << 20)) pb = 1u << 20;
    mi_prof_begin(st, MI_FAM_DCONV, 2.0 * N * g.P * ST_K * 147.0, 4.0 * (double)N * 3 * H * H + (y_dt == MID_BF16 ? 2.0 : 4.0) * (double)N * ST_K * g.P);
    hipLaunchKernelGGL(st_pad_kernel, dim3((unsigned)pb), dim3(256), 0, st, x, (u16 *)xp, g, pairs);
    u16 *wf = (u16 *)(scratch + (size_t)ST_WAVES * ST_K * ST_COLS);
    hipLaunchKernelGGL(st_wt_kernel, dim3((ST_K * ST_KRED + 255) / 256), dim3(256), 0, st, w, wf);
    const int ntiles = N * (g.P / 32);
    g.fdTpi = make_fastdiv(g.P / 32);
    const int waves = st_waves(ntiles);
    float *bn_part = st_parts(parts, waves);
    if (y_dt == MID_BF16) hipLaunchKernelGGL(st_fwd_kernel<true>, dim3(waves / 4), dim3(256), 4 * 64 * (32 * 4 + 16), st, (const u16 *)xp, wf, y, g, ntiles, bn_part);
    else hipLaunchKernelGGL(st_fwd_kernel<false>, dim3(waves / 4), dim3(256), 4 * 64 * (32 * 4 + 16), st, (const u16 *)xp, wf, y, g, ntiles, bn_part);
    mi_prof_end(st);
    MI_LAUNCH_CHECK("st_fwd_kernel");
    return 0;
}
/* the exact-fp32 pair: planes and operands fp32 (xp: mid_stem_f32_xp_bytes) */
size_t mid_stem_f32_xp_bytes(int N, int H) {
    StArgs g; st_geometry(g, N, H);
    return (size_t)N * g.img * 4 + 64;
}
int mid_stem_fwd_f32(mid_stream s, const float *x, const float *w, float *y, void *xp, size_t xp_bytes, float *scratch, size_t scratch_floats,
                     int N, int H, mid_bn_parts *parts) {
    hipStream_t st = (hipStream_t)s;
    StArgs g; st_geometry(g, N, H);
    if (xp_bytes < mid_stem_f32_xp_bytes(N, H) || scratch_floats < mid_stem_bf16_part_floats(N, H)) { mi_record_error("mid_stem_fwd_f32", "workspace too small"); return -3; }
    const size_t total = (size_t)N * g.img;
    size_t pb = (total + 255) / 256; if (pb > (1u << 20)) pb = 1u << 20;
    mi_prof_begin(st, MI_FAM_DCONV, 2.0 * N * g.P * ST_K * 147.0, 4.0 * ((double)N * 3 * H * H + (double)N * ST_K * g.P));
    hipLaunchKernelGGL(st32_pad_kernel<0>, dim3((unsigned)pb), dim3(256), 0, st, x, (float *)xp, g, total);
    float *wl = scratch + (size_t)ST_WAVES * ST_K * ST_COLS;   // 21 * 4 * 2 * 64 floats = 10752 <= ST_K * ST_KRED (11264)
    hipLaunchKernelGGL(st32_wt_kernel, dim3((21 * 4 * 2 * ST_K + 255) / 256), dim3(256), 0, st, w, wl);
    const int ntiles = N * (g.P / 32);
    g.fdTpi = make_fastdiv(g.P / 32);
    const int waves = st_waves(ntiles);
    const size_t lds = (size_t)21 * 4 * 2 * ST_K * 4 + 4 * 64 * (32 * 4 + 16);
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)st32_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            mi_record_error("st32_fwd_kernel", "cannot raise the dynamic LDS limit");
            return -1;
        }
        attr_set = 1;
    }
    float *bn_part = st_parts(parts, waves);
    hipLaunchKernelGGL(st32_fwd_kernel, dim3(waves / 4), dim3(256), lds, st, (const float *)xp, wl, y, g, ntiles, bn_part);
    mi_prof_end(st);
    MI_LAUNCH_CHECK("st32_fwd_kernel");
    return 0;
}
int mid_stem_wgrad_f32(mid_stream s, const void *xp, const float *dy, float *dw, float *scratch, size_t scratch_floats, int N, int H) {
    hipStream_t st = (hipStream_t)s;
    StArgs g; st_geometry(g, N, H);
    if (scratch_floats < mid_stem_bf16_part_floats(N, H)) { mi_record_error("mid_stem_wgrad_f32", "workspace too small"); return -3; }
    const int nchunks = N * (g.P / 8);
    g.fdTpi = make_fastdiv(g.P / 8);
    const int waves = st_waves(nchunks, ST_WAVES / 2); /* one wave per SIMD (register ring) */
    mi_prof_begin(st, MI_FAM_WGRAD, 2.0 * N * g.P * ST_K * 147.0, 4.0 * ((double)N * ST_K * g.P) + 4.0 * N * g.img);
    hipLaunchKernelGGL(st32_wgrad_kernel, dim3(waves / 4), dim3(256), 0, st, (const float *)xp, dy, scratch, g, nchunks);
    hipLaunchKernelGGL(st_wgrad_reduce_kernel, dim3((ST_K * 147 + 63) / 64), dim3(256), 0, st, scratch, dw, waves);
    mi_prof_end(st);
    MI_LAUNCH_CHECK("st32_wgrad_kernel");
    return 0;
}
/* dw (KCRS fp32) from the planes the forward left in xp and dy (bf16, or fp32 rounded to bf16 on the way in: the same operand bits) */
int mid_stem_wgrad_bf16(mid_stream s, const void *xp, const void *dy, int dy_dt, float *dw, float *scratch, size_t scratch_floats, int N, int H) {
    hipStream_t st = (hipStream_t)s;
    StArgs g; st_geometry(g, N, H);
    if (scratch_floats < mid_stem_bf16_part_floats(N, H)) { mi_record_error("mid_stem_wgrad_bf16", "workspace too small"); return -3; }
    const int nchunks = N * (g.P / 16);
    g.fdTpi = make_fastdiv(g.P / 16);
    const int waves = st_waves(nchunks, ST_WAVES / 2); /* one wave per SIMD (register ring) */
    mi_prof_begin(st, MI_FAM_WGRAD, 2.0 * N * g.P * ST_K * 147.0, (dy_dt == MID_BF16 ? 2.0 : 4.0) * ((double)N * ST_K * g.P) + 2.0 * N * g.img);
    if (dy_dt == MID_BF16) hipLaunchKernelGGL(st_wgrad_kernel<true>, dim3(waves / 4), dim3(256), 0, st, (const u16 *)xp, dy, scratch, g, nchunks);
    else hipLaunchKernelGGL(st_wgrad_kernel<false>, dim3(waves / 4), dim3(256), 0, st, (const u16 *)xp, dy, scratch, g, nchunks);
    hipLaunchKernelGGL(st_wgrad_reduce_kernel, dim3((ST_K * 147 + 63) / 64), dim3(256), 0, st, scratch, dw, waves);
    mi_prof_end(st);
    MI_LAUNCH_CHECK("st_wgrad_kernel");
    return 0;
}
}
