// kernels_igemm_bf16.hip -- the bf16-activation path (BASELINE configs[4]): convolutions as an im2col-free implicit
// GEMM on v_mfma_f32_32x32x16_bf16, fp32 accumulate.  Activations and activation gradients are stored as bf16 in the same
// NCHW tensors; parameters, parameter gradients, Adam state and BN statistics stay fp32 (the policy of
// resnet_cudnn_nchw.cu:1196-1211, TENSOR_OP_MATH_ALLOW_CONVERSION, with the storage made explicit).
//
//   forward  Y = W * im2col(X)      M = K, columns (n,ho,wo), reduction (tap, c); A = weights rounded to bf16 and re-laid
//                                   as k-step tiles [t][c/32][K][32] (one tile = one contiguous run of 16-byte loads)
//   dgrad    dX = W^T * dY          M = C, reduction (tap, k); S=2: four parity classes (blockIdx.y); A = [t][k/32][C][32]
//   wgrad    dW = dY * im2col(X)^T  M = K, columns (tap, 64-channel block) in pairs, reduction (n,ho,wo) split over
//                                   blockIdx.y; fp32 partials [split][t][k][c] reduced in a fixed order
//
// The MFMA operands want the REDUCTION index contiguous per lane (8 bf16 = one ds_read_b128): LDS images are
// [row or column][32 reduction elements], pitch 80 bytes (conflict-free ds_read_b128: 16 lanes x 20 dwords cover all 64
// banks).  In NCHW the channel -- the reduction index of forward and dgrad -- is the STRIDED dimension, so the gathered
// operand is transposed on the way into LDS: a thread gathers 16 channels of ONE pixel (2-byte loads, coalesced along the
// pixels of a wave), packs them in registers and writes two 16-byte rows pieces.  wgrad reduces over pixels (contiguous),
// so both its operands are gathered along the reduction and go to LDS element-wise.
// Epilogue of forward / dgrad: accumulators are transposed through LDS so that a lane owns one output channel and 32
// consecutive pixels -> 16-byte (8 x bf16) stores, and the forward pass leaves per-tile batch-norm statistics (count, mean,
// M2, computed from the fp32 accumulators before rounding) exactly as the fp32 kernel does.
#include "mi_common.hpp"
#include "mi_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float pf4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

enum { BG_FWD = 0, BG_DGRAD = 1, BG_WGRAD = 2 };
#define BG_BK 32
#define BG_LDB 80 /* bytes per LDS row: 32 bf16 + 16 bytes of padding */

struct BgArgs {
    int N, C, K, H, W, Ho, Wo; // KS x KS, stride S, pad KS/2
    int HW, P;
    int ncols;                 // fwd/dgrad: N*P columns
    int mtiles, tiles;
    int nhalf, cb64;           // wgrad: 64-column halves = T * C/64; C/64
    int klen;                  // wgrad: reduction length per split (multiple of 32)
    FastDiv fdP, fdWo, fdM, fdCb;
    float *bn_part;            // forward: statistics partials, three planes [bn_np][K]
    int bn_np;
    int vw;                    // fwd / dgrad-s1 epilogue: pixels per store (8, 4 or 1)
};

__device__ __forceinline__ float bg_bf2f(u16 v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint32_t bg_pack2(float a, float b) { // round to nearest even, NaN stays NaN (v_cvt_pk_bf16_f32)
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return *(uint32_t *)&r;
}
__device__ __forceinline__ u16 bg_f2bf(float a) { return (u16)(bg_pack2(a, 0.f) & 0xffffu); }

template <int MODE, int KS, int S, int WMW>
__global__ void __launch_bounds__(256)
bgemm_kernel(const u16 *__restrict__ Aop, const u16 *__restrict__ Bop, void *__restrict__ OutV, const u16 *__restrict__ addend,
             const BgArgs g) {
    constexpr int BM = 64 * WMW;
    constexpr int TN = WMW == 2 ? 2 : 1;
    constexpr int WNC = 32 * TN;
    constexpr int T = KS * KS, PAD = KS / 2;
    constexpr int NA = BM / 64;   // fwd/dgrad: 16-byte loads of A per thread and tile
    constexpr int NAS = BM / 8;   // wgrad: 2-byte loads of A per thread and tile
    extern __shared__ __attribute__((aligned(16))) unsigned char bg_smem[];
    unsigned char *As = bg_smem;                         // [2][BM][80 B]
    unsigned char *Bs = bg_smem + 2 * BM * BG_LDB;       // [2][128][80 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WMW == 2 ? wave >> 1 : 0, wn = WMW == 2 ? wave & 1 : wave;

    // ---- block -> tile (XCD-contiguous, M-tiles fastest: the blocks that share a gathered pixel tile share an L2) ----
    uint32_t L = blockIdx.x;
    {
        const uint32_t per = (uint32_t)g.tiles >> 3;
        if (L < per * 8) L = (L & 7) * per + (L >> 3);
    }
    const uint32_t ct = fd_div(L, g.fdM);
    const int m0 = (int)(L - ct * g.mtiles) * BM;
    const int n0 = (int)ct * 128;

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // ---- per-thread staging state (byte offsets; every tensor is < 2^32 bytes: mi_bgemm_supported) ----
    uint32_t b_lane = 0;
    uint32_t mask = 0;
    int ph = 0, pw = 0, ntaps = 1;
    int ntiles = 0, kend = 0;
    int hf_r[2] = {0, 0}, hf_s[2] = {0, 0}, hf_ok[2] = {0, 0};
    uint32_t hf_c[2] = {0, 0};
    if (MODE == BG_FWD || MODE == BG_DGRAD) {
        const int j = n0 + (tid & 127);
        const bool jin = j < g.ncols;
        const uint32_t jc = jin ? (uint32_t)j : (uint32_t)g.ncols - 1;
        const uint32_t n = fd_div(jc, g.fdP);
        const uint32_t p = jc - n * g.P;
        const uint32_t ho = fd_div(p, g.fdWo), wo = p - ho * g.Wo;
        const int b_k = 16 * (tid >> 7);
        if (MODE == BG_FWD) {
            b_lane = (uint32_t)(n * g.C * g.HW + (S * ho) * g.W + S * wo + b_k * g.HW) * 2u;
#pragma unroll
            for (int t = 0; t < T; t++) {
                const int hi = S * (int)ho - PAD + t / KS, wi = S * (int)wo - PAD + t % KS;
                if (jin && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) mask |= 1u << t;
            }
            ntaps = T;
            ntiles = T * (g.C / BG_BK);
        } else if (S == 1) {
            b_lane = (uint32_t)(n * g.K * g.P + p + b_k * g.P) * 2u;
#pragma unroll
            for (int t = 0; t < T; t++) {
                const int hs = (int)ho + PAD - t / KS, ws = (int)wo + PAD - t % KS;
                if (jin && hs >= 0 && hs < g.Ho && ws >= 0 && ws < g.Wo) mask |= 1u << t;
            }
            ntaps = T;
            ntiles = T * (g.K / BG_BK);
        } else {
            const int cls = 3 - (int)blockIdx.y; // heaviest class (4 taps) first
            ph = cls >> 1; pw = cls & 1;
            const int ntw = pw ? 2 : 1;
            ntaps = (ph ? 2 : 1) * ntw;
            b_lane = (uint32_t)(n * g.K * g.P + p + b_k * g.P) * 2u;
            for (int tt = 0; tt < ntaps; tt++) {
                const int th = tt / ntw, tw = tt - th * ntw;
                const int dh = (ph && th == 0) ? 1 : 0, dw = (pw && tw == 0) ? 1 : 0;
                if (jin && (int)ho + dh < g.Ho && (int)wo + dw < g.Wo) mask |= 1u << tt;
            }
            ntiles = ntaps * (g.K / BG_BK);
        }
    } else {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t u = 2u * ct + h;
            const uint32_t tp = fd_div(u, g.fdCb);
            hf_ok[h] = u < (uint32_t)g.nhalf;
            const int tap = hf_ok[h] ? (int)tp : 0;
            hf_r[h] = tap / KS; hf_s[h] = tap - KS * hf_r[h];
            hf_c[h] = hf_ok[h] ? (u - tp * g.cb64) * 64u : 0u;
        }
        const int kbeg = (int)blockIdx.y * g.klen;
        kend = min(g.N * g.P, kbeg + g.klen);
        ntiles = (kend - kbeg + BG_BK - 1) / BG_BK;
    }

    // registers of the tile in flight
    u32x4 ra4[NA];
    u16 ra[NAS], rb[16];
    int sel_a = 1, sel_b = 1;
    int ld_t = 0, ld_c0 = 0;
    int ld_k0 = (MODE == BG_WGRAD) ? (int)blockIdx.y * g.klen : 0;
    auto ldg16 = [](const char *ubase, uint32_t lane_off) -> u16 { return *(const u16 *)(ubase + lane_off); };

    auto fetch = [&]() {
        if (MODE == BG_FWD || MODE == BG_DGRAD) {
            const int t = min(ld_t, ntaps - 1); // the two drain iterations re-read the last tap
            sel_b = (mask >> t) & 1;
            const char *fa, *fb;
            uint32_t fb_lane;
            size_t bstride;
            if (MODE == BG_FWD) {
                const int r = KS == 3 ? (t * 11) >> 5 : 0, s = t - KS * r;
                fa = (const char *)(Aop + ((size_t)(t * (g.C / BG_BK) + ld_c0 / BG_BK) * g.K + m0) * BG_BK);
                fb = (const char *)(Bop + (size_t)ld_c0 * g.HW);
                fb_lane = b_lane + (uint32_t)(sel_b * ((r - PAD) * g.W + (s - PAD)) * 2); // outside: centre pixel, stored as 0
                bstride = (size_t)g.HW * 2;
                ld_c0 += BG_BK;
                if (ld_c0 == g.C) { ld_c0 = 0; ld_t++; }
            } else {
                int wt, doff;
                if (S == 1) {
                    const int r = KS == 3 ? (t * 11) >> 5 : 0, s = t - KS * r;
                    wt = t;
                    doff = (PAD - r) * g.Wo + (PAD - s);
                } else {
                    const int th = pw ? t >> 1 : t, tw = pw ? t & 1 : 0;
                    const int r = ph ? 2 * th : 1, s = pw ? 2 * tw : 1;
                    wt = 3 * r + s;
                    doff = (ph & (th ^ 1)) * g.Wo + (pw & (tw ^ 1));
                }
                fa = (const char *)(Aop + ((size_t)(wt * (g.K / BG_BK) + ld_c0 / BG_BK) * g.C + m0) * BG_BK);
                fb = (const char *)(Bop + (size_t)ld_c0 * g.P);
                fb_lane = b_lane + (uint32_t)(sel_b * doff * 2);
                bstride = (size_t)g.P * 2;
                ld_c0 += BG_BK;
                if (ld_c0 == g.K) { ld_c0 = 0; ld_t++; }
            }
#pragma unroll
            for (int q = 0; q < 16; q++) rb[q] = ldg16(fb + (size_t)q * bstride, fb_lane);
#pragma unroll
            for (int q = 0; q < NA; q++) ra4[q] = *(const u32x4 *)(fa + (size_t)(tid + 256 * q) * 16);
        } else {
            const int kk = ld_k0 + (tid & 31);
            sel_a = kk < kend;
            const uint32_t kc = sel_a ? (uint32_t)kk : (uint32_t)kend - 1;
            const uint32_t n = fd_div_ge2(kc, g.fdP);
            const uint32_t pp = kc - n * g.P;
            const uint32_t ho = fd_div_ge2(pp, g.fdWo), wo = pp - ho * g.Wo;
            const uint32_t row = (uint32_t)tid >> 5;
            const char *fa = (const char *)(Aop + (size_t)m0 * g.P);
            const uint32_t fa_lane = (n * (uint32_t)(g.K * g.P) + pp + row * g.P) * 2u;
#pragma unroll
            for (int q = 0; q < NAS; q++) ra[q] = ldg16(fa + (size_t)(8 * q) * g.P * 2, fa_lane);
            sel_b = 0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int hi = S * (int)ho - PAD + hf_r[h], wi = S * (int)wo - PAD + hf_s[h];
                const int inb = (int)((uint32_t)hi < (uint32_t)g.H) & (int)((uint32_t)wi < (uint32_t)g.W);
                const uint32_t pix = inb ? (uint32_t)(hi * g.W + wi) : (S * ho) * g.W + S * wo;
                const uint32_t fb_lane = (n * (uint32_t)(g.C * g.HW) + pix + (row + hf_c[h]) * g.HW) * 2u;
                sel_b |= (sel_a & inb & hf_ok[h]) << h;
#pragma unroll
                for (int q = 0; q < 8; q++) rb[8 * h + q] = ldg16((const char *)Bop + (size_t)(8 * q) * g.HW * 2, fb_lane);
            }
            ld_k0 += BG_BK;
        }
    };
    auto stash = [&](const int buf) {
        unsigned char *as = As + buf * (BM * BG_LDB), *bs = Bs + buf * (128 * BG_LDB);
        if (MODE == BG_WGRAD) {
            const int kx = tid & 31, row = tid >> 5;
#pragma unroll
            for (int q = 0; q < NAS; q++) *(u16 *)(as + (row + 8 * q) * BG_LDB + kx * 2) = sel_a ? ra[q] : (u16)0;
#pragma unroll
            for (int q = 0; q < 16; q++)
                *(u16 *)(bs + (row + 8 * q) * BG_LDB + kx * 2) = ((sel_b >> (q >> 3)) & 1) ? rb[q] : (u16)0;
        } else {
            const int bj = tid & 127, kh = tid >> 7;
            u32x4 lo, hi;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                lo[q] = sel_b ? ((uint32_t)rb[2 * q] | ((uint32_t)rb[2 * q + 1] << 16)) : 0u;
                hi[q] = sel_b ? ((uint32_t)rb[8 + 2 * q] | ((uint32_t)rb[8 + 2 * q + 1] << 16)) : 0u;
            }
            *(u32x4 *)(bs + bj * BG_LDB + kh * 32) = lo;
            *(u32x4 *)(bs + bj * BG_LDB + kh * 32 + 16) = hi;
#pragma unroll
            for (int q = 0; q < NA; q++) {
                const int u = tid + 256 * q;
                *(u32x4 *)(as + (u >> 2) * BG_LDB + (u & 3) * 16) = ra4[q];
            }
        }
    };

    const int fr = lane & 31, fk = lane >> 5;
    fetch();
    stash(0);
    fetch();
    __syncthreads();
    for (int it = 0; it < ntiles; it++) {
        const int buf = it & 1;
        stash(buf ^ 1); // tile it+1 (held in registers) -> the other buffer; then the registers take tile it+2
        fetch();
        const unsigned char *as = As + buf * (BM * BG_LDB) + (wm * 64 + fr) * BG_LDB + fk * 16;
        const unsigned char *bs = Bs + buf * (128 * BG_LDB) + (wn * WNC + fr) * BG_LDB + fk * 16;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            bf16x8 av[2], bv[TN];
#pragma unroll
            for (int i = 0; i < 2; i++) av[i] = *(const bf16x8 *)(as + i * 32 * BG_LDB + s * 32);
#pragma unroll
            for (int j = 0; j < TN; j++) bv[j] = *(const bf16x8 *)(bs + j * 32 * BG_LDB + s * 32);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: accumulator layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ----
    if (MODE == BG_WGRAD) {
        float *Out = (float *)OutV;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int lc = wn * WNC + j * 32 + (lane & 31);
            const int h = lc >> 6; // wave-uniform; selects (no runtime-indexed arrays: they would live in scratch)
            if (!(h ? hf_ok[1] : hf_ok[0])) continue;
            const int tap = h ? hf_r[1] * KS + hf_s[1] : hf_r[0] * KS + hf_s[0];
            const size_t coff = ((size_t)((size_t)blockIdx.y * T + tap) * g.K) * g.C + (h ? hf_c[1] : hf_c[0]) + (lc & 63);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    Out[coff + (size_t)row * g.C] = acc[i][j][r];
                }
        }
        return;
    }
    u16 *Out = (u16 *)OutV;
    const int Mdim = MODE == BG_FWD ? g.K : g.C;
    const int Pout = MODE == BG_FWD ? g.P : g.HW;
    const bool wide = g.vw > 1 && !(MODE == BG_DGRAD && S == 2);
    const bool stats = MODE == BG_FWD && g.bn_part != nullptr;
    if (wide || stats) {
        // each wave transposes 64 rows x 32 columns at a time through its own 8 KB of LDS (the tiles are dead after the
        // loop's last barrier): lane = row then owns 32 consecutive columns
        float *tw = (float *)bg_smem + wave * (32 * 64);
        const int l31 = lane & 31, row = lane;
        const int rowg = m0 + wm * 64 + row;
        const int nvw = min(WNC, max(0, g.ncols - (n0 + wn * WNC)));
        float s0 = 0.f, sd = 0.f, sq = 0.f;
#pragma unroll
        for (int j = 0; j < TN; j++) {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    const int row4 = i * 8 + rg * 2 + (lane >> 5);
                    pf4 q4 = {acc[i][j][rg * 4 + 0], acc[i][j][rg * 4 + 1], acc[i][j][rg * 4 + 2], acc[i][j][rg * 4 + 3]};
                    *(pf4 *)(tw + l31 * 64 + ((row4 ^ (l31 & 15)) << 2)) = q4;
                }
            float v[32];
#pragma unroll
            for (int c = 0; c < 32; c++) v[c] = tw[c * 64 + ((((row >> 2) ^ (c & 15))) << 2) + (row & 3)];
            if (stats) {
                if (j == 0) s0 = v[0];
                const int nvj = min(32, max(0, nvw - 32 * j));
#pragma unroll
                for (int c = 0; c < 32; c++) {
                    const float d = c < nvj ? v[c] - s0 : 0.f;
                    sd += d;
                    sq = fmaf(d, d, sq);
                }
            }
            if (wide) {
                const int colb = n0 + wn * WNC + j * 32;
                if (g.vw == 8) {
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        const int col = colb + 8 * gq;
                        if (col < g.ncols) {
                            const uint32_t n = fd_div((uint32_t)col, g.fdP);
                            const uint32_t p = (uint32_t)col - n * g.P;
                            const size_t o = ((size_t)n * Mdim + rowg) * Pout + p;
                            float w8[8];
#pragma unroll
                            for (int e = 0; e < 8; e++) w8[e] = v[8 * gq + e];
                            if (MODE == BG_DGRAD && addend) {
                                const u32x4 ad = *(const u32x4 *)(addend + o);
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    w8[2 * e] += __uint_as_float(ad[e] << 16);
                                    w8[2 * e + 1] += __uint_as_float(ad[e] & 0xffff0000u);
                                }
                            }
                            u32x4 pk;
#pragma unroll
                            for (int e = 0; e < 4; e++) pk[e] = bg_pack2(w8[2 * e], w8[2 * e + 1]);
                            *(u32x4 *)(Out + o) = pk;
                        }
                    }
                } else { // 4 pixels per store
#pragma unroll
                    for (int gq = 0; gq < 8; gq++) {
                        const int col = colb + 4 * gq;
                        if (col < g.ncols) {
                            const uint32_t n = fd_div((uint32_t)col, g.fdP);
                            const uint32_t p = (uint32_t)col - n * g.P;
                            const size_t o = ((size_t)n * Mdim + rowg) * Pout + p;
                            float w4[4];
#pragma unroll
                            for (int e = 0; e < 4; e++) w4[e] = v[4 * gq + e];
                            if (MODE == BG_DGRAD && addend) {
                                const u32x2 ad = *(const u32x2 *)(addend + o);
#pragma unroll
                                for (int e = 0; e < 2; e++) {
                                    w4[2 * e] += __uint_as_float(ad[e] << 16);
                                    w4[2 * e + 1] += __uint_as_float(ad[e] & 0xffff0000u);
                                }
                            }
                            u32x2 pk = {bg_pack2(w4[0], w4[1]), bg_pack2(w4[2], w4[3])};
                            *(u32x2 *)(Out + o) = pk;
                        }
                    }
                }
            }
        }
        if (stats) {
            constexpr int PPT = 4 / WMW;
            const float inv = nvw > 0 ? 1.0f / (float)nvw : 0.f;
            const size_t plane = (size_t)g.bn_np * g.K;
            const size_t o = (size_t)(ct * PPT + wn) * g.K + rowg;
            g.bn_part[o] = (float)nvw;
            g.bn_part[plane + o] = nvw > 0 ? s0 + sd * inv : 0.f;
            g.bn_part[2 * plane + o] = fmaxf(sq - sd * sd * inv, 0.f);
        }
    }
    if (!wide) {
        // 2-byte stores straight from the accumulators (planes whose size is not a multiple of 4, stride-2 dgrad)
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int col = n0 + wn * WNC + j * 32 + (lane & 31);
            if (col >= g.ncols) continue;
            const uint32_t n = fd_div((uint32_t)col, g.fdP);
            const uint32_t p = (uint32_t)col - n * g.P;
            size_t coff;
            if (MODE == BG_FWD) coff = (size_t)n * g.K * g.P + p;
            else if (S == 1) coff = (size_t)n * g.C * g.HW + p;
            else {
                const uint32_t a = fd_div(p, g.fdWo), b = p - a * g.Wo;
                coff = (size_t)n * g.C * g.HW + (size_t)(2 * a + ph) * g.W + 2 * b + pw;
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                float ad[16];
                if (MODE == BG_DGRAD && addend) {
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        ad[r] = bg_bf2f(addend[coff + (size_t)(m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * Pout]);
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float v = acc[i][j][r];
                    if (MODE == BG_DGRAD && addend) v += ad[r];
                    Out[coff + (size_t)row * Pout] = bg_f2bf(v);
                }
            }
        }
    }
}

// weights KCRS fp32 -> bf16 k-step tiles.  fwd: [t][c/32][K][32] (inner = c); dgrad: [t][k/32][C][32] (inner = k).
// One launch for MANY layers (table in device memory, as igemm_wt_all_kernel): block = one 32 x 32 (k, c) tile of one layer.
__global__ void __launch_bounds__(256)
bg_wt_all_kernel(const mid_wt_entry *__restrict__ entries, const int *__restrict__ tile_entry) {
    __shared__ float tile[32][9 * 32 + 1];
    const mid_wt_entry E = entries[tile_entry[blockIdx.x]];
    const int tl = (int)blockIdx.x - E.tile0, ctiles = E.C / 32, T = E.T;
    const int c0 = (tl % ctiles) * 32, k0 = (tl / ctiles) * 32, row = T * 32;
    for (int e = threadIdx.x; e < 32 * row; e += 256) {
        const int kr = e / row, x = e - kr * row;
        tile[kr][x] = E.w[((size_t)(k0 + kr) * E.C + c0) * T + x];
    }
    __syncthreads();
    u16 *fw = (u16 *)E.fwd, *dg = (u16 *)E.dgrad;
    for (int e = threadIdx.x; e < T * 1024; e += 256) {
        const int t = e >> 10, u = (e >> 5) & 31, v = e & 31; // v fastest = contiguous output dim
        if (fw) fw[((size_t)(t * (E.C / 32) + c0 / 32) * E.K + k0 + u) * 32 + v] = bg_f2bf(tile[u][v * T + t]);
        if (dg) dg[((size_t)(t * (E.K / 32) + k0 / 32) * E.C + c0 + u) * 32 + v] = bg_f2bf(tile[v][u * T + t]);
    }
}
// the same for one layer (operator layer / no table): grid (C/32, K/32)
template <int T>
__global__ void __launch_bounds__(256)
bg_wt_kernel(const float *__restrict__ w, u16 *__restrict__ fw, u16 *__restrict__ dg, int K, int C) {
    __shared__ float tile[32][T * 32 + 1];
    const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int e = threadIdx.x; e < 32 * T * 32; e += 256) {
        const int kr = e / (T * 32), x = e - kr * (T * 32);
        tile[kr][x] = w[((size_t)(k0 + kr) * C + c0) * T + x];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T * 1024; e += 256) {
        const int t = e >> 10, u = (e >> 5) & 31, v = e & 31;
        if (fw) fw[((size_t)(t * (C / 32) + c0 / 32) * K + k0 + u) * 32 + v] = bg_f2bf(tile[u][v * T + t]);
        if (dg) dg[((size_t)(t * (K / 32) + k0 / 32) * C + c0 + u) * 32 + v] = bg_f2bf(tile[v][u * T + t]);
    }
}

// fp32 <-> bf16 tensors (operator layer, tests)
__global__ void __launch_bounds__(256) bg_f2b_kernel(const float *__restrict__ in, u16 *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = bg_f2bf(in[i]);
}
__global__ void __launch_bounds__(256) bg_b2f_kernel(const u16 *__restrict__ in, float *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = bg_bf2f(in[i]);
}

// ------------------------------------------------------------------------------------------------------------
enum { BGOP_FWD = 0, BGOP_DGRAD = 1, BGOP_WGRAD = 2 };
int mi_bgemm_supported(int op, int N, int C, int H, int K, int k, int stride) {
    if (!((k == 1 && stride == 1) || (k == 3 && (stride == 1 || stride == 2)))) return 0;
    if (H % stride || H / stride < 2) return 0;
    if ((double)N * C * H * H >= 2147483648.0 || (double)N * K * (H / stride) * (H / stride) >= 2147483648.0) return 0; /* 32-bit byte offsets */
    if (op == BGOP_FWD) return C % 32 == 0 && K % 64 == 0;
    if (op == BGOP_DGRAD) return K % 32 == 0 && C % 64 == 0;
    return C % 64 == 0 && K % 64 == 0;
}
#define BG_SLOTS 768 /* 256 CUs x 3 resident workgroups (40 KB of LDS, <= 128 VGPRs) */
static int bgemm_wgrad_splits(int N, int C, int H, int K, int k, int stride) {
    const int bm = K % 128 == 0 ? 128 : 64;
    const long nhalf = (long)k * k * (C / 64);
    const long tiles = ((nhalf + 1) / 2) * (K / bm);
    const long ksteps = ((long)N * (H / stride) * (H / stride) + BG_BK - 1) / BG_BK;
    int best = 1;
    double best_eff = 0;
    for (int s = 1; s <= 512; s++) {
        if (s > 1 && ksteps / s < 64) break;
        const double waves = (double)tiles * s / BG_SLOTS;
        const double eff = waves / (double)((long)((tiles * s + BG_SLOTS - 1) / BG_SLOTS));
        if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
    }
    return best;
}
size_t mi_bgemm_part_floats(int N, int C, int H, int K, int k, int stride) {
    return (size_t)bgemm_wgrad_splits(N, C, H, K, k, stride) * k * k * K * C;
}

static void bgemm_geometry(BgArgs &g, int N, int C, int H, int K, int stride) {
    g.N = N; g.C = C; g.K = K; g.H = H; g.W = H; g.Ho = H / stride; g.Wo = H / stride;
    g.HW = H * H; g.P = g.Ho * g.Wo;
    g.ncols = N * g.P;
    g.fdP = make_fastdiv(g.P); g.fdWo = make_fastdiv(g.Wo);
    g.fdCb = make_fastdiv(1);
}
template <int MODE, int KS, int S, int WMW>
static int bgemm_launch_t(hipStream_t st, dim3 grid, const u16 *A, const u16 *B, void *out, const u16 *addend, const BgArgs &g) {
    constexpr int BM = 64 * WMW;
    constexpr size_t tiles_b = (size_t)2 * (BM + 128) * BG_LDB, ep_b = (size_t)4 * 32 * 64 * 4;
    constexpr size_t lds = tiles_b > ep_b ? tiles_b : ep_b;
    hipLaunchKernelGGL((bgemm_kernel<MODE, KS, S, WMW>), grid, dim3(256), lds, st, A, B, out, addend, g);
    return 0;
}
template <int MODE>
static int bgemm_launch(hipStream_t st, dim3 grid, const u16 *A, const u16 *B, void *out, const u16 *addend, const BgArgs &g, int k,
                        int stride, int bm) {
#define BGL(KS_, S_)                                                                              \
    if (k == KS_ && stride == S_)                                                                 \
        return bm == 128 ? bgemm_launch_t<MODE, KS_, S_, 2>(st, grid, A, B, out, addend, g)       \
                         : bgemm_launch_t<MODE, KS_, S_, 1>(st, grid, A, B, out, addend, g);
    BGL(1, 1) BGL(3, 1) BGL(3, 2)
#undef BGL
    return -2;
}
static int bgemm_fam(int k) { return k == 1 ? MI_FAM_GEMM : MI_FAM_PCONV; }
static int bgemm_pick_bm(int M, long coltiles) {
    if (M % 128) return 64;
    // 64-row tiles where 128-row ones would leave most of the chip empty (fewer than one workgroup per CU)
    return (long)(M / 128) * coltiles < 256 ? 64 : 128;
}
static int bgemm_vw(int P) { return P % 8 == 0 ? 8 : P % 4 == 0 ? 4 : 1; }

static int bg_prelayout_one(hipStream_t st, const float *w, u16 *fw, u16 *dg, int K, int C, int k) {
    if (k == 1) hipLaunchKernelGGL(bg_wt_kernel<1>, dim3(C / 32, K / 32), dim3(256), 0, st, w, fw, dg, K, C);
    else hipLaunchKernelGGL(bg_wt_kernel<9>, dim3(C / 32, K / 32), dim3(256), 0, st, w, fw, dg, K, C);
    MI_LAUNCH_CHECK("bg_wt_kernel");
    return 0;
}

int mi_bgemm_fwd(hipStream_t st, mid_workspace *ws, const u16 *x, const float *w, u16 *y, int N, int C, int H, int K, int k, int stride,
                 mid_bn_parts *parts) {
    const int T = k * k;
    const u16 *A = (const u16 *)ws->pre_fwd;
    if (!A) {
        if (!ws->wt || ws->wt_floats * 2 < (size_t)T * C * K) { mi_record_error("mi_bgemm_fwd", "workspace too small"); return -3; }
        if (bg_prelayout_one(st, w, (u16 *)ws->wt, nullptr, K, C, k)) return -1;
        A = (const u16 *)ws->wt;
    }
    BgArgs g = {};
    bgemm_geometry(g, N, C, H, K, stride);
    const int ctl = mi_cdiv(g.ncols, 128);
    const int bm = bgemm_pick_bm(K, ctl);
    g.mtiles = K / bm;
    g.tiles = g.mtiles * ctl;
    g.fdM = make_fastdiv(g.mtiles);
    g.vw = bgemm_vw(g.P);
    if (parts) {
        parts->nparts = 0;
        const int np = ctl * (bm == 128 ? 2 : 4);
        if (parts->buf && parts->floats >= (size_t)3 * np * K) { g.bn_part = parts->buf; g.bn_np = np; parts->nparts = np; }
    }
    mi_prof_begin(st, bgemm_fam(k), 2.0 * T * (double)g.ncols * C * K, 2.0 * ((double)N * C * g.HW + (double)g.ncols * K) + 4.0 * T * C * K);
    const int rc = bgemm_launch<BG_FWD>(st, dim3(g.tiles), A, x, y, nullptr, g, k, stride, bm);
    mi_prof_end(st);
    if (rc) return rc;
    MI_LAUNCH_CHECK("bgemm_kernel<fwd>");
    return 0;
}

int mi_bgemm_dgrad(hipStream_t st, mid_workspace *ws, const float *w, const u16 *dy, u16 *dx, const u16 *addend, int N, int C, int H,
                   int K, int k, int stride) {
    const int T = k * k;
    const u16 *A = (const u16 *)ws->pre_dgrad;
    if (!A) {
        if (!ws->wt || ws->wt_floats * 2 < (size_t)T * C * K) { mi_record_error("mi_bgemm_dgrad", "workspace too small"); return -3; }
        if (bg_prelayout_one(st, w, nullptr, (u16 *)ws->wt, K, C, k)) return -1;
        A = (const u16 *)ws->wt;
    }
    BgArgs g = {};
    bgemm_geometry(g, N, C, H, K, stride);
    const int ctl = mi_cdiv(g.ncols, 128);
    const int bm = bgemm_pick_bm(C, ctl);
    g.mtiles = C / bm;
    g.tiles = g.mtiles * ctl;
    g.fdM = make_fastdiv(g.mtiles);
    g.vw = bgemm_vw(g.HW);
    mi_prof_begin(st, bgemm_fam(k), 2.0 * T * (double)g.ncols * C * K,
                  2.0 * ((double)g.ncols * K + (double)N * C * g.HW * (addend ? 2 : 1)) + 4.0 * T * C * K);
    const int rc = bgemm_launch<BG_DGRAD>(st, dim3(g.tiles, stride == 2 ? 4 : 1), A, dy, dx, addend, g, k, stride, bm);
    mi_prof_end(st);
    if (rc) return rc;
    MI_LAUNCH_CHECK("bgemm_kernel<dgrad>");
    return 0;
}

int mi_igemm_wgrad_reduce(hipStream_t st, const float *part, float *dw, int K, int C, int k, int splits); // kernels_igemm.hip

int mi_bgemm_wgrad(hipStream_t st, mid_workspace *ws, const u16 *x, const u16 *dy, float *dw, int N, int C, int H, int K, int k, int stride) {
    const int T = k * k;
    const int splits = bgemm_wgrad_splits(N, C, H, K, k, stride);
    if (!ws || ws->part_floats < (size_t)splits * T * K * C) { mi_record_error("mi_bgemm_wgrad", "workspace too small"); return -3; }
    BgArgs g = {};
    bgemm_geometry(g, N, C, H, K, stride);
    const int bm = K % 128 == 0 ? 128 : 64;
    g.mtiles = K / bm;
    g.cb64 = C / 64;
    g.nhalf = T * g.cb64;
    g.fdCb = make_fastdiv(g.cb64);
    g.tiles = g.mtiles * ((g.nhalf + 1) / 2);
    g.fdM = make_fastdiv(g.mtiles);
    const int kd = N * g.P;
    g.klen = mi_cdiv(mi_cdiv(kd, splits), BG_BK) * BG_BK;
    const int used = mi_cdiv(kd, g.klen);
    mi_prof_begin(st, bgemm_fam(k), 2.0 * T * (double)kd * C * K, 2.0 * ((double)N * C * g.HW + (double)kd * K) + 4.0 * T * C * K);
    const int rc = bgemm_launch<BG_WGRAD>(st, dim3(g.tiles, used), dy, x, ws->part, nullptr, g, k, stride, bm);
    if (rc) { mi_prof_end(st); return rc; }
    const int rr = mi_igemm_wgrad_reduce(st, ws->part, dw, K, C, k, used);
    mi_prof_end(st);
    if (rr) return rr;
    MI_LAUNCH_CHECK("bgemm_kernel<wgrad>");
    return 0;
}

extern "C" {
int mid_bf16_supported(int op, int N, int C, int H, int K, int k, int stride) { return mi_bgemm_supported(op, N, C, H, K, k, stride); }
size_t mid_bf16_part_floats(int N, int C, int H, int K, int k, int stride) { return mi_bgemm_part_floats(N, C, H, K, k, stride); }
int mid_conv_prelayout_all_bf16(mid_stream s, const mid_wt_entry *entries_dev, const int *tile_entry_dev, int ntiles) {
    if (ntiles <= 0) return 0;
    hipLaunchKernelGGL(bg_wt_all_kernel, dim3(ntiles), dim3(256), 0, (hipStream_t)s, entries_dev, tile_entry_dev);
    MI_LAUNCH_CHECK("bg_wt_all_kernel");
    return 0;
}
int mid_conv_fwd_bf16(mid_stream s, mid_workspace *ws, const void *x, const float *w, void *y, int N, int C, int H, int K, int k,
                      int stride, mid_bn_parts *parts) {
    if (parts) parts->nparts = 0;
    if (!mi_bgemm_supported(BGOP_FWD, N, C, H, K, k, stride)) { mi_record_error("mid_conv_fwd_bf16", "shape not supported by the bf16 kernels"); return -2; }
    return mi_bgemm_fwd((hipStream_t)s, ws, (const u16 *)x, w, (u16 *)y, N, C, H, K, k, stride, parts);
}
int mid_conv_dgrad_bf16(mid_stream s, mid_workspace *ws, const float *w, const void *dy, void *dx, const void *addend, int N, int C,
                        int H, int K, int k, int stride) {
    if (!mi_bgemm_supported(BGOP_DGRAD, N, C, H, K, k, stride)) { mi_record_error("mid_conv_dgrad_bf16", "shape not supported by the bf16 kernels"); return -2; }
    return mi_bgemm_dgrad((hipStream_t)s, ws, w, (const u16 *)dy, (u16 *)dx, (const u16 *)addend, N, C, H, K, k, stride);
}
int mid_conv_wgrad_bf16(mid_stream s, mid_workspace *ws, const void *x, const void *dy, float *dw, int N, int C, int H, int K, int k,
                        int stride) {
    if (!mi_bgemm_supported(BGOP_WGRAD, N, C, H, K, k, stride)) { mi_record_error("mid_conv_wgrad_bf16", "shape not supported by the bf16 kernels"); return -2; }
    return mi_bgemm_wgrad((hipStream_t)s, ws, (const u16 *)x, (const u16 *)dy, dw, N, C, H, K, k, stride);
}
int mid_f32_to_bf16(mid_stream s, const float *in, void *out, size_t n) {
    size_t b = (n + 255) / 256; if (b > 65536) b = 65536; if (b < 1) b = 1;
    hipLaunchKernelGGL(bg_f2b_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)s, in, (u16 *)out, n);
    MI_LAUNCH_CHECK("bg_f2b_kernel");
    return 0;
}
int mid_bf16_to_f32(mid_stream s, const void *in, float *out, size_t n) {
    size_t b = (n + 255) / 256; if (b > 65536) b = 65536; if (b < 1) b = 1;
    hipLaunchKernelGGL(bg_b2f_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)s, (const u16 *)in, out, n);
    MI_LAUNCH_CHECK("bg_b2f_kernel");
    return 0;
}
}
