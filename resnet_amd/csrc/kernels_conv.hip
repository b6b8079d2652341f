// kernels_conv.hip -- im2col-free, LDS-tiled direct convolution for the k>1 layers (3x3 s1/s2, 7x7 s2):
// forward, dgrad and wgrad.  No MFMA here (north_star: MFMA only for the 1x1 / FC GEMMs, kernels_gemm.hip).
// Replaces doConvolution / convolutionDerivInput / convolutionDerivWeights (resnet.cu:109-281) and their
// launch wrappers (resnet.cu:1386-1429).  Activations NCHW, weights KCRS.
//
// Design (CDNA4, wave64):
//  * forward / dgrad ("tap" kernel): a workgroup owns 64*NW consecutive output pixels of the flattened
//    (n, oh, ow) space and TK output channels.  The input rows those pixels need are staged per channel
//    chunk in LDS (zero-filled halo), every lane owns ONE output pixel and keeps TK accumulators, and the
//    weights are wave-uniform: they are read with scalar loads (s_load_dwordx8/16 from a [c][tap][k]
//    re-laid copy) and enter v_fma_f32 as SGPR operands -- no LDS or VGPR traffic for weights at all.
//    dgrad is the same kernel on dY with flipped/re-laid weights; the stride-2 dgrad runs as four
//    parity classes (1, 2, 2, 4 taps) so no tap is ever multiplied by a structural zero.
//  * wgrad: every lane owns output channels k (TKL of them) and TC*k*k accumulators; the input row
//    segment is held one element per lane in a VGPR and broadcast with v_readlane (SGPR operand again);
//    dY tiles go through LDS ([k][pixel], odd pitch, conflict-free).  The N*Ho*Wo reduction is split
//    across workgroups; partial sums are reduced by a second, order-fixed kernel (deterministic).
#include <stdlib.h>
#include "mi_common.hpp"
#include "mi_device.h"

// ------------------------------------------------------------------------------------------
// weight re-layout: out[(ci*T + t)*Co + co] = w_kcrs[...], T taps taken from src_rs[t] = r*k+s
struct WtArgs {
    int Ci, Co, T, kk, C; // C = channels of the KCRS tensor (its 2nd dim)
    int transposed;       // 0: ci = c, co = k (forward).  1: ci = k, co = c (dgrad)
    unsigned char src[49];
};
__global__ void wt_relayout_kernel(const float *__restrict__ w, float *__restrict__ out, const WtArgs a) {
    const long total = (long)a.Ci * a.T * a.Co;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % a.Co);
        const long r = i / a.Co;
        const int t = (int)(r % a.T), ci = (int)(r / a.T);
        const int kf = a.transposed ? ci : co, c = a.transposed ? co : ci;
        out[i] = w[((long)kf * a.C + c) * a.kk + a.src[t]];
    }
}

// ------------------------------------------------------------------------------------------
struct DConvArgs {
    int N, Cin, Hin, Win, Cout, Hof, Wof;
    int Hsub, Wsub, osy, oy0, osx, ox0; // output sub-grid: position (i*osy+oy0, j*osx+ox0)
    int sy, sx, offy, offx;             // input row of (i, tr) = i*sy + offy + tr
    int PW, QR, Hq, CC, tbl_pad, plane, jcnt;
    uint32_t total_pix;
    FastDiv fd_Wsub, fd_Hsub, fd_Hq, fd_PW;
};

// Pipeline per channel chunk (CC channels): the chunk after next is in flight from HBM/L2 into registers while the
// current one is multiplied out of LDS buffer A and the next one sits in LDS buffer B -> ONE barrier per chunk and
// no exposed global-load latency.  Each thread stages the same JM patch positions for every channel of a chunk.
template <int NTR, int NTC, int TK, int JM, int CCM>
__global__ void __launch_bounds__(448)
dconv_kernel(const float *__restrict__ in, const float *__restrict__ wT, float *__restrict__ out,
             const float *__restrict__ addend, const DConvArgs a) {
    extern __shared__ float smem[];
    int *tbl = (int *)smem;
    float *patch0 = smem + a.tbl_pad;
    const int tid = threadIdx.x;
    const uint32_t pix0 = blockIdx.x * blockDim.x;
    const int k0 = blockIdx.y * TK;
    const int plane = a.plane, bufsz = a.CC * plane;
    const int HW = a.Hin * a.Win;

    // first patch row of this workgroup in "q space" (q = n*Hq + i*sy + tr): contiguous for a pixel range
    const uint32_t R0 = fd_div(pix0, a.fd_Wsub);
    const uint32_t n0 = fd_div(R0, a.fd_Hsub);
    const int q0 = (int)(n0 * a.Hq + (R0 - n0 * a.Hsub) * a.sy);
    for (int qr = tid; qr < a.QR; qr += blockDim.x) {
        const uint32_t q = (uint32_t)(q0 + qr);
        const uint32_t nq = fd_div(q, a.fd_Hq);
        const int ih = (int)(q - nq * a.Hq) + a.offy;
        tbl[qr] = (nq < (uint32_t)a.N && ih >= 0 && ih < a.Hin) ? (int)((nq * a.Cin * a.Hin + ih) * a.Win) : -1;
    }
    // the pixel this lane owns
    const uint32_t p = pix0 + tid;
    const bool pvalid = p < a.total_pix;
    const uint32_t pc = pvalid ? p : a.total_pix - 1;
    const uint32_t R = fd_div(pc, a.fd_Wsub);
    const int j = (int)(pc - R * a.Wsub);
    const uint32_t n = fd_div(R, a.fd_Hsub);
    const int i = (int)(R - n * a.Hsub);
    const int lbase = ((int)(n * a.Hq + i * a.sy) - q0) * a.PW + j * a.sx;
    __syncthreads();
    // patch positions this thread stages: pos = tid + 448*jj  ->  element offset inside channel 0 (or -1 = zero)
    // byte offsets (unsigned, so a load is scalar channel base + 32-bit lane offset: no 64-bit VALU address math);
    // invalid positions (halo, tail) point at element 0 and are zeroed by the validity bit when stashed
    uint32_t goff[JM], gvalid = 0;
#pragma unroll
    for (int jj = 0; jj < JM; jj++) {
        const int pos = tid + jj * 448;
        uint32_t g = 0;
        if (jj < a.jcnt && pos < plane) {
            const uint32_t qr = fd_div((uint32_t)pos, a.fd_PW);
            const int iw = pos - (int)qr * a.PW + a.offx;
            const int t = tbl[qr];
            if (t >= 0 && iw >= 0 && iw < a.Win) { g = (uint32_t)(t + iw) * 4u; gvalid |= 1u << jj; }
        }
        goff[jj] = g;
    }
    float regs[CCM][JM];
    auto issue = [&](int c0) {
#pragma unroll
        for (int c = 0; c < CCM; c++) {
            const bool cok = c < a.CC && c0 + c < a.Cin;
            const char *src = (const char *)(in + (size_t)(cok ? c0 + c : 0) * HW); // wave-uniform
#pragma unroll
            for (int jj = 0; jj < JM; jj++) regs[c][jj] = *(const float *)(src + goff[jj]); // masked when stashed
        }
    };
    auto stash = [&](float *buf) {
#pragma unroll
        for (int c = 0; c < CCM; c++) {
            if (c < a.CC) {
#pragma unroll
                for (int jj = 0; jj < JM; jj++) {
                    const int pos = tid + jj * 448;
                    if (jj < a.jcnt && pos < plane) buf[c * plane + pos] = ((gvalid >> jj) & 1u) ? regs[c][jj] : 0.f;
                }
            }
        }
    };

    float acc[TK];
#pragma unroll
    for (int t = 0; t < TK; t++) acc[t] = 0.f;

    const int nch = (a.Cin + a.CC - 1) / a.CC;
    issue(0);
    stash(patch0);
    if (nch > 1) issue(a.CC);
    __syncthreads();
    for (int ch = 0; ch < nch; ch++) {
        const float *cur = patch0 + (ch & 1) * bufsz;
        const int c0 = ch * a.CC;
        const int cc = min(a.CC, a.Cin - c0);
        for (int c = 0; c < cc; c++) {
            const float *wp = wT + (size_t)(c0 + c) * (NTR * NTC) * a.Cout + k0; // wave-uniform -> scalar loads
            const float *pp = cur + c * plane + lbase;
#pragma unroll
            for (int tr = 0; tr < NTR; tr++) {
#pragma unroll
                for (int tc = 0; tc < NTC; tc++) {
                    const float v = pp[tr * a.PW + tc];
                    const float *wq = wp + (tr * NTC + tc) * a.Cout;
#pragma unroll
                    for (int t = 0; t < TK; t++) acc[t] = fmaf(wq[t], v, acc[t]);
                }
            }
        }
        if (ch + 1 < nch) {
            stash(patch0 + ((ch + 1) & 1) * bufsz);
            if (ch + 2 < nch) issue(c0 + 2 * a.CC);
        }
        __syncthreads();
    }
    if (pvalid) {
        const int oh = i * a.osy + a.oy0, ow = j * a.osx + a.ox0;
        const size_t plane_o = (size_t)a.Hof * a.Wof;
        const size_t o = ((size_t)n * a.Cout + k0) * plane_o + (size_t)oh * a.Wof + ow;
        if (addend) {
#pragma unroll
            for (int t = 0; t < TK; t++) out[o + t * plane_o] = acc[t] + addend[o + t * plane_o];
        } else {
#pragma unroll
            for (int t = 0; t < TK; t++) out[o + t * plane_o] = acc[t];
        }
    }
}

template <int NTR, int NTC>
static int launch_dconv_t(hipStream_t st, dim3 grid, size_t lds, int tk, int jm, const float *in, const float *wT, float *out,
                          const float *addend, const DConvArgs &a) {
    dim3 block(448);
#define DCL(TK_, JM_, CCM_) hipLaunchKernelGGL((dconv_kernel<NTR, NTC, TK_, JM_, CCM_>), grid, block, lds, st, in, wT, out, addend, a)
    if (jm <= 2) { if (tk == 64) DCL(64, 2, 8); else if (tk == 32) DCL(32, 2, 8); else DCL(16, 2, 8); }
    else if (jm <= 5) { if (tk == 64) DCL(64, 5, 3); else if (tk == 32) DCL(32, 5, 3); else DCL(16, 5, 3); }
    else { if (tk == 64) DCL(64, 7, 3); else if (tk == 32) DCL(32, 7, 3); else DCL(16, 7, 3); }
#undef DCL
    return 0;
}

// one tap-kernel launch; ntr x ntc taps, weights already re-laid in wT
static int launch_dconv(hipStream_t st, const float *in, const float *wT, float *out, const float *addend, DConvArgs a,
                        int ntr, int ntc, bool prof = true) {
    const int PIX = 448;
    a.total_pix = (uint32_t)a.N * a.Hsub * a.Wsub;
    a.PW = (a.Wsub - 1) * a.sx + ntc;
    a.Hq = (a.Hsub - 1) * a.sy + ntr;
    // worst-case number of patch rows a workgroup touches
    long rb = (PIX % a.Wsub == 0) ? PIX / a.Wsub : (PIX + a.Wsub - 2) / a.Wsub + 1;
    const long totrows = (long)a.N * a.Hsub;
    if (rb > totrows) rb = totrows;
    long nimg = (rb - 1 + a.Hsub - 1) / a.Hsub + 1;
    if (PIX % a.Wsub == 0 && (a.Hsub % rb == 0)) nimg = 1;
    a.QR = (int)((rb - 1) * a.sy + ntr + (nimg - 1) * (ntr > a.sy ? ntr - a.sy : 0));
    a.tbl_pad = (a.QR + 3) & ~3;
    a.plane = a.QR * a.PW;
    a.jcnt = (a.plane + PIX - 1) / PIX;
    if (a.jcnt > 7) { mi_record_error("dconv", "input patch too large for this shape"); return -2; }
    const int ccm = a.jcnt <= 2 ? 8 : 3;
    static int lds_budget = -1; /* floats per LDS buffer; RESNET_MI_DCONV_LDS overrides */
    if (lds_budget < 0) { const char *e = getenv("RESNET_MI_DCONV_LDS"); lds_budget = e ? atoi(e) : 5120; }
    int cc = lds_budget / a.plane; // two LDS buffers
    if (cc < 1) cc = 1;
    if (cc > ccm) cc = ccm;
    if (cc > a.Cin) cc = a.Cin;
    a.CC = cc;
    const size_t lds = (size_t)(a.tbl_pad + 2 * (size_t)cc * a.plane) * 4;
    if (lds > 64 * 1024) { mi_record_error("dconv", "LDS patch too large for this shape"); return -2; }
    a.fd_Wsub = make_fastdiv(a.Wsub); a.fd_Hsub = make_fastdiv(a.Hsub);
    a.fd_Hq = make_fastdiv(a.Hq); a.fd_PW = make_fastdiv(a.PW);
    static int tk_pref = -1; /* experiment knob: RESNET_MI_DCONV_TK=64|32 */
    if (tk_pref < 0) { const char *e = getenv("RESNET_MI_DCONV_TK"); tk_pref = e ? atoi(e) : 32; }
    int tk = (a.Cout % 32 == 0) ? 32 : 16;
    if (tk_pref == 64 && a.Cout % 64 == 0) tk = 64;
    if (tk_pref == 16) tk = 16;
    if (a.Cout % 16 != 0) { mi_record_error("dconv", "channel count must be a multiple of 16"); return -2; }
    dim3 grid(mi_cdiv(a.total_pix, PIX), a.Cout / tk);
    // algorithmic work of this launch: 2*taps MACs per (pixel, cin, cout); bytes = input + weights + output once
    const double fl = 2.0 * ntr * ntc * (double)a.total_pix * a.Cin * a.Cout;
    const double by = 4.0 * ((double)a.N * a.Cin * a.Hin * a.Win / (a.osy * a.osx) + (double)ntr * ntc * a.Cin * a.Cout +
                             (double)a.total_pix * a.Cout * (addend ? 2 : 1));
    if (prof) mi_prof_begin(st, MI_FAM_DCONV, fl, by);
    int rc = -2;
#define DC(NTR_, NTC_) if (ntr == NTR_ && ntc == NTC_) rc = launch_dconv_t<NTR_, NTC_>(st, grid, lds, tk, a.jcnt, in, wT, out, addend, a);
    DC(3, 3) DC(7, 7) DC(1, 1) DC(1, 2) DC(2, 1) DC(2, 2)
#undef DC
    if (prof) mi_prof_end(st);
    if (rc) { mi_record_error("dconv", "unsupported tap shape"); return rc; }
    MI_LAUNCH_CHECK("dconv_kernel");
    return 0;
}

static int launch_wt(hipStream_t st, const float *w, float *out, int Ci, int Co, int T, int k, int C, int transposed,
                     const unsigned char *src) {
    WtArgs a;
    a.Ci = Ci; a.Co = Co; a.T = T; a.kk = k * k; a.C = C; a.transposed = transposed;
    for (int t = 0; t < T; t++) a.src[t] = src[t];
    const long total = (long)Ci * T * Co;
    int blocks = mi_cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wt_relayout_kernel, dim3(blocks), dim3(256), 0, st, w, out, a);
    MI_LAUNCH_CHECK("wt_relayout_kernel");
    return 0;
}

// ------------------------------------------------------------------------------------------
// wgrad
struct WgArgs {
    int N, C, H, W, K, Ho, Wo, pad;
    int RPC, RG, SEGW, SG;
    int chunks_total, chunks_per_split;
    FastDiv fd_SG, fd_RG;
    size_t part_stride;
};
#define WG_LDP 65

// Pipeline: while chunk ch is multiplied (dY tile in LDS buffer ch&1, x rows in registers), chunk ch+1 is in flight
// from memory into registers -> one barrier per chunk.  Pixels are walked four at a time so that one v_readlane
// window of 3*S+KS input elements serves 4*KS taps (rolling reuse along the row).
template <int KS, int S, int TKL, int TC, int NXR>
__global__ void __launch_bounds__(256)
wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ part, const WgArgs a) {
    constexpr int KT = 64 * TKL, T = KS * KS, RPCMAX = (NXR - KS) / S + 1, DYR = KT / 4, NXS = 3 * S + KS;
    extern __shared__ float dyS[]; // [2][KT][WG_LDP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = blockIdx.x * KT;
    const int cb = (blockIdx.y * 4 + wave) * TC;
    float acc[TKL][TC][T];
#pragma unroll
    for (int u = 0; u < TKL; u++)
#pragma unroll
        for (int c = 0; c < TC; c++)
#pragma unroll
            for (int t = 0; t < T; t++) acc[u][c][t] = 0.f;

    const int ch_beg = blockIdx.z * a.chunks_per_split;
    const int ch_end = min(a.chunks_total, ch_beg + a.chunks_per_split);
    float dreg[DYR], xnext[TC][NXR], xrow[TC][NXR];

    struct Geo { int n, oh0, rows, ow0, segw; };
    auto geom = [&](int ch) {
        Geo g;
        const uint32_t t1 = fd_div((uint32_t)ch, a.fd_SG);
        const int sg = ch - (int)t1 * a.SG;
        const uint32_t n = fd_div(t1, a.fd_RG);
        const int rg = (int)(t1 - n * a.RG);
        g.n = (int)n; g.oh0 = rg * a.RPC; g.rows = min(a.RPC, a.Ho - g.oh0);
        g.ow0 = sg * a.SEGW; g.segw = min(a.SEGW, a.Wo - g.ow0);
        return g;
    };
    auto issue = [&](int ch) {
        const Geo g = geom(ch);
        const int npx = g.rows * g.segw; // rows==1 or segw==Wo: pixels of a k-plane are one contiguous run
#pragma unroll
        for (int q = 0; q < DYR; q++) {
            const int kk = wave + 4 * q;
            float v = 0.f;
            if (lane < npx) v = dy[(((size_t)g.n * a.K + kb + kk) * a.Ho + g.oh0) * a.Wo + g.ow0 + lane];
            dreg[q] = v;
        }
        const int nxr = (g.rows - 1) * S + KS;
        const int iw = g.ow0 * S - a.pad + lane;
        const bool colok = iw >= 0 && iw < a.W && lane < (g.segw - 1) * S + KS;
#pragma unroll
        for (int c = 0; c < TC; c++) {
            const int cch = cb + c;
#pragma unroll
            for (int xr = 0; xr < NXR; xr++) {
                const int ih = g.oh0 * S - a.pad + xr;
                float v = 0.f;
                if (xr < nxr && cch < a.C && ih >= 0 && ih < a.H && colok)
                    v = x[(((size_t)g.n * a.C + cch) * a.H + ih) * a.W + iw];
                xnext[c][xr] = v;
            }
        }
    };

    if (ch_beg < ch_end) issue(ch_beg);
    for (int ch = ch_beg; ch < ch_end; ch++) {
        float *buf = dyS + ((ch - ch_beg) & 1) * (KT * WG_LDP);
#pragma unroll
        for (int q = 0; q < DYR; q++) buf[(wave + 4 * q) * WG_LDP + lane] = dreg[q];
#pragma unroll
        for (int c = 0; c < TC; c++)
#pragma unroll
            for (int xr = 0; xr < NXR; xr++) xrow[c][xr] = xnext[c][xr];
        const Geo g = geom(ch);
        if (ch + 1 < ch_end) issue(ch + 1);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RPCMAX; r++) {
            if (r < g.rows) {
                for (int ow4 = 0; ow4 < g.segw; ow4 += 4) {
                    float d[TKL][4];
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) {
                        const bool ok = ow4 + q4 < g.segw; // tail pixels contribute 0
#pragma unroll
                        for (int u = 0; u < TKL; u++) {
                            const float v = buf[(lane + 64 * u) * WG_LDP + r * g.segw + ow4 + q4];
                            d[u][q4] = ok ? v : 0.f;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < TC; c++)
#pragma unroll
                        for (int tr = 0; tr < KS; tr++) {
                            float xs[NXS];
#pragma unroll
                            for (int m = 0; m < NXS; m++)
                                xs[m] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xrow[c][r * S + tr]), (ow4 * S + m) & 63));
#pragma unroll
                            for (int q4 = 0; q4 < 4; q4++)
#pragma unroll
                                for (int tc = 0; tc < KS; tc++)
#pragma unroll
                                    for (int u = 0; u < TKL; u++)
                                        acc[u][c][tr * KS + tc] = fmaf(xs[q4 * S + tc], d[u][q4], acc[u][c][tr * KS + tc]);
                        }
                }
            }
        }
    }
    float *po = part + (size_t)blockIdx.z * a.part_stride;
#pragma unroll
    for (int u = 0; u < TKL; u++) {
        const int k = kb + lane + 64 * u;
#pragma unroll
        for (int c = 0; c < TC; c++) {
            const int cch = cb + c;
            if (cch < a.C) {
#pragma unroll
                for (int t = 0; t < T; t++) po[((size_t)k * a.C + cch) * T + t] = acc[u][c][t];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// wgrad, 3x3 with C % 64 == 0 ("C" formulation; the read-lane kernel above remains for the 7x7 stem).
// Lane = input channel c (64 per workgroup), wave = 8 output channels (32 per workgroup), 72 accumulators per lane.
// Per chunk of output pixels both operands go through LDS: the x patch channel-major with an odd pitch (lane c reads
// its own row, conflict-free) and the dY tile [k][pixel] read as broadcast ds_read_b128 (four pixels of one k).
// Pixels are walked four at a time with a rolling x window (3 rows x (3*S+3) columns serve 4 pixels x 9 taps);
// rows are padded to whole quads with zero dY.  Global loads of the NEXT chunk are in flight in registers.
// (A scalar-load variant of the dY operand was measured first: correct but latency-bound on cold scalar-cache
// misses, 19-38 TF; unlike dconv's weights the dY stream is unique per workgroup.)
struct WcArgs {
    int N, C, H, W, K, Ho, Wo, pad;
    int NPX, rows, segw, QPR, DP, cpp, chunks_total, chunks_per_split;
    int PR, PW, cwlog2, pitch, jcnt, xs_floats;
    FastDiv fd_cpp, fd_Wo, fd_PR, fd_Q4, fd_PW;
    size_t part_stride;
};
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define WC_DPK 36 /* dY tile: [pixel slot][32 k] padded to 36 floats (16-B aligned rows) */

template <int S>
__global__ void __launch_bounds__(256)
wgradC_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ part, const WcArgs a) {
    constexpr int NXW = 3 * S + 3;
    extern __shared__ float sm[];
    // per buffer: x patch [64][pitch] + slack (xs_floats), then the dY tile [pixel slot][WC_DPK]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb0 = blockIdx.x * 32, cb = blockIdx.y * 64;
    const int plane_o = a.Ho * a.Wo;
    // accumulators as k-pairs: v_pk_fma_f32 takes (dY[k], dY[k+1]) straight out of a ds_read_b128 and the x value
    // as a broadcast half -- no register shuffling to feed the packed FMA
    f32x2 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int kp = 0; kp < 4; kp++) acc[t][kp] = (f32x2){0.f, 0.f};
    // masked tail columns of a row read past the patch row (next row / pitch pad / slack) and multiply it by dY = 0:
    // everything they can touch must be finite, so the whole x region starts zeroed (pads are never rewritten)
    for (int i = tid; i < 2 * (a.xs_floats + 32 * WC_DPK); i += 256) sm[i] = 0.f;
    // x staging: wave w stages channels w*16 .. w*16+15; lane l owns patch positions l and l+64 (row, col fixed for
    // the whole kernel), so a slot costs one scalar channel base + one per-lane constant offset
    float pre[2][16], pred[4];
    const int patch = a.PR * a.PW;
    int prow[2], pcol[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int pos = lane + 64 * h;
        const uint32_t r = fd_div((uint32_t)pos, a.fd_PW);
        prow[h] = (int)r; pcol[h] = pos - (int)r * a.PW;
    }
    // dY tile slot of this thread: the same (row, column) for its 4 k-rows
    const int ps = tid & 31;
    const uint32_t dr = fd_div((uint32_t)ps, a.fd_Q4);
    const int dq = ps - (int)dr * (a.QPR * 4);
    const bool dslot = ps < a.rows * a.QPR * 4;

    auto issue = [&](int ch) {
        const uint32_t n = fd_div((uint32_t)ch, a.fd_cpp);
        const int p0 = (ch - (int)n * a.cpp) * a.NPX;
        const int npx = min(a.NPX, plane_o - p0);
        const uint32_t oh0 = fd_div((uint32_t)p0, a.fd_Wo);
        const int ih0 = (int)oh0 * S - a.pad, iw0 = (p0 - (int)oh0 * a.Wo) * S - a.pad;
        int off[2];
        bool ok[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int ih = ih0 + prow[h], iw = iw0 + pcol[h];
            ok[h] = lane + 64 * h < patch && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
            off[h] = ih * a.W + iw;
        }
        const float *xb = x + ((size_t)n * a.C + cb + wave * 16) * a.H * a.W; // wave-uniform
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float *xc = xb + (size_t)j * a.H * a.W;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                float v = 0.f;
                if (ok[h]) v = xc[off[h]];
                pre[h][j] = v;
            }
        }
        const int pix = (int)dr * a.segw + dq;
        const bool dok = dslot && dq < a.segw && pix < npx;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int kk = (tid >> 5) + 8 * j;
            float v = 0.f;
            if (dok) v = dy[((size_t)n * a.K + kb0 + kk) * plane_o + p0 + pix];
            pred[j] = v;
        }
    };
    // two LDS buffers (x patch + dY tile each): chunk ch+1 is stashed while other waves may still multiply chunk ch,
    // so there is ONE barrier per chunk (occupancy is VGPR-limited to 2 workgroups/CU, the second buffer is free)
    const int bufsz = a.xs_floats + 32 * WC_DPK;
    auto stash = [&](float *base) {
        float *xw_ = base + wave * 16 * a.pitch + lane;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (lane + 64 * h < patch) {
#pragma unroll
                for (int j = 0; j < 16; j++) xw_[j * a.pitch + 64 * h] = pre[h][j];
            }
        }
        if (dslot) {
            float *dd = base + a.xs_floats;
#pragma unroll
            for (int j = 0; j < 4; j++) dd[ps * WC_DPK + (tid >> 5) + 8 * j] = pred[j];
        }
    };

    const int ch_beg = blockIdx.z * a.chunks_per_split;
    const int ch_end = min(a.chunks_total, ch_beg + a.chunks_per_split);
    if (ch_beg < ch_end) issue(ch_beg);
    __syncthreads(); // zero fill above is complete
    if (ch_beg < ch_end) {
        stash(sm);
        if (ch_beg + 1 < ch_end) issue(ch_beg + 1);
    }
    __syncthreads();
    for (int ch = ch_beg; ch < ch_end; ch++) {
        const float *cur = sm + ((ch - ch_beg) & 1) * bufsz;
        const float *xl = cur + lane * a.pitch;
        const float *dw = cur + a.xs_floats + wave * 8;
        for (int r = 0; r < a.rows; r++) {
            for (int q = 0; q < a.QPR; q++) {
                float xw[3][NXW];
                const float *xb = xl + r * S * a.PW + q * 4 * S;
#pragma unroll
                for (int tr = 0; tr < 3; tr++)
#pragma unroll
                    for (int m = 0; m < NXW; m++) xw[tr][m] = xb[tr * a.PW + m];
                const float *db = dw + (r * a.QPR + q) * 4 * WC_DPK;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const f32x4 da = *(const f32x4 *)(db + j * WC_DPK), dbv = *(const f32x4 *)(db + j * WC_DPK + 4);
                    const f32x2 d2[4] = {(f32x2){da[0], da[1]}, (f32x2){da[2], da[3]}, (f32x2){dbv[0], dbv[1]}, (f32x2){dbv[2], dbv[3]}};
#pragma unroll
                    for (int tr = 0; tr < 3; tr++)
#pragma unroll
                        for (int tc = 0; tc < 3; tc++) {
                            const float xv = xw[tr][j * S + tc];
                            const f32x2 x2 = (f32x2){xv, xv};
#pragma unroll
                            for (int kp = 0; kp < 4; kp++)
                                acc[tr * 3 + tc][kp] = __builtin_elementwise_fma(d2[kp], x2, acc[tr * 3 + tc][kp]);
                        }
                }
            }
        }
        if (ch + 1 < ch_end) {
            stash(sm + ((ch + 1 - ch_beg) & 1) * bufsz);
            if (ch + 2 < ch_end) issue(ch + 2);
        }
        __syncthreads();
    }
    float *po = part + (size_t)blockIdx.z * a.part_stride;
    const int kb = kb0 + wave * 8;
#pragma unroll
    for (int k = 0; k < 8; k++)
#pragma unroll
        for (int t = 0; t < 9; t++) po[((size_t)(kb + k) * a.C + cb + lane) * 9 + t] = acc[t][k >> 1][k & 1];
}

int mi_launch_split_reduce(hipStream_t st, const float *part, float *out, long n, int splits, size_t stride);
struct WbPlan { int npx, rows, segw, qpr, cpp, chunks, splits, cps, PR, PW, cwlog2, pitch, jcnt; };
static int wgradB_plan(int N, int C, int H, int K, int k, int stride, WbPlan *p) {
    if (k != 3 || (stride != 1 && stride != 2) || C % 64 || K % 32 || H % stride) return -1;
    const int Wo = H / stride, npxmax = stride == 1 ? 28 : 14;
    int npx = 0;
    if (Wo >= npxmax) { for (int d = npxmax; d >= 4; d--) if (Wo % d == 0) { npx = d; break; } }
    else npx = Wo * (npxmax / Wo);
    if (npx < 4) return -1;
    p->rows = npx >= Wo ? npx / Wo : 1;
    p->segw = npx >= Wo ? Wo : npx;
    p->qpr = mi_cdiv(p->segw, 4);
    if (p->rows * p->qpr * 4 > 32) return -1;
    p->npx = npx;
    p->PR = (p->rows - 1) * stride + 3;
    p->PW = (p->segw - 1) * stride + 3;
    if (p->PW > 32) return -1;
    p->cwlog2 = p->PW > 16 ? 5 : 4;
    p->pitch = (p->PR * p->PW) | 1;
    p->jcnt = 0;
    if (p->PR * p->PW > 128) return -1; /* two patch positions per lane */
    p->cpp = mi_cdiv(Wo * Wo, npx);
    p->chunks = N * p->cpp;
    const int base_blocks = (K / 32) * (C / 64);
    int splits = mi_cdiv(2048, base_blocks);
    if (splits > p->chunks) splits = p->chunks;
    if (splits < 1) splits = 1;
    p->cps = mi_cdiv(p->chunks, splits);
    p->splits = mi_cdiv(p->chunks, p->cps);
    return 0;
}

// out[i] = sum_z part[z][i] in a FIXED order (deterministic): a workgroup owns 64 consecutive elements (coalesced
// 256-B rows), its 4 waves take z = w, w+4, ... with 8 loads in flight each, LDS adds the four partial sums.
__global__ void __launch_bounds__(256)
split_reduce_kernel(const float *__restrict__ part, float *__restrict__ out, long n, int splits, size_t stride) {
    __shared__ float sh[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long base = (long)blockIdx.x * 64; base < n; base += (long)gridDim.x * 64) {
        const long i = base + lane;
        float s = 0.f;
        if (i < n) {
            int z = wave;
            for (; z + 28 < splits; z += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = part[(size_t)(z + 4 * u) * stride + i];
#pragma unroll
                for (int u = 0; u < 8; u++) s += v[u];
            }
            for (; z < splits; z += 4) s += part[(size_t)z * stride + i];
        }
        sh[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && i < n) out[i] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
        __syncthreads();
    }
}

int mi_launch_split_reduce(hipStream_t st, const float *part, float *out, long n, int splits, size_t stride) {
    int blocks = mi_cdiv(n, 64);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(split_reduce_kernel, dim3(blocks), dim3(256), 0, st, part, out, n, splits, stride);
    MI_LAUNCH_CHECK("split_reduce_kernel");
    return 0;
}

struct WgPlan { int tkl, tc, rpc, rg, segw, sg, splits, cps, chunks; };
static int wgrad_plan(int N, int C, int H, int K, int k, int stride, WgPlan *p) {
    if (!((k == 3 && (stride == 1 || stride == 2)) || (k == 7 && stride == 2))) return -1;
    if (K % 64 != 0) return -1;
    const int Ho = H / stride, Wo = Ho;
    const int rpcmax = (k == 7) ? 1 : (stride == 2 ? 4 : 7); /* NXR = 9 input rows held per channel */
    p->tkl = (k == 7) ? 1 : (K >= 128 ? 2 : 1);
    p->tc = (k == 7) ? 1 : 2;
    int segmax = (64 - k) / stride + 1; // (segw-1)*stride + k <= 64
    if (segmax > 64) segmax = 64;
    if (Wo <= segmax) { p->segw = Wo; p->sg = 1; }
    else { p->sg = mi_cdiv(Wo, segmax); p->segw = mi_cdiv(Wo, p->sg); }
    int rpc = 1;
    if (p->sg == 1) { rpc = 64 / Wo; if (rpc < 1) rpc = 1; if (rpc > rpcmax) rpc = rpcmax; if (rpc > Ho) rpc = Ho; }
    p->rpc = rpc; p->rg = mi_cdiv(Ho, rpc);
    p->chunks = N * p->rg * p->sg;
    const int base_blocks = (K / (64 * p->tkl)) * mi_cdiv(C, 4 * p->tc);
    int splits = mi_cdiv(2048, base_blocks);
    if (splits > p->chunks) splits = p->chunks;
    if (splits < 1) splits = 1;
    p->cps = mi_cdiv(p->chunks, splits);
    p->splits = mi_cdiv(p->chunks, p->cps);
    return 0;
}

template <int KS, int S, int TKL, int TC, int NXR>
static int launch_wgrad_t(hipStream_t st, const float *x, const float *dy, float *part, const WgArgs &a, dim3 grid) {
    const size_t lds = (size_t)2 * 64 * TKL * WG_LDP * 4;
    mi_prof_begin(st, MI_FAM_WGRAD, 2.0 * KS * KS * (double)a.N * a.Ho * a.Wo * a.C * a.K,
                  4.0 * ((double)a.N * a.C * a.H * a.W + (double)a.N * a.K * a.Ho * a.Wo + (double)KS * KS * a.C * a.K));
    hipLaunchKernelGGL((wgrad_kernel<KS, S, TKL, TC, NXR>), grid, dim3(256), lds, st, x, dy, part, a);
    mi_prof_end(st);
    MI_LAUNCH_CHECK("wgrad_kernel");
    return 0;
}

// ------------------------------------------------------------------------------------------
// 1x1 convolutions live in kernels_gemm.hip
int mi_conv1x1_fwd(hipStream_t st, const float *x, const float *w, float *y, int N, int C, int P, int K);
int mi_conv1x1_dgrad(hipStream_t st, const float *w, const float *dy, float *dx, const float *addend, int N, int C,
                     int P, int K);
int mi_conv1x1_wgrad(hipStream_t st, mid_workspace *ws, const float *x, const float *dy, float *dw, int N, int C, int P,
                     int K);
size_t mi_conv1x1_wgrad_part_floats(int N, int C, int P, int K);
// kernels_igemm.hip: implicit GEMM on fp32 MFMA (1x1 layers, the 3x3/s2 projection shortcuts; policy in mi_igemm_supported)
enum { IGOP_FWD = 0, IGOP_DGRAD = 1, IGOP_WGRAD = 2 };
int mi_igemm_supported(int op, int N, int C, int H, int K, int k, int stride);
size_t mi_igemm_part_floats(int N, int C, int H, int K, int k, int stride);
size_t mi_igemm_tail_floats(void);
int mi_igemm_fwd(hipStream_t st, mid_workspace *ws, const float *x, const float *w, float *y, int N, int C, int H, int K, int k,
                 int stride, mid_bn_parts *parts);
int mi_igemm_dgrad(hipStream_t st, mid_workspace *ws, const float *w, const float *dy, float *dx, const float *addend, int N,
                   int C, int H, int K, int k, int stride, mid_bn_bwd_parts *fz = nullptr);
int mi_igemm_wgrad(hipStream_t st, mid_workspace *ws, const float *x, const float *dy, float *dw, int N, int C, int H, int K, int k,
                   int stride);

extern "C" {

/* re-laid weights, followed by the partial tiles of igemm's reduction-sliced tail workgroups (compute stream only) */
size_t mid_conv_ws_wt_floats(int C, int K, int k) { return (size_t)k * k * C * K + (k <= 3 ? mi_igemm_tail_floats() : 0); }

size_t mid_conv_ws_part_floats(int N, int C, int H, int K, int k, int stride) {
    if (mi_igemm_supported(IGOP_WGRAD, N, C, H, K, k, stride)) return mi_igemm_part_floats(N, C, H, K, k, stride);
    if (k == 1) return mi_conv1x1_wgrad_part_floats(N, C, H * H, K);
    WbPlan pb;
    if (!wgradB_plan(N, C, H, K, k, stride, &pb)) return pb.splits > 1 ? (size_t)pb.splits * K * C * 9 : 0;
    WgPlan p;
    if (wgrad_plan(N, C, H, K, k, stride, &p)) return 0;
    return (size_t)p.splits * K * C * k * k;
}

int mid_conv_fwd(mid_stream s, mid_workspace *ws, const float *x, const float *w, float *y, int N, int C, int H, int K,
                 int k, int stride) {
    return mid_conv_fwd_stats(s, ws, x, w, y, N, C, H, K, k, stride, nullptr);
}
int mid_conv_fwd_stats(mid_stream s, mid_workspace *ws, const float *x, const float *w, float *y, int N, int C, int H, int K,
                       int k, int stride, mid_bn_parts *parts) {
    hipStream_t st = (hipStream_t)s;
    if (parts) parts->nparts = 0;
    if (mi_igemm_supported(IGOP_FWD, N, C, H, K, k, stride)) return mi_igemm_fwd(st, ws, x, w, y, N, C, H, K, k, stride, parts);
    if (k == 1 && stride == 1) return mi_conv1x1_fwd(st, x, w, y, N, C, H * H, K);
    if (!((k == 3 && (stride == 1 || stride == 2)) || (k == 7 && stride == 2)) || H % stride) {
        mi_record_error("mid_conv_fwd", "unsupported kernel/stride");
        return -2;
    }
    if (!ws || ws->wt_floats < (size_t)k * k * C * K) { mi_record_error("mid_conv_fwd", "workspace too small"); return -3; }
    unsigned char src[49];
    for (int t = 0; t < k * k; t++) src[t] = (unsigned char)t;
    if (launch_wt(st, w, ws->wt, C, K, k * k, k, C, 0, src)) return -1;
    DConvArgs a = {};
    a.N = N; a.Cin = C; a.Hin = H; a.Win = H; a.Cout = K; a.Hof = H / stride; a.Wof = H / stride;
    a.Hsub = a.Hof; a.Wsub = a.Wof; a.osy = a.osx = 1; a.oy0 = a.ox0 = 0;
    a.sy = a.sx = stride; a.offy = a.offx = -(k / 2);
    return launch_dconv(st, x, ws->wt, y, nullptr, a, k, k);
}

/* dgrad + the reduction pass of the batch-norm backward its output feeds, where the launch takes the fp32 implicit-GEMM route with
 * stride 1: fz->nparts > 0 on return says the epilogue did it (dx then holds the gated gradient); 0 = plain dgrad was run */
int mid_conv_dgrad_bn_f32(mid_stream s, mid_workspace *ws, const float *w, const float *dy, float *dx, const float *addend,
                          int N, int C, int H, int K, int k, int stride, mid_bn_bwd_parts *fz) {
    if (fz) fz->nparts = 0;
    if (fz && stride == 1 && mi_igemm_supported(IGOP_DGRAD, N, C, H, K, k, stride))
        return mi_igemm_dgrad((hipStream_t)s, ws, w, dy, dx, addend, N, C, H, K, k, stride, fz);
    return mid_conv_dgrad(s, ws, w, dy, dx, addend, N, C, H, K, k, stride);
}
int mid_conv_dgrad(mid_stream s, mid_workspace *ws, const float *w, const float *dy, float *dx, const float *addend,
                   int N, int C, int H, int K, int k, int stride) {
    hipStream_t st = (hipStream_t)s;
    if (mi_igemm_supported(IGOP_DGRAD, N, C, H, K, k, stride)) return mi_igemm_dgrad(st, ws, w, dy, dx, addend, N, C, H, K, k, stride);
    if (k == 1 && stride == 1) return mi_conv1x1_dgrad(st, w, dy, dx, addend, N, C, H * H, K);
    if (k != 3 || (stride != 1 && stride != 2) || H % stride) { mi_record_error("mid_conv_dgrad", "unsupported kernel/stride"); return -2; }
    if (!ws || ws->wt_floats < (size_t)9 * C * K) { mi_record_error("mid_conv_dgrad", "workspace too small"); return -3; }
    const int Ho = H / stride;
    DConvArgs a = {};
    a.N = N; a.Cin = K; a.Hin = Ho; a.Win = Ho; a.Cout = C; a.Hof = H; a.Wof = H;
    unsigned char src[49];
    if (stride == 1) {
        for (int tr = 0; tr < 3; tr++)
            for (int tc = 0; tc < 3; tc++) src[tr * 3 + tc] = (unsigned char)((2 - tr) * 3 + (2 - tc));
        if (launch_wt(st, w, ws->wt, K, C, 9, 3, C, 1, src)) return -1;
        a.Hsub = H; a.Wsub = H; a.osy = a.osx = 1; a.oy0 = a.ox0 = 0; a.sy = a.sx = 1; a.offy = a.offx = -1;
        return launch_dconv(st, dy, ws->wt, dx, addend, a, 3, 3);
    }
    // stride 2: input pixel (2i+pa, 2j+pb); pa==0 -> r=1 (oh=i); pa==1 -> r=2 (oh=i), r=0 (oh=i+1).
    // The four parity classes write disjoint pixels, so they run CONCURRENTLY on four HIP streams (fork/join with
    // events): the 1- and 2-tap classes fill the CUs the 4-tap class leaves idle in its tail.
    static hipStream_t aux[3];
    static hipEvent_t ev_fork, ev_join[3];
    static int aux_ready = 0, use_aux = -1;
    if (use_aux < 0) { const char *e = getenv("RESNET_MI_DGRAD_STREAMS"); use_aux = e ? atoi(e) : 1; }
    if (use_aux && !aux_ready) {
        for (int q = 0; q < 3; q++) { (void)hipStreamCreateWithFlags(&aux[q], hipStreamNonBlocking); (void)hipEventCreateWithFlags(&ev_join[q], hipEventDisableTiming); }
        (void)hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming);
        aux_ready = 1;
    }
    float *wt = ws->wt;
    // weights of all four classes first (main stream), then fork
    float *wts[4];
    int cls = 0;
    for (int pa = 0; pa < 2; pa++)
        for (int pb = 0; pb < 2; pb++, cls++) {
            const int ntr = pa ? 2 : 1, ntc = pb ? 2 : 1;
            for (int tr = 0; tr < ntr; tr++)
                for (int tc = 0; tc < ntc; tc++) {
                    const int r = pa ? 2 - 2 * tr : 1, sc = pb ? 2 - 2 * tc : 1;
                    src[tr * ntc + tc] = (unsigned char)(r * 3 + sc);
                }
            if (launch_wt(st, w, wt, K, C, ntr * ntc, 3, C, 1, src)) return -1;
            wts[cls] = wt;
            wt += (size_t)ntr * ntc * C * K;
        }
    // the four concurrent class kernels are timed as ONE launch of the dgrad operator (fork .. join on the main stream)
    mi_prof_begin(st, MI_FAM_DCONV, 2.0 * 9 * (double)N * Ho * Ho * C * K,
                  4.0 * ((double)N * K * Ho * Ho + 9.0 * C * K + (double)N * C * H * H * (addend ? 2 : 1)));
    if (use_aux) {
        (void)hipEventRecord(ev_fork, st);
        for (int q = 0; q < 3; q++) (void)hipStreamWaitEvent(aux[q], ev_fork, 0);
    }
    cls = 0;
    // heaviest class (2x2 taps) first on the main stream
    for (int pa = 1; pa >= 0; pa--)
        for (int pb = 1; pb >= 0; pb--, cls++) {
            const int ntr = pa ? 2 : 1, ntc = pb ? 2 : 1;
            hipStream_t sq = (use_aux && cls > 0) ? aux[cls - 1] : st;
            DConvArgs b = a;
            b.Hsub = Ho; b.Wsub = Ho; b.osy = b.osx = 2; b.oy0 = pa; b.ox0 = pb; b.sy = b.sx = 1; b.offy = b.offx = 0;
            int rc = launch_dconv(sq, dy, wts[pa * 2 + pb], dx, addend, b, ntr, ntc, false);
            if (rc) return rc;
            if (use_aux && cls > 0) { (void)hipEventRecord(ev_join[cls - 1], sq); (void)hipStreamWaitEvent(st, ev_join[cls - 1], 0); }
        }
    mi_prof_end(st);
    return 0;
}

int mid_conv_wgrad(mid_stream s, mid_workspace *ws, const float *x, const float *dy, float *dw, int N, int C, int H,
                   int K, int k, int stride) {
    hipStream_t st = (hipStream_t)s;
    if (mi_igemm_supported(IGOP_WGRAD, N, C, H, K, k, stride)) return mi_igemm_wgrad(st, ws, x, dy, dw, N, C, H, K, k, stride);
    if (k == 1 && stride == 1) return mi_conv1x1_wgrad(st, ws, x, dy, dw, N, C, H * H, K);
    WbPlan pb;
    if (!wgradB_plan(N, C, H, K, k, stride, &pb)) {
        const size_t wsz9 = (size_t)K * C * 9;
        if (pb.splits > 1 && (!ws || ws->part_floats < wsz9 * pb.splits)) { mi_record_error("mid_conv_wgrad", "workspace too small"); return -3; }
        WcArgs b;
        b.N = N; b.C = C; b.H = H; b.W = H; b.K = K; b.Ho = H / stride; b.Wo = H / stride; b.pad = 1;
        b.NPX = pb.npx; b.rows = pb.rows; b.segw = pb.segw; b.QPR = pb.qpr; b.DP = pb.rows * pb.qpr * 4;
        b.cpp = pb.cpp; b.chunks_total = pb.chunks; b.chunks_per_split = pb.cps;
        b.PR = pb.PR; b.PW = pb.PW; b.cwlog2 = pb.cwlog2; b.pitch = pb.pitch; b.jcnt = pb.jcnt;
        b.xs_floats = (64 * pb.pitch + 16 + 3) & ~3;
        b.fd_cpp = make_fastdiv(pb.cpp); b.fd_Wo = make_fastdiv(b.Wo); b.fd_PR = make_fastdiv(pb.PR);
        b.fd_Q4 = make_fastdiv(pb.qpr * 4); b.fd_PW = make_fastdiv(pb.PW);
        b.part_stride = wsz9;
        float *outp = pb.splits == 1 ? dw : ws->part;
        dim3 grid(K / 32, C / 64, pb.splits);
        const size_t lds = (size_t)2 * (b.xs_floats + 32 * WC_DPK) * 4;
        mi_prof_begin(st, MI_FAM_WGRAD, 2.0 * 9 * (double)N * b.Ho * b.Wo * C * K,
                      4.0 * ((double)N * C * H * H + (double)N * K * b.Ho * b.Wo + 9.0 * C * K));
        if (stride == 1) hipLaunchKernelGGL((wgradC_kernel<1>), grid, dim3(256), lds, st, x, dy, outp, b);
        else hipLaunchKernelGGL((wgradC_kernel<2>), grid, dim3(256), lds, st, x, dy, outp, b);
        mi_prof_end(st);
        MI_LAUNCH_CHECK("wgradC_kernel");
        if (pb.splits > 1) return mi_launch_split_reduce(st, outp, dw, (long)wsz9, pb.splits, wsz9);
        return 0;
    }
    WgPlan p;
    if (wgrad_plan(N, C, H, K, k, stride, &p) || H % stride) { mi_record_error("mid_conv_wgrad", "unsupported shape"); return -2; }
    const size_t wsz = (size_t)K * C * k * k;
    if (!ws || ws->part_floats < wsz * p.splits) { mi_record_error("mid_conv_wgrad", "workspace too small"); return -3; }
    WgArgs a;
    a.N = N; a.C = C; a.H = H; a.W = H; a.K = K; a.Ho = H / stride; a.Wo = H / stride; a.pad = k / 2;
    a.RPC = p.rpc; a.RG = p.rg; a.SEGW = p.segw; a.SG = p.sg;
    a.chunks_total = p.chunks; a.chunks_per_split = p.cps;
    a.fd_SG = make_fastdiv(p.sg); a.fd_RG = make_fastdiv(p.rg);
    a.part_stride = wsz;
    float *part = p.splits == 1 ? dw : ws->part;
    dim3 grid(K / (64 * p.tkl), mi_cdiv(C, 4 * p.tc), p.splits);
    int rc = -2;
    if (k == 3 && stride == 1 && p.tkl == 2) rc = launch_wgrad_t<3, 1, 2, 2, 9>(st, x, dy, part, a, grid);
    else if (k == 3 && stride == 1 && p.tkl == 1) rc = launch_wgrad_t<3, 1, 1, 2, 9>(st, x, dy, part, a, grid);
    else if (k == 3 && stride == 2 && p.tkl == 2) rc = launch_wgrad_t<3, 2, 2, 2, 9>(st, x, dy, part, a, grid);
    else if (k == 3 && stride == 2 && p.tkl == 1) rc = launch_wgrad_t<3, 2, 1, 2, 9>(st, x, dy, part, a, grid);
    else if (k == 7 && stride == 2) rc = launch_wgrad_t<7, 2, 1, 1, 7>(st, x, dy, part, a, grid);
    if (rc) return rc;
    if (p.splits > 1) return mi_launch_split_reduce(st, part, dw, (long)wsz, p.splits, wsz);
    return 0;
}
}
