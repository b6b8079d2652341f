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
    int goff[JM];
#pragma unroll
    for (int jj = 0; jj < JM; jj++) {
        const int pos = tid + jj * 448;
        int g = -1;
        if (jj < a.jcnt && pos < plane) {
            const uint32_t qr = fd_div((uint32_t)pos, a.fd_PW);
            const int iw = pos - (int)qr * a.PW + a.offx;
            const int t = tbl[qr];
            if (t >= 0 && iw >= 0 && iw < a.Win) g = t + iw;
        }
        goff[jj] = g;
    }
    float regs[CCM][JM];
    auto issue = [&](int c0) {
#pragma unroll
        for (int c = 0; c < CCM; c++) {
            const bool cok = c < a.CC && c0 + c < a.Cin;
            const float *src = in + (size_t)(c0 + c) * HW;
#pragma unroll
            for (int jj = 0; jj < JM; jj++) {
                float v = 0.f;
                if (cok && goff[jj] >= 0) v = src[goff[jj]];
                regs[c][jj] = v;
            }
        }
    };
    auto stash = [&](float *buf) {
#pragma unroll
        for (int c = 0; c < CCM; c++) {
            if (c < a.CC) {
#pragma unroll
                for (int jj = 0; jj < JM; jj++) {
                    const int pos = tid + jj * 448;
                    if (jj < a.jcnt && pos < plane) buf[c * plane + pos] = regs[c][jj];
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
    if (jm <= 2) { if (tk == 32) DCL(32, 2, 8); else DCL(16, 2, 8); }
    else if (jm <= 5) { if (tk == 32) DCL(32, 5, 3); else DCL(16, 5, 3); }
    else { if (tk == 32) DCL(32, 7, 3); else DCL(16, 7, 3); }
#undef DCL
    return 0;
}

// one tap-kernel launch; ntr x ntc taps, weights already re-laid in wT
static int launch_dconv(hipStream_t st, const float *in, const float *wT, float *out, const float *addend, DConvArgs a,
                        int ntr, int ntc) {
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
    int cc = 5120 / a.plane; // two LDS buffers of <= 20 KB each
    if (cc < 1) cc = 1;
    if (cc > ccm) cc = ccm;
    if (cc > a.Cin) cc = a.Cin;
    a.CC = cc;
    const size_t lds = (size_t)(a.tbl_pad + 2 * (size_t)cc * a.plane) * 4;
    if (lds > 64 * 1024) { mi_record_error("dconv", "LDS patch too large for this shape"); return -2; }
    a.fd_Wsub = make_fastdiv(a.Wsub); a.fd_Hsub = make_fastdiv(a.Hsub);
    a.fd_Hq = make_fastdiv(a.Hq); a.fd_PW = make_fastdiv(a.PW);
    const int tk = (a.Cout % 32 == 0) ? 32 : 16;
    if (a.Cout % 16 != 0) { mi_record_error("dconv", "channel count must be a multiple of 16"); return -2; }
    dim3 grid(mi_cdiv(a.total_pix, PIX), a.Cout / tk);
    // algorithmic work of this launch: 2*taps MACs per (pixel, cin, cout); bytes = input + weights + output once
    const double fl = 2.0 * ntr * ntc * (double)a.total_pix * a.Cin * a.Cout;
    const double by = 4.0 * ((double)a.N * a.Cin * a.Hin * a.Win / (a.osy * a.osx) + (double)ntr * ntc * a.Cin * a.Cout +
                             (double)a.total_pix * a.Cout * (addend ? 2 : 1));
    mi_prof_begin(st, MI_FAM_DCONV, fl, by);
    int rc = -2;
#define DC(NTR_, NTC_) if (ntr == NTR_ && ntc == NTC_) rc = launch_dconv_t<NTR_, NTC_>(st, grid, lds, tk, a.jcnt, in, wT, out, addend, a);
    DC(3, 3) DC(7, 7) DC(1, 1) DC(1, 2) DC(2, 1) DC(2, 2)
#undef DC
    mi_prof_end(st);
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

template <int KS, int S, int TKL, int TC, int NXR>
__global__ void __launch_bounds__(256)
wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ part, const WgArgs a) {
    constexpr int KT = 64 * TKL, T = KS * KS, RPCMAX = (NXR - KS) / S + 1;
    extern __shared__ float dyS[]; // [KT][WG_LDP]
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
    for (int ch = ch_beg; ch < ch_end; ch++) {
        const uint32_t t1 = fd_div((uint32_t)ch, a.fd_SG);
        const int sg = ch - (int)t1 * a.SG;
        const uint32_t n = fd_div(t1, a.fd_RG);
        const int rg = (int)(t1 - n * a.RG);
        const int oh0 = rg * a.RPC, rows = min(a.RPC, a.Ho - oh0);
        const int ow0 = sg * a.SEGW, segw = min(a.SEGW, a.Wo - ow0);
        const int npx = rows * segw; // rows==1 or segw==Wo: pixels of a k-plane are one contiguous run
        __syncthreads();
        for (int kk = wave; kk < KT; kk += 4) {
            const float *src = dy + (((size_t)n * a.K + kb + kk) * a.Ho + oh0) * a.Wo + ow0;
            for (int px = lane; px < npx; px += 64) dyS[kk * WG_LDP + px] = src[px];
        }
        // input row segments, one element per lane
        float xrow[TC][NXR];
        const int nxr = (rows - 1) * S + KS;
        const int iw = ow0 * S - a.pad + lane;
        const bool colok = iw >= 0 && iw < a.W && lane < (segw - 1) * S + KS;
#pragma unroll
        for (int c = 0; c < TC; c++) {
            const int cch = cb + c;
#pragma unroll
            for (int xr = 0; xr < NXR; xr++) {
                const int ih = oh0 * S - a.pad + xr;
                float v = 0.f;
                if (xr < nxr && cch < a.C && ih >= 0 && ih < a.H && colok)
                    v = x[(((size_t)n * a.C + cch) * a.H + ih) * a.W + iw];
                xrow[c][xr] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RPCMAX; r++) {
            if (r < rows) {
                for (int ow = 0; ow < segw; ow++) {
                    float d[TKL];
#pragma unroll
                    for (int u = 0; u < TKL; u++) d[u] = dyS[(lane + 64 * u) * WG_LDP + r * segw + ow];
#pragma unroll
                    for (int c = 0; c < TC; c++)
#pragma unroll
                        for (int tr = 0; tr < KS; tr++)
#pragma unroll
                            for (int tc = 0; tc < KS; tc++) {
                                const float xs = __int_as_float(
                                    __builtin_amdgcn_readlane(__float_as_int(xrow[c][r * S + tr]), ow * S + tc));
#pragma unroll
                                for (int u = 0; u < TKL; u++)
                                    acc[u][c][tr * KS + tc] = fmaf(xs, d[u], acc[u][c][tr * KS + tc]);
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

__global__ void split_reduce_kernel(const float *__restrict__ part, float *__restrict__ out, long n, int splits,
                                    size_t stride) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < splits; z++) s += part[(size_t)z * stride + i];
        out[i] = s;
    }
}

int mi_launch_split_reduce(hipStream_t st, const float *part, float *out, long n, int splits, size_t stride) {
    int blocks = mi_cdiv(n, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_reduce_kernel, dim3(blocks), dim3(256), 0, st, part, out, n, splits, stride);
    MI_LAUNCH_CHECK("split_reduce_kernel");
    return 0;
}

struct WgPlan { int tkl, tc, rpc, rg, segw, sg, splits, cps, chunks; };
static int wgrad_plan(int N, int C, int H, int K, int k, int stride, WgPlan *p) {
    if (!((k == 3 && (stride == 1 || stride == 2)) || (k == 7 && stride == 2))) return -1;
    if (K % 64 != 0) return -1;
    const int Ho = H / stride, Wo = Ho;
    const int rpcmax = (k == 7) ? 1 : 7;
    p->tkl = (k == 7) ? 1 : (K >= 128 ? 2 : 1);
    p->tc = (k == 7) ? 1 : (stride == 2 ? 1 : 2);
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
    const size_t lds = (size_t)64 * TKL * WG_LDP * 4;
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

extern "C" {

size_t mid_conv_ws_wt_floats(int C, int K, int k) { return k == 1 ? 0 : (size_t)k * k * C * K; }

size_t mid_conv_ws_part_floats(int N, int C, int H, int K, int k, int stride) {
    if (k == 1) return mi_conv1x1_wgrad_part_floats(N, C, H * H, K);
    WgPlan p;
    if (wgrad_plan(N, C, H, K, k, stride, &p)) return 0;
    return (size_t)p.splits * K * C * k * k;
}

int mid_conv_fwd(mid_stream s, mid_workspace *ws, const float *x, const float *w, float *y, int N, int C, int H, int K,
                 int k, int stride) {
    hipStream_t st = (hipStream_t)s;
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

int mid_conv_dgrad(mid_stream s, mid_workspace *ws, const float *w, const float *dy, float *dx, const float *addend,
                   int N, int C, int H, int K, int k, int stride) {
    hipStream_t st = (hipStream_t)s;
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
    // stride 2: input pixel (2i+pa, 2j+pb); pa==0 -> r=1 (oh=i); pa==1 -> r=2 (oh=i), r=0 (oh=i+1)
    float *wt = ws->wt;
    for (int pa = 0; pa < 2; pa++)
        for (int pb = 0; pb < 2; pb++) {
            const int ntr = pa ? 2 : 1, ntc = pb ? 2 : 1;
            for (int tr = 0; tr < ntr; tr++)
                for (int tc = 0; tc < ntc; tc++) {
                    const int r = pa ? 2 - 2 * tr : 1, sc = pb ? 2 - 2 * tc : 1;
                    src[tr * ntc + tc] = (unsigned char)(r * 3 + sc);
                }
            if (launch_wt(st, w, wt, K, C, ntr * ntc, 3, C, 1, src)) return -1;
            DConvArgs b = a;
            b.Hsub = Ho; b.Wsub = Ho; b.osy = b.osx = 2; b.oy0 = pa; b.ox0 = pb; b.sy = b.sx = 1; b.offy = b.offx = 0;
            int rc = launch_dconv(st, dy, wt, dx, addend, b, ntr, ntc);
            if (rc) return rc;
            wt += (size_t)ntr * ntc * C * K;
        }
    return 0;
}

int mid_conv_wgrad(mid_stream s, mid_workspace *ws, const float *x, const float *dy, float *dw, int N, int C, int H,
                   int K, int k, int stride) {
    hipStream_t st = (hipStream_t)s;
    if (k == 1 && stride == 1) return mi_conv1x1_wgrad(st, ws, x, dy, dw, N, C, H * H, K);
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
    else if (k == 3 && stride == 2 && p.tkl == 2) rc = launch_wgrad_t<3, 2, 2, 1, 15>(st, x, dy, part, a, grid);
    else if (k == 3 && stride == 2 && p.tkl == 1) rc = launch_wgrad_t<3, 2, 1, 1, 15>(st, x, dy, part, a, grid);
    else if (k == 7 && stride == 2) rc = launch_wgrad_t<7, 2, 1, 1, 7>(st, x, dy, part, a, grid);
    if (rc) return rc;
    if (p.splits > 1) return mi_launch_split_reduce(st, part, dw, (long)wsz, p.splits, wsz);
    return 0;
}
}
