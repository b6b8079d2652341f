// kernels_igemm.hip -- convolutions as an im2col-free implicit GEMM on the CDNA4 matrix cores, fp32
// (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate).
//
// Which layers come here is a policy of mi_igemm_supported() (RESNET_MI_IGEMM, see there).  By default every 3x3 and 1x1
// convolution whose channel counts tile, forward / dgrad / wgrad -- among them the reference's PROJECTION shortcuts, which
// are 3x3 / stride-2 / pad-1 (resnet.cu:884-889, 1693-1700), not the usual 1x1: 256->512 @56, 512->1024 @28,
// 1024->2048 @14 hold 57% of the network's multiply-adds.  The stem (C = 3), shapes that do not tile and, with
// RESNET_MI_IGEMM=1, the bottleneck's own 3x3 convolutions run on the direct VALU kernels of kernels_conv.hip; FC and
// (RESNET_MI_IGEMM <= 1) 1x1 forward / dgrad on gemm_mfma_kernel.
// The "im2col" matrix exists only as the 32 x 128 tile of LDS that the current k-step multiplies, gathered straight from
// the NCHW tensor.
//
//   forward  Y[n][k][ho][wo]  = sum_{t=(r,s)} sum_c W[k][c][r][s] X[n][c][S*ho-pad+r][S*wo-pad+s]
//            M = K (out channels), N = (n,ho,wo), reduction (t,c) tap-major; A = weights re-laid [t][c][k]
//   dgrad    S=1: dX[n][c][h][w] = sum_{t,k} W[k][c][r][s] dY[n][k][h+pad-r][w+pad-s]
//            S=2: dX[n][c][2a+ph][2b+pw] = sum over the 1/2/2/4 taps that reach parity class (ph,pw); classes = blockIdx.y
//            M = C, N = (n, pixel), reduction (tap,k); A = weights re-laid [t][k][c] (1x1: the KC tensor itself)
//   wgrad    dW[k][c][t] = sum_{n,ho,wo} dY[n][k][ho][wo] X[n][c][S*ho-pad+r][S*wo-pad+s]
//            M = K, N = (t,c), reduction (n,ho,wo) split over blockIdx.y; partials [split][t][k][c], reduced (and
//            transposed to KCRS) by igemm_wgrad_reduce_kernel in a fixed order (deterministic).  64 input channels:
//            3x3 with TWO taps per 128-column tile, 1x1 as the transposed product (rows = input channels)
//   forward also leaves per-tile batch-norm statistics of its output (count, mean, M2 per channel and column group), so
//   that BN does not read the tensor a third time (RESNET_MI_BNFUSE)
//
// Tile BM x 128 x 32 with BM = 128 (4 waves of 64x64) or 64 (4 waves of 64x32), 256 threads.  LDS double-buffered, one
// barrier per k-step; tile height and a reduction-sliced last round chosen against workgroup-count quantisation
// (igemm_pick_bm, igemm_tail_plan).  Staging is software-pipelined over the 16 MFMA groups of a k-step: the LDS stores of tile i+1 (held
// in registers) ride with groups 0-7, the global loads of tile i+2 with groups 8-15.  What made this fast, in order of
// effect (b7 projection wgrad 96 -> 117 TFLOP/s; MFMA-only ceiling of this loop 138):
//   1. loads as `global_load v, v_off32, s[base]`: wave-uniform 64-bit base advanced on the scalar unit + 32-bit per-lane
//      byte offset.  Per-lane 64-bit addresses cost two v_lshl_add_u64 per load and twice the address payload; VMEM issue,
//      not memory latency, was the limiter (the same loads from an L2-resident slice were no faster);
//   2. stores and loads UNCONDITIONAL (the last two iterations move data nobody reads) and the loop entered with no load
//      in flight (one drain after the prologue): the vmcnt a store waits for is the minimum over all paths into it, and a
//      branch or the compiler-reordered prologue loads made every store drain most of the loads in flight;
//   3. no branches in the loop body at all (fd_div_ge2, arithmetic selects);
//   4. fragments of step k2+2 read ahead of the MFMAs of step k2 (sched_barrier pins it).
// Out-of-image taps load from the (always valid) centre tap's address and are zeroed when stored to LDS, so no load is
// predicated.  Block ids are re-mapped so that each XCD walks a contiguous range of tiles, M-tiles fastest: the blocks that
// share a gathered pixel tile share an L2.
#include "mi_common.hpp"
#include <type_traits>
#include "mi_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float pf4 __attribute__((ext_vector_type(4)));

enum { IG_FWD = 0, IG_DGRAD = 1, IG_WGRAD = 2 };

struct IgArgs {
    int N, C, K, H, W, Ho, Wo; // KS x KS, stride S, pad KS/2: Ho = H/S, Wo = W/S
    int HW, P;                 // H*W, Ho*Wo
    int ncols;                 // fwd/dgrad: N*P columns
    int mtiles;                // number of BM-row tiles
    int tiles;                 // mtiles * column tiles: blocks per blockIdx.y
    int ctiles;                // wgrad: C/128
    int klen;                  // wgrad: reduction length per split (multiple of 32)
    FastDiv fdP, fdWo, fdM;    // fdM: division by mtiles
    // fwd / dgrad-s1 tail slicing (igemm_tail_plan): tiles >= full are cut into tsplit reduction slices of tklen k-steps
    // whose partial tiles go to tailbuf and are summed in slice order by igemm_tail_reduce_kernel
    int full, tsplit, tklen, cpt; // cpt: k-steps per tap
    FastDiv fdTs, fdCpt;
    float *tailbuf;
    // forward: batch-norm statistics of the output fused into the epilogue (three planes [bn_np][K] of count, mean, M2 over
    // the columns of a (tile, 128/PPT-column group); PPT = 2 for 128-row tiles, 4 for 64-row tiles); nullptr = off
    float *bn_part;
    int bn_np;
    // dgrad (stride 1) that also does the REDUCTION pass of the batch-norm backward its output feeds (as the bf16 kernel does): the
    // stored gradient is gated by bnb_mask > 0, and per (64-row wave tile, channel) the sums of g and g (x - mean) over the wave's
    // columns go to bnb_part (two planes [bnb_np][C]); nullptr = off
    const float *bnb_x, *bnb_mask, *bnb_mean;
    float *bnb_part;
    int bnb_np;
};

// sum over the 32 lanes of each half of a wave (lanes 0-31 / 32-63), on the vector ALU alone (DPP: no LDS crossbar traffic).  The
// total of a half ends up in its lanes 16..31 (row_bcast15 hands a row's last lane to the next row).
__device__ __forceinline__ float ig_half_sum32(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)); // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)); // row_mirror: every lane = its row's 16
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, true)); // row_bcast15 into rows 1 and 3
    return v;
}

// (count, mean, M2) of one value per lane over the first nv lanes of each 32-lane half: shifted sums about the half's
// first value (no cancellation), butterfly over the half.  Result valid in every lane of the half.
__device__ __forceinline__ void ig_half_stats(float x, int nv, int l31, float &mean, float &m2) {
    const float s = __shfl(x, (int)(threadIdx.x & 32), 64);
    const float d = l31 < nv ? x - s : 0.f;
    float sd = d, sq = d * d;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { sd += __shfl_xor(sd, o, 64); sq += __shfl_xor(sq, o, 64); }
    const float inv = nv > 0 ? 1.0f / (float)nv : 0.f;
    mean = s + sd * inv;
    m2 = fmaxf(sq - sd * sd * inv, 0.f);
    if (nv <= 0) { mean = 0.f; m2 = 0.f; }
}
// Chan merge of two groups (na, ma, qa) <- (nb, mb, qb)
__device__ __forceinline__ void ig_merge(float &na, float &ma, float &qa, float nb, float mb, float qb) {
    const float n = na + nb;
    if (n > 0.f) {
        const float dlt = mb - ma, f = nb / n;
        ma = ma + dlt * f;
        qa = qa + qb + dlt * dlt * na * f;
    }
    na = n;
}

#define IG_BK 32
#ifdef IG_STAMP /* diagnostic build only (tools/variant.sh): per-workgroup s_memtime stamps of the kernel's phases */
__device__ unsigned long long ig_stamps[8 * 16384];
#define IG_T(slot) do { if (threadIdx.x == 0 && blockIdx.x < 16384 && blockIdx.y == 0) ig_stamps[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define IG_T(slot) do { } while (0)
#endif
#ifndef IG_ABLATE
#define IG_ABLATE 0 /* experiments only: 1 = no staging after the first tile, 4 = no global loads */
#endif

template <int MODE, int KS, int S, int WMW, bool VBP = false>
__global__ void __launch_bounds__(256)
igemm_kernel(const float *__restrict__ Aop, const float *__restrict__ Bop, float *__restrict__ Out,
             const float *__restrict__ addend, const IgArgs g) {
    constexpr int BM = 64 * WMW;                 // rows per block
    constexpr int TN = WMW == 2 ? 2 : 1;         // 32-column MFMA tiles per wave (wave tile 64 x 32*TN)
    constexpr int WNC = 32 * TN;                 // columns per wave
    constexpr int T = KS * KS, PAD = KS / 2;
    constexpr int LDA = (MODE == IG_WGRAD) ? BM + 1 : BM + 4; // +1: conflict-free transposed scalar stores; +4: 16-B rows
    constexpr int LDB = (MODE == IG_WGRAD) ? 129 : 132;
    constexpr int NA4 = BM / 32;                 // fwd/dgrad: float4 loads of A per thread and tile
    constexpr int NAS = BM / 8;                  // wgrad: scalar loads of A per thread and tile
    // 1x1 forward / dgrad with Ho*Wo % 4 == 0: four consecutive columns are four consecutive pixels of one image, so the
    // B operand is staged with 16-byte loads / ds_write_b128 (4 per thread and k-step instead of 16 scalars)
    constexpr bool VB = VBP && KS == 1 && MODE != IG_WGRAD;
    // wgrad with 64 input channels: a 128-column tile holds TWO taps x 64 channels (columns 0-63 tap 2*ct, 64-127 tap 2*ct+1)
    constexpr bool CT64 = VBP && MODE == IG_WGRAD;
    extern __shared__ __attribute__((aligned(16))) float ig_smem[];
    float *As = ig_smem;                   // [2][32][LDA]
    float *Bs = ig_smem + 2 * IG_BK * LDA; // [2][32][LDB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WMW == 2 ? wave >> 1 : 0, wn = WMW == 2 ? wave & 1 : wave;

    IG_T(0);
    // ---- block -> tile (XCD-contiguous, M-tiles fastest); blocks past g.full are reduction slices of the last tiles ----
    uint32_t L = blockIdx.x;
    int tail_id = -1, it0 = 0, nt_slice = 0;
    uint32_t wg_split = blockIdx.y;
    if (MODE != IG_WGRAD && g.tsplit > 1 && L >= (uint32_t)g.full) {
        tail_id = (int)(L - (uint32_t)g.full);
        const uint32_t tl = fd_div((uint32_t)tail_id, g.fdTs);
        const int z = tail_id - (int)tl * g.tsplit;
        L = (uint32_t)g.full + tl;
        it0 = z * g.tklen;
        nt_slice = g.tklen;
    } else if (MODE == IG_WGRAD && gridDim.y >= 8) {
        // wgrad with at least as many splits as XCDs: (tile, split) from a SPLIT-major list cut into eight contiguous pieces, one per XCD
        // (workgroups go to the XCDs round-robin by linear id): all the tiles that reduce over one stretch of pixels run on one XCD at
        // about the same time, so each operand byte is fetched from HBM by one L2
        const uint32_t T_ = gridDim.x, tot = T_ * gridDim.y, per = tot >> 3;
        uint32_t V = blockIdx.y * T_ + blockIdx.x;
        if (V < per * 8) V = (V & 7) * per + (V >> 3);
        wg_split = V / T_;
        L = V - wg_split * T_;
    } else {
        const uint32_t lim = MODE != IG_WGRAD && g.tsplit > 1 ? (uint32_t)g.full : (uint32_t)g.tiles;
        const uint32_t per = lim >> 3;
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

    // ---- per-thread staging state ----
    // FWD / DGRAD: A = NA4 x float4 along M, B = 16 gathered scalars of ONE column (rows 16*b_k + q: neighbouring rows
    //              pair up into ds_write2_b32)
    // WGRAD:       A = NAS, B = 16 scalars of ONE reduction index (tid & 31), rows/cols (tid >> 5) + 8q
    // Global addresses are a wave-uniform 64-bit base (SGPRs, advanced on the scalar unit) plus a 32-bit per-lane byte
    // offset; every tensor is < 2^32 bytes (mi_igemm_supported).
    uint32_t a_lane = 0, b_lane = 0;
    uint32_t mask = 0; // bit t: (class) tap t of this thread's column lies inside the image
    int ph = 0, pw = 0, ntaps = 1;
    int wg_t = 0, wg_r = 0, wg_s = 0, cblk = 0, kend = 0;
    int ntiles = 0;
    if (MODE == IG_FWD || MODE == IG_DGRAD) {
        const int Mdim = MODE == IG_FWD ? g.K : g.C;
        const int a_i = (tid & (BM / 4 - 1)) * 4, a_k = tid / (BM / 4);
        a_lane = (uint32_t)(a_k * Mdim + a_i) * 4u;
        const int j = VB ? n0 + (tid & 31) * 4 : n0 + (tid & 127);
        const bool jin = j < g.ncols; // VB: ncols % 4 == 0, the four columns are in or out together
        // columns past the end decode to a column that exists (masked below).  VB: to the START of the last group of four -- a 16-byte
        // load from the last pixel itself would run 12 bytes past the tensor when it is the last channel of the last image
        // (found as a memory fault once the allocation behind such a tensor happened to be unmapped)
        const uint32_t jc = jin ? (uint32_t)j : (uint32_t)g.ncols - (VB ? 4u : 1u);
        const uint32_t n = fd_div(jc, g.fdP);
        const uint32_t p = jc - n * g.P;
        const uint32_t ho = fd_div(p, g.fdWo), wo = p - ho * g.Wo;
        const int b_k = VB ? (tid >> 5) : 16 * (tid >> 7); // first B row of this thread (VB: rows b_k + 8q)
        if (MODE == IG_FWD) {
            // centre tap (S*ho, S*wo) is always inside the image
            b_lane = (uint32_t)(n * g.C * g.HW + (S * ho) * g.W + S * wo + b_k * g.HW) * 4u;
#pragma unroll
            for (int t = 0; t < T; t++) {
                const int hi = S * (int)ho - PAD + t / KS, wi = S * (int)wo - PAD + t % KS;
                if (jin && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) mask |= 1u << t;
            }
            ntaps = T;
            ntiles = T * (g.C / IG_BK);
        } else if (S == 1) {
            // one class, all taps: source pixel (h + PAD - r, w + PAD - s); (h, w) itself is the centre tap
            b_lane = (uint32_t)(n * g.K * g.P + p + b_k * g.P) * 4u;
#pragma unroll
            for (int t = 0; t < T; t++) {
                const int hs = (int)ho + PAD - t / KS, ws = (int)wo + PAD - t % KS;
                if (jin && hs >= 0 && hs < g.Ho && ws >= 0 && ws < g.Wo) mask |= 1u << t;
            }
            ntaps = T;
            ntiles = T * (g.K / IG_BK);
        } else {
            const int cls = 3 - (int)blockIdx.y; // heaviest class (4 taps) first
            ph = cls >> 1; pw = cls & 1;
            const int ntw = pw ? 2 : 1;
            ntaps = (ph ? 2 : 1) * ntw;
            b_lane = (uint32_t)(n * g.K * g.P + p + b_k * g.P) * 4u; // (a, b) itself is always a valid source pixel
            for (int tt = 0; tt < ntaps; tt++) {
                const int th = tt / ntw, tw = tt - th * ntw;
                const int dh = (ph && th == 0) ? 1 : 0, dw = (pw && tw == 0) ? 1 : 0;
                if (jin && (int)ho + dh < g.Ho && (int)wo + dw < g.Wo) mask |= 1u << tt;
            }
            ntiles = ntaps * (g.K / IG_BK);
        }
    } else {
        if (CT64) { wg_t = 2 * (int)ct; cblk = 0; }
        else {
            wg_t = (int)(ct / (uint32_t)g.ctiles);
            cblk = (int)(ct - (uint32_t)wg_t * g.ctiles) * 128;
        }
        wg_r = wg_t / KS; wg_s = wg_t - KS * wg_r;
        const int kbeg = (int)wg_split * g.klen;
        kend = min(g.N * g.P, kbeg + g.klen);
        ntiles = (kend - kbeg + IG_BK - 1) / IG_BK;
    }

    pf4 ra4[NA4];
    float ra[NAS], rb[16];
    pf4 rb4[4];
    int sel_a = 1, sel_b = 1;   // whether the values held in ra / rb are real (else they are stored to LDS as 0)
    int ld_t = 0, ld_c0 = 0;    // next tile to fetch: tap (class tap) and first reduction channel
    if (MODE != IG_WGRAD && tail_id >= 0) { // a reduction slice: k-steps [it0, it0 + nt_slice) of the tile
        ld_t = (int)fd_div((uint32_t)it0, g.fdCpt);
        ld_c0 = (it0 - ld_t * g.cpt) * IG_BK;
        ntiles = min(nt_slice, ntiles - it0);
    }
    int ld_k0 = (MODE == IG_WGRAD) ? (int)wg_split * g.klen : 0;
    const char *fa = nullptr, *fb = nullptr; // wave-uniform bases of the tile being fetched (set by part 0)
    uint32_t fa_lane = 0, fb_lane = 0;       // per-lane byte offsets of the tile being fetched
    uint32_t fb_lane2 = 0;                   // CT64: the second tap's offset (sel_b bit 1 = its validity)
    auto ldg = [](const char *ubase, uint32_t lane_off) -> float { return *(const float *)(ubase + lane_off); };
    auto ldg4 = [](const char *ubase, uint32_t lane_off) -> pf4 { return *(const pf4 *)(ubase + lane_off); };

    // Staging is cut into 8 parts so that it can be spread over the 16 MFMA groups of a tile.
    auto fetch_into = [&](const int p, pf4 *ra4, float *ra, float *rb, pf4 *rb4, int &sel_a, int &sel_b) {
        if (MODE == IG_FWD || MODE == IG_DGRAD) {
            if (p == 0) {
                // past the last tile (the two drain iterations fetch unconditionally) the last tap is simply read again
                const int t = min(ld_t, ntaps - 1);
                sel_b = (mask >> t) & 1;
                if (MODE == IG_FWD) {
                    const int r = KS == 3 ? (t * 11) >> 5 : 0, s = t - KS * r;
                    fa = (const char *)(Aop + (size_t)(t * g.C + ld_c0) * g.K + m0);
                    fb = (const char *)(Bop + (size_t)ld_c0 * g.HW);
                    fb_lane = b_lane + (uint32_t)(sel_b * ((r - PAD) * g.W + (s - PAD)) * 4); // outside: centre pixel, stored as 0
                    ld_c0 += IG_BK;
                    if (ld_c0 == g.C) { ld_c0 = 0; ld_t++; }
                } else {
                    int wt, doff; // weight tap, source-pixel offset
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
                    fa = (const char *)(Aop + (size_t)(wt * g.K + ld_c0) * g.C + m0);
                    fb = (const char *)(Bop + (size_t)ld_c0 * g.P);
                    fb_lane = b_lane + (uint32_t)(sel_b * doff * 4);
                    ld_c0 += IG_BK;
                    if (ld_c0 == g.K) { ld_c0 = 0; ld_t++; }
                }
            }
            const size_t bstride = (MODE == IG_FWD ? (size_t)g.HW : (size_t)g.P) * 4;
            const size_t astride = (MODE == IG_FWD ? (size_t)g.K : (size_t)g.C) * 4;
            if (VB) { if ((p & 1) == 0) rb4[p >> 1] = ldg4(fb + (size_t)(4 * p) * bstride, fb_lane); }
            else {
                rb[2 * p] = ldg(fb + (size_t)(2 * p) * bstride, fb_lane);
                rb[2 * p + 1] = ldg(fb + (size_t)(2 * p + 1) * bstride, fb_lane);
            }
            // A rows a_k + (32 / NA4) * q: spread over the parts
            if (p % (8 / NA4) == 0) ra4[p / (8 / NA4)] = ldg4(fa + (size_t)(4 * p) * astride, a_lane);
        } else {
            if (p == 0) {
                const int kk = ld_k0 + (tid & 31);
                sel_a = kk < kend;
                const uint32_t kc = sel_a ? (uint32_t)kk : (uint32_t)kend - 1;
                const uint32_t n = fd_div_ge2(kc, g.fdP); // P, Wo >= 2 (mi_igemm_supported); no branches in here
                const uint32_t pp = kc - n * g.P;
                const uint32_t ho = fd_div_ge2(pp, g.fdWo), wo = pp - ho * g.Wo;
                const int hi = S * (int)ho - PAD + wg_r, wi = S * (int)wo - PAD + wg_s;
                const int inb = (int)((uint32_t)hi < (uint32_t)g.H) & (int)((uint32_t)wi < (uint32_t)g.W);
                sel_b = sel_a & inb;
                const uint32_t row = (uint32_t)tid >> 5;
                fa = (const char *)(Aop + (size_t)m0 * g.P);
                fb = (const char *)(Bop + (size_t)cblk * g.HW);
                fa_lane = (n * (uint32_t)(g.K * g.P) + pp + row * g.P) * 4u;
                const uint32_t pix = inb ? (uint32_t)(hi * g.W + wi) : (S * ho) * g.W + S * wo;
                fb_lane = (n * (uint32_t)(g.C * g.HW) + pix + row * g.HW) * 4u;
                if (CT64) {
                    const int t1 = wg_t + 1, r1 = t1 / KS, s1 = t1 - KS * r1;
                    const int hj = S * (int)ho - PAD + r1, wj = S * (int)wo - PAD + s1;
                    const int inb1 = (int)((uint32_t)hj < (uint32_t)g.H) & (int)((uint32_t)wj < (uint32_t)g.W) & (int)(t1 < T);
                    const uint32_t pix1 = inb1 ? (uint32_t)(hj * g.W + wj) : (S * ho) * g.W + S * wo;
                    fb_lane2 = (n * (uint32_t)(g.C * g.HW) + pix1 + row * g.HW) * 4u;
                    sel_b |= (sel_a & inb1) << 1;
                }
                ld_k0 += IG_BK;
            }
            if (NAS == 16) {
                ra[2 * p] = ldg(fa + (size_t)(16 * p) * g.P * 4, fa_lane);
                ra[2 * p + 1] = ldg(fa + (size_t)(16 * p + 8) * g.P * 4, fa_lane);
            } else ra[p] = ldg(fa + (size_t)(8 * p) * g.P * 4, fa_lane);
            if (CT64) {
                rb[2 * p] = ldg(fb + (size_t)(16 * (p & 3)) * g.HW * 4, p < 4 ? fb_lane : fb_lane2);
                rb[2 * p + 1] = ldg(fb + (size_t)(16 * (p & 3) + 8) * g.HW * 4, p < 4 ? fb_lane : fb_lane2);
            } else {
                rb[2 * p] = ldg(fb + (size_t)(16 * p) * g.HW * 4, fb_lane);
                rb[2 * p + 1] = ldg(fb + (size_t)(16 * p + 8) * g.HW * 4, fb_lane);
            }
        }
    };
    auto fetch_part = [&](const int p) { fetch_into(p, ra4, ra, rb, rb4, sel_a, sel_b); };
    auto stash_from = [&](const int buf, const int p, const pf4 *ra4, const float *ra, const float *rb, const pf4 *rb4, const int sel_a, const int sel_b) {
        float *as = As + buf * IG_BK * LDA, *bs = Bs + buf * IG_BK * LDB;
        if (MODE == IG_WGRAD) {
            const int kx = tid & 31, row = tid >> 5;
            if (NAS == 16) {
                as[kx * LDA + row + 16 * p] = sel_a ? ra[2 * p] : 0.f;
                as[kx * LDA + row + 16 * p + 8] = sel_a ? ra[2 * p + 1] : 0.f;
            } else as[kx * LDA + row + 8 * p] = sel_a ? ra[p] : 0.f;
            const int sb = CT64 ? (sel_b >> (p >> 2)) & 1 : sel_b;
            bs[kx * LDB + row + 16 * p] = sb ? rb[2 * p] : 0.f;
            bs[kx * LDB + row + 16 * p + 8] = sb ? rb[2 * p + 1] : 0.f;
        } else {
            if (VB) {
                if ((p & 1) == 0) {
                    const pf4 z = {0.f, 0.f, 0.f, 0.f};
                    *(pf4 *)(bs + ((tid >> 5) + 4 * p) * LDB + (tid & 31) * 4) = sel_b ? rb4[p >> 1] : z;
                }
            } else {
                const int bj = tid & 127, b_k = 16 * (tid >> 7);
                bs[(b_k + 2 * p) * LDB + bj] = sel_b ? rb[2 * p] : 0.f;
                bs[(b_k + 2 * p + 1) * LDB + bj] = sel_b ? rb[2 * p + 1] : 0.f;
            }
            if (p % (8 / NA4) == 0) {
                const int a_i = (tid & (BM / 4 - 1)) * 4, a_k = tid / (BM / 4);
                *(pf4 *)(as + (a_k + 4 * p) * LDA + a_i) = ra4[p / (8 / NA4)];
            }
        }
    };
    auto stash_part = [&](const int buf, const int p) { stash_from(buf, p, ra4, ra, rb, rb4, sel_a, sel_b); };

    const int fr = lane & 31, fk = lane >> 5;
    {
        // Prologue: tiles 0 and 1 are fetched back to back into two register sets (one memory latency, not two in a row);
        // tile 0 goes to LDS from the temporary set, tile 1 stays in the loop's registers.
        pf4 pa4[NA4], pb4[4];
        float pa[NAS], pb[16];
        int psel_a = 1, psel_b = 1;
#pragma unroll
        for (int p = 0; p < 8; p++) fetch_into(p, pa4, pa, pb, pb4, psel_a, psel_b);
#pragma unroll
        for (int p = 0; p < 8; p++) fetch_part(p);
#pragma unroll
        for (int p = 0; p < 8; p++) stash_from(0, p, pa4, pa, pb, pb4, psel_a, psel_b);
    }
    // The loop is entered with NO load in flight: every register of tile 1 passes through an empty asm, so the compiler
    // drains vmcnt here, once (header note 2).
    if (MODE == IG_WGRAD) {
#pragma unroll
        for (int q = 0; q < NAS; q++) asm volatile("" : "+v"(ra[q]));
    } else {
#pragma unroll
        for (int q = 0; q < NA4; q++) asm volatile("" : "+v"(ra4[q]));
    }
#pragma unroll
    for (int q = 0; q < 16; q++) { if (!VB) asm volatile("" : "+v"(rb[q])); }
#pragma unroll
    for (int q = 0; q < 4; q++) { if (VB) asm volatile("" : "+v"(rb4[q])); }
    __syncthreads();
    IG_T(1);
    // One k-step: the registers hold tile it+1, which goes to the other buffer (STASH), then they are refilled with tile it+2
    // (FETCH).  The last two k-steps run without the fetch / without both (peeled below): a short reduction (2-8 k-steps for the
    // 1x1 expansions) no longer moves two k-steps of operands nobody reads, and the main loop stays free of branches.
    auto kstep = [&](auto st_tag, auto ft_tag, const int buf) {
        constexpr bool st = IG_ABLATE != 1 && decltype(st_tag)::value;
        constexpr bool ft = IG_ABLATE != 1 && IG_ABLATE != 4 && decltype(ft_tag)::value;
        const float *as = As + buf * IG_BK * LDA + fk * LDA + wm * 64 + fr;
        const float *bs = Bs + buf * IG_BK * LDB + fk * LDB + wn * WNC + fr;
        // fragments of step k2+2 are read while the MFMAs of step k2 run
        float av[2][2], bv[2][TN];
#pragma unroll
        for (int i = 0; i < 2; i++) av[0][i] = as[i * 32];
#pragma unroll
        for (int j = 0; j < TN; j++) bv[0][j] = bs[j * 32];
#pragma unroll
        for (int k2 = 0; k2 < IG_BK; k2 += 2) {
            const int cur = (k2 >> 1) & 1, grp = k2 >> 1;
            if (k2 + 2 < IG_BK) {
#pragma unroll
                for (int i = 0; i < 2; i++) av[cur ^ 1][i] = as[(k2 + 2) * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; j++) bv[cur ^ 1][j] = bs[(k2 + 2) * LDB + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0); // keep the reads ahead of the MFMAs (the scheduler would sink them to their use)
            if (grp < 8) { if (st) stash_part(buf ^ 1, grp); }
            else { if (ft) fetch_part(grp - 8); }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    {
        constexpr std::true_type Y{};
        constexpr std::false_type NO{};
        int it = 0;
        for (; it + 2 < ntiles; it++) kstep(Y, Y, it & 1);
        if (it + 1 < ntiles) { kstep(Y, NO, it & 1); it++; }
        if (it < ntiles) kstep(NO, NO, it & 1);
    }

    IG_T(2);
    // ---- epilogue: accumulator layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ----
    if (MODE != IG_WGRAD && tail_id >= 0) {
        float *tb = g.tailbuf + (size_t)tail_id * (BM * 128);
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    tb[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 128 + wn * WNC + j * 32 + (lane & 31)] = acc[i][j][r];
        return;
    }
    if (MODE == IG_DGRAD && S == 1 && g.bnb_part) {
        // dgrad + the reduction pass of the batch norm backward that consumes it: g = (mask > 0 ? acc (+ addend) : 0) is what gets
        // stored, and the sums of g and g (x - mean) over this wave's WNC columns go out per channel (row).  A half-wave holds 32
        // columns of ONE row per accumulator register, so a row's sum is a 32-lane DPP reduction; the TN column blocks are
        // added in the lane first.  Columns past the end contribute zero and store nothing.
        constexpr int PPT = 4 / WMW;
        if ((g.P & 3) == 0) {
            // Planes a multiple of 4 pixels (56^2, 28^2, 14^2): the tile goes through a wave-private LDS image [32 rows][WNC columns]
            // (two halves in turn; the operand tiles are dead after the loop's last barrier) and is drained with lanes running ALONG a
            // row -- 16 bytes per lane, CPR lanes per row: every access to x / mask / addend / dx is a run of whole 128-byte lines,
            // and a row's sums are a DPP reduction over its CPR lanes.  (First form, kept below for the other planes: lane = column,
            // 4-byte accesses -- four times the memory instructions; it cost more than the pass it replaced.)
            constexpr int CPR = WNC / 4, RPI = 64 / CPR, PITCH = WNC + 4;
            float *img = ig_smem + wave * (32 * PITCH);
            const int c4 = (lane % CPR) * 4, r0 = lane / CPR;
            const int col = n0 + wn * WNC + c4;
            const bool cok = col < g.ncols;                     // (ncols % 4 == 0: the four columns are in or out together)
            const uint32_t jc = cok ? (uint32_t)col : 0u;
            const uint32_t nimg = fd_div(jc, g.fdP);
            const size_t coff = (size_t)nimg * g.C * g.HW + (jc - nimg * g.P);
            const size_t pplane = (size_t)g.bnb_np * g.C;
#pragma unroll
            for (int i = 0; i < 2; i++) {
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        img[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * PITCH + j * 32 + (lane & 31)] = acc[i][j][r];
                // (a wave reads only what it wrote: the LDS operations of a wave complete in order)
#pragma unroll
                for (int ps = 0; ps < 32 / RPI; ps++) {
                    const int rl = ps * RPI + r0;
                    const int row = m0 + wm * 64 + i * 32 + rl;
                    const size_t o = coff + (size_t)row * g.HW;
                    const pf4 z = {0.f, 0.f, 0.f, 0.f};
                    pf4 v = *(const pf4 *)(img + rl * PITCH + c4);
                    pf4 xv = z, mv = z;
                    if (cok) {
                        if (addend) v += *(const pf4 *)(addend + o);
                        xv = *(const pf4 *)(g.bnb_x + o);
                        mv = *(const pf4 *)(g.bnb_mask + o);
                    }
                    const float mean = g.bnb_mean[row];
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        v[e] = (cok && mv[e] > 0.f) ? v[e] : 0.f;
                        s1 += v[e];
                        s2 = fmaf(v[e], xv[e] - mean, s2);
                    }
                    if (cok) *(pf4 *)(Out + o) = v;
                    // sum over the CPR lanes of the row (CPR = 16: one DPP row; 8: half of one)
                    s1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0xB1, 0xF, 0xF, true));
                    s2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0xB1, 0xF, 0xF, true));
                    s1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x4E, 0xF, 0xF, true));
                    s2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0x4E, 0xF, 0xF, true));
                    s1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x141, 0xF, 0xF, true));
                    s2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0x141, 0xF, 0xF, true));
                    if (CPR == 16) {
                        s1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x140, 0xF, 0xF, true));
                        s2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0x140, 0xF, 0xF, true));
                    }
                    if ((lane % CPR) == 0) {
                        const size_t po = (size_t)(ct * PPT + wn) * g.C + row;
                        g.bnb_part[po] = s1;
                        g.bnb_part[pplane + po] = s2;
                    }
                }
            }
            return;
        }
        size_t coffj[TN];
        bool cokj[TN];
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int col = n0 + wn * WNC + j * 32 + (lane & 31);
            cokj[j] = col < g.ncols;
            const uint32_t jc = cokj[j] ? (uint32_t)col : 0u;
            const uint32_t n = fd_div(jc, g.fdP);
            coffj[j] = (size_t)n * g.C * g.HW + (jc - n * g.P);
        }
        const size_t rstride = (size_t)g.HW;
        const size_t pplane = (size_t)g.bnb_np * g.C;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            float s1[16], s2[16], mn[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                s1[r] = 0.f; s2[r] = 0.f;
                mn[r] = g.bnb_mean[m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
            }
#pragma unroll
            for (int j = 0; j < TN; j++) {
                float ad[16], xv[16], mv[16];
#pragma unroll
                for (int r = 0; r < 16; r++) { // all reads first: one latency
                    const size_t o = coffj[j] + (size_t)(m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * rstride;
                    ad[r] = (addend && cokj[j]) ? addend[o] : 0.f;
                    xv[r] = cokj[j] ? g.bnb_x[o] : 0.f;
                    mv[r] = cokj[j] ? g.bnb_mask[o] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float gv = mv[r] > 0.f ? acc[i][j][r] + ad[r] : 0.f;
                    if (cokj[j]) Out[coffj[j] + (size_t)row * rstride] = gv;
                    s1[r] += gv;
                    s2[r] = fmaf(gv, xv[r] - mn[r], s2[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float t1 = ig_half_sum32(s1[r]), t2 = ig_half_sum32(s2[r]);
                if ((lane & 31) == 31) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const size_t po = (size_t)(ct * PPT + wn) * g.C + row;
                    g.bnb_part[po] = t1;
                    g.bnb_part[pplane + po] = t2;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int col = n0 + wn * WNC + j * 32 + (lane & 31);
        size_t coff, rstride;
        bool cok;
        if (MODE == IG_WGRAD) {
            const int lc = wn * WNC + j * 32 + (lane & 31);      // column of the tile
            const int tap = CT64 ? wg_t + (lc >> 6) : wg_t;
            cok = tap < T;
            coff = ((size_t)((size_t)wg_split * T + tap) * g.K) * g.C + (CT64 ? (lc & 63) : cblk + lc);
            rstride = (size_t)g.C;
        } else {
            cok = col < g.ncols;
            const uint32_t jc = cok ? (uint32_t)col : 0u;
            const uint32_t n = fd_div(jc, g.fdP);
            const uint32_t p = jc - n * g.P;
            if (MODE == IG_FWD) { coff = (size_t)n * g.K * g.P + p; rstride = (size_t)g.P; }
            else if (S == 1) { coff = (size_t)n * g.C * g.HW + p; rstride = (size_t)g.HW; }
            else {
                const uint32_t a = fd_div(p, g.fdWo), b = p - a * g.Wo;
                coff = (size_t)n * g.C * g.HW + (size_t)(2 * a + ph) * g.W + 2 * b + pw;
                rstride = (size_t)g.HW;
            }
        }
        if (!cok) continue;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            float ad[16];
            if (MODE == IG_DGRAD && addend) {
                // all 16 reads first: one latency, not sixteen
#pragma unroll
                for (int r = 0; r < 16; r++)
                    ad[r] = addend[coff + (size_t)(m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * rstride];
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[i][j][r];
                if (MODE == IG_DGRAD && addend) v += ad[r];
                Out[coff + (size_t)row * rstride] = v;
            }
        }
    }

    IG_T(3);
    // ---- forward: batch-norm statistics of this tile (the separate read of the whole output is gone) ----
    // Each wave transposes its 64 x WNC accumulator tile through LDS (16-byte stores, XOR-swizzled: conflict-free both ways)
    // so that lane = row; the lane then walks its row's WNC values alone -- no cross-lane traffic (a butterfly per row cost
    // 700 ds_bpermute per wave and made the step 5 % slower than the separate pass it replaced).
    if (MODE == IG_FWD && g.bn_part) {
        constexpr int PPT = 4 / WMW;               // column groups per tile: one per wave column (wn)
        float *tw = ig_smem + wave * (WNC * 64);   // [WNC cols][64 rows]; the LDS tiles are dead after the loop's last barrier
        const int l31 = lane & 31;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int col = j * 32 + l31;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    const int row4 = i * 8 + rg * 2 + (lane >> 5); // rows 4*row4 .. 4*row4+3  (row = i*32 + 8*rg + 4*(lane>>5) + (r&3))
                    pf4 q4 = {acc[i][j][rg * 4 + 0], acc[i][j][rg * 4 + 1], acc[i][j][rg * 4 + 2], acc[i][j][rg * 4 + 3]};
                    *(pf4 *)(tw + col * 64 + ((row4 ^ (col & 15)) << 2)) = q4;
                }
        }
        // (each wave reads only what it wrote: no workgroup barrier; the LDS ops of a wave complete in order)
        const int nv = min(WNC, max(0, g.ncols - (n0 + wn * WNC))); // valid columns of this wave (a prefix)
        const int row = lane;
        const float s0 = tw[(((row >> 2) ^ 0) << 2) + (row & 3)];   // column 0: the shift
        float sd = 0.f, sq = 0.f;
#pragma unroll 8
        for (int c = 0; c < WNC; c++) {
            const float x = tw[c * 64 + ((((row >> 2) ^ (c & 15))) << 2) + (row & 3)];
            const float d = c < nv ? x - s0 : 0.f;
            sd += d;
            sq = fmaf(d, d, sq);
        }
        const float inv = nv > 0 ? 1.0f / (float)nv : 0.f;
        const size_t plane = (size_t)g.bn_np * g.K;
        const size_t o = (size_t)(ct * PPT + wn) * g.K + m0 + wm * 64 + row;
        g.bn_part[o] = (float)nv;
        g.bn_part[plane + o] = nv > 0 ? s0 + sd * inv : 0.f;
        g.bn_part[2 * plane + o] = fmaxf(sq - sd * sd * inv, 0.f);
    }
    IG_T(4);
}

// Sum of the reduction slices of the tail tiles, in slice order (deterministic), written where the tile's own epilogue
// would have put it.  grid = (tail tiles, BM / 16): a workgroup owns a 16-row slab of one tile, a thread one column and
// 8 rows of it -- the 8 x tsplit loads of a thread are independent (the first form, one thread per column looping over
// all rows, was latency-bound: 76 us per launch).
template <int MODE, int BM>
__global__ void __launch_bounds__(256)
igemm_tail_reduce_kernel(float *__restrict__ Out, const float *__restrict__ addend, const IgArgs g) {
    const uint32_t L = (uint32_t)g.full + blockIdx.x;
    const uint32_t ct = fd_div(L, g.fdM);
    const int m0 = (int)(L - ct * g.mtiles) * BM;
    const int col = (int)ct * 128 + (threadIdx.x & 127);
    const bool cok = col < g.ncols;
    const uint32_t jc = cok ? (uint32_t)col : 0u;
    const uint32_t n = fd_div(jc, g.fdP);
    const uint32_t p = jc - n * g.P;
    size_t coff, rstride;
    if (MODE == IG_FWD) { coff = (size_t)n * g.K * g.P + p; rstride = (size_t)g.P; }
    else { coff = (size_t)n * g.C * g.HW + p; rstride = (size_t)g.HW; }
    const int row0 = (int)blockIdx.y * 16 + (int)(threadIdx.x >> 7) * 8;
    const float *tb = g.tailbuf + (size_t)blockIdx.x * g.tsplit * (BM * 128) + (size_t)row0 * 128 + (threadIdx.x & 127);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = 0.f;
    for (int z = 0; z < g.tsplit; z++) {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] += tb[(size_t)z * (BM * 128) + i * 128];
    }
    if (cok) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const size_t o = coff + (size_t)(m0 + row0 + i) * rstride;
            float r = v[i];
            if (MODE == IG_DGRAD && addend) r += addend[o];
            Out[o] = r;
        }
    }
    if (MODE == IG_FWD && g.bn_part) { // same partials as the main kernel's epilogue writes for an unsliced tile
        constexpr int PPT = BM == 128 ? 2 : 4;
        const int lane = threadIdx.x & 63, l31 = lane & 31;
        const int c32 = (int)ct * 128 + (int)(threadIdx.x & 127 & ~31); // first column of this lane's 32-column half
        const int nv = min(32, max(0, g.ncols - c32));
        const size_t plane = (size_t)g.bn_np * g.K;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float mj, qj, nj = (float)nv;
            ig_half_stats(v[i], nv, l31, mj, qj);
            int pp = (int)(threadIdx.x & 127) >> 5; // 32-column group of the tile
            if (PPT == 2) { // merge the two halves of the wave: 64-column groups
                const float n2 = __shfl_xor(nj, 32, 64), m2o = __shfl_xor(mj, 32, 64), q2 = __shfl_xor(qj, 32, 64);
                if (lane == 0) ig_merge(nj, mj, qj, n2, m2o, q2);
                pp >>= 1;
            }
            if (PPT == 2 ? lane == 0 : l31 == 0) {
                const size_t o = (size_t)(ct * PPT + pp) * g.K + m0 + row0 + i;
                g.bn_part[o] = nj;
                g.bn_part[plane + o] = mj;
                g.bn_part[2 * plane + o] = qj;
            }
        }
    }
}

// weights KCRS -> [t][c][k] (forward A operand) or [t][k][c] (dgrad A operand); 32 x 32 (k, c) tiles through LDS
template <int T>
__global__ void __launch_bounds__(256)
igemm_wt_kernel(const float *__restrict__ w, float *__restrict__ out, int K, int C, int to_tck) {
    __shared__ float tile[32][T * 32 + 1];
    const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    // row k of the tile: 32 channels x T taps = contiguous floats of w
    for (int e = threadIdx.x; e < 32 * T * 32; e += 256) {
        const int kr = e / (T * 32), x = e - kr * (T * 32);
        tile[kr][x] = w[((size_t)(k0 + kr) * C + c0) * T + x];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T * 32 * 32; e += 256) {
        const int t = e >> 10, u = (e >> 5) & 31, v = e & 31; // v fastest = contiguous output dim
        if (to_tck) out[((size_t)t * C + c0 + u) * K + k0 + v] = tile[v][u * T + t];
        else out[((size_t)t * K + k0 + u) * C + c0 + v] = tile[u][v * T + t];
    }
}

// The same re-layout for MANY layers in one launch (mid_conv_prelayout_all: the trainer runs it once at the start of a
// forward pass instead of 71 small launches in front of the convolutions): block = one 32 x 32 (k, c) tile of one layer.
__global__ void __launch_bounds__(256)
igemm_wt_all_kernel(const mid_wt_entry *__restrict__ entries, const int *__restrict__ tile_entry) {
    __shared__ float tile[32][9 * 32 + 1];
    const mid_wt_entry E = entries[tile_entry[blockIdx.x]];
    const int tl = (int)blockIdx.x - E.tile0, ctiles = E.C / 32, T = E.T;
    const int c0 = (tl % ctiles) * 32, k0 = (tl / ctiles) * 32, row = T * 32;
    for (int e = threadIdx.x; e < 32 * row; e += 256) {
        const int kr = e / row, x = e - kr * row;
        tile[kr][x] = E.w[((size_t)(k0 + kr) * E.C + c0) * T + x];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T * 1024; e += 256) {
        const int t = e >> 10, u = (e >> 5) & 31, v = e & 31; // v fastest = contiguous output dim
        if (E.fwd) E.fwd[((size_t)t * E.C + c0 + u) * E.K + k0 + v] = tile[v][u * T + t];
        if (E.dgrad) E.dgrad[((size_t)t * E.K + k0 + u) * E.C + c0 + v] = tile[u][v * T + t];
    }
}

// dW[k][c][t] = sum_z part[z][t][k][c] in ascending z (deterministic)
template <int T>
__global__ void __launch_bounds__(256)
igemm_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dw, long KC, int splits) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= KC) return;
    float s[T];
#pragma unroll
    for (int t = 0; t < T; t++) s[t] = 0.f;
    constexpr int U = T == 1 ? 8 : 2; // partials of U splits in flight per thread (one split per trip was latency-bound)
    int z = 0;
    for (; z + U <= splits; z += U) {
        float v[U][T];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int t = 0; t < T; t++) v[u][t] = part[((long)(z + u) * T + t) * KC + i];
#pragma unroll
        for (int u = 0; u < U; u++) // ascending split order, as before
#pragma unroll
            for (int t = 0; t < T; t++) s[t] += v[u][t];
    }
    for (; z < splits; z++)
#pragma unroll
        for (int t = 0; t < T; t++) s[t] += part[((long)z * T + t) * KC + i];
#pragma unroll
    for (int t = 0; t < T; t++) dw[i * T + t] = s[t];
}

// The same sum for the launches the one-thread-per-output form leaves the chip idle on (K C / 256 workgroups: 64 for a 64 x 256 layer
// with 100+ splits to walk): G threads per output, each adds a CONTIGUOUS group of splits in ascending order, the G sums are then added
// in ascending group order -- a fixed tree, so still deterministic (not the same bits as the flat order).
template <int T, int G>
__global__ void __launch_bounds__(256)
igemm_wgrad_reduce_g_kernel(const float *__restrict__ part, float *__restrict__ dw, long KC, int splits) {
    constexpr int OPB = 256 / G;                   // outputs per workgroup: OPB consecutive floats of a partial plane per group
    __shared__ float sh[G][OPB][T];
    const int o = threadIdx.x % OPB, gq = threadIdx.x / OPB;
    const long i = (long)blockIdx.x * OPB + o;
    const int chunk = (splits + G - 1) / G, z0 = gq * chunk, z1 = min(splits, z0 + chunk);
    float s[T];
#pragma unroll
    for (int t = 0; t < T; t++) s[t] = 0.f;
    if (i < KC) {
        constexpr int U = T == 1 ? 8 : 2;
        int z = z0;
        for (; z + U <= z1; z += U) {
            float v[U][T];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int t = 0; t < T; t++) v[u][t] = part[((long)(z + u) * T + t) * KC + i];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int t = 0; t < T; t++) s[t] += v[u][t];
        }
        for (; z < z1; z++)
#pragma unroll
            for (int t = 0; t < T; t++) s[t] += part[((long)z * T + t) * KC + i];
    }
#pragma unroll
    for (int t = 0; t < T; t++) sh[gq][o][t] = s[t];
    __syncthreads();
    if (gq == 0 && i < KC) {
#pragma unroll
        for (int t = 0; t < T; t++) {
            float r = sh[0][o][t];
            for (int q = 1; q < G; q++) r += sh[q][o][t];
            dw[i * T + t] = r;
        }
    }
}
static void igemm_wgrad_reduce_launch(hipStream_t st, const float *part, float *dw, long KC, int k, int splits) {
    // groups of splits per output when one thread per output would leave most CUs without a workgroup
    static int on = -1;
    if (on < 0) { const char *e = getenv("RESNET_MI_WGRAD_REDUCE_G"); on = e ? atoi(e) : 1; }
    const bool grouped = on && splits >= 16 && mi_cdiv(KC, 256) < 1024;
    if (k == 1) {
        if (grouped) hipLaunchKernelGGL((igemm_wgrad_reduce_g_kernel<1, 8>), dim3(mi_cdiv(KC, 32)), dim3(256), 0, st, part, dw, KC, splits);
        else hipLaunchKernelGGL(igemm_wgrad_reduce_kernel<1>, dim3(mi_cdiv(KC, 256)), dim3(256), 0, st, part, dw, KC, splits);
    } else {
        if (grouped) hipLaunchKernelGGL((igemm_wgrad_reduce_g_kernel<9, 8>), dim3(mi_cdiv(KC, 32)), dim3(256), 0, st, part, dw, KC, splits);
        else hipLaunchKernelGGL(igemm_wgrad_reduce_kernel<9>, dim3(mi_cdiv(KC, 256)), dim3(256), 0, st, part, dw, KC, splits);
    }
}

// dW[k][c] = sum_z part[z][c][k] (1x1, transposed product)
__global__ void __launch_bounds__(256)
igemm_wgrad_reduce_t_kernel(const float *__restrict__ part, float *__restrict__ dw, int K, int C, int splits) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x; // over [c][k]: coalesced reads
    if (i >= (long)K * C) return;
    float s = 0.f;
    int z = 0;
    for (; z + 8 <= splits; z += 8) { // eight partials in flight, summed in ascending split order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = part[(long)(z + u) * K * C + i];
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; z < splits; z++) s += part[(long)z * K * C + i];
    const int c = (int)(i / K), kk = (int)(i - (long)c * K);
    dw[(long)kk * C + c] = s;
}

// ------------------------------------------------------------------------------------------------------------
static int igemm_mode(void) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("RESNET_MI_IGEMM"); on = e ? atoi(e) : 2; }
    return on;
}
enum { IGOP_FWD = 0, IGOP_DGRAD = 1, IGOP_WGRAD = 2 };
// Policy + shape gate.  RESNET_MI_IGEMM:
//   0  off: every convolution on the older kernels (direct VALU 3x3/7x7, gemm_mfma_kernel 1x1)
//   1  1x1 weight gradients and the 3x3/s2 projection shortcuts (K >= 2C) only: the bottleneck's own 3x3 convolutions
//      stay on the direct VALU kernels, 1x1 forward / dgrad on gemm_mfma_kernel
//   2  (default) every 3x3 and 1x1 convolution whose channel counts tile.  Measured on MI355X the 3x3 layers are
//      compute-bound, not HBM-bound (fp32 3x3 at C >= 64: >= 288 flop per byte), and the matrix cores run them at
//      90-121 TFLOP/s against 60-75 on the vector ALUs.  1x1 forward / dgrad gain nothing from this kernel's staging
//      pipeline (reductions of 2..64 k-steps: bound by prologue, epilogue and workgroup-count quantisation) but 8 %
//      from its tile-height choice and sliced tail round (igemm_pick_bm, igemm_tail_plan): 19.0 -> 16.8 ms/step.
int mi_igemm_supported(int op, int N, int C, int H, int K, int k, int stride) {
    const int mode = igemm_mode();
    if (!mode) return 0;
    if (!((k == 1 && stride == 1) || (k == 3 && (stride == 1 || stride == 2)))) return 0;
    if (H % stride || H / stride < 2) return 0;
    if (k == 1 && op != IGOP_WGRAD && mode < 2) return 0;
    if (k == 3 && mode == 1 && !(stride == 2 && K >= 2 * C && C >= 256)) return 0; // projections only
    if ((double)N * C * H * H >= 1073741824.0 || (double)N * K * (H / stride) * (H / stride) >= 1073741824.0) return 0; /* 32-bit byte offsets */
    if (op == IGOP_FWD) return C % 32 == 0 && K % 64 == 0;
    if (op == IGOP_DGRAD) return K % 32 == 0 && C % 64 == 0;
    if (C % 128 == 0 && K % 64 == 0) return 1;
    if (k == 3 && C == 64 && K % 64 == 0) return 1;      // two taps per column tile (CT64)
    return k == 1 && C % 64 == 0 && K % 128 == 0; // 1x1 with 64 input channels: computed as the transposed product (below)
}
#define IG_SLOTS 512              /* 256 CUs x 2 resident workgroups of 128-row tiles (67.6 KB of LDS each) */
#define IG_SLOTS64 768            /* 64-row tiles: 51 KB of LDS, three per CU */
static int igemm_wgrad_splits(int N, int C, int H, int K, int k, int stride) {
    const int bm = K % 128 == 0 ? 128 : 64;
    const long tiles = (C == 64 && k == 3 ? 5L : (long)k * k * (C / 128)) * (K / bm);
    const long ksteps = ((long)N * (H / stride) * (H / stride) + IG_BK - 1) / IG_BK;
    const long slots = bm == 128 ? IG_SLOTS : IG_SLOTS64; // resident workgroups on the chip
    int best = 1;
    double best_eff = 0;
    for (int s = 1; s <= 512; s++) {
        if (s > 1 && ksteps / s < 48) break;
        const double waves = (double)tiles * s / slots;
        const double eff = waves / (double)((long)((tiles * s + slots - 1) / slots));
        if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
    }
    return best;
}
// 1x1 weight gradient with C % 128 != 0 (the 64-channel layers): dW^T[c][k] = sum X[c] dY[k] -- the same kernel with the two
// tensors in each other's role (rows = the 64 input channels as a 64-row tile, columns = output channels)
static bool igemm_wgrad_swapped(int C, int K, int k) { return k == 1 && C % 128 != 0; }
size_t mi_igemm_part_floats(int N, int C, int H, int K, int k, int stride) {
    if (igemm_wgrad_swapped(C, K, k)) return (size_t)igemm_wgrad_splits(N, K, H, C, k, stride) * K * C;
    return (size_t)igemm_wgrad_splits(N, C, H, K, k, stride) * k * k * K * C;
}

#define IG_TAIL_FLOATS ((size_t)IG_SLOTS * 128 * 128) /* partial-tile buffer: one slice per slot (768 x 64 x 128 fits too), 33.5 MB */
size_t mi_igemm_tail_floats(void) { return igemm_mode() ? IG_TAIL_FLOATS : 0; }
extern "C" int mid_igemm_mode(void) { return igemm_mode(); } /* RESNET_MI_IGEMM: 0 = no matrix cores anywhere */
// Workgroup-count quantisation: `tiles` equal workgroups on IG_SLOTS resident slots run in ceil(tiles / IG_SLOTS) rounds, so
// a last round that fills only a fraction of the chip costs a whole round (ResNet-50 at N=256: the 1024->2048 @14
// projection has 1568 tiles = 3.06 rounds).  The `rem` tiles of that last round are cut into s = IG_SLOTS / rem slices along
// the reduction, each slice a workgroup of its own: the round shrinks to 1/s of its length (b13 projection forward
// 104.9 -> 111.3 TFLOP/s).  Measured and rejected: slicing a last round that is more than half full into more than
// IG_SLOTS slices (several short rounds) -- no gain over leaving it whole (3x3 forward 8.8 vs 8.8 ms/step, projections
// 12.3 vs 12.1).
static void igemm_tail_plan(IgArgs &g, int ksteps, float *tailbuf, int bm) {
    const int slots = bm == 128 ? IG_SLOTS : IG_SLOTS64;
    g.full = g.tiles; g.tsplit = 1; g.tklen = ksteps; g.tailbuf = tailbuf;
    g.fdTs = make_fastdiv(1);
    static int on = -1;
    if (on < 0) { const char *e = getenv("RESNET_MI_IGEMM_TAIL"); on = e ? atoi(e) : 1; }
    if (!on || !tailbuf) return;
    const int rem = g.tiles % slots;
    if (rem == 0) return;
    if (rem * 2 > slots) return;                       // the last round is at least half full
    int s = slots / rem;                               // slices per tail tile: rem * s <= slots
    if (s > 16) s = 16;
    while (s > 1 && ksteps / s < 8) s--;               // a slice is at least 8 k-steps
    if (s < 2) return;
    g.tklen = (ksteps + s - 1) / s;
    g.tsplit = (ksteps + g.tklen - 1) / g.tklen;       // no empty slices
    g.full = g.tiles - rem;
    g.fdTs = make_fastdiv(g.tsplit);
}

// Rows per workgroup tile.  64-row tiles do ~10% less work per cycle than 128-row ones but halve the granularity: they win
// where the 128-row grid ends in a mostly empty round that is too full to slice (e.g. 1024->256 @14: 784 tiles = 1.53
// rounds -> two rounds; 1568 half tiles = 3.06 rounds with a sliced tail).
static int igemm_pick_bm(int M, int coltiles, int ksteps, int k) {
    if (M % 128) return 64;
    double best = 0;
    int pick = 128;
    for (int bm = 128; bm >= 64; bm -= 64) {
        const long B = (long)(M / bm) * coltiles;
        const long slots = bm == 128 ? IG_SLOTS : IG_SLOTS64;
        const long rem = B % slots;
        double tail = 0;
        if (rem) {
            int s = (int)(slots / rem);
            if (s > 16) s = 16;
            while (s > 1 && ksteps / s < 8) s--;
            tail = rem * 2 > slots || s < 2 ? 1.0 : 1.0 / s + 0.03;
        }
        // time of a round: three 64-row workgroups share a CU (3 x 0.5 = 1.5 units of a 128-row workgroup's work, at ~10%
        // lower efficiency) against two 128-row ones (2 units)
        // (fitted on the per-layer sweep: 3x3 layers run fastest with 0.78-0.83, 1x1 with anything <= 0.70 -- their short
        // reductions favour the finer tiles nearly everywhere)
        const double t = ((double)(B / slots) + tail) * (bm == 128 ? 1.0 : k == 1 ? 0.70 : 0.80);
        if (bm == 128 || t < best * 0.97) { if (bm == 128 || t < best) { best = t; pick = bm; } }
    }
    static int force = -1;
    if (force < 0) { const char *e = getenv("RESNET_MI_IGEMM_BM"); force = e ? atoi(e) : 0; }
    if (force == 64 || (force == 128 && M % 128 == 0)) return force;
    return pick;
}

static void igemm_geometry(IgArgs &g, int N, int C, int H, int K, int stride) {
    g.N = N; g.C = C; g.K = K; g.H = H; g.W = H; g.Ho = H / stride; g.Wo = H / stride;
    g.HW = H * H; g.P = g.Ho * g.Wo;
    g.ncols = N * g.P;
    g.fdP = make_fastdiv(g.P); g.fdWo = make_fastdiv(g.Wo);
}
template <int MODE, int KS, int S, int WMW, bool VB = false>
static int igemm_launch_t(hipStream_t st, dim3 grid, const float *A, const float *B, float *out, const float *addend, const IgArgs &g) {
    constexpr int BM = 64 * WMW;
    constexpr int LDA = (MODE == IG_WGRAD) ? BM + 1 : BM + 4, LDB = (MODE == IG_WGRAD) ? 129 : 132;
    constexpr size_t lds = (size_t)2 * IG_BK * (LDA + LDB) * sizeof(float);
    static int attr_set = 0;
    if (!attr_set) {
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute((const void *)igemm_kernel<MODE, KS, S, WMW, VB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            mi_record_error("igemm_kernel", "cannot raise the dynamic LDS limit");
            return -1;
        }
        attr_set = 1;
    }
    hipLaunchKernelGGL((igemm_kernel<MODE, KS, S, WMW, VB>), grid, dim3(256), lds, st, A, B, out, addend, g);
    return 0;
}
template <int MODE>
static int igemm_launch(hipStream_t st, dim3 grid, const float *A, const float *B, float *out, const float *addend, const IgArgs &g,
                        int k, int stride, int bm) {
#define IGL(KS_, S_)                                                                                      \
    if (k == KS_ && stride == S_)                                                                         \
        return bm == 128 ? igemm_launch_t<MODE, KS_, S_, 2>(st, grid, A, B, out, addend, g)               \
                         : igemm_launch_t<MODE, KS_, S_, 1>(st, grid, A, B, out, addend, g);
    if (MODE != IG_WGRAD && k == 1 && stride == 1 && g.P % 4 == 0)
        return bm == 128 ? igemm_launch_t<MODE, 1, 1, 2, true>(st, grid, A, B, out, addend, g)
                         : igemm_launch_t<MODE, 1, 1, 1, true>(st, grid, A, B, out, addend, g);
    if (MODE == IG_WGRAD && k == 3 && g.C == 64) {
        if (stride == 1) return bm == 128 ? igemm_launch_t<MODE, 3, 1, 2, true>(st, grid, A, B, out, addend, g) : igemm_launch_t<MODE, 3, 1, 1, true>(st, grid, A, B, out, addend, g);
        return bm == 128 ? igemm_launch_t<MODE, 3, 2, 2, true>(st, grid, A, B, out, addend, g) : igemm_launch_t<MODE, 3, 2, 1, true>(st, grid, A, B, out, addend, g);
    }
    IGL(1, 1) IGL(3, 1) IGL(3, 2)
#undef IGL
    return -2;
}
static int igemm_fam(int k) { return k == 1 ? MI_FAM_GEMM : MI_FAM_PCONV; }

int mi_igemm_fwd(hipStream_t st, mid_workspace *ws, const float *x, const float *w, float *y, int N, int C, int H, int K, int k,
                 int stride, mid_bn_parts *parts) {
    const int T = k * k;
    if (!ws || ws->wt_floats < (size_t)T * C * K) { mi_record_error("mi_igemm_fwd", "workspace too small"); return -3; }
    const float *A = ws->pre_fwd; // re-laid [t][c][k] by mid_conv_prelayout_all, else done here
    if (!A) {
        if (k == 1) hipLaunchKernelGGL(igemm_wt_kernel<1>, dim3(C / 32, K / 32), dim3(256), 0, st, w, ws->wt, K, C, 1);
        else hipLaunchKernelGGL(igemm_wt_kernel<9>, dim3(C / 32, K / 32), dim3(256), 0, st, w, ws->wt, K, C, 1);
        MI_LAUNCH_CHECK("igemm_wt_kernel");
        A = ws->wt;
    }
    IgArgs g = {};
    igemm_geometry(g, N, C, H, K, stride);
    const int bm = igemm_pick_bm(K, mi_cdiv(g.ncols, 128), T * (C / IG_BK), k);
    g.mtiles = K / bm;
    g.tiles = g.mtiles * mi_cdiv(g.ncols, 128);
    g.fdM = make_fastdiv(g.mtiles);
    g.cpt = C / IG_BK; g.fdCpt = make_fastdiv(g.cpt);
    igemm_tail_plan(g, T * g.cpt, ws->wt_floats >= (size_t)T * C * K + IG_TAIL_FLOATS ? ws->wt + (size_t)T * C * K : nullptr, bm);
    if (parts) {
        parts->nparts = 0;
        const int np = mi_cdiv(g.ncols, 128) * (bm == 128 ? 2 : 4);
        if (parts->buf && parts->floats >= (size_t)3 * np * K) { g.bn_part = parts->buf; g.bn_np = np; parts->nparts = np; }
    }
    mi_prof_begin(st, igemm_fam(k), 2.0 * T * (double)g.ncols * C * K,
                  4.0 * ((double)N * C * g.HW + (double)T * C * K + (double)g.ncols * K));
    int rc = igemm_launch<IG_FWD>(st, dim3(g.full + (g.tiles - g.full) * g.tsplit), A, x, y, nullptr, g, k, stride, bm);
    if (!rc && g.tsplit > 1) {
        if (bm == 128) hipLaunchKernelGGL((igemm_tail_reduce_kernel<IG_FWD, 128>), dim3(g.tiles - g.full, 8), dim3(256), 0, st, y, nullptr, g);
        else hipLaunchKernelGGL((igemm_tail_reduce_kernel<IG_FWD, 64>), dim3(g.tiles - g.full, 4), dim3(256), 0, st, y, nullptr, g);
    }
    mi_prof_end(st);
    if (rc) return rc;
    MI_LAUNCH_CHECK("igemm_kernel<fwd>");
    return 0;
}

// fz (optional): the reduction pass of the batch-norm backward that consumes dx, fused into this kernel's epilogue (stride 1);
// fz->nparts > 0 on return says it did (dx then holds the GATED gradient)
int mi_igemm_dgrad(hipStream_t st, mid_workspace *ws, const float *w, const float *dy, float *dx, const float *addend, int N,
                   int C, int H, int K, int k, int stride, mid_bn_bwd_parts *fz) {
    if (fz) fz->nparts = 0;
    const int T = k * k;
    const float *A = w; // 1x1: the KC tensor is already [k][c]
    if (k == 3) {
        if (!ws || ws->wt_floats < (size_t)T * C * K) { mi_record_error("mi_igemm_dgrad", "workspace too small"); return -3; }
        if (ws->pre_dgrad) A = ws->pre_dgrad;
        else {
            hipLaunchKernelGGL(igemm_wt_kernel<9>, dim3(C / 32, K / 32), dim3(256), 0, st, w, ws->wt, K, C, 0);
            MI_LAUNCH_CHECK("igemm_wt_kernel");
            A = ws->wt;
        }
    }
    IgArgs g = {};
    igemm_geometry(g, N, C, H, K, stride);
    const int bm = stride == 1 ? igemm_pick_bm(C, mi_cdiv(g.ncols, 128), T * (K / IG_BK), k) : (C % 128 == 0 ? 128 : 64);
    g.mtiles = C / bm;
    g.tiles = g.mtiles * mi_cdiv(g.ncols, 128);
    g.fdM = make_fastdiv(g.mtiles);
    g.cpt = K / IG_BK; g.fdCpt = make_fastdiv(g.cpt);
    bool fused = false;
    if (fz && fz->buf && stride == 1) {
        const int np = mi_cdiv(g.ncols, 128) * (bm == 128 ? 2 : 4);
        if (fz->floats >= (size_t)2 * np * C) {
            g.bnb_x = (const float *)fz->x; g.bnb_mask = (const float *)fz->mask; g.bnb_mean = fz->means;
            g.bnb_part = fz->buf; g.bnb_np = np; fz->nparts = np;
            fused = true;
        }
    }
    // (stride 2: the four parity classes of unequal length already fill the rounds; no slicing.  With the BN' reduction in the
    // epilogue every tile is launched whole: the slices' second stage does not carry it)
    igemm_tail_plan(g, T * g.cpt, !fused && stride == 1 && ws && ws->wt_floats >= (size_t)T * C * K + IG_TAIL_FLOATS ? ws->wt + (size_t)T * C * K : nullptr, bm);
    mi_prof_begin(st, igemm_fam(k), 2.0 * T * (double)g.ncols * C * K,
                  4.0 * ((double)g.ncols * K + (double)T * C * K + (double)N * C * g.HW * ((addend ? 2 : 1) + (fused ? 2 : 0))));
    int rc = igemm_launch<IG_DGRAD>(st, dim3(g.full + (g.tiles - g.full) * g.tsplit, stride == 2 ? 4 : 1), A, dy, dx, addend, g, k, stride, bm);
    if (!rc && g.tsplit > 1) {
        if (bm == 128) hipLaunchKernelGGL((igemm_tail_reduce_kernel<IG_DGRAD, 128>), dim3(g.tiles - g.full, 8), dim3(256), 0, st, dx, addend, g);
        else hipLaunchKernelGGL((igemm_tail_reduce_kernel<IG_DGRAD, 64>), dim3(g.tiles - g.full, 4), dim3(256), 0, st, dx, addend, g);
    }
    mi_prof_end(st);
    if (rc) return rc;
    MI_LAUNCH_CHECK("igemm_kernel<dgrad>");
    return 0;
}

int mi_igemm_wgrad(hipStream_t st, mid_workspace *ws, const float *x, const float *dy, float *dw, int N, int C, int H, int K, int k,
                   int stride) {
    if (igemm_wgrad_swapped(C, K, k)) {
        const int splits = igemm_wgrad_splits(N, K, H, C, 1, 1);
        if (!ws || ws->part_floats < (size_t)splits * K * C) { mi_record_error("mi_igemm_wgrad", "workspace too small"); return -3; }
        IgArgs g = {};
        igemm_geometry(g, N, /*columns:*/ K, H, /*rows:*/ C, 1);
        const int bm = C % 128 == 0 ? 128 : 64;
        g.mtiles = C / bm;
        g.ctiles = K / 128;
        g.tiles = g.mtiles * g.ctiles;
        g.fdM = make_fastdiv(g.mtiles);
        const int kd = N * g.P;
        g.full = g.tiles; g.tsplit = 1; g.fdTs = make_fastdiv(1); g.cpt = 1; g.fdCpt = make_fastdiv(1);
        g.klen = mi_cdiv(mi_cdiv(kd, splits), IG_BK) * IG_BK;
        const int used = mi_cdiv(kd, g.klen);
        mi_prof_begin(st, igemm_fam(1), 2.0 * (double)kd * C * K, 4.0 * ((double)N * C * g.HW + (double)kd * K + (double)C * K));
        const int rc = igemm_launch<IG_WGRAD>(st, dim3(g.tiles, used), /*rows from*/ x, /*columns from*/ dy, ws->part, nullptr, g, 1, 1, bm);
        mi_prof_end(st);
        if (rc) return rc;
        MI_LAUNCH_CHECK("igemm_kernel<wgrad, transposed>");
        hipLaunchKernelGGL(igemm_wgrad_reduce_t_kernel, dim3(mi_cdiv((long)K * C, 256)), dim3(256), 0, st, ws->part, dw, K, C, used);
        MI_LAUNCH_CHECK("igemm_wgrad_reduce_t_kernel");
        return 0;
    }
    const int T = k * k;
    const int splits = igemm_wgrad_splits(N, C, H, K, k, stride);
    if (!ws || ws->part_floats < (size_t)splits * T * K * C) { mi_record_error("mi_igemm_wgrad", "workspace too small"); return -3; }
    IgArgs g = {};
    igemm_geometry(g, N, C, H, K, stride);
    const int bm = K % 128 == 0 ? 128 : 64;
    g.mtiles = K / bm;
    g.ctiles = C / 128;
    g.tiles = C == 64 ? g.mtiles * ((T + 1) / 2) : g.mtiles * T * g.ctiles; // C == 64: two taps per column tile
    g.fdM = make_fastdiv(g.mtiles);
    const int kd = N * g.P;
    g.full = g.tiles; g.tsplit = 1; g.fdTs = make_fastdiv(1); g.cpt = 1; g.fdCpt = make_fastdiv(1);
    g.klen = mi_cdiv(mi_cdiv(kd, splits), IG_BK) * IG_BK;
    const int used = mi_cdiv(kd, g.klen);
    mi_prof_begin(st, igemm_fam(k), 2.0 * T * (double)kd * C * K,
                  4.0 * ((double)N * C * g.HW + (double)kd * K + (double)T * C * K));
    const int rc = igemm_launch<IG_WGRAD>(st, dim3(g.tiles, used), dy, x, ws->part, nullptr, g, k, stride, bm);
    mi_prof_end(st);
    if (rc) return rc;
    MI_LAUNCH_CHECK("igemm_kernel<wgrad>");
    const long KC = (long)K * C;
    igemm_wgrad_reduce_launch(st, ws->part, dw, KC, k, used);
    MI_LAUNCH_CHECK("igemm_wgrad_reduce_kernel");
    return 0;
}

// second stage of a weight gradient whose partials are [split][t][k][c] (also used by the bf16 kernel)
int mi_igemm_wgrad_reduce(hipStream_t st, const float *part, float *dw, int K, int C, int k, int splits) {
    const long KC = (long)K * C;
    igemm_wgrad_reduce_launch(st, part, dw, KC, k, splits);
    MI_LAUNCH_CHECK("igemm_wgrad_reduce_kernel");
    return 0;
}

#ifdef IG_STAMP
extern "C" int mi_debug_igemm_stamps(unsigned long long *dst, int nblocks) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ig_stamps), sizeof(unsigned long long) * 8 * (size_t)nblocks) == hipSuccess ? 0 : -1;
}
#endif

// Host-only view of the launch planners (no GPU touched): which route a convolution takes and with what grid.
// out[0] route taken (1 implicit GEMM, 0 other kernels), [1] rows per tile, [2] tiles, [3] tiles launched whole, [4] reduction
// slices per tail tile, [5] k-steps per slice, [6] wgrad splits, [7] workgroups launched (per class / split), [8] k-steps
extern "C" int mid_igemm_plan(int op, int N, int C, int H, int K, int k, int stride, int out[9]) {
    for (int i = 0; i < 9; i++) out[i] = 0;
    if (!mi_igemm_supported(op, N, C, H, K, k, stride)) return 0;
    out[0] = 1;
    const int T = k * k;
    IgArgs g = {};
    if (op == IGOP_WGRAD) {
        const bool sw = igemm_wgrad_swapped(C, K, k);
        const int rows = sw ? C : K, cols = sw ? K : C;
        igemm_geometry(g, N, cols, H, rows, stride);
        const int bm = rows % 128 == 0 ? 128 : 64;
        const int splits = sw ? igemm_wgrad_splits(N, K, H, C, 1, 1) : igemm_wgrad_splits(N, C, H, K, k, stride);
        const int tiles = sw ? (rows / bm) * (cols / 128) : (C == 64 && k == 3 ? (rows / bm) * ((T + 1) / 2) : (rows / bm) * T * (C / 128));
        const int kd = N * g.P, klen = mi_cdiv(mi_cdiv(kd, splits), IG_BK) * IG_BK;
        out[1] = bm; out[2] = tiles; out[3] = tiles; out[4] = 1; out[5] = klen / IG_BK; out[6] = mi_cdiv(kd, klen); out[7] = tiles; out[8] = mi_cdiv(kd, IG_BK);
        return 1;
    }
    igemm_geometry(g, N, C, H, K, stride);
    const int M = op == IGOP_FWD ? K : C, red = op == IGOP_FWD ? C : K;
    const int ksteps = T * (red / IG_BK);
    const int bm = (op == IGOP_FWD || stride == 1) ? igemm_pick_bm(M, mi_cdiv(g.ncols, 128), ksteps, k) : (M % 128 == 0 ? 128 : 64);
    g.mtiles = M / bm;
    g.tiles = g.mtiles * mi_cdiv(g.ncols, 128);
    float dummy;
    igemm_tail_plan(g, ksteps, (op == IGOP_FWD || stride == 1) ? &dummy : nullptr, bm);
    out[1] = bm; out[2] = g.tiles; out[3] = g.full; out[4] = g.tsplit; out[5] = g.tklen; out[6] = 1;
    out[7] = g.full + (g.tiles - g.full) * g.tsplit; out[8] = ksteps;
    return 1;
}

extern "C" int mid_conv_prelayout_all(mid_stream s, const mid_wt_entry *entries_dev, const int *tile_entry_dev, int ntiles) {
    if (ntiles <= 0) return 0;
    hipLaunchKernelGGL(igemm_wt_all_kernel, dim3(ntiles), dim3(256), 0, (hipStream_t)s, entries_dev, tile_entry_dev);
    MI_LAUNCH_CHECK("igemm_wt_all_kernel");
    return 0;
}
extern "C" void mid_conv_prelayout_needs(int N, int C, int H, int K, int k, int stride, int *need_fwd, int *need_dgrad) {
    *need_fwd = mi_igemm_supported(IGOP_FWD, N, C, H, K, k, stride) ? 1 : 0;
    *need_dgrad = (k == 3 && mi_igemm_supported(IGOP_DGRAD, N, C, H, K, k, stride)) ? 1 : 0; // 1x1 dgrad reads the KC tensor itself
}
