// kernels_gemm.hip -- fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered
// fma chain) for the GEMM-shaped layers only: the 1x1 convolutions (forward, dgrad, wgrad) and the FC
// layer (matMul + the two transposed forms, resnet.cu:70-101, 1482-1509).
//
// One kernel template.  C[M x N] = A[M x K] * B[K x N] with generic element addressing so that the
// NCHW tensors are consumed in place:
//   forward  Y[n][k][p]  = sum_c W[k][c]  X[n][c][p]     M=K_out, N=(n,p) "batched column", K=C
//   dgrad    dX[n][c][p] = sum_k W[k][c]  dY[n][k][p]    M=C,     N=(n,p),                 K=K_out
//   wgrad    dW[k][c]    = sum_(n,p) dY[n][k][p] X[n][c][p]   M=K_out, N=C, K=(n,p) "batched K", split over z
// Tile 128x128x32 (or 64x128x32), 256 threads = 4 waves, each wave a 64x64 (64x32) block of 32x32 MFMA
// tiles; operands staged k-major in LDS (pitch 130: conflict-free for both the staging writes and the
// lane=row fragment reads), 16-byte global loads where the contiguous dim allows, next tile prefetched into registers while the current one is multiplied.
#include "mi_common.hpp"
#include "mi_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    int M, N, K;
    // A element (i, kk):  a_sm*i + a_sk*kk            (a_kb: kk is (n,p): n*a_sb + p + a_sm*i)
    long a_sm, a_sk, a_sb;
    // B element (kk, j):  b_sk*kk + b_sn*j            (b_nb: j is (n,p): n*b_sb + p + b_sk*kk)
    //                                                 (b_kb: kk is (n,p): n*b_sb + p + b_sn*j)
    long b_sk, b_sn, b_sb;
    // C element (i, j):   c_sm*i + c_sn*j             (c_nb: j is (n,p): n*c_sb + p + c_sm*i)
    long c_sm, c_sn, c_sb;
    int P;            // plane size for batched dims
    FastDiv fdP;
    int klen;         // K range per blockIdx.z (split-K)
    long c_split;     // C offset per split
};

enum { BATCH_NONE = 0, BATCH_N = 1, BATCH_K = 2 };

#define G_BK 32
#define G_LDK 129 /* pitch of a K-contiguous operand staged transposed (4 scalar stores per float4: conflict-free) */
#define G_LDN 132 /* pitch of an M/N-contiguous operand staged with ds_write_b128 (16-B aligned rows) */

typedef float gf4 __attribute__((ext_vector_type(4)));

// A_KC: A's contiguous dim is K (else M).  B_KC: B's contiguous dim is K (else N).
// VA / VB: the operand may be fetched with 16-byte loads along its contiguous dim (sizes % 4 == 0, 16-B aligned).
template <int BM, int BATCH, bool A_KC, bool B_KC, bool VA, bool VB>
__global__ void __launch_bounds__(256)
gemm_mfma_kernel(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                 const float *__restrict__ addend, const GemmArgs g) {
    constexpr int BN = 128;
    constexpr int WM = 64, WN = (BM == 128) ? 64 : 32;     // per-wave tile
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int LDA = A_KC ? G_LDK : G_LDN, LDB = B_KC ? G_LDK : G_LDN;
    constexpr int AV = BM * G_BK / 1024, BV = BN * G_BK / 1024;  // float4 per thread per tile
    constexpr int AE = BM * G_BK / 256, BE = BN * G_BK / 256;    // scalars per thread per tile
    __shared__ __attribute__((aligned(16))) float As[G_BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[G_BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (BM == 128) ? (wave >> 1) : 0, wn = (BM == 128) ? (wave & 1) : wave;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * g.klen, kend = min(g.K, kbeg + g.klen);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // Per-thread staging coordinates and the kk-independent part of every element offset, computed ONCE:
    // inside the k loop an address costs one multiply-add (plus one fast division per k-step when K is the batched dim).
    constexpr int NA = VA ? AV : AE, NB = VB ? BV : BE;
    int ai[NA], ak[NA], bj[NB], bk[NB];
    long aoff[NA], boff[NB];
#pragma unroll
    for (int q = 0; q < NA; q++) {
        if (VA) {
            if (A_KC) { ak[q] = (tid & 7) * 4; ai[q] = (tid >> 3) + 32 * q; }
            else { ai[q] = (tid & (BM / 4 - 1)) * 4; ak[q] = tid / (BM / 4) + (1024 / BM) * q; }
        } else {
            if (A_KC) { ak[q] = tid & 31; ai[q] = (tid >> 5) + 8 * q; }
            else { ai[q] = tid % BM; ak[q] = tid / BM + (256 / BM) * q; }
        }
        aoff[q] = (long)(m0 + ai[q]) * g.a_sm;
    }
#pragma unroll
    for (int q = 0; q < NB; q++) {
        if (VB) {
            if (B_KC) { bk[q] = (tid & 7) * 4; bj[q] = (tid >> 3) + 32 * q; }
            else { bj[q] = (tid & 31) * 4; bk[q] = (tid >> 5) + 8 * q; }
        } else {
            if (B_KC) { bk[q] = tid & 31; bj[q] = (tid >> 5) + 8 * q; }
            else { bj[q] = tid & 127; bk[q] = (tid >> 7) + 2 * q; }
        }
        const int col = min(n0 + bj[q], g.N - 1);
        if (BATCH == BATCH_N) { const uint32_t n = fd_div((uint32_t)col, g.fdP); boff[q] = (long)n * g.b_sb + (col - (long)n * g.P); }
        else boff[q] = (long)col * g.b_sn;
    }
    // kk-dependent part (batched K: the reduction index is (image, pixel))
    auto koff_a = [&](int kk) -> long {
        if (BATCH == BATCH_K) { const uint32_t n = fd_div((uint32_t)kk, g.fdP); return (long)n * g.a_sb + (kk - (long)n * g.P); }
        return (long)kk * g.a_sk;
    };
    auto koff_b = [&](int kk) -> long {
        if (BATCH == BATCH_K) { const uint32_t n = fd_div((uint32_t)kk, g.fdP); return (long)n * g.b_sb + (kk - (long)n * g.P); }
        return (long)kk * g.b_sk;
    };

    gf4 ra4[VA ? AV : 1], rb4[VB ? BV : 1];
    float ra[VA ? 1 : AE], rb[VB ? 1 : BE];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int q = 0; q < NA; q++) {
            const int kk = k0 + ak[q];
            const bool ok = m0 + ai[q] < g.M && kk < kend;
            if (VA) { gf4 v = {0.f, 0.f, 0.f, 0.f}; if (ok) v = *(const gf4 *)(A + aoff[q] + koff_a(kk)); ra4[q] = v; }
            else { float v = 0.f; if (ok) v = A[aoff[q] + koff_a(kk)]; ra[q] = v; }
        }
#pragma unroll
        for (int q = 0; q < NB; q++) {
            const int kk = k0 + bk[q];
            const bool ok = n0 + bj[q] < g.N && kk < kend;
            if (VB) { gf4 v = {0.f, 0.f, 0.f, 0.f}; if (ok) v = *(const gf4 *)(B + boff[q] + koff_b(kk)); rb4[q] = v; }
            else { float v = 0.f; if (ok) v = B[boff[q] + koff_b(kk)]; rb[q] = v; }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int q = 0; q < NA; q++) {
            if (VA && A_KC) {
#pragma unroll
                for (int e = 0; e < 4; e++) As[(ak[q] + e) * LDA + ai[q]] = ra4[q][e];
            } else if (VA) *(gf4 *)(As + ak[q] * LDA + ai[q]) = ra4[q];
            else As[ak[q] * LDA + ai[q]] = ra[q];
        }
#pragma unroll
        for (int q = 0; q < NB; q++) {
            if (VB && B_KC) {
#pragma unroll
                for (int e = 0; e < 4; e++) Bs[(bk[q] + e) * LDB + bj[q]] = rb4[q][e];
            } else if (VB) *(gf4 *)(Bs + bk[q] * LDB + bj[q]) = rb4[q];
            else Bs[bk[q] * LDB + bj[q]] = rb[q];
        }
    };

    load_tile(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += G_BK) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (k0 + G_BK < kend) load_tile(k0 + G_BK);
        const int fr = lane & 31, fk = lane >> 5;
        // fragments of step k2+2 are read before the MFMAs of step k2 are issued (sched_barrier pins the order: the
        // scheduler would otherwise sink the LDS reads to their use and expose their latency every step)
        float av[2][TM], bv[2][TN];
#pragma unroll
        for (int i = 0; i < TM; i++) av[0][i] = As[fk * LDA + wm * WM + i * 32 + fr];
#pragma unroll
        for (int j = 0; j < TN; j++) bv[0][j] = Bs[fk * LDB + wn * WN + j * 32 + fr];
#pragma unroll
        for (int k2 = 0; k2 < G_BK; k2 += 2) {
            const int cur = (k2 >> 1) & 1;
            if (k2 + 2 < G_BK) {
#pragma unroll
                for (int i = 0; i < TM; i++) av[cur ^ 1][i] = As[(k2 + 2 + fk) * LDA + wm * WM + i * 32 + fr];
#pragma unroll
                for (int j = 0; j < TN; j++) bv[cur ^ 1][j] = Bs[(k2 + 2 + fk) * LDB + wn * WN + j * 32 + fr];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    float *Cz = C + (long)blockIdx.z * g.c_split;
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int col = n0 + wn * WN + j * 32 + (lane & 31);
        if (col >= g.N) continue;
        long coff;
        if (BATCH == BATCH_N) { const uint32_t n = fd_div((uint32_t)col, g.fdP); coff = (long)n * g.c_sb + (col - (long)n * g.P); }
        else coff = (long)col * g.c_sn;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            float ad[16];
            if (addend) { // all 16 reads first: one memory latency per tile row block, not sixteen
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    ad[r] = row < g.M ? addend[coff + (long)row * g.c_sm] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.M) {
                    const long o = coff + (long)row * g.c_sm;
                    float v = acc[i][j][r];
                    if (addend) v += ad[r];
                    Cz[o] = v;
                }
            }
        }
    }
}

template <int BATCH, bool A_KC, bool B_KC>
static int launch_gemm(hipStream_t st, const float *A, const float *B, float *C, const float *addend, GemmArgs g,
                       int splits, bool va, bool vb) {
    g.fdP = make_fastdiv(g.P > 0 ? g.P : 1);
    if (splits < 1) splits = 1;
    int klen = mi_cdiv(g.K, splits);
    klen = mi_cdiv(klen, G_BK) * G_BK;
    g.klen = klen;
    splits = mi_cdiv(g.K, klen);
    const int bm = g.M <= 64 ? 64 : 128;
    dim3 grid(mi_cdiv(g.N, 128), mi_cdiv(g.M, bm), splits), block(256);
    mi_prof_begin(st, MI_FAM_GEMM, 2.0 * (double)g.M * g.N * g.K,
                  4.0 * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N * (addend ? 2 : 1)));
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) va = vb = false;
#define GL(BM_, VA_, VB_) hipLaunchKernelGGL((gemm_mfma_kernel<BM_, BATCH, A_KC, B_KC, VA_, VB_>), grid, block, 0, st, A, B, C, addend, g)
    if (bm == 64) { if (va && vb) GL(64, true, true); else if (va) GL(64, true, false); else GL(64, false, false); }
    else { if (va && vb) GL(128, true, true); else if (va) GL(128, true, false); else GL(128, false, false); }
#undef GL
    mi_prof_end(st);
    MI_LAUNCH_CHECK("gemm_mfma_kernel");
    return splits;
}

int mi_launch_split_reduce(hipStream_t st, const float *part, float *out, long n, int splits, size_t stride);

// Y[n][k][p] = sum_c W[k][c] X[n][c][p]
int mi_conv1x1_fwd(hipStream_t st, const float *x, const float *w, float *y, int N, int C, int P, int K) {
    GemmArgs g = {};
    g.M = K; g.N = N * P; g.K = C; g.P = P;
    g.a_sm = C; g.a_sk = 1;
    g.b_sk = P; g.b_sb = (long)C * P;
    g.c_sm = P; g.c_sb = (long)K * P;
    return launch_gemm<BATCH_N, true, false>(st, w, x, y, nullptr, g, 1, C % 4 == 0, P % 4 == 0) > 0 ? 0 : -1;
}
// dX[n][c][p] = sum_k W[k][c] dY[n][k][p] (+ addend)
int mi_conv1x1_dgrad(hipStream_t st, const float *w, const float *dy, float *dx, const float *addend, int N, int C,
                     int P, int K) {
    GemmArgs g = {};
    g.M = C; g.N = N * P; g.K = K; g.P = P;
    g.a_sm = 1; g.a_sk = C;
    g.b_sk = P; g.b_sb = (long)K * P;
    g.c_sm = P; g.c_sb = (long)C * P;
    return launch_gemm<BATCH_N, false, false>(st, w, dy, dx, addend, g, 1, C % 4 == 0, P % 4 == 0) > 0 ? 0 : -1;
}
static int wgrad1x1_splits(int N, int C, int P, int K) {
    const int tiles = mi_cdiv(C, 128) * mi_cdiv(K, K <= 64 ? 64 : 128);
    int splits = mi_cdiv(1024, tiles);
    const long kd = (long)N * P;
    const int maxs = (int)(kd / 256 > 0 ? kd / 256 : 1);
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    return splits;
}
size_t mi_conv1x1_wgrad_part_floats(int N, int C, int P, int K) {
    return (size_t)wgrad1x1_splits(N, C, P, K) * K * C;
}
// dW[k][c] = sum_(n,p) dY[n][k][p] X[n][c][p]
int mi_conv1x1_wgrad(hipStream_t st, mid_workspace *ws, const float *x, const float *dy, float *dw, int N, int C, int P,
                     int K) {
    GemmArgs g = {};
    g.M = K; g.N = C; g.K = N * P; g.P = P;
    g.a_sm = P; g.a_sb = (long)K * P;
    g.b_sn = P; g.b_sb = (long)C * P;
    g.c_sm = C; g.c_sn = 1;
    g.c_split = (long)K * C;
    int splits = wgrad1x1_splits(N, C, P, K);
    if (splits > 1 && (!ws || ws->part_floats < (size_t)splits * K * C)) {
        mi_record_error("mi_conv1x1_wgrad", "workspace too small");
        return -3;
    }
    float *out = splits > 1 ? ws->part : dw;
    const int used = launch_gemm<BATCH_K, true, true>(st, dy, x, out, nullptr, g, splits, P % 4 == 0, P % 4 == 0);
    if (used < 0) return -1;
    if (splits > 1) return mi_launch_split_reduce(st, out, dw, (long)K * C, used, (size_t)K * C);
    return 0;
}

extern "C" {
int mid_gemm_nn(mid_stream s, const float *A, const float *B, float *out, int m, int k, int n) {
    GemmArgs g = {};
    g.M = m; g.N = n; g.K = k; g.P = 1;
    g.a_sm = k; g.a_sk = 1; g.b_sk = n; g.b_sn = 1; g.c_sm = n; g.c_sn = 1;
    return launch_gemm<BATCH_NONE, true, false>((hipStream_t)s, A, B, out, nullptr, g, 1, k % 4 == 0, n % 4 == 0) > 0 ? 0 : -1;
}
int mid_gemm_tn(mid_stream s, const float *At, const float *B, float *out, int m, int k, int n) {
    GemmArgs g = {};
    g.M = m; g.N = n; g.K = k; g.P = 1;
    g.a_sm = 1; g.a_sk = m; g.b_sk = n; g.b_sn = 1; g.c_sm = n; g.c_sn = 1;
    return launch_gemm<BATCH_NONE, false, false>((hipStream_t)s, At, B, out, nullptr, g, 1, m % 4 == 0, n % 4 == 0) > 0 ? 0 : -1;
}
int mid_gemm_nt(mid_stream s, const float *A, const float *Bt, float *out, int m, int k, int n) {
    GemmArgs g = {};
    g.M = m; g.N = n; g.K = k; g.P = 1;
    g.a_sm = k; g.a_sk = 1; g.b_sk = 1; g.b_sn = k; g.c_sm = n; g.c_sn = 1;
    return launch_gemm<BATCH_NONE, true, true>((hipStream_t)s, A, Bt, out, nullptr, g, 1, k % 4 == 0, k % 4 == 0) > 0 ? 0 : -1;
}
}
