// kernels_cl_bf16.hip -- the 3x3 / stride-2 forward convolution of the bf16 path on CHANNEL-LAST operands (round 3).
//
// Why: kernels_igemm_bf16.hip gathers its pixel operand from NCHW, where the reduction index (the channel) is the strided one: every
// k-step costs 16-byte loads, a register transpose (v_perm) and 8-byte LDS writes per thread, nine times per channel chunk (once per
// tap), and with 64 x 64 wave tiles the loop moves 1.5 KB of LDS traffic per MFMA against a budget of 1 KB at the matrix peak
// (128 B/clk/CU against one 32x32x16 MFMA per 8 clk per CU): it runs at 0.20-0.29 of the bf16 MFMA peak.
// Here the stride-2 layers' input is re-laid ONCE per forward pass as zero-padded parity planes with the channel LAST,
//     Xp[n][q = 2 pr + pc][a][b][c],   a in [0, Ho], b in [0, Wo],   Xp[..][a][b][c] = x[n][c][2 (a - 1) + pr][2 (b - 1) + pc],
// row a = 0 and column b = 0 zero (the halo of the taps that look up / left), so that for tap (r, s) the operand row of output pixel
// (oh, ow) is the 2 C bytes at
//     base(n, oh, ow) + D(r, s),      D = ((q(r, s) Hp + (r > 0)) Wp + (s > 0)) C,   q = 2 ((r + 1) & 1) + ((s + 1) & 1)
// -- one per-thread base per tile, one wave-uniform constant per tap, no masks, and the reduction index contiguous: exactly the MFMA
// operand.  Both operand tiles ([128 rows][64 reduction elements] = rows of 128 bytes) go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no transposes, no ds_write instructions), XOR-swizzled through the SOURCE address
// (the DMA writes lane-linear) so that the ds_read_b128 fragment reads are conflict-free; two LDS buffers, the DMA of k-step i + 1 in
// flight under the MFMAs of k-step i, one barrier per k-step.  Product pixel-major (pixels = accumulator rows) as in the NCHW kernel:
// a lane ends with 4 consecutive pixels of one channel per accumulator quad -- lane-local BN statistics, and NCHW stores of whole
// 128-byte lines through a wave-private LDS image.
#include <stdlib.h>
#include "mi_common.hpp"
#include <type_traits>
#include "mi_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

struct ClArgs {
    int N, C, K, Ho, Wo, Hp, Wp, P; // Hp = Ho + 1, Wp = Wo + 1, P = Ho * Wo
    int ncols, mtiles, tiles;
    FastDiv fdP, fdWo, fdM;
    float *bn_part;                 // statistics partials, three planes [bn_np][K] (count, mean, M2), or nullptr
    int bn_np;
    int vw;                         // pixels per output store: 8 or 4 (P % vw == 0)
};

__device__ __forceinline__ uint32_t cl_pack2(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return *(uint32_t *)&r;
}
// 16-byte chunk `ch` of row `row` of a [rows][128 bytes] LDS image sits at chunk ch ^ key(row): rows r and r + 1 share a 256-byte bank
// row, so the key changes every second row -- the 16 lanes of a ds_read_b128 group (16 consecutive rows, one chunk) then cover all 16
// slots of the two-row bank period
__device__ __forceinline__ int cl_key(int row) { return (row >> 1) & 7; }

// x [N][C][H][W] bf16 -> Xp (see the header); interior only: the halo row / column are zeroed once, when the buffer is made.
// block = (image, output row a - 1, 64-channel chunk): two input rows of 64 channels through LDS, out as 128-byte channel runs.
__global__ void __launch_bounds__(256)
cl_s2d_kernel(const u16 *__restrict__ x, u16 *__restrict__ xp, int C, int H, int W, int Hp, int Wp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cl_smem[];
    u16 *tile = (u16 *)cl_smem;                   // [pr][pc][b'][64 channels], pitch 64 + 8 elements
    constexpr int PT = 72;
    const int Wo = W / 2;
    const int c0 = blockIdx.x * 64, a1 = blockIdx.y, n = blockIdx.z; // a1 = a - 1
    const int pairs = 2 * 64 * Wo;                // (pr, channel, b') 4-byte pairs
    for (int e = threadIdx.x; e < pairs; e += 256) {
        const int bq = e % Wo, t = e / Wo, cc = t & 63, pr = t >> 6;
        const uint32_t v = *(const uint32_t *)(x + (((size_t)n * C + c0 + cc) * H + 2 * a1 + pr) * W + 2 * bq);
        tile[((pr * 2 + 0) * Wo + bq) * PT + cc] = (u16)(v & 0xffffu);
        tile[((pr * 2 + 1) * Wo + bq) * PT + cc] = (u16)(v >> 16);
    }
    __syncthreads();
    const int pieces = 4 * Wo * 8;                // (q, b', 16-byte piece of the 64 channels)
    for (int e = threadIdx.x; e < pieces; e += 256) {
        const int pc8 = e & 7, t = e >> 3, bq = t % Wo, q = t / Wo;
        const u32x4 v = *(const u32x4 *)(tile + (q * Wo + bq) * PT + pc8 * 8);
        *(u32x4 *)(xp + ((((size_t)n * 4 + q) * Hp + a1 + 1) * Wp + bq + 1) * C + c0 + pc8 * 8) = v;
    }
}

// MB: 32-row M blocks per wave (2: workgroup tile 128 x 128, wave tile 64 x 64; 4: 256 x 128, wave tile 128 x 64)
template <int MB>
__global__ void __launch_bounds__(256)
cl_fwd_kernel(const u16 *__restrict__ Aop, const u16 *__restrict__ Xp, u16 *__restrict__ Out, const ClArgs g) {
    constexpr int BM = 64 * MB;                    // rows (output channels) per workgroup
    constexpr int NAU = BM / 32;                   // 16-byte DMA pieces of the A tile per thread
    constexpr int ABYTES = BM * 128, BBYTES = 128 * 128, BUF = ABYTES + BBYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char cl_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- block -> tile (XCD-contiguous, M-tiles fastest: the blocks that share a pixel tile share an L2) ----
    uint32_t L = blockIdx.x;
    {
        const uint32_t per = (uint32_t)g.tiles >> 3;
        if (L < per * 8) L = (L & 7) * per + (L >> 3);
    }
    const uint32_t ct = fd_div(L, g.fdM);
    const int m0 = (int)(L - ct * g.mtiles) * BM, n0 = (int)ct * 128;

    f32x16 acc[MB][2];
#pragma unroll
    for (int i = 0; i < MB; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // ---- DMA state: piece p = tid + 256 u of a tile lands at LDS byte 16 p (lane-linear), i.e. row p >> 3, chunk p & 7; it is
    // loaded from chunk (p & 7) ^ key(row) of that row's 128 source bytes ----
    const int prow = tid >> 3, pchunk = tid & 7;
    uint32_t boff[4];                              // byte offset of the pixel row's channel 0 in Xp (tap 0, 0 not yet added)
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int row = prow + 32 * u;
        const uint32_t j = (uint32_t)min(n0 + row, g.ncols - 1);   // columns past the end: a pixel that exists (never stored)
        const uint32_t n = fd_div(j, g.fdP), p = j - n * g.P;
        const uint32_t oh = fd_div(p, g.fdWo), ow = p - oh * g.Wo;
        boff[u] = (((n * 4u) * g.Hp + oh) * g.Wp + ow) * (uint32_t)g.C * 2u + (uint32_t)((pchunk ^ cl_key(row)) * 16);
    }
    uint32_t aoff[NAU];
#pragma unroll
    for (int u = 0; u < NAU; u++) {
        const int row = prow + 32 * u;
        aoff[u] = (uint32_t)(row * 128 + ((pchunk ^ cl_key(row)) * 16));
    }
    const int cpt = g.C / 64;                      // k-steps per tap
    const int ntiles = 9 * cpt;
    int ld_t = 0, ld_c = 0;
    auto issue = [&](const int buf) {
        const int r = (ld_t * 11) >> 5, s = ld_t - 3 * r;
        const int q = 2 * ((r + 1) & 1) + ((s + 1) & 1);
        const uint32_t d = (uint32_t)(((q * g.Hp + (r > 0)) * g.Wp + (s > 0)) * g.C + ld_c * 64) * 2u;
        const unsigned char *fa = (const unsigned char *)(Aop + ((size_t)(ld_t * cpt + ld_c) * g.K + m0) * 64);
        const unsigned char *fb = (const unsigned char *)Xp + d;
        unsigned char *la = cl_smem + buf * BUF + wave * 1024, *lb = cl_smem + buf * BUF + ABYTES + wave * 1024;
#pragma unroll
        for (int u = 0; u < NAU; u++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(fa + aoff[u]),
                                             (__attribute__((address_space(3))) void *)(la + u * 4096), 16, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; u++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(fb + boff[u]),
                                             (__attribute__((address_space(3))) void *)(lb + u * 4096), 16, 0, 0);
        if (++ld_c == cpt) { ld_c = 0; ld_t++; }
    };
    const int fr = lane & 31, fk = lane >> 5;
    auto compute = [&](const int buf) {
        const unsigned char *as = cl_smem + buf * BUF, *bs = as + ABYTES;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            bf16x8 av[MB], bv[2];
#pragma unroll
            for (int i = 0; i < MB; i++) {
                const int row = wm * (32 * MB) + i * 32 + fr;
                av[i] = *(const bf16x8 *)(as + row * 128 + (((2 * s + fk) ^ cl_key(row)) * 16));
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int row = wn * 64 + j * 32 + fr;
                bv[j] = *(const bf16x8 *)(bs + row * 128 + (((2 * s + fk) ^ cl_key(row)) * 16));
            }
#pragma unroll
            for (int i = 0; i < MB; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j], av[i], acc[i][j], 0, 0, 0);
        }
    };

    issue(0);
    for (int it = 0; it < ntiles; it++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this thread's pieces of tile `it` have landed ...
        __syncthreads();   // ... and so have everyone's; everyone is done with the other buffer
        if (it + 1 < ntiles) issue((it + 1) & 1);
        compute(it & 1);
    }
    __syncthreads();       // the epilogue re-uses the operand buffers

    // ---- epilogue: accumulator (i, j): rows = the 32 pixels of column block j, row (r & 3) + 8 (r >> 2) + 4 (lane >> 5);
    // column = channel m0 + wm 32 MB + i 32 + (lane & 31).  A quad r = 4 q .. 4 q + 3 is 4 consecutive pixels of one image (P % 4 == 0) ----
    const int l31 = lane & 31, hh = lane >> 5;
    uint32_t okm = 0;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int col = n0 + wn * 64 + j * 32 + 8 * q + 4 * hh;
            if (col < g.ncols) { okm |= 1u << (j * 4 + q); cnt += 4; }
        }
    if (g.bn_part) { // lane-local statistics of the wave's 32 MB channels x 64 pixels (a lane holds 32 pixels of ONE channel per i)
#pragma unroll
        for (int i = 0; i < MB; i++) {
            const int m = m0 + wm * (32 * MB) + i * 32 + l31;
            const float s0 = __shfl(acc[i][0][0], l31, 64); // the wave's first pixel (a real one if any is)
            float sd = 0.f, sq = 0.f;
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const bool ok = (okm >> (j * 4 + q)) & 1u;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float dlt = ok ? acc[i][j][4 * q + e] - s0 : 0.f;
                        sd += dlt;
                        sq = fmaf(dlt, dlt, sq);
                    }
                }
            sd += __shfl_xor(sd, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            const int nvw = cnt + __shfl_xor(cnt, 32, 64);
            if (hh == 0) {
                const float inv = nvw > 0 ? 1.0f / (float)nvw : 0.f;
                const size_t plane = (size_t)g.bn_np * g.K;
                const size_t o = (size_t)(ct * 2 + wn) * g.K + m;
                g.bn_part[o] = (float)nvw;
                g.bn_part[plane + o] = nvw > 0 ? s0 + sd * inv : 0.f;
                g.bn_part[2 * plane + o] = fmaxf(sq - sd * sd * inv, 0.f);
            }
        }
    }
    // stores through a wave-private LDS image [64 channels][64 pixels] (bf16, pitch 144 B), 64 channels at a time, read back with
    // lanes running ALONG a channel row: a wave instruction writes whole 128-byte lines of y
    constexpr int PITCH = 64 * 2 + 16;
    unsigned char *img = cl_smem + wave * (64 * PITCH);
#pragma unroll
    for (int half = 0; half < MB / 2; half++) {
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
            const int i = half * 2 + ii;
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    u32x2 v = {cl_pack2(acc[i][j][4 * q], acc[i][j][4 * q + 1]), cl_pack2(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3])};
                    *(u32x2 *)(img + (ii * 32 + l31) * PITCH + (j * 32 + 8 * q + 4 * hh) * 2) = v;
                }
        }
        auto drain = [&](auto cpx_tag) {
            constexpr int CPX = decltype(cpx_tag)::value; // pixels per lane and store
            constexpr int CPR = 64 / CPX, RPP = 64 / CPR;
            const int c = lane % CPR, r0 = lane / CPR;
            const int col = n0 + wn * 64 + c * CPX;
            const uint32_t cc = (uint32_t)min(col, g.ncols - 1);
            const uint32_t n = fd_div(cc, g.fdP), pp = cc - n * g.P;
            const bool ok = col < g.ncols;
            const uint32_t obase = n * (uint32_t)(g.K * g.P) + pp;
#pragma unroll
            for (int ps = 0; ps < 64 / RPP; ps++) {
                const int row = ps * RPP + r0;
                const size_t o = (size_t)(obase + (uint32_t)(m0 + wm * (32 * MB) + half * 64 + row) * (uint32_t)g.P);
                const unsigned char *sp = img + row * PITCH + c * CPX * 2;
                if constexpr (CPX == 8) { const u32x4 v = *(const u32x4 *)sp; if (ok) *(u32x4 *)(Out + o) = v; }
                else { const u32x2 v = *(const u32x2 *)sp; if (ok) *(u32x2 *)(Out + o) = v; }
            }
        };
        if (g.vw == 8) drain(std::integral_constant<int, 8>{});
        else drain(std::integral_constant<int, 4>{});
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
extern "C" {
/* shapes the channel-last forward covers: 3x3, stride 2, channels a multiple of 64, output planes a multiple of 4 pixels */
int mid_cl_fwd_supported(int N, int C, int H, int K) {
    if (H % 2 || H < 4 || H > 240) return 0;
    const int Ho = H / 2, P = Ho * Ho;
    if (C % 64 || K % 128 || P % 4) return 0;
    if ((double)N * 4 * (Ho + 1) * (Ho + 1) * C * 2 >= 4294000000.0 || (double)N * K * P >= 2147480000.0) return 0; /* 32-bit byte offsets */
    return 1;
}
size_t mid_cl_xp_bytes(int N, int C, int H) { return (size_t)N * 4 * (H / 2 + 1) * (H / 2 + 1) * C * 2; }
/* x (bf16 NCHW) -> padded channel-last parity planes; the halo of xp must be zero (zero the buffer once when it is allocated) */
int mid_cl_s2d(mid_stream s, const void *x, void *xp, int N, int C, int H) {
    const int Ho = H / 2;
    const size_t lds = (size_t)4 * Ho * 72 * 2;
    hipLaunchKernelGGL(cl_s2d_kernel, dim3(C / 64, Ho, N), dim3(256), lds, (hipStream_t)s, (const u16 *)x, (u16 *)xp, C, H, H, Ho + 1, Ho + 1);
    MI_LAUNCH_CHECK("cl_s2d_kernel");
    return 0;
}
/* y (bf16 NCHW) = conv3x3 stride 2 of the planes xp with the weights a_tiles = bf16 k-step tiles [t][c/64][K][64] (the forward
 * layout of mid_conv_prelayout_all_bf16); parts (optional): BN statistics partials as mid_conv_fwd_bf16 leaves them */
int mid_cl_fwd(mid_stream s, const void *xp, const void *a_tiles, void *y, int N, int C, int H, int K, mid_bn_parts *parts) {
    hipStream_t st = (hipStream_t)s;
    if (parts) parts->nparts = 0;
    if (!mid_cl_fwd_supported(N, C, H, K)) { mi_record_error("mid_cl_fwd", "shape not covered"); return -2; }
    ClArgs g = {};
    g.N = N; g.C = C; g.K = K; g.Ho = H / 2; g.Wo = H / 2; g.Hp = g.Ho + 1; g.Wp = g.Wo + 1; g.P = g.Ho * g.Wo;
    g.ncols = N * g.P;
    g.fdP = make_fastdiv(g.P); g.fdWo = make_fastdiv(g.Wo);
    g.vw = g.P % 8 == 0 ? 8 : 4;
    const int ctl = mi_cdiv(g.ncols, 128);
    static int mb_force = -1;
    if (mb_force < 0) { const char *e = getenv("RESNET_MI_CL_MB"); mb_force = e ? atoi(e) : 0; }
    int mb = (K % 256 == 0 && (long)(K / 256) * ctl >= 512) ? 4 : 2; // 256-row tiles where they still fill the chip twice over
    if (mb_force == 2 || (mb_force == 4 && K % 256 == 0)) mb = mb_force;
    const int bm = 64 * mb;
    g.mtiles = K / bm; g.tiles = g.mtiles * ctl; g.fdM = make_fastdiv(g.mtiles);
    if (parts && parts->buf && parts->floats >= (size_t)3 * ctl * 2 * K) { g.bn_part = parts->buf; g.bn_np = ctl * 2; parts->nparts = ctl * 2; }
    const size_t lds = (size_t)2 * (bm * 128 + 128 * 128);
    mi_prof_begin(st, MI_FAM_PCONV, 2.0 * 9 * (double)N * g.P * C * K, 2.0 * ((double)N * C * H * H + (double)N * g.P * K) + 4.0 * 9 * C * K);
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)cl_fwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 * 128 + 128 * 128)) != hipSuccess ||
            hipFuncSetAttribute((const void *)cl_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 * 128 + 128 * 128)) != hipSuccess) {
            mi_record_error("cl_fwd_kernel", "cannot raise the dynamic LDS limit");
            return -1;
        }
        attr_set = 1;
    }
    if (mb == 4) hipLaunchKernelGGL(cl_fwd_kernel<4>, dim3(g.tiles), dim3(256), lds, st, (const u16 *)a_tiles, (const u16 *)xp, (u16 *)y, g);
    else hipLaunchKernelGGL(cl_fwd_kernel<2>, dim3(g.tiles), dim3(256), lds, st, (const u16 *)a_tiles, (const u16 *)xp, (u16 *)y, g);
    mi_prof_end(st);
    MI_LAUNCH_CHECK("cl_fwd_kernel");
    return 0;
}
}
