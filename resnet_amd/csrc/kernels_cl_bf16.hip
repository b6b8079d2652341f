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
// (buffer_load_dwordx4 ... lds: no staging registers, no transposes, no ds_write instructions), XOR-swizzled through the SOURCE address
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

int mi_igemm_wgrad_reduce(hipStream_t st, const float *part, float *dw, int K, int C, int k, int splits); // kernels_igemm.hip

struct ClArgs {
    int Cin, M;                     // reduction channels per tap, output channels (rows of the product)
    int GH, GW, P, ncols;           // output grid per image, P = GH * GW, ncols = N * P
    int img_rows, Wp;               // padded input: rows per image (all planes), padded row length; pixel (n, y, x) of the grid reads
                                    // the Cin channels at ((n img_rows + y) Wp + x) Cin, shifted by a per-tap constant
    int ntaps;
    uint32_t tap_delta[9];          // byte offset of tap t's operand row from the grid pixel's
    int tap_w[9];                   // its weight tile: A = [tap_w][Cin / 64][M][64]
    int mtiles, tiles;
    FastDiv fdP, fdGW, fdM;
    float *bn_part;                 // forward: statistics partials, three planes [bn_np][M] (count, mean, M2), or nullptr
    int bn_np;
    int vw;                         // pixels per output store: 8 / 4 (P % vw == 0), or 1 (any plane: 2-byte stores)
    const u16 *addend;              // dgrad: shortcut gradient added before the one rounding to bf16 (same shape as the output), or nullptr
};

__device__ __forceinline__ uint32_t cl_pack2(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return *(uint32_t *)&r;
}
__device__ __forceinline__ float cl_bf2f(u16 v) { return __uint_as_float((uint32_t)v << 16); }
// 16-byte chunk `ch` of row `row` of a [rows][128 bytes] LDS image sits at chunk ch ^ key(row): rows r and r + 1 share a 256-byte bank
// row, so the key changes every second row -- the 16 lanes of a ds_read_b128 group (16 consecutive rows, one chunk) then cover all 16
// slots of the two-row bank period
__device__ __forceinline__ int cl_key(int row) { return (row >> 1) & 7; }
// One 16-byte piece per lane, global -> LDS, by BUFFER addressing (buffer_load_dwordx4 ... offen lds): the lane's part of the address is one
// 32-bit VGPR, the wave-uniform part one SGPR -- against the flat form (global_load_lds: a 64-bit address pair per lane, built with vector
// adds for every piece) the eight DMA instructions of a k-step cost 565 instead of 876 cycles of a wave's issue time.  A lane whose offset
// is past num_records gets ZEROS written to LDS (tools/ubench/buf_lds_oob.hip): CL_OOB is how a piece beyond a plane asks for zeros.
#define CL_OOB 0xfffffff0u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cl_rsrc(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void cl_dma16(const __amdgpu_buffer_rsrc_t r, unsigned char *lds, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, (int)voff, (int)soff, 0, 0);
}

// x [N][C][H][W] bf16 -> channel-last, zero-padded (interior only: the halo is zeroed once, when the buffer is made).
//   par = 0: one plane   Xc[n][y + yoff][x + xoff][c], Hp x Wp  (3x3 stride 1: yoff = xoff = 1, Hp = H + 2; dY of a stride-2 dgrad:
//            yoff = xoff = 0, Hp = H + 1 -- its taps look down / right)
//   par = 1: four parity planes  Xp[n][2 pr + pc][a][b][c], Hp = H/2 + 1:  x[n][c][2 (a - 1) + pr][2 (b - 1) + pc]  (3x3 stride 2)
// Tiled for the memory system: a block moves 64 channels x 64 consecutive pixels of one image -- in: 128-byte
// runs of a channel plane (16-byte loads at any 2-byte alignment: gfx950 serves them at the streaming rate), out: 128-byte channel runs.
typedef u32x4 __attribute__((aligned(2))) u32x4_u2;
__global__ void __launch_bounds__(256)
cl_relayout64_kernel(const u16 *__restrict__ x, u16 *__restrict__ xp, int C, int P, int W, int Hp, int Wp, int yoff, int xoff, FastDiv fdW, int par) {
    __shared__ __attribute__((aligned(16))) u16 tile[64 * 72];  // [pixel][64 channels], pitch 72
    const int c0 = blockIdx.x * 64, p0 = blockIdx.y * 64, n = blockIdx.z;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int idx = threadIdx.x + 256 * u, cc = idx >> 3, part = idx & 7;
        // (pixels past the plane's end are loaded from the next plane / the tensor's guard bytes and never stored)
        const u32x4 v = *(const u32x4_u2 *)(x + ((size_t)n * C + c0 + cc) * P + p0 + part * 8);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            tile[(part * 8 + 2 * e) * 72 + cc] = (u16)(v[e] & 0xffffu);
            tile[(part * 8 + 2 * e + 1) * 72 + cc] = (u16)(v[e] >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int idx = threadIdx.x + 256 * u, px = idx >> 3, piece = idx & 7;
        const int p = p0 + px;
        if (p < P) {
            const uint32_t y = fd_div((uint32_t)p, fdW), xx = (uint32_t)p - y * W;
            // par: the four parity planes of a stride-2 forward (Hp = H / 2 + 1): plane 2 (y & 1) + (x & 1), row (y >> 1) + 1, column (x >> 1) + 1
            const size_t o = par ? ((((size_t)n * 4 + 2 * (y & 1) + (xx & 1)) * Hp + (y >> 1) + 1) * Wp + (xx >> 1) + 1) * C
                                 : (((size_t)n * Hp + y + yoff) * Wp + xx + xoff) * C;
            *(u32x4 *)(xp + o + c0 + piece * 8) = *(const u32x4 *)(tile + px * 72 + piece * 8);
        }
    }
}

// Out[m][col] = sum over taps t, channels c:  A[tap_w[t]][c][m] * In[pixel(col) + tap_delta[t]][c]      (NCHW output, bf16)
// WM: waves along M.  2: workgroup tile 128 (M) x 128 (pixels), waves 2 x 2; 1: 64 x 256, waves 1 x 4 (64-channel layers).  Wave tile 64 x 64.
// NBUF: operand buffers (2: the DMA of k-step i + 1 runs under the MFMAs of k-step i, one barrier per k-step; 1: half the LDS, two
// barriers per k-step, the overlap comes from the other workgroups of the CU)
#ifdef CL_STAMP /* diagnostic build only (tools/variant.sh stamp "-DCL_STAMP" kernels_cl_bf16.hip; tools/diag/cl_stamps.py): where wave 0 of a workgroup spends its cycles */
__device__ unsigned long long cl_stamps[8 * 16384];
#define CL_NOW() __builtin_amdgcn_s_memtime()
#define CL_ACC(var, t_from, t_to) do { (var) += (t_to) - (t_from); } while (0)
#else
#define CL_NOW() 0ull
#define CL_ACC(var, t_from, t_to) do { } while (0)
#endif
template <int WM, int NBUF>
__global__ void __launch_bounds__(256)
cl_conv_kernel(const u16 *__restrict__ Aop, const u16 *__restrict__ In, u16 *__restrict__ Out, const ClArgs g) {
    constexpr int BM = 64 * WM, BN = 256 / WM;     // 128 x 128 or 64 x 256
    constexpr int NAU = BM / 32, NBU = BN / 32;    // 16-byte DMA pieces per thread and tile
    constexpr int ABYTES = BM * 128, BBYTES = BN * 128, BUF = ABYTES + BBYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char cl_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WM == 2 ? wave >> 1 : 0, wn = WM == 2 ? wave & 1 : wave;

    // ---- block -> tile (XCD-contiguous, M-tiles fastest: the blocks that share a pixel tile share an L2) ----
    uint32_t L = blockIdx.x;
    {
        const uint32_t per = (uint32_t)g.tiles >> 3;
        if (L < per * 8) L = (L & 7) * per + (L >> 3);
    }
    const uint32_t ct = fd_div(L, g.fdM);
    const int m0 = (int)(L - ct * g.mtiles) * BM, n0 = (int)ct * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // ---- DMA state: piece p = tid + 256 u of a tile lands at LDS byte 16 p (lane-linear), i.e. row p >> 3, chunk p & 7; it is
    // loaded from chunk (p & 7) ^ key(row) of that row's 128 source bytes ----
    const int prow = tid >> 3, pchunk = tid & 7;
    uint32_t boff[NBU];                            // byte offset of the pixel row's channel 0 (no tap shift yet)
#pragma unroll
    for (int u = 0; u < NBU; u++) {
        const int row = prow + 32 * u;
        const uint32_t j = (uint32_t)min(n0 + row, g.ncols - 1);   // columns past the end: a pixel that exists (never stored)
        const uint32_t n = fd_div(j, g.fdP), p = j - n * g.P;
        const uint32_t y = fd_div(p, g.fdGW), xx = p - y * g.GW;
        boff[u] = ((n * (uint32_t)g.img_rows + y) * g.Wp + xx) * (uint32_t)g.Cin * 2u + (uint32_t)((pchunk ^ cl_key(row)) * 16);
    }
    uint32_t aoff[NAU];
#pragma unroll
    for (int u = 0; u < NAU; u++) {
        const int row = prow + 32 * u;
        aoff[u] = (uint32_t)(row * 128 + ((pchunk ^ cl_key(row)) * 16));
    }
    const int cpt = g.Cin / 64;                    // k-steps per tap
    const int ntiles = g.ntaps * cpt;
    int ld_t = 0, ld_c = 0;
    // the per-thread part of a piece's address is a CONSTANT (aoff / boff), the k-step's part one SGPR: no vector arithmetic per piece
    const __amdgpu_buffer_rsrc_t rA = cl_rsrc(Aop, 0xffffff00u), rB = cl_rsrc(In, 0xffffff00u);   // (every offset is in range: no bound needed)
    auto issue = [&](const int buf) {
        const uint32_t sa = (uint32_t)(((g.tap_w[ld_t] * cpt + ld_c) * g.M + m0) * 128);
        const uint32_t sb = g.tap_delta[ld_t] + (uint32_t)(ld_c * 128);
        unsigned char *la = cl_smem + buf * BUF + wave * 1024, *lb = cl_smem + buf * BUF + ABYTES + wave * 1024;
#pragma unroll
        for (int u = 0; u < NAU; u++)
            cl_dma16(rA, la + u * 4096, aoff[u], sa);
#pragma unroll
        for (int u = 0; u < NBU; u++) cl_dma16(rB, lb + u * 4096, boff[u], sb);
        // taps fastest: the nine k-steps of one 64-channel chunk read the same 128-byte lines of neighbouring pixels back to back, so the
        // re-reads hit in L2 (channel chunks fastest streamed BN x Cin x 2 bytes per workgroup between two uses of a line: 16 MB per XCD
        // at 1024 channels, four times its L2, and every tap went back to memory)
        if (++ld_t == g.ntaps) { ld_t = 0; ld_c++; }
    };
    const int fr = lane & 31, fk = lane >> 5;
    auto compute = [&](const int buf) {
        const unsigned char *as = cl_smem + buf * BUF, *bs = as + ABYTES;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            bf16x8 av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int row = wm * 64 + i * 32 + fr;
                av[i] = *(const bf16x8 *)(as + row * 128 + (((2 * s + fk) ^ cl_key(row)) * 16));
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int row = wn * 64 + j * 32 + fr;
                bv[j] = *(const bf16x8 *)(bs + row * 128 + (((2 * s + fk) ^ cl_key(row)) * 16));
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j], av[i], acc[i][j], 0, 0, 0);
        }
    };

    [[maybe_unused]] unsigned long long st_begin = CL_NOW(), st_wait = 0, st_bar = 0, st_issue = 0, st_comp = 0;
    if constexpr (NBUF == 2) {
        issue(0);
        for (int it = 0; it < ntiles; it++) {
            [[maybe_unused]] const unsigned long long t0 = CL_NOW();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this thread's pieces of tile `it` have landed ...
            [[maybe_unused]] const unsigned long long t1 = CL_NOW();
            __syncthreads();   // ... and so have everyone's; everyone is done with the other buffer
            [[maybe_unused]] const unsigned long long t2 = CL_NOW();
            if (it + 1 < ntiles) issue((it + 1) & 1);
            [[maybe_unused]] const unsigned long long t3 = CL_NOW();
            compute(it & 1);
            [[maybe_unused]] const unsigned long long t4 = CL_NOW();
            CL_ACC(st_wait, t0, t1); CL_ACC(st_bar, t1, t2); CL_ACC(st_issue, t2, t3); CL_ACC(st_comp, t3, t4);
        }
    } else {
        for (int it = 0; it < ntiles; it++) {
            issue(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    }
    __syncthreads();       // the epilogue re-uses the operand buffers
#ifdef CL_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 16384) { // (the epilogue's own time = the launch's tail; not stamped)
        unsigned long long *o = cl_stamps + (size_t)blockIdx.x * 8;
        o[0] = st_begin; o[1] = CL_NOW(); o[2] = st_wait; o[3] = st_bar; o[4] = st_issue; o[5] = st_comp; o[6] = (unsigned long long)ntiles;
    }
#endif

    // ---- epilogue: accumulator (i, j): rows = the 32 pixels of column block j, row (r & 3) + 8 (r >> 2) + 4 (lane >> 5);
    // column = channel m0 + wm 64 + i 32 + (lane & 31) ----
    const int l31 = lane & 31, hh = lane >> 5;
    const int colw = n0 + wn * 64;                 // the wave's first column
    if (g.bn_part) { // lane-local statistics of the wave's 64 channels x 64 pixels (a lane holds 32 pixels of ONE channel per i)
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) cnt += max(0, min(4, g.ncols - (colw + j * 32 + 8 * q + 4 * hh)));
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int m = m0 + wm * 64 + i * 32 + l31;
            const float s0 = __shfl(acc[i][0][0], l31, 64); // the wave's first pixel (a real one if any is)
            float sd = 0.f, sq = 0.f;
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int col = colw + j * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
                    const float dlt = col < g.ncols ? acc[i][j][r] - s0 : 0.f;
                    sd += dlt;
                    sq = fmaf(dlt, dlt, sq);
                }
            sd += __shfl_xor(sd, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            const int nvw = cnt + __shfl_xor(cnt, 32, 64);
            if (hh == 0) {
                constexpr int PPT = BN / 64;           // 64-column wave groups per tile
                const float inv = nvw > 0 ? 1.0f / (float)nvw : 0.f;
                const size_t plane = (size_t)g.bn_np * g.M;
                const size_t o = (size_t)(ct * PPT + wn) * g.M + m;
                g.bn_part[o] = (float)nvw;
                g.bn_part[plane + o] = nvw > 0 ? s0 + sd * inv : 0.f;
                g.bn_part[2 * plane + o] = fmaxf(sq - sd * sd * inv, 0.f);
            }
        }
    }
    if (g.vw == 1) {
        // planes that are not a multiple of 4 pixels (7 x 7): 2-byte stores straight from the accumulators, (image, pixel) per element
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int col = colw + j * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
                if (col >= g.ncols) continue;
                const uint32_t n = fd_div((uint32_t)col, g.fdP), pp = (uint32_t)col - n * g.P;
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const size_t o = ((size_t)n * g.M + m0 + wm * 64 + i * 32 + l31) * g.P + pp;
                    float v = acc[i][j][r];
                    if (g.addend) v += cl_bf2f(g.addend[o]);
                    Out[o] = (u16)(cl_pack2(v, 0.f) & 0xffffu);
                }
            }
        return;
    }
    // stores through a wave-private LDS image [64 channels][64 pixels] (fp32 when a shortcut gradient is added before the one
    // rounding, else bf16), read back with lanes running ALONG a channel row: a wave instruction writes whole 128-byte lines
    if (g.addend) {
        constexpr int PITCH = 64 * 4 + 16;         // fp32 image, 32 channel rows at a time
        unsigned char *img = cl_smem + wave * (32 * PITCH);
#pragma unroll
        for (int i = 0; i < 2; i++) {
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    typedef float pf4 __attribute__((ext_vector_type(4)));
                    pf4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    *(pf4 *)(img + l31 * PITCH + (j * 32 + 8 * q + 4 * hh) * 4) = v;
                }
            // 4 pixels per lane: 16 lanes along a row, 4 rows per instruction
            const int c = lane & 15, r0 = lane >> 4;
            const int col = colw + c * 4;
            const uint32_t cc = (uint32_t)min(col, g.ncols - 1);
            const uint32_t n = fd_div(cc, g.fdP), pp = cc - n * g.P;
            const bool ok = col < g.ncols;
            const uint32_t obase = n * (uint32_t)(g.M * g.P) + pp;
#pragma unroll
            for (int ps = 0; ps < 8; ps++) {
                typedef float pf4 __attribute__((ext_vector_type(4)));
                const int row = ps * 4 + r0;
                const size_t o = (size_t)(obase + (uint32_t)(m0 + wm * 64 + i * 32 + row) * (uint32_t)g.P);
                const pf4 v = *(const pf4 *)(img + row * PITCH + c * 16);
                if (ok) {
                    const u32x2 ad = *(const u32x2 *)(g.addend + o);
                    u32x2 st = {cl_pack2(v[0] + __uint_as_float(ad[0] << 16), v[1] + __uint_as_float(ad[0] & 0xffff0000u)),
                                cl_pack2(v[2] + __uint_as_float(ad[1] << 16), v[3] + __uint_as_float(ad[1] & 0xffff0000u))};
                    *(u32x2 *)(Out + o) = st;
                }
            }
        }
        return;
    }
    constexpr int PITCH = 64 * 2 + 16;
    unsigned char *img = cl_smem + wave * (64 * PITCH);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                u32x2 v = {cl_pack2(acc[i][j][4 * q], acc[i][j][4 * q + 1]), cl_pack2(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3])};
                *(u32x2 *)(img + (i * 32 + l31) * PITCH + (j * 32 + 8 * q + 4 * hh) * 2) = v;
            }
    auto drain = [&](auto cpx_tag) {
        constexpr int CPX = decltype(cpx_tag)::value; // pixels per lane and store
        constexpr int CPR = 64 / CPX, RPP = 64 / CPR;
        const int c = lane % CPR, r0 = lane / CPR;
        const int col = colw + c * CPX;
        const uint32_t cc = (uint32_t)min(col, g.ncols - 1);
        const uint32_t n = fd_div(cc, g.fdP), pp = cc - n * g.P;
        const bool ok = col < g.ncols;
        const uint32_t obase = n * (uint32_t)(g.M * g.P) + pp;
#pragma unroll
        for (int ps = 0; ps < 64 / RPP; ps++) {
            const int row = ps * RPP + r0;
            const size_t o = (size_t)(obase + (uint32_t)(m0 + wm * 64 + row) * (uint32_t)g.P);
            const unsigned char *sp = img + row * PITCH + c * CPX * 2;
            if constexpr (CPX == 8) { const u32x4 v = *(const u32x4 *)sp; if (ok) *(u32x4 *)(Out + o) = v; }
            else { const u32x2 v = *(const u32x2 *)sp; if (ok) *(u32x2 *)(Out + o) = v; }
        }
    };
    if (g.vw == 8) drain(std::integral_constant<int, 8>{});
    else drain(std::integral_constant<int, 4>{});
}

// Stride-2 dgrad.  dx (2a + ph, 2b + pw) takes the taps whose parity matches: ph = 0 -> r = 1 (dY row a); ph = 1 -> r = 0 (row a + 1) and
// r = 2 (row a); the same in the columns.  A workgroup owns ONE row parity (blockIdx.y) and BOTH column parities of a 128 (channels) x 128
// (grid pixels (n, a, b)) tile: two accumulator sets, filled one after the other from the same re-laid dY (K channels last, one zero row
// / column at the far end: the taps look down / right) -- so that in the epilogue the two results of a grid pixel are neighbours,
// dx(.., 2b) and dx(.., 2b + 1), and a lane stores runs of 8 / 4 consecutive output pixels instead of scattering 2-byte elements at
// stride 2 (what the NCHW kernel's four parity classes do: a third of the b3 projection's dgrad was its epilogue).
struct ClD2Args {
    int K, C;                       // reduction channels (of dY), output channels (of dx)
    int Ho, Wo, P, ncols;           // dY grid per image, P = Ho * Wo, ncols = N * P
    int Wp;                         // padded row length of the re-laid dY (Wo + 1); rows per image Ho + 1
    int mtiles, tiles;
    FastDiv fdP, fdWo, fdM;
    int cpx;                        // grid pixels per lane and store in the epilogue: 4 / 2 / 1 (Wo % cpx == 0)
};
__global__ void __launch_bounds__(256, 2) // (two waves per SIMD: 128 accumulator registers + operands must fit 256)
cl_dgrad2_kernel(const u16 *__restrict__ Aop, const u16 *__restrict__ In, u16 *__restrict__ Out, const ClD2Args g) {
    constexpr int ABYTES = 128 * 128, BBYTES = 128 * 128, BUF = ABYTES + BBYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char cl_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ph = 1 - (int)blockIdx.y;            // the heavier row parity (two row taps) first
    uint32_t L = blockIdx.x;
    {
        const uint32_t per = (uint32_t)g.tiles >> 3;
        if (L < per * 8) L = (L & 7) * per + (L >> 3);
    }
    const uint32_t ct = fd_div(L, g.fdM);
    const int m0 = (int)(L - ct * g.mtiles) * 128, n0 = (int)ct * 128;

    f32x16 acc[2][2][2];                            // [column parity][i][j]
#pragma unroll
    for (int w = 0; w < 2; w++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[w][i][j][r] = 0.f;

    const int prow = tid >> 3, pchunk = tid & 7;
    uint32_t boff[4], aoff[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int row = prow + 32 * u;
        const uint32_t j = (uint32_t)min(n0 + row, g.ncols - 1);
        const uint32_t n = fd_div(j, g.fdP), p = j - n * g.P;
        const uint32_t a = fd_div(p, g.fdWo), b = p - a * g.Wo;
        boff[u] = ((n * (uint32_t)(g.Ho + 1) + a) * g.Wp + b) * (uint32_t)g.K * 2u + (uint32_t)((pchunk ^ cl_key(row)) * 16);
        aoff[u] = (uint32_t)(row * 128 + ((pchunk ^ cl_key(row)) * 16));
    }
    // tap list: column parity 0 (s = 1) first, then column parity 1 (s = 0: dY column b + 1, s = 2: column b); per row tap
    const int nrt = ph ? 2 : 1;                    // row taps: ph = 1: r = 0 (row a + 1), r = 2 (row a); ph = 0: r = 1
    const int cpt = g.K / 64;
    const int nt0 = nrt * cpt, ntiles = 3 * nrt * cpt; // k-steps of column parity 0 / in all
    int ld_i = 0, ld_c = 0;                        // tap index in the list (0 .. 3 nrt - 1), channel chunk
    const __amdgpu_buffer_rsrc_t rA = cl_rsrc(Aop, 0xffffff00u), rB = cl_rsrc(In, 0xffffff00u);
    auto issue = [&](const int buf) {
        // tap ld_i: [0, nrt): pw = 0, row tap ld_i; [nrt, 3 nrt): pw = 1, row tap (ld_i - nrt) >> 1, column tap (ld_i - nrt) & 1
        const int pw1 = ld_i >= nrt;
        const int rt = pw1 ? (ld_i - nrt) >> 1 : ld_i, ctp = pw1 ? (ld_i - nrt) & 1 : 0;
        const int r = ph ? 2 * rt : 1, dh = ph ? 1 - rt : 0;
        const int sx = pw1 ? 2 * ctp : 1, dw = pw1 ? 1 - ctp : 0;
        const uint32_t sa = (uint32_t)((((3 * r + sx) * cpt + ld_c) * g.C + m0) * 128);
        const uint32_t sb = (uint32_t)((dh * g.Wp + dw) * g.K + ld_c * 64) * 2u;
        unsigned char *la = cl_smem + buf * BUF + wave * 1024, *lb = cl_smem + buf * BUF + ABYTES + wave * 1024;
#pragma unroll
        for (int u = 0; u < 4; u++) cl_dma16(rA, la + u * 4096, aoff[u], sa);
#pragma unroll
        for (int u = 0; u < 4; u++) cl_dma16(rB, lb + u * 4096, boff[u], sb);
        // taps fastest within a column parity (see cl_conv_kernel)
        if (ld_i < nrt) { if (++ld_i == nrt) { ld_i = 0; if (++ld_c == cpt) { ld_c = 0; ld_i = nrt; } } }
        else if (++ld_i == 3 * nrt) { ld_i = nrt; ld_c++; }
    };
    const int fr = lane & 31, fk = lane >> 5;
    auto compute = [&](const int buf, auto w_tag) {
        constexpr int W = decltype(w_tag)::value;
        const unsigned char *as = cl_smem + buf * BUF, *bs = as + ABYTES;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            bf16x8 av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int row = wm * 64 + i * 32 + fr;
                av[i] = *(const bf16x8 *)(as + row * 128 + (((2 * s + fk) ^ cl_key(row)) * 16));
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int row = wn * 64 + j * 32 + fr;
                bv[j] = *(const bf16x8 *)(bs + row * 128 + (((2 * s + fk) ^ cl_key(row)) * 16));
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[W][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j], av[i], acc[W][i][j], 0, 0, 0);
        }
    };
    issue(0);
    int it = 0;
    for (; it < nt0; it++) {                       // column parity 0 (the DMA pipeline runs on into parity 1's first k-step)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        issue((it + 1) & 1);
        compute(it & 1, std::integral_constant<int, 0>{});
    }
    for (; it < ntiles; it++) {                    // column parity 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (it + 1 < ntiles) issue((it + 1) & 1);
        compute(it & 1, std::integral_constant<int, 1>{});
    }
    __syncthreads();

    // ---- epilogue: 32 channels at a time through a wave-private LDS image [32 channels][64 grid pixels][2 column parities] (bf16:
    // 256 bytes per row + 16), drained with lanes running along a channel row ----
    const int l31 = lane & 31, hh = lane >> 5;
    constexpr int PITCH = 64 * 4 + 16;
    unsigned char *img = cl_smem + wave * (32 * PITCH);
    const int colw = n0 + wn * 64;
    const int H = 2 * g.Ho, Wd = 2 * g.Wo;
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) { // 4 consecutive grid pixels x 2 parities = 8 consecutive output pixels (where they share a row)
                u32x4 v = {cl_pack2(acc[0][i][j][4 * q], acc[1][i][j][4 * q]), cl_pack2(acc[0][i][j][4 * q + 1], acc[1][i][j][4 * q + 1]),
                           cl_pack2(acc[0][i][j][4 * q + 2], acc[1][i][j][4 * q + 2]), cl_pack2(acc[0][i][j][4 * q + 3], acc[1][i][j][4 * q + 3])};
                *(u32x4 *)(img + l31 * PITCH + (j * 32 + 8 * q + 4 * hh) * 4) = v;
            }
        auto drain = [&](auto cpx_tag) {
            constexpr int CPX = decltype(cpx_tag)::value; // grid pixels per lane: 2 CPX output pixels per store
            constexpr int CPR = 64 / CPX, RPP = 64 / CPR;
            const int c = lane % CPR, r0 = lane / CPR;
            const int col = colw + c * CPX;
            const bool ok = col < g.ncols;               // (ncols % CPX == 0: Wo % CPX == 0)
            const uint32_t cc = ok ? (uint32_t)col : 0u;
            const uint32_t n = fd_div(cc, g.fdP), p = cc - n * g.P;
            const uint32_t a = fd_div(p, g.fdWo), b = p - a * g.Wo;
            const size_t obase = ((size_t)n * g.C * H + (2 * a + ph)) * Wd + 2 * b;
#pragma unroll
            for (int ps = 0; ps < 32 / RPP; ps++) {
                const int row = ps * RPP + r0;
                const size_t o = obase + (size_t)(m0 + wm * 64 + i * 32 + row) * H * Wd;
                const unsigned char *sp = img + row * PITCH + c * CPX * 4;
                if constexpr (CPX == 4) { const u32x4 v = *(const u32x4 *)sp; if (ok) *(u32x4 *)(Out + o) = v; }
                else if constexpr (CPX == 2) { const u32x2 v = *(const u32x2 *)sp; if (ok) *(u32x2 *)(Out + o) = v; }
                else { const uint32_t v = *(const uint32_t *)sp; if (ok) *(uint32_t *)(Out + o) = v; }
            }
        };
        if (g.cpx == 4) drain(std::integral_constant<int, 4>{});
        else if (g.cpx == 2) drain(std::integral_constant<int, 2>{});
        else drain(std::integral_constant<int, 1>{});
    }
}

// Weight gradient:  dW[t][k][c] = sum over pixels  dY[k][pix] * X[pix + D_t][c].  The reduction runs over pixels, which is the CONTIGUOUS index of dY
// (NCHW: the A operand as it lies, 128-byte rows of 64 pixels) and the ROW index of the channel-last input: the B fragments (8 consecutive
// pixels of one channel per lane) are read TRANSPOSED from a [64 pixels][128 channels] LDS image with ds_read_b64_tr_b16 (per 16-lane group a
// 4 row x 16 column block, delivered column-major), 256-byte rows XOR-swizzled so that both the DMA fill and the transposed reads are
// conflict-free.  Workgroup tile 128 (k) x 128 (c) of ONE tap; the reduction = 64-pixel tiles that never cross an image (the tail tile of a
// plane is padded: its missing pixels read a zero page on the B side, so whatever the A side picks up beyond the plane -- finite data of the
// next plane -- is multiplied by zero), split over blockIdx.y; fp32 partials [split][t][k][c], summed in a fixed order by the second stage.
typedef short cl_s16x4 __attribute__((ext_vector_type(4)));
struct ClWgArgs {
    int K, C;                       // dY channels (rows), input channels (columns)
    int P, GW;                      // output pixels per image, output row length
    int img_rows, Wp;               // padded input geometry as ClArgs
    int ptiles;                     // reduction steps (CLW_KPX pixels) per image
    int rtiles;                     // reduction tiles in all = N * ptiles
    int rlen;                       // reduction tiles per split
    int ctiles, mtiles;             // C / 128, K / 128
    uint32_t tap_delta[9];
    FastDiv fdGW, fdPt, fdM, fdT, fd9;
    uint32_t tiles, total8;         // output tiles (with taps); (tiles * splits) rounded down to a multiple of 8
    uint32_t a_bytes, b_bytes;      // num_records of the two buffer descriptors (a piece past a plane is given an offset beyond them: zeros)
};
__device__ __forceinline__ int cl_keyb(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
constexpr int CLW_KPX = 64;                        // pixels per reduction step
__global__ void __launch_bounds__(256, 2)
cl_wgrad_kernel(const u16 *__restrict__ dY, const u16 *__restrict__ Xc, float *__restrict__ part, const ClWgArgs g) {
    constexpr int ABYTES = 128 * 128, BBYTES = 64 * 256, BUF = ABYTES + BBYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char cl_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroups go to the 8 XCDs round-robin by their linear id, and each XCD has its own L2: XCD x is given a CONTIGUOUS eighth of the
    // (split, tile) list, tiles (column block, row block, tap) fastest, so that all the tiles that read one stretch of pixels run on one
    // XCD at about the same time and each operand byte is fetched from HBM by one L2.
    uint32_t L = blockIdx.x;
    if (L < g.total8) L = (L & 7u) * (g.total8 >> 3) + (L >> 3);
    const uint32_t split = fd_div(L, g.fdT), tile = L - split * g.tiles;
    const uint32_t tc = fd_div(tile, g.fd9), t = tile - tc * 9u;
    const uint32_t ct = fd_div(tc, g.fdM);
    const int m0 = (int)(tc - ct * g.mtiles) * 128, c0 = (int)ct * 128;
    const int r_beg = (int)split * g.rlen, r_end = min(g.rtiles, r_beg + g.rlen);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // A pieces: p = tid + 256 u -> row p >> 3 (dY channel), chunk p & 7 (8 pixels); B pieces: p -> pixel row p >> 4, chunk p & 15 (8 channels).
    // Addresses are kept incremental: a reduction tile starts 64 pixels after the one before it (or at pixel 0 of the next image), so the
    // per-thread parts are constants (A) or advance by a constant step with one carry (B); the image / tile origin is wave-uniform.
    const int arow = tid >> 3, achunk = tid & 7;
    const int brow = tid >> 4, bchunk = tid & 15;
    uint32_t aoff[4];
    int alim[4];                                   // the piece holds pixels of the plane while p0 < alim
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int row = arow + 32 * u, sc = achunk ^ cl_key(row);  // LDS chunk achunk of the row holds source chunk sc
        aoff[u] = (uint32_t)(((m0 + row) * g.P + sc * 8) * 2);
        alim[u] = g.P - sc * 8;
    }
    const uint32_t q64 = fd_div(64u, g.fdGW), r64 = 64u - q64 * g.GW;  // 64 pixels further = q64 rows and r64 columns further
    const uint32_t rowb = (uint32_t)g.Wp * g.C * 2u, colb = (uint32_t)g.C * 2u;
    uint32_t boff0[4], bx0[4], boff[4], bx[4];
    const uint32_t tdelta = g.tap_delta[t] + (uint32_t)c0 * 2u;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int row = brow + 16 * u, sc = bchunk ^ cl_keyb(row);
        const uint32_t y = fd_div((uint32_t)row, g.fdGW), xx = (uint32_t)row - y * g.GW;     // pixel `row` of an image (tile 0)
        boff0[u] = y * rowb + xx * colb + tdelta + (uint32_t)(sc * 16);
        bx0[u] = xx;
    }
    int ld_r = r_beg;
    uint32_t ld_n = fd_div((uint32_t)r_beg, g.fdPt);
    int ld_p0 = (r_beg - (int)ld_n * g.ptiles) * 64;
#pragma unroll
    for (int u = 0; u < 4; u++) { // state of the first tile of this split: pixel ld_p0 + row
        const int row = brow + 16 * u, sc = bchunk ^ cl_keyb(row);
        const uint32_t pp = (uint32_t)(ld_p0 + row);
        const uint32_t y = fd_div(pp, g.fdGW), xx = pp - y * g.GW;
        boff[u] = y * rowb + xx * colb + tdelta + (uint32_t)(sc * 16);
        bx[u] = xx;
    }
    const int tin0 = ld_p0 >> 6;
    const __amdgpu_buffer_rsrc_t rA = cl_rsrc(dY, g.a_bytes), rB = cl_rsrc(Xc, g.b_bytes);
    auto issue = [&](const int buf) {
        const uint32_t sa = (uint32_t)(((size_t)ld_n * g.K * g.P + ld_p0) * 2);   // wave-uniform (tensors < 2^32 bytes: mid_cl_supported)
        const uint32_t sb = ld_n * (uint32_t)g.img_rows * rowb;
        unsigned char *la = cl_smem + buf * BUF + wave * 1024, *lb = cl_smem + buf * BUF + ABYTES + wave * 1024;
#pragma unroll
        for (int u = 0; u < 4; u++) cl_dma16(rA, la + u * 4096, ld_p0 < alim[u] ? aoff[u] : CL_OOB, sa);           // beyond the plane: zeros
#pragma unroll
        for (int u = 0; u < 4; u++) cl_dma16(rB, lb + u * 4096, ld_p0 + brow + 16 * u < g.P ? boff[u] : CL_OOB, sb);
        // the next tile: 64 pixels on, or pixel 0 of the next image
        ld_r++;
        ld_p0 += 64;
        if (ld_p0 >= g.ptiles * 64) {
            ld_p0 = 0; ld_n++;
#pragma unroll
            for (int u = 0; u < 4; u++) { boff[u] = boff0[u]; bx[u] = bx0[u]; }
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                uint32_t nx = bx[u] + r64;
                const uint32_t carry = nx >= (uint32_t)g.GW ? 1u : 0u;
                nx -= carry * (uint32_t)g.GW;
                boff[u] += (q64 + carry) * rowb + (nx - bx[u]) * colb;   // (unsigned wrap-around is the subtraction it stands for)
                bx[u] = nx;
            }
        }
    };
    const int fr = lane & 31, fk = lane >> 5;
    const int tq = (lane & 15) >> 2, tp = lane & 3, tcb = (lane >> 4) & 1; // transposed read: row within the 4-row block, column quad, 16-column half
    auto compute_n = [&](const int buf, const int nsub, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const unsigned char *as = cl_smem + buf * BUF, *bs = as + ABYTES;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (!FULL && s >= nsub) break;
            bf16x8 av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int row = wm * 64 + i * 32 + fr;
                av[i] = *(const bf16x8 *)(as + row * 128 + (((2 * s + fk) ^ cl_key(row)) * 16));
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int ch0 = (wn * 64 + j * 32) / 8 + 2 * tcb + (tp >> 1); // 16-byte chunk of the image row this lane addresses
                cl_s16x4 lo, hi;
                {
                    const int row = 16 * s + 8 * fk + tq;
                    lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cl_s16x4 *)(bs + row * 256 + ((ch0 ^ cl_keyb(row)) * 16) + 8 * (tp & 1)));
                }
                {
                    const int row = 16 * s + 8 * fk + 4 + tq;
                    hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cl_s16x4 *)(bs + row * 256 + ((ch0 ^ cl_keyb(row)) * 16) + 8 * (tp & 1)));
                }
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bv[j] = *(bf16x8 *)&v;
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    // full tiles run in an inner loop of their own (one code path: the accumulators stay where they are); the partial tail tile of an
    // image, whose 16-pixel sub-steps past the end of the plane are all zeros, multiplies only the sub-steps that hold pixels.
    // (Measured and dropped: 32-pixel steps in four LDS stages with counted vmcnt waits and no drain -- 8 % slower, the extra barriers
    // cost more than the deeper prefetch returns.)
    if (r_beg < r_end) {
        issue(0);
        const int full = g.P >> 6, tail_sub = ((g.P & 63) + 15) >> 4;
        int it = r_beg, tin = tin0;
        while (it < r_end) {
            const int nf = min(r_end - it, full - tin);
            for (int k = 0; k < nf; k++, it++) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (it + 1 < r_end) issue((it + 1 - r_beg) & 1);
                compute_n((it - r_beg) & 1, 4, std::true_type{});
            }
            tin += max(nf, 0);
            if (it < r_end && tin == full && g.ptiles > full) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (it + 1 < r_end) issue((it + 1 - r_beg) & 1);
                compute_n((it - r_beg) & 1, tail_sub, std::false_type{});
                it++; tin++;
            }
            if (tin >= g.ptiles) tin = 0;
        }
    }
    // partials [split][t][k][c]: accumulator (i, j): column = c0 + wn 64 + j 32 + (lane & 31), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float *o = part + ((size_t)((size_t)split * 9 + t) * g.K) * g.C + c0 + wn * 64 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                o[(size_t)(m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * g.C + j * 32] = acc[i][j][r];
}

// Weight gradient with BOTH operands channel-last (stride 2): A = the dY planes the stride-2 dgrad already uses ([N][Ho+1][Wo+1][K], zero row /
// column at the far end), B = the input's parity planes.  The reduction runs over the N * Ho * Wo pixels as ONE flat list in tiles of 64 --
// a tile may straddle images, so planes of 49 or 196 pixels waste nothing -- and every 16-byte DMA piece is addressed through its own
// pixel's (image, row, column), kept incrementally per thread (the four pixel rows a thread stages are the same for A and for B).  Both
// fragments are transposed reads (ds_read_b64_tr_b16) of [64 pixels][128 channels] LDS images.  Tiles, splits, XCD placement, two-stage
// fixed-order sum as cl_wgrad_kernel.
struct ClWg2Args {
    int K, C;
    int GH, GW;                     // output plane
    int R;                          // reduction length = N * GH * GW pixels
    int rtiles, rlen;               // 64-pixel tiles in all; per split
    int img_rows, Wp;               // parity-plane geometry of the input (as ClArgs)
    uint32_t a_rowb, a_imgb, a_off; // bytes of one dY plane row / one dY image; byte offset of pixel (0, 0) in a plane (stride 1: halo of 1)
    int ctiles, mtiles;
    uint32_t tap_delta[9];
    FastDiv fdGW, fdP, fdM, fdT, fd9;
    uint32_t tiles, total8;
    uint32_t a_bytes, b_bytes;      // num_records of the two buffer descriptors
};
__global__ void __launch_bounds__(256, 2)
cl_wgrad2_kernel(const u16 *__restrict__ dYc, const u16 *__restrict__ Xc, float *__restrict__ part, const ClWg2Args g) {
    constexpr int ABYTES = 64 * 256, BUF = 2 * ABYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char cl_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    uint32_t L = blockIdx.x;
    if (L < g.total8) L = (L & 7u) * (g.total8 >> 3) + (L >> 3);
    const uint32_t split = fd_div(L, g.fdT), tile = L - split * g.tiles;
    const uint32_t tc = fd_div(tile, g.fd9), t = tile - tc * 9u;
    const uint32_t ct = fd_div(tc, g.fdM);
    const int m0 = (int)(tc - ct * g.mtiles) * 128, c0 = (int)ct * 128;
    const int r_beg = (int)split * g.rlen, r_end = min(g.rtiles, r_beg + g.rlen);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // piece p = tid + 256 u of either image: pixel row p >> 4, LDS chunk p & 15, which holds source chunk (p & 15) ^ key(row) (8 channels)
    const int brow = tid >> 4, bchunk = tid & 15;
    const uint32_t rowb = (uint32_t)g.Wp * g.C * 2u, colb = (uint32_t)g.C * 2u, imgb = (uint32_t)g.img_rows * rowb;
    const uint32_t acolb = (uint32_t)g.K * 2u;
    const uint32_t q64 = fd_div(64u, g.fdGW), r64 = 64u - q64 * g.GW;
    const uint32_t tdelta = g.tap_delta[t] + (uint32_t)c0 * 2u;
    uint32_t rn[4], ry[4], rx[4], lanec[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int row = brow + 16 * u;
        const uint32_t j = (uint32_t)r_beg * 64u + (uint32_t)row;      // may lie past the end: such rows load the zero page
        const uint32_t n = fd_div(j, g.fdP), pp = j - n * (uint32_t)(g.GH * g.GW);
        const uint32_t y = fd_div(pp, g.fdGW);
        rn[u] = n; ry[u] = y; rx[u] = pp - y * g.GW;
        lanec[u] = (uint32_t)((bchunk ^ cl_keyb(row)) * 16);
    }
    int ld_j0 = r_beg * 64;
    const __amdgpu_buffer_rsrc_t rA = cl_rsrc(dYc, g.a_bytes), rB = cl_rsrc(Xc, g.b_bytes);
    auto issue = [&](const int buf) {
        unsigned char *la = cl_smem + buf * BUF + wave * 1024, *lb = la + ABYTES;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool ok = ld_j0 + brow + 16 * u < g.R;
            const uint32_t va = rn[u] * g.a_imgb + ry[u] * g.a_rowb + rx[u] * acolb + g.a_off + (uint32_t)m0 * 2u + lanec[u];
            const uint32_t vb = rn[u] * imgb + ry[u] * rowb + rx[u] * colb + tdelta + lanec[u];
            cl_dma16(rA, la + u * 4096, ok ? va : CL_OOB, 0u);   // rows past the end of the list: zeros
            cl_dma16(rB, lb + u * 4096, ok ? vb : CL_OOB, 0u);
        }
        ld_j0 += 64;
#pragma unroll
        for (int u = 0; u < 4; u++) { // 64 pixels on: q64 rows and r64 columns further, carried into the row and the image
            uint32_t nx = rx[u] + r64;
            const uint32_t carry = nx >= (uint32_t)g.GW ? 1u : 0u;
            rx[u] = nx - carry * (uint32_t)g.GW;
            uint32_t ny = ry[u] + q64 + carry;
            while (ny >= (uint32_t)g.GH) { ny -= (uint32_t)g.GH; rn[u]++; }
            ry[u] = ny;
        }
    };
    const int fk = lane >> 5;
    const int tq = (lane & 15) >> 2, tp = lane & 3, tcb = (lane >> 4) & 1;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    auto frag = [&](const unsigned char *img, const int s, const int chan0) -> bf16x8 { // 8 consecutive pixels of one channel per lane
        const int ch0 = chan0 / 8 + 2 * tcb + (tp >> 1);
        const int row0 = 16 * s + 8 * fk + tq, row1 = row0 + 4;
        const cl_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cl_s16x4 *)(img + row0 * 256 + ((ch0 ^ cl_keyb(row0)) * 16) + 8 * (tp & 1)));
        const cl_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cl_s16x4 *)(img + row1 * 256 + ((ch0 ^ cl_keyb(row1)) * 16) + 8 * (tp & 1)));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return *(bf16x8 *)&v;
    };
    auto compute = [&](const int buf) {
        const unsigned char *as = cl_smem + buf * BUF, *bs = as + ABYTES;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            bf16x8 av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; i++) av[i] = frag(as, s, wm * 64 + i * 32);
#pragma unroll
            for (int j = 0; j < 2; j++) bv[j] = frag(bs, s, wn * 64 + j * 32);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    if (r_beg < r_end) {
        issue(0);
        for (int it = r_beg; it < r_end; it++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (it + 1 < r_end) issue((it + 1 - r_beg) & 1);
            compute((it - r_beg) & 1);
        }
    }
    float *o = part + ((size_t)((size_t)split * 9 + t) * g.K) * g.C + c0 + wn * 64 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                o[(size_t)(m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * g.C + j * 32] = acc[i][j][r];
}

// Weight gradient of a 1x1 convolution, both operands as they lie (bf16 NCHW):  dW[k][c] = sum over images and pixels dY[n][k][p] * X[n][c][p].
// The reduction index (pixels) is the contiguous one of BOTH operands, so both stage by LDS-DMA into [128 channels][64 pixels] images of
// 128-byte rows (XOR-swizzled 16-byte chunks) and both fragments are plain ds_read_b128.  Tiles, splits, XCD placement and the two-stage
// fixed-order sum as cl_wgrad_kernel.  A plane that is not a multiple of 8 pixels (P % 8 == 4) ends in a chunk of 4 pixels: that chunk is
// loaded 4 pixels EARLY on both sides (pixels P-8 .. P-1: nothing is read past a row, so nothing past the tensors), and the dY fragment that
// holds it has its first four elements -- pixels already counted by the chunk before -- set to zero.
struct PwWgArgs {
    int K, C, P;
    int ptiles, rtiles, rlen;       // 64-pixel tiles per image; reduction tiles in all = N * ptiles; per split
    int mtiles;                     // K / 128
    FastDiv fdPt, fdM, fdT;
    uint32_t tiles, total8;         // output tiles; (tiles * splits) rounded down to a multiple of 8
    uint32_t a_bytes, b_bytes;      // num_records of the two buffer descriptors
};
__global__ void __launch_bounds__(256, 2)
pw_wgrad_kernel(const u16 *__restrict__ dY, const u16 *__restrict__ X, float *__restrict__ part, const PwWgArgs g) {
    constexpr int ABYTES = 128 * 128, BUF = 2 * ABYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char cl_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroups go to the 8 XCDs round-robin by their linear id, and each XCD has its own L2: XCD x is given a CONTIGUOUS eighth of the
    // (split, tile) list, tiles fastest, so that all the tiles that read one stretch of pixels run on one XCD at about the same time and
    // each operand byte is fetched from HBM by one L2, once (tile-fastest over round-robin XCDs fetched the narrower operand C/128 times)
    uint32_t L = blockIdx.x;
    if (L < g.total8) L = (L & 7u) * (g.total8 >> 3) + (L >> 3);
    const uint32_t split = fd_div(L, g.fdT), tile = L - split * g.tiles;
    const uint32_t ct = fd_div(tile, g.fdM);
    const int m0 = (int)(tile - ct * g.mtiles) * 128, c0 = (int)ct * 128;
    const int r_beg = (int)split * g.rlen, r_end = min(g.rtiles, r_beg + g.rlen);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // pieces of 16 bytes: p = tid + 256 u -> LDS row p >> 3 (channel), LDS chunk p & 7, which holds source chunk sc = chunk ^ key (8 pixels)
    const int arow = tid >> 3, achunk = tid & 7;
    const int rem8 = g.P & 7;
    uint32_t aoff[4], boff[4];
    int scx[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int row = arow + 32 * u, sc = achunk ^ cl_key(row);
        aoff[u] = (uint32_t)(((m0 + row) * g.P + sc * 8) * 2);
        boff[u] = (uint32_t)(((c0 + row) * g.P + sc * 8) * 2);
        scx[u] = sc * 8;
    }
    int ld_r = r_beg;
    uint32_t ld_n = fd_div((uint32_t)r_beg, g.fdPt);
    int ld_p0 = (r_beg - (int)ld_n * g.ptiles) * 64;
    const int tin0 = ld_p0 >> 6;
    const __amdgpu_buffer_rsrc_t rA = cl_rsrc(dY, g.a_bytes), rB = cl_rsrc(X, g.b_bytes);
    auto issue = [&](const int buf) {
        const uint32_t sa = (uint32_t)((size_t)ld_n * g.K * g.P * 2), sb = (uint32_t)((size_t)ld_n * g.C * g.P * 2);   // wave-uniform: the image
        const uint32_t pb = (uint32_t)ld_p0 * 2u;  // the tile's first pixel goes into the LANE offset: the bound check sees only that part, and
        unsigned char *la = cl_smem + buf * BUF + wave * 1024, *lb = la + ABYTES;   // "8 bytes early" must not take it below zero
        const int left = g.P - ld_p0;             // pixels of the plane from this tile's first one on
#pragma unroll
        for (int u = 0; u < 4; u++) {
            // whole chunk inside the plane: as it lies; the 4-pixel end chunk: 4 pixels (8 bytes) early; beyond the plane: zeros
            const int room = left - scx[u];
            cl_dma16(rA, la + u * 4096, room >= 8 ? aoff[u] + pb : room > 0 ? aoff[u] + pb - 8u : CL_OOB, sa);
            cl_dma16(rB, lb + u * 4096, room >= 8 ? boff[u] + pb : room > 0 ? boff[u] + pb - 8u : CL_OOB, sb);
        }
        ld_r++;
        ld_p0 += 64;
        if (ld_p0 >= g.ptiles * 64) { ld_p0 = 0; ld_n++; }
    };
    const int fr = lane & 31, fk = lane >> 5;
    // nsub: 16-pixel sub-steps that hold pixels; cut: chunk (of the tile's eight) that is the shifted 4-pixel end chunk, or -1
    auto compute_n = [&](const int buf, const int nsub, const int cut, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const unsigned char *as = cl_smem + buf * BUF, *bs = as + ABYTES;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (!FULL && s >= nsub) break;
            bf16x8 av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int row = wm * 64 + i * 32 + fr;
                av[i] = *(const bf16x8 *)(as + row * 128 + (((2 * s + fk) ^ cl_key(row)) * 16));
                if (!FULL && 2 * s + fk == cut) { u32x4 v = *(u32x4 *)&av[i]; v[0] = 0u; v[1] = 0u; av[i] = *(bf16x8 *)&v; }
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int row = wn * 64 + j * 32 + fr;
                bv[j] = *(const bf16x8 *)(bs + row * 128 + (((2 * s + fk) ^ cl_key(row)) * 16));
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    if (r_beg < r_end) {
        issue(0);
        const int full = g.P >> 6, tail = g.P & 63, tail_sub = (tail + 15) >> 4, tail_cut = rem8 ? tail >> 3 : -1;
        int it = r_beg, tin = tin0;
        while (it < r_end) {
            const int nf = min(r_end - it, full - tin);
            for (int k = 0; k < nf; k++, it++) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (it + 1 < r_end) issue((it + 1 - r_beg) & 1);
                compute_n((it - r_beg) & 1, 4, -1, std::true_type{});
            }
            tin += max(nf, 0);
            if (it < r_end && tin == full && g.ptiles > full) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (it + 1 < r_end) issue((it + 1 - r_beg) & 1);
                compute_n((it - r_beg) & 1, tail_sub, tail_cut, std::false_type{});
                it++; tin++;
            }
            if (tin >= g.ptiles) tin = 0;
        }
    }
    // partials [split][k][c]
    float *o = part + ((size_t)split * g.K) * g.C + c0 + wn * 64 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                o[(size_t)(m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * g.C + j * 32] = acc[i][j][r];
}

// ---------------------------------------------------------------------------------------------------------------------------
static int cl_launch(hipStream_t st, const u16 *A, const u16 *In, u16 *Out, ClArgs &g, int N, double flops, double bytes) {
    g.ncols = N * g.P;
    g.fdP = make_fastdiv(g.P); g.fdGW = make_fastdiv(g.GW);
    g.vw = g.P % 8 == 0 ? 8 : g.P % 4 == 0 ? 4 : 1;
    static int force = -1;
    if (force < 0) { const char *e = getenv("RESNET_MI_CL_NBUF"); force = e ? atoi(e) : 0; }
    const int nbuf = force == 1 ? 1 : 2;
    const int wmv = g.M % 128 == 0 ? 2 : 1;
    const int bm = 64 * wmv, bn = 256 / wmv;
    const int ctl = mi_cdiv(g.ncols, bn);
    g.mtiles = g.M / bm; g.tiles = g.mtiles * ctl; g.fdM = make_fastdiv(g.mtiles);
    if (g.bn_part) g.bn_np = ctl * (bn / 64);
    size_t lds = (size_t)nbuf * (bm * 128 + bn * 128);
    const size_t img = (size_t)4 * 64 * 144;       // the epilogue's four wave images (fp32 form: 4 x 32 x 272, smaller)
    if (lds < img) lds = img;
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)cl_conv_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + 128) * 128) != hipSuccess ||
            hipFuncSetAttribute((const void *)cl_conv_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (64 + 256) * 128) != hipSuccess) {
            mi_record_error("cl_conv_kernel", "cannot raise the dynamic LDS limit");
            return -1;
        }
        attr_set = 1;
    }
    mi_prof_begin(st, MI_FAM_PCONV, flops, bytes);
#define CL_LAUNCH(WM_, NB_) hipLaunchKernelGGL((cl_conv_kernel<WM_, NB_>), dim3(g.tiles), dim3(256), lds, st, A, In, Out, g)
    if (wmv == 2 && nbuf == 2) CL_LAUNCH(2, 2);
    else if (wmv == 2) CL_LAUNCH(2, 1);
    else if (nbuf == 2) CL_LAUNCH(1, 2);
    else CL_LAUNCH(1, 1);
#undef CL_LAUNCH
    mi_prof_end(st);
    MI_LAUNCH_CHECK("cl_conv_kernel");
    return 0;
}

extern "C" {
/* diagnostic (-DCL_STAMP builds): the per-workgroup stamps of the last cl_conv_kernel launches; -1 in a normal build */
int mi_debug_cl_stamps(unsigned long long *dst, int nblocks) {
#ifdef CL_STAMP
    if (nblocks > 16384) nblocks = 16384;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(cl_stamps), sizeof(unsigned long long) * 8 * (size_t)nblocks) == hipSuccess ? 0 : -1;
#else
    (void)dst; (void)nblocks;
    return -1;
#endif
}
/* shapes the channel-last kernels cover: 3x3, stride 1 or 2, channel counts a multiple of 64 (op 0 forward, 1 dgrad: stride 1 only) */
int mid_cl_supported(int op, int N, int C, int H, int K, int stride) {
    if (stride != 1 && stride != 2) return 0;
    if (op == 1 && stride != 1) return 0;
    if (H % stride || H / stride < 2 || H > 240) return 0;
    if (C % 64 || K % 64) return 0;
    const double inb = (double)N * (stride == 2 ? 4.0 * (H / 2 + 1) * (H / 2 + 1) : (double)(H + 2) * (H + 2)) * (op == 0 ? C : K) * 2;
    if (inb >= 4294000000.0 || (double)N * K * (H / stride) * (H / stride) >= 2147480000.0 || (double)N * C * H * H >= 2147480000.0) return 0; /* 32-bit byte offsets */
    return 1;
}
/* bytes of the re-laid operand: forward input (stride 1: one plane with a halo of 1; stride 2: four parity planes), or (op 1) the
 * output gradient of a stride-1 dgrad (one plane with a halo of 1, K channels) */
size_t mid_cl_operand_bytes(int op, int N, int C, int H, int K, int stride) {
    if (op == 1) return (size_t)N * (H + 2) * (H + 2) * K * 2;
    if (stride == 2) return (size_t)N * 4 * (H / 2 + 1) * (H / 2 + 1) * C * 2;
    return (size_t)N * (H + 2) * (H + 2) * C * 2;
}
/* x (bf16 NCHW, C channels, H x H) -> the padded channel-last operand; the halo of xp must be zero (zero the buffer once when it is made).
 * parity != 0: the four parity planes of a stride-2 forward; else one plane with a halo of 1 */
int mid_cl_relayout(mid_stream s, const void *x, void *xp, int N, int C, int H, int parity) {
    /* counted in the 3x3 family's time (no flops of its own): it is part of what the channel-last route costs */
    mi_prof_begin((hipStream_t)s, MI_FAM_PCONV, 0.0, 4.0 * (double)N * C * H * H);
    if (parity) {
        const int Ho = H / 2;
        // (the row-pair kernel cl_relayout_kernel<1> moved 3.0 TB/s; the 64 x 64 tile form with parity addressing on the way out does 3.8+)
        hipLaunchKernelGGL(cl_relayout64_kernel, dim3(C / 64, mi_cdiv(H * H, 64), N), dim3(256), 0, (hipStream_t)s, (const u16 *)x, (u16 *)xp, C, H * H, H, Ho + 1, Ho + 1, 0, 0, make_fastdiv(H), 1);
    } else {
        hipLaunchKernelGGL(cl_relayout64_kernel, dim3(C / 64, mi_cdiv(H * H, 64), N), dim3(256), 0, (hipStream_t)s, (const u16 *)x, (u16 *)xp, C, H * H, H, H + 2, H + 2, 1, 1, make_fastdiv(H), 0);
    }
    mi_prof_end((hipStream_t)s);
    MI_LAUNCH_CHECK("cl_relayout_kernel");
    return 0;
}
/* dY (bf16 NCHW, K channels, Ho x Ho) -> channel-last with one zero row / column at the far end: the operand of the stride-2 dgrad */
int mid_cl_relayout_end(mid_stream s, const void *dy, void *dyp, int N, int K, int Ho) {
    mi_prof_begin((hipStream_t)s, MI_FAM_PCONV, 0.0, 4.0 * (double)N * K * Ho * Ho);
    hipLaunchKernelGGL(cl_relayout64_kernel, dim3(K / 64, mi_cdiv(Ho * Ho, 64), N), dim3(256), 0, (hipStream_t)s, (const u16 *)dy, (u16 *)dyp, K, Ho * Ho, Ho, Ho + 1, Ho + 1, 0, 0,
                       make_fastdiv(Ho), 0);
    mi_prof_end((hipStream_t)s);
    MI_LAUNCH_CHECK("cl_relayout64_kernel");
    return 0;
}
size_t mid_cl_dgrad2_operand_bytes(int N, int K, int Ho) { return (size_t)N * (Ho + 1) * (Ho + 1) * K * 2; }
int mid_cl_dgrad2_supported(int N, int C, int H, int K) {
    if (H % 2 || H < 4 || H > 240 || C % 128 || K % 64) return 0;
    if ((double)N * (H / 2 + 1) * (H / 2 + 1) * K * 2 >= 4294000000.0 || (double)N * C * H * H >= 2147480000.0) return 0;
    return 1;
}
/* dx (bf16 NCHW, C channels, H x H) = the 3x3 stride-2 dgrad of dyp (mid_cl_relayout_end of dY, K channels, H/2 x H/2) with
 * a_tiles = the dgrad k-step tiles [t][k/64][C][64].  Every element of dx is written (no addend). */
int mid_cl_dgrad2(mid_stream s, const void *dyp, const void *a_tiles, void *dx, int N, int C, int H, int K) {
    hipStream_t st = (hipStream_t)s;
    if (!mid_cl_dgrad2_supported(N, C, H, K)) { mi_record_error("mid_cl_dgrad2", "shape not covered"); return -2; }
    ClD2Args g = {};
    g.K = K; g.C = C; g.Ho = H / 2; g.Wo = H / 2; g.P = g.Ho * g.Wo; g.ncols = N * g.P; g.Wp = g.Wo + 1;
    g.fdP = make_fastdiv(g.P); g.fdWo = make_fastdiv(g.Wo);
    g.cpx = g.Wo % 4 == 0 ? 4 : g.Wo % 2 == 0 ? 2 : 1;
    g.mtiles = C / 128; g.tiles = g.mtiles * mi_cdiv(g.ncols, 128); g.fdM = make_fastdiv(g.mtiles);
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)cl_dgrad2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess) { mi_record_error("cl_dgrad2_kernel", "cannot raise the dynamic LDS limit"); return -1; }
        attr_set = 1;
    }
    mi_prof_begin(st, MI_FAM_PCONV, 2.0 * 9 * (double)N * g.P * C * K, 2.0 * ((double)N * K * g.P + (double)N * C * H * H) + 4.0 * 9 * C * K);
    hipLaunchKernelGGL(cl_dgrad2_kernel, dim3(g.tiles, 2), dim3(256), 65536, st, (const u16 *)a_tiles, (const u16 *)dyp, (u16 *)dx, g);
    mi_prof_end(st);
    MI_LAUNCH_CHECK("cl_dgrad2_kernel");
    return 0;
}
static int cl_wgrad_splits(int N, int C, int K, int P) {
    const long tiles = (long)(K / 128) * (C / 128) * 9, rt = (long)N * ((P + CLW_KPX - 1) / CLW_KPX);
    int best = 1;
    double best_eff = 0;
    for (int sp = 1; sp <= 256; sp++) {
        if (sp > 1 && rt / sp < 24) break;
        const double waves = (double)tiles * sp / 512.0;
        const double eff = waves / (double)((long)((tiles * sp + 511) / 512));
        if (eff > best_eff + 0.02) { best_eff = eff; best = sp; }
    }
    return best;
}
/* weight gradient on the channel-last input: 3x3, stride 1 or 2; C % 128, K % 128, output planes a multiple of 4 pixels */
int mid_cl_wgrad_supported(int N, int C, int H, int K, int stride) {
    if (!mid_cl_supported(0, N, C, H, K, stride)) return 0;
    const int P = (H / stride) * (H / stride);
    return C % 128 == 0 && K % 128 == 0 && P % 4 == 0;
}
size_t mid_cl_wgrad_part_floats(int N, int C, int H, int K, int stride) {
    const int P = (H / stride) * (H / stride);
    return (size_t)cl_wgrad_splits(N, C, K, P) * 9 * K * C;
}
/* dw (fp32 KCRS) from dy (bf16 NCHW, K channels, H/stride planes) and xp = the forward's re-laid input (mid_cl_relayout); part: scratch of
 * mid_cl_wgrad_part_floats floats */
int mid_cl_wgrad(mid_stream s, const void *xp, const void *dy, float *dw, float *part, size_t part_floats, int N, int C, int H, int K, int stride) {
    hipStream_t st = (hipStream_t)s;
    if (!mid_cl_wgrad_supported(N, C, H, K, stride)) { mi_record_error("mid_cl_wgrad", "shape not covered"); return -2; }
    ClWgArgs g = {};
    g.K = K; g.C = C; g.GW = H / stride; g.P = g.GW * g.GW;
    if (stride == 2) {
        const int Hp = g.GW + 1;
        g.img_rows = 4 * Hp; g.Wp = Hp;
        for (int t = 0; t < 9; t++) {
            const int r = t / 3, sx = t % 3, q = 2 * ((r + 1) & 1) + ((sx + 1) & 1);
            g.tap_delta[t] = (uint32_t)(((q * Hp + (r > 0)) * Hp + (sx > 0)) * C) * 2u;
        }
    } else {
        g.img_rows = H + 2; g.Wp = H + 2;
        for (int t = 0; t < 9; t++) g.tap_delta[t] = (uint32_t)(((t / 3) * g.Wp + (t % 3)) * C) * 2u;
    }
    g.ptiles = (g.P + CLW_KPX - 1) / CLW_KPX; g.rtiles = N * g.ptiles;
    const int splits = cl_wgrad_splits(N, C, K, g.P);
    if (part_floats < (size_t)splits * 9 * K * C) { mi_record_error("mid_cl_wgrad", "workspace too small"); return -3; }
    g.rlen = mi_cdiv(g.rtiles, splits);
    const int used = mi_cdiv(g.rtiles, g.rlen);
    g.ctiles = C / 128; g.mtiles = K / 128;
    g.fdGW = make_fastdiv(g.GW); g.fdPt = make_fastdiv(g.ptiles); g.fdM = make_fastdiv(g.mtiles); g.fdT = make_fastdiv(g.mtiles * g.ctiles * 9); g.fd9 = make_fastdiv(9);
    g.tiles = (uint32_t)(g.mtiles * g.ctiles * 9); g.total8 = (uint32_t)(g.mtiles * g.ctiles * 9 * used) & ~7u;
    g.a_bytes = (uint32_t)((size_t)N * K * g.P * 2); g.b_bytes = (uint32_t)mid_cl_operand_bytes(0, N, C, H, K, stride);
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)cl_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess) { mi_record_error("cl_wgrad_kernel", "cannot raise the dynamic LDS limit"); return -1; }
        attr_set = 1;
    }
    mi_prof_begin(st, MI_FAM_PCONV, 2.0 * 9 * (double)N * g.P * C * K, 2.0 * ((double)N * C * H * H + (double)N * g.P * K) + 4.0 * 9 * C * K);
    hipLaunchKernelGGL(cl_wgrad_kernel, dim3(g.mtiles * g.ctiles * 9 * used), dim3(256), 65536, st, (const u16 *)dy, (const u16 *)xp, part, g);
    const int rr = mi_igemm_wgrad_reduce(st, part, dw, K, C, 3, used);
    mi_prof_end(st);
    if (rr) return rr;
    MI_LAUNCH_CHECK("cl_wgrad_kernel");
    return 0;
}
static int cl_wgrad2_splits(int N, int C, int K, int P) {
    const long tiles = (long)(K / 128) * (C / 128) * 9, rt = ((long)N * P + 63) / 64;
    int best = 1;
    double best_eff = 0;
    for (int sp = 1; sp <= 256; sp++) {
        if (sp > 1 && rt / sp < 24) break;
        const double waves = (double)tiles * sp / 512.0;
        const double eff = waves / (double)((long)((tiles * sp + 511) / 512));
        if (eff > best_eff + 0.02) { best_eff = eff; best = sp; }
    }
    return best;
}
/* weight gradient of a 3x3 layer from the channel-last planes of BOTH operands.  stride 2: xp = mid_cl_relayout(parity) of the input,
 * dyp = mid_cl_relayout_end of dY (what mid_cl_dgrad2 reads); stride 1: xp = the input with a halo of 1, dyp = dY with a halo of 1 (what
 * mid_cl_dgrad reads).  C % 128, K % 128, any plane size */
int mid_cl_wgrad2_supported(int N, int C, int H, int K, int stride) {
    if (!mid_cl_supported(0, N, C, H, K, stride)) return 0;
    if (C % 128 || K % 128) return 0;
    const int Ho = H / stride, Hq = stride == 2 ? Ho + 1 : Ho + 2;
    if ((double)N * Hq * Hq * K * 2 >= 4294000000.0 || (double)N * Ho * Ho + 64 >= 2147480000.0) return 0;
    return 1;
}
size_t mid_cl_wgrad2_part_floats(int N, int C, int H, int K, int stride) { return (size_t)cl_wgrad2_splits(N, C, K, (H / stride) * (H / stride)) * 9 * K * C; }
int mid_cl_wgrad2(mid_stream s, const void *xp, const void *dyp, float *dw, float *part, size_t part_floats, int N, int C, int H, int K, int stride) {
    hipStream_t st = (hipStream_t)s;
    if (!mid_cl_wgrad2_supported(N, C, H, K, stride)) { mi_record_error("mid_cl_wgrad2", "shape not covered"); return -2; }
    ClWg2Args g = {};
    g.K = K; g.C = C; g.GH = H / stride; g.GW = H / stride; g.R = N * g.GH * g.GW;
    if (stride == 2) {
        const int Hp = g.GW + 1;
        g.img_rows = 4 * Hp; g.Wp = Hp;
        for (int t = 0; t < 9; t++) {
            const int r = t / 3, sx = t % 3, q = 2 * ((r + 1) & 1) + ((sx + 1) & 1);
            g.tap_delta[t] = (uint32_t)(((q * Hp + (r > 0)) * Hp + (sx > 0)) * C) * 2u;
        }
        g.a_rowb = (uint32_t)Hp * K * 2u; g.a_imgb = (uint32_t)Hp * g.a_rowb; g.a_off = 0;
    } else {
        const int Hp = H + 2;
        g.img_rows = Hp; g.Wp = Hp;
        for (int t = 0; t < 9; t++) g.tap_delta[t] = (uint32_t)(((t / 3) * Hp + (t % 3)) * C) * 2u;
        g.a_rowb = (uint32_t)Hp * K * 2u; g.a_imgb = (uint32_t)Hp * g.a_rowb; g.a_off = g.a_rowb + (uint32_t)K * 2u;
    }
    g.rtiles = mi_cdiv(g.R, 64);
    const int splits = cl_wgrad2_splits(N, C, K, g.GH * g.GW);
    if (part_floats < (size_t)splits * 9 * K * C) { mi_record_error("mid_cl_wgrad2", "workspace too small"); return -3; }
    g.rlen = mi_cdiv(g.rtiles, splits);
    const int used = mi_cdiv(g.rtiles, g.rlen);
    g.ctiles = C / 128; g.mtiles = K / 128;
    g.fdGW = make_fastdiv(g.GW); g.fdP = make_fastdiv(g.GH * g.GW); g.fdM = make_fastdiv(g.mtiles); g.fdT = make_fastdiv(g.mtiles * g.ctiles * 9); g.fd9 = make_fastdiv(9);
    g.tiles = (uint32_t)(g.mtiles * g.ctiles * 9); g.total8 = (uint32_t)(g.mtiles * g.ctiles * 9 * used) & ~7u;
    g.a_bytes = (uint32_t)((size_t)N * g.a_imgb); g.b_bytes = (uint32_t)mid_cl_operand_bytes(0, N, C, H, K, stride);
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)cl_wgrad2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess) { mi_record_error("cl_wgrad2_kernel", "cannot raise the dynamic LDS limit"); return -1; }
        attr_set = 1;
    }
    mi_prof_begin(st, MI_FAM_PCONV, 2.0 * 9 * (double)g.R * C * K, 2.0 * ((double)N * C * H * H + (double)g.R * K) + 4.0 * 9 * C * K);
    hipLaunchKernelGGL(cl_wgrad2_kernel, dim3(g.mtiles * g.ctiles * 9 * used), dim3(256), 65536, st, (const u16 *)dyp, (const u16 *)xp, part, g);
    const int rr = mi_igemm_wgrad_reduce(st, part, dw, K, C, 3, used);
    mi_prof_end(st);
    if (rr) return rr;
    MI_LAUNCH_CHECK("cl_wgrad2_kernel");
    return 0;
}
static int pw_wgrad_splits(int N, int C, int K, int P) {
    const long tiles = (long)(K / 128) * (C / 128), rt = (long)N * ((P + 63) / 64);
    int best = 1;
    double best_eff = 0;
    for (int sp = 1; sp <= 512; sp++) {
        if (sp > 1 && rt / sp < 16) break;
        const double waves = (double)tiles * sp / 512.0;
        const double eff = waves / (double)((long)((tiles * sp + 511) / 512));
        if (eff > best_eff + 0.02) { best_eff = eff; best = sp; }
    }
    return best;
}
/* weight gradient of a 1x1 stride-1 convolution from the NCHW bf16 tensors: C % 128, K % 128, planes a multiple of 4 pixels */
int mid_pw_wgrad_supported(int N, int C, int H, int K) {
    const long P = (long)H * H;
    if (C % 128 || K % 128 || P % 4 || P < 8) return 0;
    if ((double)N * C * P >= 2147480000.0 || (double)N * K * P >= 2147480000.0) return 0;
    if ((double)K * P * 2 >= 4294000000.0 || (double)C * P * 2 >= 4294000000.0) return 0;
    return 1;
}
size_t mid_pw_wgrad_part_floats(int N, int C, int H, int K) { return (size_t)pw_wgrad_splits(N, C, K, H * H) * K * C; }
int mid_pw_wgrad(mid_stream s, const void *x, const void *dy, float *dw, float *part, size_t part_floats, int N, int C, int H, int K) {
    hipStream_t st = (hipStream_t)s;
    if (!mid_pw_wgrad_supported(N, C, H, K)) { mi_record_error("mid_pw_wgrad", "shape not covered"); return -2; }
    PwWgArgs g = {};
    g.K = K; g.C = C; g.P = H * H;
    g.ptiles = (g.P + 63) / 64; g.rtiles = N * g.ptiles;
    const int splits = pw_wgrad_splits(N, C, K, g.P);
    if (part_floats < (size_t)splits * K * C) { mi_record_error("mid_pw_wgrad", "workspace too small"); return -3; }
    g.rlen = mi_cdiv(g.rtiles, splits);
    const int used = mi_cdiv(g.rtiles, g.rlen);
    g.mtiles = K / 128;
    const int tiles = g.mtiles * (C / 128);
    g.fdPt = make_fastdiv(g.ptiles); g.fdM = make_fastdiv(g.mtiles); g.fdT = make_fastdiv(tiles);
    g.tiles = (uint32_t)tiles; g.total8 = (uint32_t)(tiles * used) & ~7u;
    g.a_bytes = (uint32_t)((size_t)N * K * g.P * 2); g.b_bytes = (uint32_t)((size_t)N * C * g.P * 2);
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)pw_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess) { mi_record_error("pw_wgrad_kernel", "cannot raise the dynamic LDS limit"); return -1; }
        attr_set = 1;
    }
    mi_prof_begin(st, MI_FAM_GEMM, 2.0 * (double)N * g.P * C * K, 2.0 * ((double)N * C * g.P + (double)N * g.P * K) + 4.0 * C * K);
    hipLaunchKernelGGL(pw_wgrad_kernel, dim3(tiles * used), dim3(256), 65536, st, (const u16 *)dy, (const u16 *)x, part, g);
    const int rr = mi_igemm_wgrad_reduce(st, part, dw, K, C, 1, used);
    mi_prof_end(st);
    if (rr) return rr;
    MI_LAUNCH_CHECK("pw_wgrad_kernel");
    return 0;
}
/* y (bf16 NCHW) = conv3x3 (stride 1 or 2, pad 1) of the re-laid input xp with a_tiles = the forward k-step tiles [t][c/64][K][64]
 * (mid_conv_prelayout_all_bf16 / mid_bf16_prelayout_fwd); parts (optional): BN statistics partials as mid_conv_fwd_bf16 leaves them */
int mid_cl_fwd(mid_stream s, const void *xp, const void *a_tiles, void *y, int N, int C, int H, int K, int stride, mid_bn_parts *parts) {
    if (parts) parts->nparts = 0;
    if (!mid_cl_supported(0, N, C, H, K, stride)) { mi_record_error("mid_cl_fwd", "shape not covered"); return -2; }
    ClArgs g = {};
    g.Cin = C; g.M = K; g.GH = H / stride; g.GW = H / stride; g.P = g.GH * g.GW; g.ntaps = 9;
    if (stride == 2) {
        const int Hp = g.GH + 1;
        g.img_rows = 4 * Hp; g.Wp = Hp;
        for (int t = 0; t < 9; t++) {
            const int r = t / 3, sx = t % 3, q = 2 * ((r + 1) & 1) + ((sx + 1) & 1);
            g.tap_delta[t] = (uint32_t)(((q * Hp + (r > 0)) * Hp + (sx > 0)) * C) * 2u;
            g.tap_w[t] = t;
        }
    } else {
        g.img_rows = H + 2; g.Wp = H + 2;
        for (int t = 0; t < 9; t++) { g.tap_delta[t] = (uint32_t)(((t / 3) * g.Wp + (t % 3)) * C) * 2u; g.tap_w[t] = t; }
    }
    const int bn = K % 128 == 0 ? 128 : 256;
    const int ctl = mi_cdiv(N * g.P, bn);
    if (parts && parts->buf && parts->floats >= (size_t)3 * ctl * (bn / 64) * K) { g.bn_part = parts->buf; parts->nparts = ctl * (bn / 64); }
    return cl_launch((hipStream_t)s, (const u16 *)a_tiles, (const u16 *)xp, (u16 *)y, g, N, 2.0 * 9 * (double)N * g.P * C * K,
                     2.0 * ((double)N * C * H * H + (double)N * g.P * K) + 4.0 * 9 * C * K);
}
/* 1x1 convolution on a DENSE channel-last input ([N * H * H][C], mid_cl_relayout_dense): the same kernel with one tap -- both operands
 * reduction-contiguous, nothing to transpose.  The form every 1x1 layer takes once activations are kept channel-last (DESIGN.md, next
 * round); here an operator, to measure that claim on the benchmark's shapes. */
int mid_cl_pw_supported(int N, int C, int H, int K) {
    if (C % 64 || K % 64 || H < 2 || H > 240) return 0;
    if ((double)N * H * H * C * 2 >= 4294000000.0 || (double)N * K * H * H >= 2147480000.0) return 0;
    return 1;
}
int mid_cl_relayout_dense(mid_stream s, const void *x, void *xp, int N, int C, int H) {
    hipLaunchKernelGGL(cl_relayout64_kernel, dim3(C / 64, mi_cdiv(H * H, 64), N), dim3(256), 0, (hipStream_t)s, (const u16 *)x, (u16 *)xp, C, H * H, H, H, H, 0, 0, make_fastdiv(H), 0);
    MI_LAUNCH_CHECK("cl_relayout64_kernel");
    return 0;
}
int mid_cl_pw_fwd(mid_stream s, const void *xc, const void *a_tiles, void *y, int N, int C, int H, int K, mid_bn_parts *parts) {
    if (parts) parts->nparts = 0;
    if (!mid_cl_pw_supported(N, C, H, K)) { mi_record_error("mid_cl_pw_fwd", "shape not covered"); return -2; }
    ClArgs g = {};
    g.Cin = C; g.M = K; g.GH = H; g.GW = H; g.P = H * H; g.ntaps = 1;
    g.img_rows = H; g.Wp = H;
    g.tap_delta[0] = 0; g.tap_w[0] = 0;
    const int bn = K % 128 == 0 ? 128 : 256;
    const int ctl = mi_cdiv(N * g.P, bn);
    if (parts && parts->buf && parts->floats >= (size_t)3 * ctl * (bn / 64) * K) { g.bn_part = parts->buf; parts->nparts = ctl * (bn / 64); }
    return cl_launch((hipStream_t)s, (const u16 *)a_tiles, (const u16 *)xc, (u16 *)y, g, N, 2.0 * (double)N * g.P * C * K,
                     2.0 * ((double)N * C * g.P + (double)N * g.P * K) + 4.0 * C * K);
}
/* dx (bf16 NCHW, C channels) = the stride-1 dgrad of dyp = dY re-laid with a halo of 1 (K channels), a_tiles = the dgrad k-step tiles
 * [t][k/64][C][64]; addend (optional, bf16 NCHW like dx; may be dx itself): added before the one rounding */
int mid_cl_dgrad(mid_stream s, const void *dyp, const void *a_tiles, void *dx, const void *addend, int N, int C, int H, int K) {
    if (!mid_cl_supported(1, N, C, H, K, 1)) { mi_record_error("mid_cl_dgrad", "shape not covered"); return -2; }
    ClArgs g = {};
    g.Cin = K; g.M = C; g.GH = H; g.GW = H; g.P = H * H; g.ntaps = 9;
    g.img_rows = H + 2; g.Wp = H + 2;
    for (int t = 0; t < 9; t++) { // dx(h, w) takes tap (r, s) from dY(h + 1 - r, w + 1 - s): padded row h + 2 - r
        g.tap_delta[t] = (uint32_t)(((2 - t / 3) * g.Wp + (2 - t % 3)) * K) * 2u;
        g.tap_w[t] = t;
    }
    g.addend = (const u16 *)addend;
    return cl_launch((hipStream_t)s, (const u16 *)a_tiles, (const u16 *)dyp, (u16 *)dx, g, N, 2.0 * 9 * (double)N * g.P * C * K,
                     2.0 * ((double)N * K * g.P + (double)N * C * g.P * (addend ? 2 : 1)) + 4.0 * 9 * C * K);
}
}
