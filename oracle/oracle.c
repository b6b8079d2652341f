/*
 * oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see oracle.h; parity unpinned by
 * reference fixtures).  Each function cites the reference lines it restates.
 *
 * Numerics: accumulators are `acc_t` = float (liboracle_f32.so, the faithful
 * sequential-fp32 restatement; a*b+c written as fmaf, which is what nvcc's
 * default -fmad=true does to the reference's `+=` lines) or double
 * (liboracle_f64.so, -DORC_ACC_DOUBLE, the "true value" used to set tolerances).
 * OpenMP only ever splits INDEPENDENT outputs, so the per-output summation
 * order is the reference's loop order for any thread count.
 */
#include "oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORC_ACC_DOUBLE
typedef double acc_t;
#define FMA(a, b, c) ((acc_t)(a) * (acc_t)(b) + (c))
#define SQRT(x) sqrt(x)
#define POW(x, y) pow(x, y)
#define EXP(x) exp(x)
int orc_acc_is_double(void) { return 1; }
#else
typedef float acc_t;
#define FMA(a, b, c) fmaf((a), (b), (c))
#define SQRT(x) sqrtf(x)
#define POW(x, y) powf(x, y)
#define EXP(x) expf(x)
int orc_acc_is_double(void) { return 0; }
#endif

static int g_threads = 1;
void orc_set_threads(int n) {
    g_threads = n < 1 ? 1 : n;
#ifdef _OPENMP
    omp_set_num_threads(g_threads);
#endif
}
int orc_get_threads(void) { return g_threads; }

/* ------------------------------------------------------------------ */
/* resnet.cu:109-156 doConvolution.  y[n,oh,ow,k] = sum over (row_offset, col_offset,
 * in_channel) in THAT loop order of w[k][c][r][s] * x[n, st*oh+r-k/2, st*ow+s-k/2, c];
 * zero outside; Ho = H/stride.  KB output filters are carried together (independent
 * chains, same per-output order) purely for CPU speed. */
#define KB 8
void orc_conv_fwd(const float *x, const float *w, int H, int k, int C, int K, int stride, int N, float *y) {
    const int Ho = H / stride, half = k / 2, ksz = k * k * C;
    const long rows = (long)N * Ho;
#pragma omp parallel for schedule(static)
    for (long row = 0; row < rows; row++) {
        const int n = (int)(row / Ho), oh = (int)(row % Ho);
        for (int ow = 0; ow < Ho; ow++) {
            float *yo = y + (((size_t)n * Ho + oh) * Ho + ow) * K;
            for (int k0 = 0; k0 < K; k0 += KB) {
                const int kb = (K - k0) < KB ? (K - k0) : KB;
                acc_t acc[KB];
                for (int j = 0; j < KB; j++) acc[j] = 0;
                for (int ro = -half; ro <= half; ro++) {
                    for (int co = -half; co <= half; co++) {
                        const int ih = stride * oh + ro, iw = stride * ow + co;
                        const int inb = !(ih < 0 || ih >= H || iw < 0 || iw >= H);
                        const float *xi = x + (((size_t)n * H + (inb ? ih : 0)) * H + (inb ? iw : 0)) * C;
                        const float *wk = w + (size_t)k0 * ksz + k * (ro + half) + (co + half);
                        for (int c = 0; c < C; c++) {
                            const float xv = inb ? xi[c] : 0.0f;
                            const float *wc = wk + (size_t)k * k * c;
                            for (int j = 0; j < kb; j++) acc[j] = FMA(wc[(size_t)j * ksz], xv, acc[j]);
                        }
                    }
                }
                for (int j = 0; j < kb; j++) yo[k0 + j] = (float)acc[j];
            }
        }
    }
}

/* resnet.cu:166-219 convolutionDerivInput: for each input element, loop out_filt, row_offset,
 * col_offset; candidate output = (ih/stride + ro, iw/stride + co); tap index
 * kr = ih - oh*stride + half must lie in [0,k) (zero term added otherwise, kept: adding 0.0f
 * is exact).  to_add accumulates into dx (residual join, :212-217). */
void orc_conv_dgrad(const float *w, const float *dy, int H, int k, int C, int K, int stride, int N, int to_add,
                    float *dx) {
    const int Ho = H / stride, half = k / 2, ksz = k * k * C;
    const long rows = (long)N * H;
#pragma omp parallel for schedule(static)
    for (long row = 0; row < rows; row++) {
        const int n = (int)(row / H), ih = (int)(row % H);
        acc_t *acc = (acc_t *)malloc(sizeof(acc_t) * (size_t)C);
        for (int iw = 0; iw < H; iw++) {
            for (int c = 0; c < C; c++) acc[c] = 0;
            const int ohs = ih / stride, ows = iw / stride;
            for (int kf = 0; kf < K; kf++) {
                for (int ro = -half; ro <= half; ro++) {
                    for (int co = -half; co <= half; co++) {
                        const int oh = ohs + ro, ow = ows + co;
                        const int kr = ih - oh * stride + half, kc = iw - ow * stride + half;
                        if (kr < 0 || kr >= k || kc < 0 || kc >= k || oh < 0 || oh >= Ho || ow < 0 || ow >= Ho)
                            continue; /* reference adds an exact 0 here */
                        const float d = dy[(((size_t)n * Ho + oh) * Ho + ow) * K + kf];
                        const float *wp = w + (size_t)kf * ksz + k * kr + kc;
                        /* reference: total += w*d  (product rounded, then added: the product is
                         * stored in out_spatial_val_deriv first, resnet.cu:204-206) */
                        for (int c = 0; c < C; c++) {
#ifdef ORC_ACC_DOUBLE
                            acc[c] += (double)wp[(size_t)k * k * c] * (double)d;
#else
                            float prod = wp[(size_t)k * k * c] * d;
                            acc[c] += prod;
#endif
                        }
                    }
                }
            }
            float *o = dx + (((size_t)n * H + ih) * H + iw) * C;
            if (to_add)
                for (int c = 0; c < C; c++) o[c] = (float)((acc_t)o[c] + acc[c]);
            else
                for (int c = 0; c < C; c++) o[c] = (float)acc[c];
        }
        free(acc);
    }
}

/* resnet.cu:227-281 convolutionDerivWeights: one serial sum over (s, out_row, out_col) per
 * weight; product stored then added (:274-276). */
void orc_conv_wgrad(const float *x, const float *dy, int H, int k, int C, int K, int stride, int N, float *dw) {
    const int Ho = H / stride, half = k / 2, ksz = k * k * C;
    const long items = (long)K * k * k;
#pragma omp parallel for schedule(dynamic, 1)
    for (long it = 0; it < items; it++) {
        const int kf = (int)(it / (k * k)), kr = (int)((it / k) % k), kc = (int)(it % k);
        acc_t *acc = (acc_t *)calloc((size_t)C, sizeof(acc_t));
        for (int s = 0; s < N; s++) {
            for (int oh = 0; oh < Ho; oh++) {
                const int ih = stride * oh + kr - half;
                if (ih < 0 || ih >= H) continue;
                for (int ow = 0; ow < Ho; ow++) {
                    const int iw = stride * ow + kc - half;
                    if (iw < 0 || iw >= H) continue;
                    const float d = dy[(((size_t)s * Ho + oh) * Ho + ow) * K + kf];
                    const float *xi = x + (((size_t)s * H + ih) * H + iw) * C;
                    for (int c = 0; c < C; c++) {
#ifdef ORC_ACC_DOUBLE
                        acc[c] += (double)xi[c] * (double)d;
#else
                        float prod = xi[c] * d;
                        acc[c] += prod;
#endif
                    }
                }
            }
        }
        for (int c = 0; c < C; c++) dw[(size_t)kf * ksz + (size_t)k * k * c + k * kr + kc] = (float)acc[c];
        free(acc);
    }
}

/* resnet.cu:289-342 doBatchNormAndActivate: per channel, three passes over (s,i,j). */
void orc_bn_fwd(const float *x, const float *gamma, const float *beta, int H, int C, int N, float eps, float *means,
                float *vars, float *xhat, float *normalized, float *activated, int to_activate) {
    const size_t M = (size_t)N * H * H;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; c++) {
        acc_t sum = 0;
        for (size_t i = 0; i < M; i++) sum += x[i * C + c];
        const acc_t mean = sum / (acc_t)(N * H * H);
        means[c] = (float)mean;
        acc_t vs = 0;
        for (size_t i = 0; i < M; i++) {
            acc_t d = (acc_t)x[i * C + c] - mean;
            vs = FMA(d, d, vs);
        }
        const acc_t var = vs / (acc_t)(N * H * H);
        vars[c] = (float)var;
        const acc_t sd = SQRT(var + (acc_t)eps);
        for (size_t i = 0; i < M; i++) {
            acc_t t = ((acc_t)x[i * C + c] - mean) / sd;
            acc_t nv = FMA((acc_t)gamma[c], t, (acc_t)beta[c]);
            if (xhat) xhat[i * C + c] = (float)t;
            if (normalized) normalized[i * C + c] = (float)nv;
            if (activated) activated[i * C + c] = to_activate ? fmaxf((float)nv, 0.0f) : (float)nv;
        }
    }
}

/* resnet.cu:350-426 activationAndBatchNormDeriv.  xhat may be NULL (recomputed from x). */
void orc_bn_bwd(const float *x, const float *gamma, int H, int C, int N, float eps, const float *means,
                const float *vars, const float *xhat, const float *activated, const float *dy, float *dxhat,
                float *dgamma, float *dbeta, float *dx, int to_activate_deriv) {
    const size_t M = (size_t)N * H * H;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; c++) {
        const acc_t n_samples = (acc_t)(N * H * H);
        const acc_t g = gamma[c], mean = means[c], var = vars[c];
        const acc_t sd = SQRT(var + (acc_t)eps);
        acc_t dG = 0, dB = 0;
        for (size_t i = 0; i < M; i++) {
            const size_t idx = i * C + c;
            if (to_activate_deriv && activated[idx] <= 0) {
                dxhat[idx] = 0;
            } else {
                const acc_t d = dy[idx];
                const acc_t t = xhat ? (acc_t)xhat[idx] : ((acc_t)x[idx] - mean) / sd;
                dxhat[idx] = (float)(d * g);
                dG = FMA(d, t, dG);
                dB += d;
            }
        }
        dgamma[c] = (float)dG;
        dbeta[c] = (float)dB;
        acc_t dVar = 0, dMean = 0, pvd = 0;
        const acc_t p15 = (acc_t)(-0.5 * (double)POW(var + (acc_t)eps, (acc_t)-1.5));
        const acc_t nrs = (acc_t)(-1.0 / (double)sd);
        for (size_t i = 0; i < M; i++) {
            const size_t idx = i * C + c;
            const acc_t nd = dxhat[idx], df = (acc_t)x[idx] - mean;
            dVar = FMA(nd * df, p15, dVar);
            dMean = FMA(nd, nrs, dMean);
            pvd = FMA((acc_t)-2, df, pvd);
        }
        dMean += dVar * pvd / n_samples;
        for (size_t i = 0; i < M; i++) {
            const size_t idx = i * C + c;
            const acc_t nd = dxhat[idx], df = (acc_t)x[idx] - mean;
            dx[idx] = (float)(nd * (-1 * nrs) + dVar * (2 * df) / n_samples + dMean / n_samples);
        }
    }
}

/* resnet.cu:433-471 doMaxPool: window centred at stride*o, skip OOB, strict '>' (first max wins),
 * init -1024, flat NHWC argmax index. */
void orc_maxpool_fwd(const float *x, int k, int stride, int N, int Hin, int C, int *max_inds, float *y) {
    const int Ho = Hin / stride, half = k / 2;
#pragma omp parallel for schedule(static)
    for (int s = 0; s < N; s++)
        for (int oh = 0; oh < Ho; oh++)
            for (int ow = 0; ow < Ho; ow++)
                for (int c = 0; c < C; c++) {
                    float mv = -1024;
                    int mi = -1024;
                    for (int ro = -half; ro <= half; ro++)
                        for (int co = -half; co <= half; co++) {
                            const int ih = stride * oh + ro, iw = stride * ow + co;
                            if (ih < 0 || ih >= Hin || iw < 0 || iw >= Hin) continue;
                            const int ii = ((s * Hin + ih) * Hin + iw) * C + c;
                            if (x[ii] > mv) { mv = x[ii]; mi = ii; }
                        }
                    const int oi = ((s * Ho + oh) * Ho + ow) * C + c;
                    max_inds[oi] = mi;
                    y[oi] = mv;
                }
}

/* resnet.cu:476-494 maxPoolDeriv after the memset at :2186: plain (non-atomic) scatter.  Overlapping
 * windows race in the reference (hazard h5); the oracle fixes ONE valid execution: outputs are
 * visited in (s,oh,ow,c) order and the last writer wins. */
void orc_maxpool_bwd(const int *max_inds, const float *dy, int Hin, int stride, int C, int N, float *dx) {
    const int Ho = Hin / stride;
    memset(dx, 0, sizeof(float) * (size_t)N * Hin * Hin * C);
    const size_t n = (size_t)N * Ho * Ho * C;
    for (size_t i = 0; i < n; i++) dx[max_inds[i]] = dy[i];
}

/* resnet.cu:500-517 / 522-542 */
void orc_avgpool_fwd(const float *x, int H, int C, int N, float *y) {
    for (int s = 0; s < N; s++)
        for (int c = 0; c < C; c++) {
            acc_t sum = 0;
            for (int i = 0; i < H * H; i++) sum += x[((size_t)s * H * H + i) * C + c];
            y[(size_t)s * C + c] = (float)(sum / (acc_t)(H * H));
        }
}
void orc_avgpool_bwd(const float *dy, int C, int N, int H, float *dx) {
    for (int s = 0; s < N; s++)
        for (int c = 0; c < C; c++) {
            const float v = (float)((acc_t)dy[(size_t)s * C + c] / (acc_t)(H * H));
            for (int i = 0; i < H * H; i++) dx[((size_t)s * H * H + i) * C + c] = v;
        }
}
/* resnet.cu:59-65, 545-564 */
void orc_add(int n, const float *a, const float *b, float *o) { for (int i = 0; i < n; i++) o[i] = a[i] + b[i]; }
void orc_relu(int n, const float *x, float *o) { for (int i = 0; i < n; i++) o[i] = fmaxf(0.0f, x[i]); }
void orc_relu_deriv(int n, const float *x, const float *up, float *o) {
    for (int i = 0; i < n; i++) o[i] = x[i] > 0 ? up[i] : 0.0f;
}
/* resnet.cu:70-85 matMul (m x k)(k x n) row-major, serial z loop; 90-101 transpose */
void orc_matmul(const float *M, const float *Nn, int m, int k, int n, float *out) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < m; r++)
        for (int c = 0; c < n; c++) {
            acc_t v = 0;
            for (int z = 0; z < k; z++) v = FMA(M[(size_t)r * k + z], Nn[(size_t)z * n + c], v);
            out[(size_t)r * n + c] = (float)v;
        }
}
void orc_transpose(const float *in, int rows, int cols, float *out) {
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) out[(size_t)c * rows + r] = in[(size_t)r * cols + c];
}
/* resnet_cudnn.cu:572-583 (max-subtracted; resnet.cu:569-580 is the same without the max) */
void orc_softmax(const float *x, int N, int L, float *out) {
    for (int i = 0; i < N; i++) {
        float mx = x[(size_t)i * L];
        for (int j = 0; j < L; j++) if (x[(size_t)i * L + j] > mx) mx = x[(size_t)i * L + j];
        acc_t sum = 0;
        for (int j = 0; j < L; j++) sum += EXP((acc_t)(x[(size_t)i * L + j] - mx));
        for (int j = 0; j < L; j++) out[(size_t)i * L + j] = (float)(EXP((acc_t)(x[(size_t)i * L + j] - mx)) / sum);
    }
}
void orc_softmax_unstable(const float *x, int N, int L, float *out) {
    for (int i = 0; i < N; i++) {
        acc_t sum = 0;
        for (int j = 0; j < L; j++) sum += EXP((acc_t)x[(size_t)i * L + j]);
        for (int j = 0; j < L; j++) out[(size_t)i * L + j] = (float)(EXP((acc_t)x[(size_t)i * L + j]) / sum);
    }
}
/* resnet.cu:597-602; NO 1/N (the averaging kernel is commented out, :1806-1811) */
void orc_ce_deriv(float *d, const int *labels, int L, int N) {
    for (int i = 0; i < N; i++) d[(size_t)i * L + labels[i]] -= 1;
}
/* resnet.cu:605-662 updateMeans/updateVars/updateParams incl. NaN/Inf guards (prints dropped) */
void orc_adam(int n, float *p, const float *g, float *m, float *v, float lr, float wd, float b1, float b2,
              float cur_b1, float cur_b2, float eps) {
    for (int i = 0; i < n; i++) {
        if (!(isnan(g[i]) || isinf(g[i]))) {
            float gd = g[i] + wd * p[i];
            m[i] = b1 * m[i] + (1 - b1) * gd;
        }
        if (!(isnan(g[i]) || isinf(g[i]))) {
            float gd = g[i] + wd * p[i];
            v[i] = b2 * v[i] + (1 - b2) * gd * gd;
        }
        float ma = m[i] / (1 - cur_b1), va = v[i] / (1 - cur_b2), old = p[i];
        float np = old - (lr * (ma / (sqrtf(va) + eps)) + wd * old);
        if (isnan(np) || isinf(np)) np = old;
        p[i] = np;
    }
}
/* resnet.cu:3363-3383 host loss (sum of -logf) and "wrong" count */
float orc_loss(const float *pred, const int *labels, int N, int L, int *n_wrong) {
    float loss = 0;
    int wrong = 0;
    for (int s = 0; s < N; s++) loss += -1 * logf(pred[(size_t)s * L + labels[s]]);
    for (int s = 0; s < N; s++) {
        float pc = pred[(size_t)s * L + labels[s]];
        for (int c = 0; c < L; c++)
            if (c != labels[s] && pred[(size_t)s * L + c] >= pc) { wrong++; break; }
    }
    if (n_wrong) *n_wrong = wrong;
    return loss;
}

/* ================================================================== */
/* Whole network.                                                      */
typedef struct {
    int H, k, C, K, stride, relu; /* H = input spatial */
    int iw, ig, ib;               /* location indices of weight / gamma / beta */
    float *conv_out, *means, *vars, *xhat, *normalized, *activated;
    float *d_conv_out, *d_xhat, *d_activated; /* d_activated = upstream deriv wrt this unit's output */
} Unit;

typedef struct {
    Unit red, spa, exp, proj;
    int has_proj, Hin, Cin, Hout, Cout;
    float *output, *output_activated, *d_output, *d_output_activated;
} Block;

#define MAXT 1024
struct OrcNet {
    int input, ikd, icf, ics, mpd, mps, nb, final_depth, output, N;
    int *red_flags;
    int n_loc, *sizes;
    float **p, **g, **m, **v;
    float lr, wd, b1, b2, cur_b1, cur_b2, eps;
    float *images; int *labels;
    Unit stem; int *max_inds; float *pool_out, *d_pool_out;
    Block *blocks;
    float *avg, *d_avg, *fc_out, *pred, *d_fc_out;
    int nt; char *tname[MAXT]; void *tptr[MAXT]; size_t tsize[MAXT]; int tshape[MAXT][4];
};

static void *reg(OrcNet *n, const char *name, size_t count, int N, int H, int W, int C) {
    void *p = calloc(count ? count : 1, 4);
    if (!p || n->nt >= MAXT) { fprintf(stderr, "oracle: alloc failure\n"); abort(); }
    n->tname[n->nt] = strdup(name); n->tptr[n->nt] = p; n->tsize[n->nt] = count;
    n->tshape[n->nt][0] = N; n->tshape[n->nt][1] = H; n->tshape[n->nt][2] = W; n->tshape[n->nt][3] = C;
    n->nt++;
    return p;
}
static int add_loc(OrcNet *n, int size) {
    int i = n->n_loc++;
    n->sizes = (int *)realloc(n->sizes, sizeof(int) * n->n_loc);
    n->p = (float **)realloc(n->p, sizeof(float *) * n->n_loc); n->g = (float **)realloc(n->g, sizeof(float *) * n->n_loc);
    n->m = (float **)realloc(n->m, sizeof(float *) * n->n_loc); n->v = (float **)realloc(n->v, sizeof(float *) * n->n_loc);
    n->sizes[i] = size;
    n->p[i] = (float *)calloc(size, 4); n->g[i] = (float *)calloc(size, 4);
    n->m[i] = (float *)calloc(size, 4); n->v[i] = (float *)calloc(size, 4);
    return i;
}
/* unit: registers conv_out ("<a>_applied"), activated ("<b>") and their derivs, BN caches */
static void unit_init(OrcNet *n, Unit *u, int H, int k, int C, int K, int stride, int relu, const char *pre,
                      const char *applied, const char *act, const char *bn) {
    char nm[256];
    u->H = H; u->k = k; u->C = C; u->K = K; u->stride = stride; u->relu = relu;
    u->iw = add_loc(n, k * k * C * K); u->ig = add_loc(n, K); u->ib = add_loc(n, K);
    for (int i = 0; i < K; i++) n->p[u->ig][i] = 1.0f; /* gamma = 1, beta = 0 (resnet.cu:733) */
    const int Ho = H / stride; const size_t sz = (size_t)n->N * Ho * Ho * K;
    snprintf(nm, sizeof nm, "%s%s", pre, applied); u->conv_out = (float *)reg(n, nm, sz, n->N, Ho, Ho, K);
    snprintf(nm, sizeof nm, "d:%s%s", pre, applied); u->d_conv_out = (float *)reg(n, nm, sz, n->N, Ho, Ho, K);
    snprintf(nm, sizeof nm, "%s%s", pre, act); u->activated = (float *)reg(n, nm, sz, n->N, Ho, Ho, K);
    snprintf(nm, sizeof nm, "d:%s%s", pre, act); u->d_activated = (float *)reg(n, nm, sz, n->N, Ho, Ho, K);
    snprintf(nm, sizeof nm, "batch_norms/%s/means", bn); u->means = (float *)reg(n, nm, K, 0, 0, 0, K);
    snprintf(nm, sizeof nm, "batch_norms/%s/vars", bn); u->vars = (float *)reg(n, nm, K, 0, 0, 0, K);
    snprintf(nm, sizeof nm, "batch_norms/%s/normalized_temp", bn); u->xhat = (float *)reg(n, nm, sz, n->N, Ho, Ho, K);
    snprintf(nm, sizeof nm, "batch_norms/%s/normalized", bn); u->normalized = (float *)reg(n, nm, sz, n->N, Ho, Ho, K);
    snprintf(nm, sizeof nm, "d:batch_norms/%s/normalized_temp", bn); u->d_xhat = (float *)reg(n, nm, sz, n->N, Ho, Ho, K);
}

/* Parameter / location order follows init_model_parameters (resnet.cu:805-949); n_locations is COUNTED
 * (hazard h3: the reference's 16+9*n formula, :819, is only right for exactly 4 projections). */
OrcNet *orc_net_create(int input, int ikd, int icf, int ics, int mpd, int mps, int nb, const int *flags,
                       int final_depth, int output, int batch) {
    OrcNet *n = (OrcNet *)calloc(1, sizeof(OrcNet));
    n->input = input; n->ikd = ikd; n->icf = icf; n->ics = ics; n->mpd = mpd; n->mps = mps; n->nb = nb;
    n->final_depth = final_depth; n->output = output; n->N = batch;
    n->red_flags = (int *)malloc(sizeof(int) * (nb > 0 ? nb : 1));
    memcpy(n->red_flags, flags, sizeof(int) * nb);
    n->lr = 1e-4f; n->wd = 0; n->b1 = 0.9f; n->b2 = 0.999f; n->cur_b1 = 1; n->cur_b2 = 1; n->eps = 1e-7f;
    n->images = (float *)reg(n, "input", (size_t)batch * input * input * 3, batch, input, input, 3);
    n->labels = (int *)reg(n, "correct_classes", batch, 0, 0, 0, 0);
    unit_init(n, &n->stem, input, ikd, 3, icf, ics, 1, "", "init_conv_applied", "init_conv_activated", "init");
    const int Hs = input / ics, Hp = Hs / mps;
    n->max_inds = (int *)reg(n, "max_inds", (size_t)batch * Hp * Hp * icf, batch, Hp, Hp, icf);
    n->pool_out = (float *)reg(n, "init_convblock_input", (size_t)batch * Hp * Hp * icf, batch, Hp, Hp, icf);
    n->d_pool_out = (float *)reg(n, "d:init_convblock_input", (size_t)batch * Hp * Hp * icf, batch, Hp, Hp, icf);
    n->blocks = (Block *)calloc(nb > 0 ? nb : 1, sizeof(Block));
    int inc = icf, H = input / 4, red = icf, ex = 4 * icf; /* resnet.cu:857-862 (input/4 hard-wired) */
    for (int i = 0; i < nb; i++) {
        int stride = 1;
        if (flags[i] == 1) { stride = 2; red *= 2; ex *= 2; }
        Block *b = &n->blocks[i];
        char pre[64], bn[64];
        snprintf(pre, sizeof pre, "conv_blocks/%02d/", i);
        b->Hin = H; b->Cin = inc; b->Hout = H / stride; b->Cout = ex;
        snprintf(bn, sizeof bn, "%02d/reduced", i);
        unit_init(n, &b->red, H, 1, inc, red, 1, 1, pre, "reduction_applied", "reduction_activated", bn);
        snprintf(bn, sizeof bn, "%02d/spatial", i);
        unit_init(n, &b->spa, H, 3, red, red, stride, 1, pre, "spatial_applied", "spatial_activated", bn);
        snprintf(bn, sizeof bn, "%02d/expanded", i);
        unit_init(n, &b->exp, H / stride, 1, red, ex, 1, 0, pre, "expanded_applied", "expanded_post_norm", bn);
        b->has_proj = inc != ex; /* resnet.cu:780 */
        if (b->has_proj) {
            snprintf(bn, sizeof bn, "%02d/projected", i);
            unit_init(n, &b->proj, H, stride == 2 ? 3 : 1, inc, ex, stride, 0, pre, "transformed_residual",
                      "post_projection_norm_vals", bn);
        }
        const size_t osz = (size_t)batch * b->Hout * b->Hout * ex;
        char nm[128];
        snprintf(nm, sizeof nm, "%scombined_output", pre); b->output = (float *)reg(n, nm, osz, batch, b->Hout, b->Hout, ex);
        snprintf(nm, sizeof nm, "d:%scombined_output", pre); b->d_output = (float *)reg(n, nm, osz, batch, b->Hout, b->Hout, ex);
        snprintf(nm, sizeof nm, "%soutput_activated", pre); b->output_activated = (float *)reg(n, nm, osz, batch, b->Hout, b->Hout, ex);
        snprintf(nm, sizeof nm, "d:%soutput_activated", pre); b->d_output_activated = (float *)reg(n, nm, osz, batch, b->Hout, b->Hout, ex);
        if (flags[i] == 1) H /= 2;
        inc = ex;
    }
    add_loc(n, ex * output); /* fully_connected [expanded_depth][output], resnet.cu:933-943 */
    n->avg = (float *)reg(n, "final_avg_pool", (size_t)batch * final_depth, 0, 0, 0, 0);
    n->d_avg = (float *)reg(n, "d:final_avg_pool", (size_t)batch * final_depth, 0, 0, 0, 0);
    n->fc_out = (float *)reg(n, "fc_output", (size_t)batch * output, 0, 0, 0, 0);
    n->d_fc_out = (float *)reg(n, "d:fc_output", (size_t)batch * output, 0, 0, 0, 0);
    n->pred = (float *)reg(n, "softmax", (size_t)batch * output, 0, 0, 0, 0);
    return n;
}
void orc_net_destroy(OrcNet *n) {
    if (!n) return;
    for (int i = 0; i < n->nt; i++) { free(n->tname[i]); free(n->tptr[i]); }
    for (int i = 0; i < n->n_loc; i++) { free(n->p[i]); free(n->g[i]); free(n->m[i]); free(n->v[i]); }
    free(n->p); free(n->g); free(n->m); free(n->v); free(n->sizes); free(n->blocks); free(n->red_flags); free(n);
}
int orc_net_n_locations(const OrcNet *n) { return n->n_loc; }
int orc_net_location_size(const OrcNet *n, int i) { return n->sizes[i]; }
float *orc_net_param(OrcNet *n, int i) { return n->p[i]; }
float *orc_net_grad(OrcNet *n, int i) { return n->g[i]; }
float *orc_net_mean(OrcNet *n, int i) { return n->m[i]; }
float *orc_net_var(OrcNet *n, int i) { return n->v[i]; }
void orc_net_set_hyper(OrcNet *n, float lr, float wd, float b1, float b2, float eps) {
    n->lr = lr; n->wd = wd; n->b1 = b1; n->b2 = b2; n->eps = eps;
}
void orc_net_set_batch(OrcNet *n, const float *im, const int *lab) {
    memcpy(n->images, im, sizeof(float) * (size_t)n->N * n->input * n->input * 3);
    memcpy(n->labels, lab, sizeof(int) * n->N);
}
int orc_net_n_tensors(const OrcNet *n) { return n->nt; }
const char *orc_net_tensor_name(const OrcNet *n, int i) { return n->tname[i]; }
size_t orc_net_tensor_size(const OrcNet *n, int i) { return n->tsize[i]; }
void *orc_net_tensor_ptr(OrcNet *n, int i) { return n->tptr[i]; }
int orc_net_find_tensor(const OrcNet *n, const char *name) {
    for (int i = 0; i < n->nt; i++) if (!strcmp(n->tname[i], name)) return i;
    return -1;
}
void orc_net_tensor_shape(const OrcNet *n, int i, int s[4]) { for (int j = 0; j < 4; j++) s[j] = n->tshape[i][j]; }

static void unit_fwd(OrcNet *n, Unit *u, const float *in) {
    orc_conv_fwd(in, n->p[u->iw], u->H, u->k, u->C, u->K, u->stride, n->N, u->conv_out);
    orc_bn_fwd(u->conv_out, n->p[u->ig], n->p[u->ib], u->H / u->stride, u->K, n->N, n->eps, u->means, u->vars, u->xhat,
               u->normalized, u->activated, u->relu);
}
/* BN' then conv' (dgrad into dx with to_add, wgrad) -- prepareAndDoActivationAndBatchNormDeriv +
 * prepreAndDoConvolutionDeriv (resnet.cu:1399-1480) */
static void unit_bwd(OrcNet *n, Unit *u, const float *in, float *dx, int to_add, int want_dx) {
    orc_bn_bwd(u->conv_out, n->p[u->ig], u->H / u->stride, u->K, n->N, n->eps, u->means, u->vars, u->xhat, u->activated,
               u->d_activated, u->d_xhat, n->g[u->ig], n->g[u->ib], u->d_conv_out, u->relu);
    if (want_dx) orc_conv_dgrad(n->p[u->iw], u->d_conv_out, u->H, u->k, u->C, u->K, u->stride, n->N, to_add, dx);
    orc_conv_wgrad(in, u->d_conv_out, u->H, u->k, u->C, u->K, u->stride, n->N, n->g[u->iw]);
}

/* forward_pass, resnet.cu:1526-1775 */
void orc_net_forward(OrcNet *n) {
    unit_fwd(n, &n->stem, n->images);
    orc_maxpool_fwd(n->stem.activated, n->mpd, n->mps, n->N, n->input / n->ics, n->icf, n->max_inds, n->pool_out);
    const float *bin = n->pool_out;
    for (int i = 0; i < n->nb; i++) {
        Block *b = &n->blocks[i];
        unit_fwd(n, &b->red, bin);
        unit_fwd(n, &b->spa, b->red.activated);
        unit_fwd(n, &b->exp, b->spa.activated);
        const float *res = bin;
        if (b->has_proj) { unit_fwd(n, &b->proj, bin); res = b->proj.activated; }
        const int sz = n->N * b->Hout * b->Hout * b->Cout;
        orc_add(sz, b->exp.activated, res, b->output);
        orc_relu(sz, b->output, b->output_activated);
        bin = b->output_activated;
    }
    Block *last = &n->blocks[n->nb - 1];
    /* resnet.cu:1732 uses the last block's incoming_spatial_dim (a strided last block is not expressible) */
    orc_avgpool_fwd(last->output_activated, last->Hin, n->final_depth, n->N, n->avg);
    orc_matmul(n->avg, n->p[n->n_loc - 1], n->N, n->final_depth, n->output, n->fc_out);
    orc_softmax(n->fc_out, n->N, n->output, n->pred);
}
float orc_net_loss(OrcNet *n, int *n_wrong) { return orc_loss(n->pred, n->labels, n->N, n->output, n_wrong); }

/* backwards_pass, resnet.cu:1777-2248 with the spatial BN' call of resnet_cudnn.cu:2365-2366 */
void orc_net_backward(OrcNet *n) {
    const int N = n->N, L = n->output, D = n->final_depth;
    memcpy(n->d_fc_out, n->pred, sizeof(float) * (size_t)N * L);
    orc_ce_deriv(n->d_fc_out, n->labels, L, N);
    /* FC wgrad = transpose(pooled) x dlogits; FC dgrad = dlogits x transpose(W)  (:1823,:1830) */
    float *tmp = (float *)malloc(sizeof(float) * (size_t)(N > L ? N : L) * D);
    orc_transpose(n->avg, N, D, tmp);
    orc_matmul(tmp, n->d_fc_out, D, N, L, n->g[n->n_loc - 1]);
    orc_transpose(n->p[n->n_loc - 1], D, L, tmp);
    orc_matmul(n->d_fc_out, tmp, N, L, D, n->d_avg);
    free(tmp);
    Block *last = &n->blocks[n->nb - 1];
    orc_avgpool_bwd(n->d_avg, D, N, last->Hin, last->d_output_activated);
    for (int i = n->nb - 1; i >= 0; i--) {
        Block *b = &n->blocks[i];
        const float *bin = i == 0 ? n->pool_out : n->blocks[i - 1].output_activated;
        float *dbin = i == 0 ? n->d_pool_out : n->blocks[i - 1].d_output_activated;
        const int osz = N * b->Hout * b->Hout * b->Cout;
        orc_relu_deriv(osz, b->output, b->d_output_activated, b->d_output);
        if (b->has_proj) {
            memcpy(b->proj.d_activated, b->d_output, sizeof(float) * (size_t)osz);
            unit_bwd(n, &b->proj, bin, dbin, 0, 1);
        } else {
            /* setVal 0 then addVec (:2003-2004) == copy */
            memcpy(dbin, b->d_output, sizeof(float) * (size_t)N * b->Hin * b->Hin * b->Cin);
        }
        memcpy(b->exp.d_activated, b->d_output, sizeof(float) * (size_t)osz);
        unit_bwd(n, &b->exp, b->spa.activated, b->spa.d_activated, 0, 1);
        unit_bwd(n, &b->spa, b->red.activated, b->red.d_activated, 0, 1);
        unit_bwd(n, &b->red, bin, dbin, 1, 1);
    }
    orc_maxpool_bwd(n->max_inds, n->d_pool_out, n->input / n->ics, n->mps, n->icf, N, n->stem.d_activated);
    unit_bwd(n, &n->stem, n->images, NULL, 0, 0);
}

/* update_parameters, resnet.cu:2910-2987: decays advance BEFORE use; locations walked last->first;
 * then gradients (and the batch buffers) are zeroed. */
void orc_net_update(OrcNet *n) {
    const float cb1 = n->cur_b1 * n->b1, cb2 = n->cur_b2 * n->b2;
    for (int i = n->n_loc - 1; i >= 0; i--)
        orc_adam(n->sizes[i], n->p[i], n->g[i], n->m[i], n->v[i], n->lr, n->wd, n->b1, n->b2, cb1, cb2, n->eps);
    for (int i = 0; i < n->n_loc; i++) memset(n->g[i], 0, sizeof(float) * (size_t)n->sizes[i]);
    memset(n->images, 0, sizeof(float) * (size_t)n->N * n->input * n->input * 3);
    memset(n->labels, 0, sizeof(int) * n->N);
    n->cur_b1 = cb1; n->cur_b2 = cb2;
}
