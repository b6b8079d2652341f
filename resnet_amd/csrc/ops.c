/*
 * ops.c -- operator layer of the C-ABI (the prepareAndDo* wrappers of resnet.cu:1386-1509 as callable
 * entry points): each call runs one kernel family on the library's compute stream with a private
 * workspace and returns after the stream has drained.  Used by the parity tests to drive every HIP
 * kernel on its own; the trainer itself calls the mid_* launchers directly (no sync, shared workspace).
 */
#include <stdlib.h>
#include "mi_host.h"

/* MI_GUARD bytes of slack on both sides: the bf16 convolution operators read tap-shifted operands with 16-byte loads that may
 * start a few elements before / end a few elements past a tensor (masked lanes); tensors handed to mi_op_*_bf16 must come
 * from here */
void *mi_malloc(size_t bytes) {
    char *p = (char *)mid_malloc(bytes + 2 * MI_GUARD);
    return p ? p + MI_GUARD : NULL;
}
void mi_free(void *p) { if (p) mid_free((char *)p - MI_GUARD); }

static int finish(int rc) {
    mid_stream_sync(mi_global()->compute);
    if (rc) return rc;
    return mid_last_error()[0] ? -1 : 0;
}
static int ws_make(mid_workspace *ws, size_t wt, size_t part) {
    ws->wt_floats = wt; ws->part_floats = part;
    ws->pre_fwd = ws->pre_dgrad = NULL;
    ws->s2d = NULL; ws->s2d_bytes = 0; ws->s2d_valid = 0;
    ws->wt = wt ? (float *)mid_malloc(wt * sizeof(float)) : NULL;
    ws->part = part ? (float *)mid_malloc(part * sizeof(float)) : NULL;
    return (wt && !ws->wt) || (part && !ws->part);
}
static void ws_free(mid_workspace *ws) { mid_free(ws->wt); mid_free(ws->part); }

int mi_op_conv_fwd(const float *x, const float *w, float *y, int N, int C, int H, int K, int k, int stride) {
    mid_workspace ws;
    if (ws_make(&ws, mid_conv_ws_wt_floats(C, K, k), 0)) return -3;
    int rc = finish(mid_conv_fwd(mi_global()->compute, &ws, x, w, y, N, C, H, K, k, stride));
    ws_free(&ws);
    return rc;
}
int mi_op_conv_dgrad(const float *w, const float *dy, float *dx, int N, int C, int H, int K, int k, int stride, int to_add) {
    mid_workspace ws;
    if (ws_make(&ws, mid_conv_ws_wt_floats(C, K, k), 0)) return -3;
    int rc = finish(mid_conv_dgrad(mi_global()->compute, &ws, w, dy, dx, to_add ? dx : NULL, N, C, H, K, k, stride));
    ws_free(&ws);
    return rc;
}
int mi_op_conv_wgrad(const float *x, const float *dy, float *dw, int N, int C, int H, int K, int k, int stride) {
    mid_workspace ws;
    if (ws_make(&ws, 0, mid_conv_ws_part_floats(N, C, H, K, k, stride))) return -3;
    int rc = finish(mid_conv_wgrad(mi_global()->compute, &ws, x, dy, dw, N, C, H, K, k, stride));
    ws_free(&ws);
    return rc;
}
int mi_op_bn_fwd(const float *x, const float *gamma, const float *beta, float *means, float *vars, float *y, int N, int C,
                 int H, float eps, int relu) {
    float *ws = (float *)mid_malloc(mid_bn_ws_floats(C) * sizeof(float));
    int rc = finish(mid_bn_fwd(mi_global()->compute, ws, x, gamma, beta, NULL, means, vars, y, NULL, NULL, N, C, H * H, eps, relu));
    mid_free(ws);
    return rc;
}
int mi_op_bn_fwd_add_relu(const float *x, const float *gamma, const float *beta, const float *residual, float *means,
                          float *vars, float *y, int N, int C, int H, float eps) {
    float *ws = (float *)mid_malloc(mid_bn_ws_floats(C) * sizeof(float));
    int rc = finish(mid_bn_fwd(mi_global()->compute, ws, x, gamma, beta, residual, means, vars, y, NULL, NULL, N, C, H * H, eps, 0));
    mid_free(ws);
    return rc;
}
int mi_op_bn_bwd(const float *x, const float *gamma, const float *beta, const float *means, const float *vars,
                 const float *dy, const float *mask_src, float *dx, float *dgamma, float *dbeta, int N, int C, int H,
                 float eps, int mask_mode) {
    float *ws = (float *)mid_malloc(mid_bn_ws_floats(C) * sizeof(float));
    int rc = finish(mid_bn_bwd(mi_global()->compute, ws, x, gamma, beta, means, vars, dy, mask_src, dx, dgamma, dbeta, N, C, H * H, eps, mask_mode));
    mid_free(ws);
    return rc;
}
int mi_op_bn_bwd_gate(const float *x, const float *gamma, const float *beta, const float *means, const float *vars,
                      const float *dy, const float *mask_src, float *gated_out, float *dx, float *dgamma, float *dbeta, int N,
                      int C, int H, float eps) {
    float *ws = (float *)mid_malloc(mid_bn_ws_floats(C) * sizeof(float));
    int rc = finish(mid_bn_bwd_gate(mi_global()->compute, ws, x, gamma, beta, means, vars, dy, mask_src, gated_out, dx, dgamma, dbeta, N, C, H * H, eps));
    mid_free(ws);
    return rc;
}
int mi_op_maxpool_fwd(const float *x, float *y, int *max_inds, int N, int C, int H, int k, int stride) {
    return finish(mid_maxpool_fwd(mi_global()->compute, x, y, max_inds, N, C, H, k, stride));
}
int mi_op_maxpool_bwd(const int *max_inds, const float *dy, float *dx, int N, int C, int H, int k, int stride) {
    return finish(mid_maxpool_bwd(mi_global()->compute, max_inds, dy, dx, N, C, H, k, stride));
}
int mi_op_avgpool_fwd(const float *x, float *y, int N, int C, int H) { return finish(mid_avgpool_fwd(mi_global()->compute, x, y, N, C, H * H)); }
int mi_op_avgpool_bwd(const float *dy, float *dx, int N, int C, int H) { return finish(mid_avgpool_bwd(mi_global()->compute, dy, dx, N, C, H * H)); }
int mi_op_relu_deriv(const float *x, const float *up, float *out, size_t n) { return finish(mid_relu_deriv(mi_global()->compute, x, up, out, n)); }
int mi_op_matmul(const float *A, const float *B, float *out, int m, int k, int n) { return finish(mid_gemm_nn(mi_global()->compute, A, B, out, m, k, n)); }
int mi_op_matmul_lt(const float *A_kxm, const float *B, float *out, int m, int k, int n) { return finish(mid_gemm_tn(mi_global()->compute, A_kxm, B, out, m, k, n)); }
int mi_op_matmul_rt(const float *A, const float *B_nxk, float *out, int m, int k, int n) { return finish(mid_gemm_nt(mi_global()->compute, A, B_nxk, out, m, k, n)); }
int mi_op_softmax(const float *x, float *out, int N, int L) { return finish(mid_softmax(mi_global()->compute, x, out, N, L)); }
int mi_op_ce_deriv(const float *pred, const int *labels, float *d, int N, int L) { return finish(mid_ce_deriv(mi_global()->compute, pred, labels, d, N, L)); }
int mi_op_adam(float *p, const float *g, float *m, float *v, size_t n, float lr, float wd, float b1, float b2, float cur_b1,
               float cur_b2, float eps, int *nan_flag_dev) {
    return finish(mid_adam(mi_global()->compute, p, (float *)g, m, v, n, lr, wd, b1, b2, cur_b1, cur_b2, eps, nan_flag_dev, 0, NULL, 0, 0));
}
int mi_op_nhwc_to_nchw(const float *in, float *out, int N, int H, int W, int C) { return finish(mid_nhwc_to_nchw(mi_global()->compute, in, out, N, H, W, C)); }
int mi_op_fill_uniform(float *out, size_t n, uint64_t seed, float lo, float hi) { return finish(mid_fill_uniform(mi_global()->compute, out, n, seed, 0, lo, hi)); }
int mi_debug_poison_lds(void) { return finish(mid_lds_poison(mi_global()->compute)); }
int mi_debug_conv_plan(int op, int N, int C, int H, int K, int k, int stride, int out[9]) { return mid_igemm_plan(op, N, C, H, K, k, stride, out); }

/* ---- typed operators: activation tensors as bf16 (MI_DTYPE_BF16), arithmetic in fp32 ---- */
int mi_op_convert(const void *in, int in_dt, void *out, int out_dt, size_t n) {
    if (in_dt == MID_F32 && out_dt == MID_BF16) return finish(mid_f32_to_bf16(mi_global()->compute, (const float *)in, out, n));
    if (in_dt == MID_BF16 && out_dt == MID_F32) return finish(mid_bf16_to_f32(mi_global()->compute, in, (float *)out, n));
    return -2;
}
int mi_op_conv_fwd_bf16(const void *x, const float *w, void *y, int N, int C, int H, int K, int k, int stride) {
    mid_workspace ws;
    if (ws_make(&ws, (size_t)k * k * C * K, 0)) return -3;
    void *par = NULL;
    if (stride == 2) { ws.s2d_bytes = (size_t)N * C * H * H * 2; par = mi_malloc(ws.s2d_bytes); ws.s2d = par; }
    int rc = finish(mid_conv_fwd_bf16(mi_global()->compute, &ws, x, w, y, N, C, H, K, k, stride, NULL));
    mi_free(par);
    ws_free(&ws);
    return rc;
}
int mi_op_conv_dgrad_bf16(const float *w, const void *dy, void *dx, int N, int C, int H, int K, int k, int stride, int to_add) {
    mid_workspace ws;
    if (ws_make(&ws, (size_t)k * k * C * K, 0)) return -3;
    int rc = finish(mid_conv_dgrad_bf16(mi_global()->compute, &ws, w, dy, dx, to_add ? dx : NULL, N, C, H, K, k, stride));
    ws_free(&ws);
    return rc;
}
int mi_op_conv_wgrad_bf16(const void *x, const void *dy, float *dw, int N, int C, int H, int K, int k, int stride) {
    mid_workspace ws;
    if (!mid_bf16_supported(2, N, C, H, K, k, stride)) return -2;
    if (ws_make(&ws, 0, mid_bf16_part_floats(N, C, H, K, k, stride))) return -3;
    void *par = NULL;
    if (stride == 2) { ws.s2d_bytes = (size_t)N * C * H * H * 2; par = mi_malloc(ws.s2d_bytes); ws.s2d = par; }
    int rc = finish(mid_conv_wgrad_bf16(mi_global()->compute, &ws, x, dy, dw, N, C, H, K, k, stride));
    mi_free(par);
    ws_free(&ws);
    return rc;
}
/* dgrad of one convolution followed by the backward of the batch norm (+ReLU) in front of it, the way backwards_pass chains them in
 * bf16 storage (prepreAndDoConvolutionDeriv + activationAndBatchNormDeriv, resnet.cu:1399-1429, 1455-1480): where the launch allows, the
 * dgrad gates its output by mask > 0 and does the BN' reduction pass in its epilogue.  All image tensors bf16.
 * Returns < 0 on error, else the number of partial rows the dgrad left (0 = the separate reduction pass ran). */
int mi_op_conv_dgrad_bn_bwd_bf16(const float *w, const void *dy, const void *addend, void *gated, int N, int C, int H, int K, int k, int stride,
                                 const void *bn_x, const void *mask, const float *gamma, const float *beta, const float *means,
                                 const float *vars, float eps, void *bn_dx, float *dgamma, float *dbeta) {
    mid_workspace ws;
    if (ws_make(&ws, (size_t)k * k * C * K, 0)) return -3;
    mid_bn_bwd_parts fz = {bn_x, mask, means, NULL, mid_bn_parts_floats(N, C, H), 0};
    fz.buf = (float *)mid_malloc(fz.floats * sizeof(float));
    float *bws = (float *)mid_malloc(mid_bn_ws_floats(C) * sizeof(float));
    if (addend) mid_memcpy_d2d(gated, addend, (size_t)N * C * H * H * 2, mi_global()->compute);
    int rc = mid_conv_dgrad_bn_bf16(mi_global()->compute, &ws, w, dy, gated, addend ? gated : NULL, N, C, H, K, k, stride, &fz);
    if (!rc) {
        if (fz.nparts > 0)
            rc = mid_bn_bwd_parts_t(mi_global()->compute, bws, &fz, bn_x, MID_BF16, gamma, beta, means, vars, gated, MID_BF16, bn_dx, dgamma, dbeta, N, C, H * H, eps);
        else { /* the unfused chain: BN' gates by the mask itself (mode 3 writes the gated gradient where the fused form leaves it) */
            void *tmp = mi_malloc((size_t)N * C * H * H * 2);
            rc = mid_bn_bwd_t(mi_global()->compute, bws, bn_x, MID_BF16, gamma, beta, means, vars, gated, mask, tmp, MID_BF16, bn_dx, dgamma, dbeta, N, C, H * H, eps, 3);
            if (!rc) mid_memcpy_d2d(gated, tmp, (size_t)N * C * H * H * 2, mi_global()->compute);
            rc = finish(rc);
            mi_free(tmp);
        }
    }
    rc = finish(rc);
    mid_free(bws);
    mid_free(fz.buf);
    ws_free(&ws);
    return rc < 0 ? rc : fz.nparts;
}
/* the same chain in fp32 storage (backwards_pass, fp32 trainer): the stride-1 layers on the implicit-GEMM route do the reduction pass of
 * the batch-norm backward in the dgrad's epilogue */
int mi_op_conv_dgrad_bn_bwd_f32(const float *w, const float *dy, const float *addend, float *gated, int N, int C, int H, int K, int k, int stride,
                                const float *bn_x, const float *mask, const float *gamma, const float *beta, const float *means,
                                const float *vars, float eps, float *bn_dx, float *dgamma, float *dbeta) {
    mid_workspace ws;
    if (ws_make(&ws, mid_conv_ws_wt_floats(C, K, k), 0)) return -3;
    mid_bn_bwd_parts fz = {bn_x, mask, means, NULL, mid_bn_parts_floats(N, C, H), 0};
    fz.buf = (float *)mid_malloc(fz.floats * sizeof(float));
    float *bws = (float *)mid_malloc(mid_bn_ws_floats(C) * sizeof(float));
    const size_t bytes = (size_t)N * C * H * H * 4;
    if (addend) mid_memcpy_d2d(gated, addend, bytes, mi_global()->compute);
    int rc = mid_conv_dgrad_bn_f32(mi_global()->compute, &ws, w, dy, gated, addend ? gated : NULL, N, C, H, K, k, stride, &fz);
    if (!rc) {
        if (fz.nparts > 0)
            rc = mid_bn_bwd_parts_t(mi_global()->compute, bws, &fz, bn_x, MID_F32, gamma, beta, means, vars, gated, MID_F32, bn_dx, dgamma, dbeta, N, C, H * H, eps);
        else { /* the unfused chain: BN' gates by the mask itself (mode 3 writes the gated gradient where the fused form leaves it) */
            float *tmp = (float *)mid_malloc(bytes);
            rc = mid_bn_bwd_t(mi_global()->compute, bws, bn_x, MID_F32, gamma, beta, means, vars, gated, mask, tmp, MID_F32, bn_dx, dgamma, dbeta, N, C, H * H, eps, 3);
            if (!rc) mid_memcpy_d2d(gated, tmp, bytes, mi_global()->compute);
            rc = finish(rc);
            mid_free(tmp);
        }
    }
    rc = finish(rc);
    mid_free(bws);
    mid_free(fz.buf);
    ws_free(&ws);
    return rc < 0 ? rc : fz.nparts;
}
/* the bf16-mode stem (7x7 stride 2, 3 -> 64): x, y, dy fp32 tensors; image and weights rounded to bf16 inside */
int mi_op_stem_fwd_bf16(const float *x, const float *w, float *y, int N, int H) {
    if (!mid_stem_bf16_supported(3, H, 64, 7, 2)) return -2;
    const size_t xb = mid_stem_bf16_xp_bytes(N, H), sf = mid_stem_bf16_part_floats(N, H);
    void *xp = mi_malloc(xb);
    float *sc = (float *)mid_malloc(sf * sizeof(float));
    int rc = (!xp || !sc) ? -3 : finish(mid_stem_fwd_bf16(mi_global()->compute, x, w, y, MID_F32, xp, xb, sc, sf, N, H, NULL));
    mid_free(sc);
    mi_free(xp);
    return rc;
}
int mi_op_stem_wgrad_bf16(const float *x, const float *w, const float *dy, float *dw, int N, int H) {
    if (!mid_stem_bf16_supported(3, H, 64, 7, 2)) return -2;
    const size_t xb = mid_stem_bf16_xp_bytes(N, H), sf = mid_stem_bf16_part_floats(N, H);
    void *xp = mi_malloc(xb);
    float *sc = (float *)mid_malloc(sf * sizeof(float));
    float *y = (float *)mid_malloc((size_t)N * 64 * (H / 2) * (H / 2) * sizeof(float));
    int rc = (!xp || !sc || !y) ? -3 : mid_stem_fwd_bf16(mi_global()->compute, x, w, y, MID_F32, xp, xb, sc, sf, N, H, NULL); /* leaves the padded planes in xp */
    if (!rc) rc = mid_stem_wgrad_bf16(mi_global()->compute, xp, dy, MID_F32, dw, sc, sf, N, H);
    rc = finish(rc);
    mid_free(y);
    mid_free(sc);
    mi_free(xp);
    return rc;
}
/* ... and in exact fp32 (the stem of the fp32 storage mode where RESNET_MI_IGEMM allows the matrix cores) */
int mi_op_stem_fwd_f32(const float *x, const float *w, float *y, int N, int H) {
    if (!mid_stem_bf16_supported(3, H, 64, 7, 2)) return -2;
    const size_t xb = mid_stem_f32_xp_bytes(N, H), sf = mid_stem_bf16_part_floats(N, H);
    void *xp = mi_malloc(xb);
    float *sc = (float *)mid_malloc(sf * sizeof(float));
    int rc = (!xp || !sc) ? -3 : finish(mid_stem_fwd_f32(mi_global()->compute, x, w, y, xp, xb, sc, sf, N, H, NULL));
    mid_free(sc);
    mi_free(xp);
    return rc;
}
int mi_op_stem_wgrad_f32(const float *x, const float *w, const float *dy, float *dw, int N, int H) {
    if (!mid_stem_bf16_supported(3, H, 64, 7, 2)) return -2;
    const size_t xb = mid_stem_f32_xp_bytes(N, H), sf = mid_stem_bf16_part_floats(N, H);
    void *xp = mi_malloc(xb);
    float *sc = (float *)mid_malloc(sf * sizeof(float));
    float *y = (float *)mid_malloc((size_t)N * 64 * (H / 2) * (H / 2) * sizeof(float));
    int rc = (!xp || !sc || !y) ? -3 : mid_stem_fwd_f32(mi_global()->compute, x, w, y, xp, xb, sc, sf, N, H, NULL); /* leaves the padded planes in xp */
    if (!rc) rc = mid_stem_wgrad_f32(mi_global()->compute, xp, dy, dw, sc, sf, N, H);
    rc = finish(rc);
    mid_free(y);
    mid_free(sc);
    mi_free(xp);
    return rc;
}
/* A convolution and the batch norm behind it the way forward_pass runs the pair (resnet.cu:1386-1396 + 1431-1453): the
 * statistics come out of the convolution's own epilogue (fp32 accumulators) where the layer runs on the implicit GEMM, and
 * from a pass over conv_out where it does not.  dt = storage type of x, conv_out and y. */
int mi_op_conv_bn_fwd_t(const void *x, const float *w, void *conv_out, int dt, const float *gamma, const float *beta, float *means,
                        float *vars, void *y, int N, int C, int H, int K, int k, int stride, float eps, int relu) {
    mid_workspace ws;
    const int Ho = H / stride;
    if (ws_make(&ws, dt == MID_BF16 ? (size_t)k * k * C * K : mid_conv_ws_wt_floats(C, K, k), 0)) return -3;
    mid_bn_parts parts = {NULL, mid_bn_parts_floats(N, K, Ho), 0};
    parts.buf = (float *)mid_malloc(parts.floats * sizeof(float));
    float *bws = (float *)mid_malloc(mid_bn_ws_floats(K) * sizeof(float));
    void *par = NULL;
    int rc;
    if (dt == MID_BF16) {
        if (stride == 2) { ws.s2d_bytes = (size_t)N * C * H * H * 2; par = mi_malloc(ws.s2d_bytes); ws.s2d = par; }
        rc = mid_conv_fwd_bf16(mi_global()->compute, &ws, x, w, conv_out, N, C, H, K, k, stride, &parts);
    } else rc = mid_conv_fwd_stats(mi_global()->compute, &ws, (const float *)x, w, (float *)conv_out, N, C, H, K, k, stride, &parts);
    if (!rc) rc = mid_bn_fwd_t(mi_global()->compute, bws, &parts, conv_out, dt, gamma, beta, NULL, means, vars, y, dt, NULL, NULL, N, K, Ho * Ho, eps, relu);
    rc = finish(rc);
    mi_free(par);
    mid_free(bws);
    mid_free(parts.buf);
    ws_free(&ws);
    return rc < 0 ? rc : parts.nparts; /* > 0: the statistics were fused (number of partial rows), 0: separate pass */
}
int mi_op_bn_fwd_t(const void *x, int x_dt, const float *gamma, const float *beta, const void *residual, float *means, float *vars,
                   void *y, int a_dt, int N, int C, int H, float eps, int relu) {
    float *ws = (float *)mid_malloc(mid_bn_ws_floats(C) * sizeof(float));
    int rc = finish(mid_bn_fwd_t(mi_global()->compute, ws, NULL, x, x_dt, gamma, beta, residual, means, vars, y, a_dt, NULL, NULL, N, C, H * H, eps, relu));
    mid_free(ws);
    return rc;
}
/* BN (+ residual) + ReLU of a bf16 tensor with the output written twice, as forward_pass does in front of a 3x3: y (bf16 NCHW) and ycl =
 * the same values channel-last -- par = 0: one zero-padded plane [N][H+2][H+2][C] (stride-1 3x3); par = 1: the four parity planes
 * [N][2x2][H/2+1][H/2+1][C] of a stride-2 3x3 (the caller zeroes ycl once: only the interior is written) */
int mi_op_bn_fwd_cl_bf16(const void *x, const float *gamma, const float *beta, const void *residual, float *means, float *vars, void *y, void *ycl,
                         int N, int C, int H, float eps, int par) {
    float *ws = (float *)mid_malloc(mid_bn_ws_floats(C) * sizeof(float));
    mid_bn_set_cl_out(ycl, par ? -H : H);
    int rc = finish(mid_bn_fwd_t(mi_global()->compute, ws, NULL, x, MID_BF16, gamma, beta, residual, means, vars, y, MID_BF16, NULL, NULL, N, C, H * H, eps,
                                 residual ? 0 : 1));
    mid_free(ws);
    return rc;
}
int mi_op_bn_bwd_t(const void *x, int x_dt, const float *gamma, const float *beta, const float *means, const float *vars,
                   const void *dy, const void *mask_src, void *gated_out, int a_dt, void *dx, float *dgamma, float *dbeta, int N, int C,
                   int H, float eps, int mask_mode) {
    float *ws = (float *)mid_malloc(mid_bn_ws_floats(C) * sizeof(float));
    int rc = finish(mid_bn_bwd_t(mi_global()->compute, ws, x, x_dt, gamma, beta, means, vars, dy, mask_src, gated_out, a_dt, dx, dgamma, dbeta, N, C, H * H, eps, mask_mode));
    mid_free(ws);
    return rc;
}
int mi_op_bn_apply_t(const void *x, int x_dt, const float *gamma, const float *beta, const void *residual, const float *means,
                     const float *vars, void *y, int a_dt, int N, int C, int H, float eps, int relu) {
    return finish(mid_bn_apply_t(mi_global()->compute, x, x_dt, gamma, beta, residual, means, vars, y, a_dt, N, C, H * H, eps, relu));
}
int mi_op_maxpool_fwd_t(const void *x, void *y, int dt, int *max_inds, int N, int C, int H, int k, int stride) {
    return finish(mid_maxpool_fwd_t(mi_global()->compute, x, y, dt, max_inds, N, C, H, k, stride));
}
int mi_op_maxpool_bwd_t(const int *max_inds, const void *dy, void *dx, int dt, int N, int C, int H, int k, int stride) {
    return finish(mid_maxpool_bwd_t(mi_global()->compute, max_inds, dy, dx, dt, N, C, H, k, stride));
}
int mi_op_avgpool_fwd_t(const void *x, int dt, float *y, int N, int C, int H) { return finish(mid_avgpool_fwd_t(mi_global()->compute, x, dt, y, N, C, H * H)); }
int mi_op_avgpool_bwd_t(const float *dy, void *dx, int dt, int N, int C, int H) { return finish(mid_avgpool_bwd_t(mi_global()->compute, dy, dx, dt, N, C, H * H)); }
int mi_bf16_conv_supported(int op, int N, int C, int H, int K, int k, int stride) { return mid_bf16_supported(op, N, C, H, K, k, stride); }
/* 1: the 1x1 weight gradient of this shape runs on the LDS-DMA kernel (pw_wgrad_kernel) inside mi_op_conv_wgrad_bf16 / the bf16 trainer */
int mi_bf16_pw_wgrad_supported(int N, int C, int H, int K) { return mid_pw_wgrad_supported(N, C, H, K); }
void mi_clear_error(void) { mid_clear_error(); }

/* the device-side merge of cross-replica batch norm on R replicas held by one process (see mid_bn_debug_merge): lets a one-GPU
 * box check the merge against whole-batch statistics with two DIFFERENT replicas.  All pointers device, [R][C]; sums_out [R][2C]
 * or NULL; (means, vars) or (dgamma, dbeta) may be NULL to run one half only. */
int mi_debug_bn_merge(int R, int C, float *means, float *vars, float *dgamma, float *dbeta, float *sums_out) {
    float *tmp = (float *)mid_malloc((size_t)R * 2 * C * sizeof(float));
    if (!tmp) return -3;
    int rc = finish(mid_bn_debug_merge(mi_global()->compute, R, C, means, vars, dgamma, dbeta, sums_out, tmp));
    mid_free(tmp);
    return rc;
}

/* the 3x3 convolutions of the bf16 path on channel-last zero-padded operands (kernels_cl_bf16.hip): operand re-laid, weights re-laid,
 * then the LDS-DMA kernel.  Same tensors and semantics as mi_op_conv_fwd_bf16 / mi_op_conv_dgrad_bf16 with k = 3.  -2: shape not covered */
int mi_op_conv_fwd_bf16_cl(const void *x, const float *w, void *y, int N, int C, int H, int K, int stride) {
    if (!mid_cl_supported(0, N, C, H, K, stride)) return -2;
    mid_stream st = mi_global()->compute;
    const size_t xb = mid_cl_operand_bytes(0, N, C, H, K, stride);
    void *xp = mid_malloc(xb), *at = mid_malloc((size_t)9 * C * K * 2);
    if (!xp || !at) { mid_free(xp); mid_free(at); return -3; }
    mid_memset(xp, 0, xb, st);
    int rc = mid_bf16_prelayout_fwd(st, w, at, K, C, 3);
    if (!rc) rc = mid_cl_relayout(st, x, xp, N, C, H, stride == 2);
    if (!rc) rc = mid_cl_fwd(st, xp, at, y, N, C, H, K, stride, NULL);
    rc = finish(rc);
    mid_free(xp); mid_free(at);
    return rc;
}
/* 1x1 forward with the input re-laid dense channel-last (one tap of the channel-last kernel: both operands reduction-contiguous) */
int mi_op_conv1x1_fwd_bf16_cl(const void *x, const float *w, void *y, int N, int C, int H, int K) {
    if (!mid_cl_pw_supported(N, C, H, K)) return -2;
    mid_stream st = mi_global()->compute;
    const size_t xb = (size_t)N * H * H * C * 2 + 4096;
    void *xp = mid_malloc(xb), *at = mid_malloc((size_t)C * K * 2);
    if (!xp || !at) { mid_free(xp); mid_free(at); return -3; }
    int rc = mid_bf16_prelayout_fwd(st, w, at, K, C, 1);
    if (!rc) rc = mid_cl_relayout_dense(st, x, xp, N, C, H);
    if (!rc) rc = mid_cl_pw_fwd(st, xp, at, y, N, C, H, K, NULL);
    rc = finish(rc);
    mid_free(xp); mid_free(at);
    return rc;
}
int mi_op_conv_dgrad_bf16_cl(const float *w, const void *dy, void *dx, int N, int C, int H, int K, int stride, int to_add) {
    if (stride == 2) {
        if (to_add || !mid_cl_dgrad2_supported(N, C, H, K)) return -2;
        mid_stream st2 = mi_global()->compute;
        const size_t yb2 = mid_cl_dgrad2_operand_bytes(N, K, H / 2);
        void *dyp2 = mid_malloc(yb2), *at2 = mid_malloc((size_t)9 * C * K * 2);
        if (!dyp2 || !at2) { mid_free(dyp2); mid_free(at2); return -3; }
        mid_memset(dyp2, 0, yb2, st2);
        int rc2 = mid_bf16_prelayout_dgrad(st2, w, at2, K, C, 3);
        if (!rc2) rc2 = mid_cl_relayout_end(st2, dy, dyp2, N, K, H / 2);
        if (!rc2) rc2 = mid_cl_dgrad2(st2, dyp2, at2, dx, N, C, H, K);
        rc2 = finish(rc2);
        mid_free(dyp2); mid_free(at2);
        return rc2;
    }
    if (!mid_cl_supported(1, N, C, H, K, 1)) return -2;
    mid_stream st = mi_global()->compute;
    const size_t yb = mid_cl_operand_bytes(1, N, C, H, K, 1);
    void *dyp = mid_malloc(yb), *at = mid_malloc((size_t)9 * C * K * 2);
    if (!dyp || !at) { mid_free(dyp); mid_free(at); return -3; }
    mid_memset(dyp, 0, yb, st);
    int rc = mid_bf16_prelayout_dgrad(st, w, at, K, C, 3);
    if (!rc) rc = mid_cl_relayout(st, dy, dyp, N, K, H, 0);
    if (!rc) rc = mid_cl_dgrad(st, dyp, at, dx, to_add ? dx : NULL, N, C, H, K);
    rc = finish(rc);
    mid_free(dyp); mid_free(at);
    return rc;
}

/* 3x3 weight gradient from the channel-last planes of BOTH operands (cl_wgrad2_kernel): stride 2 = the input's parity planes and the dY
 * planes of the stride-2 dgrad; stride 1 = both with a halo of 1 */
int mi_op_conv_wgrad_bf16_cl2(const void *x, const void *dy, float *dw, int N, int C, int H, int K, int stride) {
    if (!mid_cl_wgrad2_supported(N, C, H, K, stride)) return -2;
    mid_stream st = mi_global()->compute;
    const int Ho = H / stride;
    const size_t xb = mid_cl_operand_bytes(0, N, C, H, K, stride), yb = stride == 2 ? mid_cl_dgrad2_operand_bytes(N, K, Ho) : mid_cl_operand_bytes(1, N, C, H, K, 1),
                 pf = mid_cl_wgrad2_part_floats(N, C, H, K, stride);
    void *xp = mid_malloc(xb), *dyp = mid_malloc(yb);
    float *part = (float *)mid_malloc(pf * sizeof(float));
    if (!xp || !dyp || !part) { mid_free(xp); mid_free(dyp); mid_free(part); return -3; }
    mid_memset(xp, 0, xb, st);
    mid_memset(dyp, 0, yb, st);
    int rc = mid_cl_relayout(st, x, xp, N, C, H, stride == 2);
    if (!rc) rc = stride == 2 ? mid_cl_relayout_end(st, dy, dyp, N, K, Ho) : mid_cl_relayout(st, dy, dyp, N, K, H, 0);
    if (!rc) rc = mid_cl_wgrad2(st, xp, dyp, dw, part, pf, N, C, H, K, stride);
    rc = finish(rc);
    mid_free(xp); mid_free(dyp); mid_free(part);
    return rc;
}
int mi_op_conv_wgrad_bf16_cl(const void *x, const void *dy, float *dw, int N, int C, int H, int K, int stride) {
    if (!mid_cl_wgrad_supported(N, C, H, K, stride)) return -2;
    mid_stream st = mi_global()->compute;
    const size_t xb = mid_cl_operand_bytes(0, N, C, H, K, stride), pf = mid_cl_wgrad_part_floats(N, C, H, K, stride);
    void *xp = mid_malloc(xb);
    float *part = (float *)mid_malloc(pf * sizeof(float));
    if (!xp || !part) { mid_free(xp); mid_free(part); return -3; }
    mid_memset(xp, 0, xb, st);
    int rc = mid_cl_relayout(st, x, xp, N, C, H, stride == 2);
    if (!rc) rc = mid_cl_wgrad(st, xp, dy, dw, part, pf, N, C, H, K, stride);
    rc = finish(rc);
    mid_free(xp); mid_free(part);
    return rc;
}
