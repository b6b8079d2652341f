/*
 * mi_device.h -- the thin header layer between the C host code (trainer.c, loader.c, ...) and the
 * HIP side (runtime.hip + kernels_*.hip).  Host .c files include only this; they never see HIP
 * headers.  It plays the role of the cuda_runtime.h include of resnet.cu:5-7 plus the
 * prepareAndDo* launch wrappers of resnet.cu:1386-1509.
 * All launches are asynchronous on the given stream; errors are recorded (mid_last_error).
 */
#ifndef MI_DEVICE_H
#define MI_DEVICE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void *mid_stream;
typedef void *mid_event;

/* storage type of an activation tensor (parameters, gradients and statistics are always fp32) */
enum { MID_F32 = 0, MID_BF16 = 1 };

/* ---- runtime ---- */
int mid_device_count(void);
int mid_set_device(int dev);
const char *mid_last_error(void);
void mid_clear_error(void);
void *mid_malloc(size_t bytes);
void mid_free(void *p);
void *mid_malloc_host(size_t bytes); /* pinned */
void mid_free_host(void *p);
void mid_memcpy_h2d(void *dst, const void *src, size_t bytes, mid_stream s);
void mid_memcpy_d2h(void *dst, const void *src, size_t bytes, mid_stream s);
void mid_memcpy_d2d(void *dst, const void *src, size_t bytes, mid_stream s);
void mid_memset(void *dst, int byte, size_t bytes, mid_stream s);
mid_stream mid_stream_create(void);
mid_stream mid_stream_create_low_priority(void);
void mid_stream_destroy(mid_stream s);
void mid_stream_sync(mid_stream s);
void mid_device_sync(void);
mid_event mid_event_create(void);
void mid_event_destroy(mid_event e);
void mid_event_record(mid_event e, mid_stream s);
void mid_stream_wait_event(mid_stream s, mid_event e);
float mid_event_elapsed_ms(mid_event a, mid_event b);
void mid_event_sync(mid_event e);

/* ---- optional per-kernel-family timing (HIP events around each launch, on the launch stream) ----
 * families: 0 direct conv fwd/dgrad, 1 direct conv wgrad, 2 MFMA GEMM (1x1 conv, FC), 3 batch norm, 4 other */
void mid_prof_enable(int on);
void mid_prof_reset(void);
/* resolves pending events (synchronises) and returns totals since the last reset */
void mid_prof_get(int family, long *launches, double *ms, double *flops, double *bytes);

/* ---- workspace a conv launch may use (transformed weights, split partials) ---- */
typedef struct {
    float *wt;        /* transformed-weight scratch */
    size_t wt_floats;
    float *part;      /* split-reduction partial sums */
    size_t part_floats;
    /* weights of the NEXT call already re-laid by mid_conv_prelayout_all (else NULL: the call re-lays them itself into wt) */
    const float *pre_fwd, *pre_dgrad;
    /* bf16 path, stride-2 layers: scratch for x re-laid as four parity planes per channel (>= 2 bytes per element of x, with the
     * guard bytes of every bf16 tensor on both sides); NULL = those layers gather element-wise */
    void *s2d;
    size_t s2d_bytes;
    int s2d_valid; /* s2d already holds the parity planes of this call's x (the forward pass of the same layer wrote them) */
} mid_workspace;

/* One launch that re-lays the weights of many convolutions for the implicit-GEMM kernel: fwd = [t][c][k], dgrad = [t][k][c]
 * (either may be NULL).  entries / tile_entry live in device memory; tile0 = first 32x32 (k, c) tile of the entry in the
 * launch's flat tile numbering, tile_entry[tile] = its entry. */
typedef struct {
    const float *w;
    float *fwd, *dgrad;
    int K, C, T, tile0;
} mid_wt_entry;
int mid_conv_prelayout_all(mid_stream s, const mid_wt_entry *entries_dev, const int *tile_entry_dev, int ntiles);
/* which pre-laid forms the implicit-GEMM route of this layer would use (0/1 each); both 0 = layer not on that route */
void mid_conv_prelayout_needs(int N, int C, int H, int K, int k, int stride, int *need_fwd, int *need_dgrad);

/* ---- convolution (NCHW activations, KCRS weights), square images/kernels, pad k/2, Ho = H/stride ---- */
/* Routed by shape and RESNET_MI_IGEMM (kernels_igemm.hip: mi_igemm_supported): MFMA implicit GEMM for the 3x3 / 1x1 shapes
 * that tile, plain MFMA GEMM for other 1x1, direct LDS-tiled VALU kernels for the 7x7 stem and the rest.
 * Returns 0 ok, <0 unsupported shape. */
int mid_conv_fwd(mid_stream s, mid_workspace *ws, const float *x, const float *w, float *y, int N, int C, int H, int K,
                 int k, int stride);
/* Batch-norm statistics fused into the producing convolution: when the layer runs on the implicit-GEMM kernel every
 * workgroup also writes per-channel (count, mean, M2) partials of its output tile into parts->buf (three planes of
 * [nparts][K]); parts->nparts comes back > 0.  0 = not fused: run the separate statistics pass (mid_bn_fwd). */
typedef struct {
    float *buf;
    size_t floats;
    int nparts;
} mid_bn_parts;
size_t mid_bn_parts_floats(int N, int K, int Ho);
int mid_conv_fwd_stats(mid_stream s, mid_workspace *ws, const float *x, const float *w, float *y, int N, int C, int H, int K,
                       int k, int stride, mid_bn_parts *parts);
/* dx = dgrad (+ addend when addend != NULL; addend may alias dx) */
int mid_conv_dgrad(mid_stream s, mid_workspace *ws, const float *w, const float *dy, float *dx, const float *addend,
                   int N, int C, int H, int K, int k, int stride);
int mid_conv_wgrad(mid_stream s, mid_workspace *ws, const float *x, const float *dy, float *dw, int N, int C, int H,
                   int K, int k, int stride);
/* sizes a workspace must have for the given layer (floats) */
size_t mid_conv_ws_wt_floats(int C, int K, int k);
size_t mid_conv_ws_part_floats(int N, int C, int H, int K, int k, int stride);

/* ---- bf16-activation path (kernels_igemm_bf16.hip): x / y / dy / dx are bf16 NCHW, weights fp32 KCRS (rounded to bf16 when
 * re-laid), weight gradients fp32.  ws->pre_fwd / pre_dgrad point at bf16 k-step tiles when mid_conv_prelayout_all_bf16 made
 * them; otherwise the call re-lays the weights itself into ws->wt. ---- */
int mid_bf16_supported(int op, int N, int C, int H, int K, int k, int stride); /* op 0 fwd, 1 dgrad, 2 wgrad */
size_t mid_bf16_part_floats(int N, int C, int H, int K, int k, int stride);
int mid_conv_prelayout_all_bf16(mid_stream s, const mid_wt_entry *entries_dev, const int *tile_entry_dev, int ntiles);
int mid_conv_fwd_bf16(mid_stream s, mid_workspace *ws, const void *x, const float *w, void *y, int N, int C, int H, int K, int k,
                      int stride, mid_bn_parts *parts);
/* The REDUCTION pass of a batch-norm backward fused into the dgrad that produces its dy (bf16 kernels, pixel-major epilogue):
 * the dgrad stores g = (mask > 0 ? dx : 0) instead of dx and leaves per-tile sums of g and g (x - mean) per channel in buf (two
 * planes [nparts][C]); nparts comes back > 0 when the launch could do it.  mid_bn_bwd_parts_t then finishes that batch norm
 * (merge, finalize, apply) without reading dy for the sums. */
typedef struct {
    const void *x;       /* the BN's input (the convolution output it normalised), bf16, the shape of the dgrad output */
    const void *mask;    /* the tensor whose sign gates: that BN's activated output, or the block output for the expansion BN */
    const float *means;
    float *buf;
    size_t floats;
    int nparts;
} mid_bn_bwd_parts;
int mid_conv_dgrad_bn_bf16(mid_stream s, mid_workspace *ws, const float *w, const void *dy, void *dx, const void *addend, int N, int C,
                           int H, int K, int k, int stride, mid_bn_bwd_parts *fz);
/* the same for fp32 storage (kernels_igemm.hip): stride-1 layers on the implicit-GEMM route; otherwise a plain dgrad, nparts = 0 */
int mid_conv_dgrad_bn_f32(mid_stream s, mid_workspace *ws, const float *w, const float *dy, float *dx, const float *addend, int N, int C,
                          int H, int K, int k, int stride, mid_bn_bwd_parts *fz);
int mid_bn_bwd_parts_t(mid_stream s, float *stats_ws, const mid_bn_bwd_parts *parts, const void *x, int x_dt, const float *gamma,
                       const float *beta, const float *means, const float *vars, const void *dy_gated, int a_dt, void *dx, float *dgamma,
                       float *dbeta, int N, int C, int P, float eps);
int mid_conv_dgrad_bf16(mid_stream s, mid_workspace *ws, const float *w, const void *dy, void *dx, const void *addend, int N, int C,
                        int H, int K, int k, int stride);
int mid_conv_wgrad_bf16(mid_stream s, mid_workspace *ws, const void *x, const void *dy, float *dw, int N, int C, int H, int K, int k,
                        int stride);
/* ---- 3x3 convolutions of the bf16 path on channel-last, zero-padded operands (kernels_cl_bf16.hip): both operands global -> LDS by
 * LDS-DMA, no transposes, no masks.  op 0 forward (stride 1 or 2), op 1 dgrad (stride 1) ---- */
int mid_bf16_prelayout_fwd(mid_stream s, const float *w, void *out, int K, int C, int k);
int mid_bf16_prelayout_dgrad(mid_stream s, const float *w, void *out, int K, int C, int k);
int mid_cl_supported(int op, int N, int C, int H, int K, int stride);
size_t mid_cl_operand_bytes(int op, int N, int C, int H, int K, int stride);
/* x (bf16 NCHW, C channels, H x H) -> padded channel-last; parity != 0: the four parity planes of a stride-2 forward.  The halo of xp
 * must be zero: zero the buffer once when it is made (the kernel writes the interior only) */
int mid_cl_relayout(mid_stream s, const void *x, void *xp, int N, int C, int H, int parity);
int mid_cl_fwd(mid_stream s, const void *xp, const void *a_tiles, void *y, int N, int C, int H, int K, int stride, mid_bn_parts *parts);
int mid_cl_dgrad(mid_stream s, const void *dyp, const void *a_tiles, void *dx, const void *addend, int N, int C, int H, int K);
/* weight gradient from dY (NCHW) and the forward's re-laid input: transposed LDS reads on the pixel-major operand */
int mid_cl_wgrad_supported(int N, int C, int H, int K, int stride);
size_t mid_cl_wgrad_part_floats(int N, int C, int H, int K, int stride);
void mid_bn_set_cl_out(void *ycl, int H); /* one-shot: the next mid_bn_fwd_t / mid_bn_apply_t also writes its (bf16; ReLU or + residual, ReLU) output channel-last: H > 0 one plane with a halo of 1, H < 0 the four parity planes of a stride-2 3x3 over |H| x |H| */
int mid_cl_pw_supported(int N, int C, int H, int K);
int mid_cl_relayout_dense(mid_stream s, const void *x, void *xp, int N, int C, int H);
int mid_cl_pw_fwd(mid_stream s, const void *xc, const void *a_tiles, void *y, int N, int C, int H, int K, mid_bn_parts *parts);
int mid_cl_wgrad2_supported(int N, int C, int H, int K, int stride);
size_t mid_cl_wgrad2_part_floats(int N, int C, int H, int K, int stride);
int mid_cl_wgrad2(mid_stream s, const void *xp, const void *dyp, float *dw, float *part, size_t part_floats, int N, int C, int H, int K, int stride);
int mid_pw_wgrad_supported(int N, int C, int H, int K);
size_t mid_pw_wgrad_part_floats(int N, int C, int H, int K);
int mid_pw_wgrad(mid_stream s, const void *x, const void *dy, float *dw, float *part, size_t part_floats, int N, int C, int H, int K);
int mid_cl_wgrad(mid_stream s, const void *xp, const void *dy, float *dw, float *part, size_t part_floats, int N, int C, int H, int K, int stride);
/* stride-2 dgrad: dY (K channels, H/2 x H/2) re-laid channel-last with one zero row / column at the far end, both column parities of dx in
 * one workgroup (dense stores) */
int mid_cl_dgrad2_supported(int N, int C, int H, int K);
size_t mid_cl_dgrad2_operand_bytes(int N, int K, int Ho);
int mid_cl_relayout_end(mid_stream s, const void *dy, void *dyp, int N, int K, int Ho);
int mid_cl_dgrad2(mid_stream s, const void *dyp, const void *a_tiles, void *dx, int N, int C, int H, int K);
/* the 7x7 stride-2 stem (3 -> 64 channels) on the bf16 matrix cores (kernels_stem_bf16.hip): image and weights rounded to
 * bf16, fp32 accumulation, fp32 output / output gradient.  xp = the image as zero-padded parity planes (written by the
 * forward, read again by the weight gradient); scratch = wave partials + re-laid weights (mid_stem_bf16_part_floats). */
int mid_igemm_mode(void); /* RESNET_MI_IGEMM (0 = the convolutions stay off the matrix cores) */
/* ... and in exact fp32 (v_mfma_f32_32x32x2_f32) for the fp32 storage mode; same shape rule, same scratch, fp32 planes */
size_t mid_stem_f32_xp_bytes(int N, int H);
int mid_stem_fwd_f32(mid_stream s, const float *x, const float *w, float *y, void *xp, size_t xp_bytes, float *scratch, size_t scratch_floats,
                     int N, int H, mid_bn_parts *parts);
int mid_stem_wgrad_f32(mid_stream s, const void *xp, const float *dy, float *dw, float *scratch, size_t scratch_floats, int N, int H);
int mid_stem_bf16_supported(int C, int H, int K, int k, int stride);
size_t mid_stem_bf16_xp_bytes(int N, int H);
size_t mid_stem_bf16_part_floats(int N, int H);
int mid_stem_fwd_bf16(mid_stream s, const float *x, const float *w, void *y, int y_dt, void *xp, size_t xp_bytes, float *scratch, size_t scratch_floats,
                      int N, int H, mid_bn_parts *parts); /* parts (optional): BN statistics partials of y, as mid_conv_fwd_stats leaves them */
int mid_stem_wgrad_bf16(mid_stream s, const void *xp, const void *dy, int dy_dt, float *dw, float *scratch, size_t scratch_floats, int N, int H);
int mid_f32_to_bf16(mid_stream s, const float *in, void *out, size_t n);
int mid_bf16_to_f32(mid_stream s, const void *in, float *out, size_t n);

/* host-only: route and grid the launch planners choose for a convolution (kernels_igemm.hip); op 0 fwd, 1 dgrad, 2 wgrad */
int mid_igemm_plan(int op, int N, int C, int H, int K, int k, int stride, int out[9]);

/* ---- plain GEMMs for the FC layer (row-major) ---- */
/* out[m x n] = A[m x k] B[k x n] */
int mid_gemm_nn(mid_stream s, const float *A, const float *B, float *out, int m, int k, int n);
/* out[m x n] = At^T B, At is [k x m] */
int mid_gemm_tn(mid_stream s, const float *At, const float *B, float *out, int m, int k, int n);
/* out[m x n] = A Bt^T, Bt is [n x k] */
int mid_gemm_nt(mid_stream s, const float *A, const float *Bt, float *out, int m, int k, int n);

/* ---- batch norm (training mode, biased variance, per-replica statistics) ---- */
/* stats_ws: >= mid_bn_ws_floats(C) floats of scratch */
size_t mid_bn_ws_floats(int C);
/* y = [relu](gamma * (x-mean)/sqrt(var+eps) + beta) [+ residual, relu]; writes means/vars.
 * residual != NULL => y = relu(bn(x) + residual).  xhat_out / norm_out optional (full-store). */
int mid_bn_fwd(mid_stream s, float *stats_ws, const float *x, const float *gamma, const float *beta,
               const float *residual, float *means, float *vars, float *y, float *xhat_out, float *norm_out, int N,
               int C, int P, float eps, int relu);
/* the same with the statistics taken from the partials a convolution left (parts->nparts > 0), else from x */
int mid_bn_fwd_parts(mid_stream s, float *stats_ws, const mid_bn_parts *parts, const float *x, const float *gamma,
                     const float *beta, const float *residual, float *means, float *vars, float *y, float *xhat_out,
                     float *norm_out, int N, int C, int P, float eps, int relu);
/* mask_mode 0 none; 1 recompute own ReLU mask from x; 2 gate dy by mask_src > 0 */
int mid_bn_bwd(mid_stream s, float *stats_ws, const float *x, const float *gamma, const float *beta,
               const float *means, const float *vars, const float *dy, const float *mask_src, float *dx,
               float *dgamma, float *dbeta, int N, int C, int P, float eps, int mask_mode);

/* mask_mode 2 that also writes gated_out = (mask_src > 0 ? dy : 0), the tensor doActivationDeriv (resnet.cu:1934) would
 * have produced in a pass of its own; gated_out may not alias dx */
int mid_bn_bwd_gate(mid_stream s, float *stats_ws, const float *x, const float *gamma, const float *beta,
                    const float *means, const float *vars, const float *dy, const float *mask_src, float *gated_out,
                    float *dx, float *dgamma, float *dbeta, int N, int C, int P, float eps);

/* the same operators over typed tensors: x_dt = storage type of the convolution output x (and of dx), a_dt = of the
 * activation-side tensors (y, residual, dy, mask_src, gated_out).  Pairs: (f32,f32), (bf16,bf16), (f32,bf16). */
int mid_bn_fwd_t(mid_stream s, float *stats_ws, const mid_bn_parts *parts, const void *x, int x_dt, const float *gamma,
                 const float *beta, const void *residual, float *means, float *vars, void *y, int a_dt, float *xhat_out,
                 float *norm_out, int N, int C, int P, float eps, int relu);
/* cross-replica batch-norm statistics through `comm` (an RCCL communicator of its own); NULL = per-replica (the reference) */
void mid_bn_set_sync(void *comm, int world, float *tmp, size_t tmp_floats, int force);
/* test aid: the sync-BN merge kernels on R replicas held by one process (the all-reduce replaced by a sum kernel); see kernels_bn.hip */
int mid_bn_debug_merge(mid_stream s, int R, int C, float *means, float *vars, float *dgamma, float *dbeta, float *sums_out, float *tmp);
int mid_bn_stats_t(mid_stream s, float *stats_ws, const void *x, int x_dt, float *means, float *vars, int N, int C, int P);
int mid_bn_apply_t(mid_stream s, const void *x, int x_dt, const float *gamma, const float *beta, const void *residual,
                   const float *means, const float *vars, void *y, int a_dt, int N, int C, int P, float eps, int relu);
/* mask_mode 0..2 as mid_bn_bwd; 3 = mid_bn_bwd_gate */
int mid_bn_bwd_t(mid_stream s, float *stats_ws, const void *x, int x_dt, const float *gamma, const float *beta, const float *means,
                 const float *vars, const void *dy, const void *mask_src, void *gated_out, int a_dt, void *dx, float *dgamma,
                 float *dbeta, int N, int C, int P, float eps, int mask_mode);
int mid_maxpool_fwd_t(mid_stream s, const void *x, void *y, int dt, int *max_inds, int N, int C, int H, int k, int stride);
int mid_maxpool_bwd_t(mid_stream s, const int *max_inds, const void *dy, void *dx, int dt, int N, int C, int H, int k, int stride);
int mid_avgpool_fwd_t(mid_stream s, const void *x, int dt, float *y, int N, int C, int P);
int mid_avgpool_bwd_t(mid_stream s, const float *dy, void *dx, int dt, int N, int C, int P);

/* ---- pools, elementwise, loss, optimizer ---- */
int mid_maxpool_fwd(mid_stream s, const float *x, float *y, int *max_inds, int N, int C, int H, int k, int stride);
int mid_maxpool_bwd(mid_stream s, const int *max_inds, const float *dy, float *dx, int N, int C, int H, int k,
                    int stride);
int mid_avgpool_fwd(mid_stream s, const float *x, float *y, int N, int C, int P);
int mid_avgpool_bwd(mid_stream s, const float *dy, float *dx, int N, int C, int P);
int mid_relu_deriv(mid_stream s, const float *x, const float *up, float *out, size_t n);
int mid_add_relu(mid_stream s, const float *a, const float *b, float *sum_out, float *act_out, size_t n);
int mid_softmax(mid_stream s, const float *x, float *out, int N, int L);
int mid_ce_deriv(mid_stream s, const float *pred, const int *labels, float *d, int N, int L);
/* fused updateMeans+updateVars+updateParams (resnet.cu:605-662).  On NaN/Inf *nan_flag (device int) becomes the highest offending
 * locations[] index + 1 (check_errors, resnet.cu:2879-2907): loc_off_dev = n_loc + 1 arena offsets (floats) of the tensors, base =
 * arena offset of p[0]; loc_off_dev NULL: the flag becomes 1. */
int mid_adam(mid_stream s, float *p, float *g, float *m, float *v, size_t n, float lr, float wd, float b1,
             float b2, float cur_b1, float cur_b2, float eps, int *nan_flag, int zero_grads, const size_t *loc_off_dev, int n_loc,
             size_t base);
int mid_nhwc_to_nchw(mid_stream s, const float *in, float *out, int N, int H, int W, int C);
int mid_nchw_to_nhwc(mid_stream s, const float *in, float *out, int N, int C, int H, int W);
/* splitmix64 counter streams on device (synthetic batches): uniform in [lo,hi) / labels mod n_classes */
int mid_fill_uniform(mid_stream s, float *out, size_t n, uint64_t seed, uint64_t offset, float lo, float hi);
int mid_lds_poison(mid_stream s); /* test aid: fills LDS of every CU with NaNs */
int mid_fill_labels(mid_stream s, int *out, size_t n, uint64_t seed, uint64_t offset, int n_classes);

/* ---- RCCL (resolved with dlopen at first use) ---- */
int mid_rccl_unique_id_bytes(void);
int mid_rccl_get_unique_id(void *out, int bytes);
void *mid_rccl_comm_init(int rank, int world, const void *unique_id, int bytes);
int mid_rccl_allreduce_sum(void *comm, float *buf, size_t count, mid_stream s);
void mid_rccl_comm_destroy(void *comm);
void mid_rccl_comm_abort(void *comm);

#ifdef __cplusplus
}
#endif
#endif
