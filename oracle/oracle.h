/*
 * oracle.h -- CPU oracle for the als244/ResNet training hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under resnet_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * What it is: a plain-C, sequential-fp32 restatement of the reference's
 * CUDA kernels (reference: resnet.cu:44-662), of its pass orchestration
 * (resnet.cu:1526-1775 forward, 1777-2248 backward, 2910-2987 Adam), with the
 * two defects the reference itself fixed in resnet_cudnn.cu (stable softmax
 * resnet_cudnn.cu:572-583; the spatial BN backward call that resnet.cu:2060-2083
 * forgot, resnet_cudnn.cu:2365-2366).  Layout is the reference's: activations
 * NHWC, weights KCRS, FC weights [in][out].
 *
 * PARITY UNPINNED by reference fixtures: the reference is CUDA-only (cannot be
 * built or run here), ships no golden vectors for this path, and its curand
 * weight stream cannot be reproduced.  What pins this oracle instead:
 *   - the reference's own self-test definitions and tolerances
 *     (resnet.cu:2990-3107 matmul 1e-5 / transpose exact; 3109-3218 conv 1e-4),
 *   - labels.buffer (format of the label file),
 *   - an independent cross-check against torch-CPU autograd (tests/),
 *   - a double-accumulation build of the same code (liboracle_f64.so).
 */
#ifndef RESNET_ORACLE_H
#define RESNET_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-kernel restatements (NHWC activations, KCRS weights) ---- */
void orc_set_threads(int n);           /* OpenMP threads over independent outputs; 1 = reference-serial */
int  orc_get_threads(void);
int  orc_acc_is_double(void);

void orc_conv_fwd(const float *x, const float *w, int H, int k, int C, int K, int stride, int N, float *y);
void orc_conv_dgrad(const float *w, const float *dy, int H, int k, int C, int K, int stride, int N, int to_add, float *dx);
void orc_conv_wgrad(const float *x, const float *dy, int H, int k, int C, int K, int stride, int N, float *dw);
void orc_bn_fwd(const float *x, const float *gamma, const float *beta, int H, int C, int N, float eps,
                float *means, float *vars, float *xhat, float *normalized, float *activated, int to_activate);
void orc_bn_bwd(const float *x, const float *gamma, int H, int C, int N, float eps,
                const float *means, const float *vars, const float *xhat, const float *activated,
                const float *dy, float *dxhat, float *dgamma, float *dbeta, float *dx, int to_activate_deriv);
void orc_maxpool_fwd(const float *x, int k, int stride, int N, int Hin, int C, int *max_inds, float *y);
void orc_maxpool_bwd(const int *max_inds, const float *dy, int Hin, int stride, int C, int N, float *dx);
void orc_avgpool_fwd(const float *x, int H, int C, int N, float *y);
void orc_avgpool_bwd(const float *dy, int C, int N, int H, float *dx);
void orc_add(int n, const float *a, const float *b, float *o);
void orc_relu(int n, const float *x, float *o);
void orc_relu_deriv(int n, const float *x, const float *up, float *o);
void orc_matmul(const float *M, const float *Nn, int m, int k, int n, float *out);
void orc_transpose(const float *in, int rows, int cols, float *out);
void orc_softmax(const float *x, int N, int L, float *out);
void orc_softmax_unstable(const float *x, int N, int L, float *out);
void orc_ce_deriv(float *d, const int *labels, int L, int N);
void orc_adam(int n, float *p, const float *g, float *m, float *v, float lr, float wd, float b1, float b2,
              float cur_b1, float cur_b2, float eps);
float orc_loss(const float *pred, const int *labels, int N, int L, int *n_wrong);

/* ---- whole network (forward_pass / backwards_pass / update_parameters) ---- */
typedef struct OrcNet OrcNet;

OrcNet *orc_net_create(int input, int init_kernel_dim, int init_conv_filters, int init_conv_stride,
                       int init_maxpool_dim, int init_maxpool_stride, int n_conv_blocks,
                       const int *is_block_spatial_reduction, int final_depth, int output, int batch);
void    orc_net_destroy(OrcNet *);
int     orc_net_n_locations(const OrcNet *);
int     orc_net_location_size(const OrcNet *, int i);
float  *orc_net_param(OrcNet *, int i);
float  *orc_net_grad(OrcNet *, int i);
float  *orc_net_mean(OrcNet *, int i);
float  *orc_net_var(OrcNet *, int i);
void    orc_net_set_hyper(OrcNet *, float lr, float wd, float b1, float b2, float eps);
void    orc_net_set_batch(OrcNet *, const float *images_nhwc, const int *labels);
void    orc_net_forward(OrcNet *);
float   orc_net_loss(OrcNet *, int *n_wrong);
void    orc_net_backward(OrcNet *);
void    orc_net_update(OrcNet *);
/* tensor table: every activation / activation-derivative by the reference's dump name
 * (resnet.cu:2321-2680), e.g. "conv_blocks/00/spatial_applied"; derivs use prefix "d:" */
int     orc_net_n_tensors(const OrcNet *);
const char *orc_net_tensor_name(const OrcNet *, int i);
size_t  orc_net_tensor_size(const OrcNet *, int i);
void   *orc_net_tensor_ptr(OrcNet *, int i);
int     orc_net_find_tensor(const OrcNet *, const char *name);
/* geometry of tensor i as (N,H,W,C); 0s when not an image tensor */
void    orc_net_tensor_shape(const OrcNet *, int i, int shape[4]);

#ifdef __cplusplus
}
#endif
#endif
