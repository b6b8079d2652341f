/*
 * resnet_mi.h -- C-ABI of libresnet_mi.so: the MI355X (gfx950) drop-in for the trainer
 * surface of als244/ResNet.
 *
 * The reference's headers declare only structs (resnet.h:4-215, resnet_cudnn.h:4-216); its
 * entry points are C++ functions defined in resnet.cu.  This header keeps the struct layouts
 * field-for-field in the reference's order (so code written against resnet.h keeps compiling and
 * every offset is unchanged) and declares the entry points with C linkage, each citing the
 * definition it replaces.  `bool` parameters became `int`; curandGenerator_t* became an opaque
 * seed handle (MiRng*); the cudnnHandle_t member of resnet_cudnn.h:213 is the opaque
 * `backend_ctx` slot (HIP streams, workspaces, RCCL communicator).
 *
 * Device memory layout: activations NCHW fp32 (north_star), weights KCRS as the reference
 * (resnet.cu:140), FC weights [in][out] (resnet.cu:1751-1759).  All device pointers below are
 * HIP device pointers owned by the trainer; nothing is freed before destroy_trainer().
 */
#ifndef RESNET_MI_H
#define RESNET_MI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- reference struct layouts (resnet.h) ---------------- */
typedef struct { /* resnet.h:4-9 */
    char **labels;
    char **synsets;
    int *counts;
    int n_classes;
} Class_Metadata;

typedef struct { /* resnet.h:11-33 */
    int input;
    int init_kernel_dim;
    int init_conv_filters;
    int init_conv_stride;
    int init_maxpool_dim;
    int init_maxpool_stride;
    int n_conv_blocks;
    int *is_block_spatial_reduction; /* caller-owned */
    int final_depth;
    int output;
} Dims;

typedef struct { /* resnet.h:35-40 */
    int spatial_dim;
    int depth;
    float *gamma;
    float *beta;
} BatchNorm;

typedef struct { /* resnet.h:43-74 */
    int incoming_filters;
    int incoming_spatial_dim;
    int reduced_depth;
    int expanded_depth;
    int stride;
    float *depth_reduction;
    BatchNorm *norm_depth_reduction;
    float *spatial;
    BatchNorm *norm_spatial;
    float *depth_expansion;
    BatchNorm *norm_expansion;
    float *projection; /* NULL when incoming_filters == expanded_depth */
    BatchNorm *norm_projection;
} ConvBlock;

typedef struct { /* resnet.h:78-88 */
    float *init_conv_layer;
    BatchNorm *norm_init_conv;
    ConvBlock **conv_blocks;
    float *fully_connected;
    float **locations; /* every tensor, order of resnet.cu:838-943 */
    int *sizes;
    int n_locations; /* COUNTED (3 + 9n + 3*#projections + 1), not the reference's 16+9n (:819) */
} Params;

typedef struct { /* resnet.h:90-97 */
    int input_size;
    int feature_size;
    float *means;
    float *vars;
    float *normalized_temp; /* x-hat; NULL unless the trainer stores full activations */
    float *normalized;      /* BN output before ReLU; NULL unless full-store */
} Cache_BatchNorm;

typedef struct { /* resnet.h:99-133 */
    int incoming_filters;
    int incoming_spatial_dim;
    int reduced_depth;
    int expanded_depth;
    int stride;
    float *post_reduced;
    Cache_BatchNorm *norm_post_reduced;
    float *post_reduced_activated;
    float *post_spatial;
    Cache_BatchNorm *norm_post_spatial;
    float *post_spatial_activated;
    float *post_expanded;
    Cache_BatchNorm *norm_post_expanded;
    float *post_expanded_norm_vals;
    float *transformed_residual;
    Cache_BatchNorm *norm_post_projection;
    float *post_projection_norm_vals;
    float *output;           /* pre-ReLU sum; aliases output_activated unless full-store */
    float *output_activated;
} Activation_ConvBlock;

typedef struct { /* resnet.h:137-152 */
    float *init_conv_applied;
    Cache_BatchNorm *norm_init_conv;
    float *init_conv_activated;
    int *max_inds; /* flat NCHW argmax index */
    float *init_convblock_input;
    Activation_ConvBlock **activation_conv_blocks;
    int n_conv_blocks;
    float *final_conv_output_pooled;
    float *linear_output;
} Activations;

typedef struct { /* resnet.h:154-157 */
    Dims *dims;
    Params *params;
} ResNet;

typedef struct { /* resnet.h:160-166 */
    Activations *activations;
    float *pred;
    float *pred_cpu; /* valid when forward_pass returns */
} Forward_Buffer;

typedef struct { /* resnet.h:168-174 */
    float *output_layer_deriv;
    Params *param_derivs;
    Params *prev_means;
    Params *prev_vars;
    Activations *activation_derivs;
} Backprop_Buffer;

typedef struct { /* resnet.h:176-192 */
    int image_dim;
    int image_size;
    int n_images;
    int cur_shard_id;
    int cur_batch_in_shard;
    int shard_n_images;
    float *full_shard_images;
    int *full_shard_correct_classes;
    float *images_float_cpu; /* pinned */
    float *images;           /* device, NCHW */
    int *correct_classes_cpu; /* pinned; valid after load_new_batch */
    int *correct_classes;
} Batch;

typedef struct { /* resnet.h:195-215, with resnet_cudnn.h:213's handle slot */
    ResNet *model;
    Batch *cur_batch;
    Forward_Buffer *forward_buffer;
    Backprop_Buffer *backprop_buffer;
    float learning_rate;
    float weight_decay;
    float base_mean_decay;
    float base_var_decay;
    float cur_mean_decay;
    float cur_var_decay;
    float eps;
    int batch_size;
    int n_epochs;
    int cur_dump_id;
    int cur_epoch;
    float *loss_per_epoch;
    float *accuracy_per_epoch;
    int init_loaded;
    void *backend_ctx; /* resnet_cudnn.h:213 cudnnHandle_t -> opaque MiCtx* */
    const char *dump_dir;
} Train_ResNet;

typedef struct MiRng MiRng; /* stands in for curandGenerator_t (resnet.cu:3264-3267) */

/* ---------------- entry points the reference's main() calls ---------------- */
/* resnet.cu:666 */
Dims *init_dimensions(int input, int init_kernel_dim, int init_conv_filters, int init_conv_stride,
                      int init_maxpool_dim, int init_maxpool_stride, int n_conv_blocks,
                      int *is_block_spatial_reduction, int final_depth, int output);
/* curandCreateGenerator + curandSetPseudoRandomGeneratorSeed, resnet.cu:3266-3267 */
MiRng *mi_rng_create(uint64_t seed);
void mi_rng_destroy(MiRng *);
/* resnet.cu:951 (Glorot-normal var 2/(fan_in+fan_out), FC var 1e-4, gamma 1, beta 0) */
ResNet *init_resnet(Dims *dims, MiRng *gen);
/* resnet.cu:1196 */
Batch *init_general_batch(int n_images, int image_size, int image_dim, int shard_n_images);
/* resnet.cu:1157 */
Train_ResNet *init_trainer(ResNet *model, Batch *cur_batch, int batch_size, float learning_rate, float weight_decay,
                           float mean_decay, float var_decay, float eps, int n_epochs, const char *dump_dir);
/* resnet_cudnn.cu:1160 (same, with the handle argument; `handle` is ignored) */
Train_ResNet *init_trainer_cudnn_abi(ResNet *model, Batch *cur_batch, int batch_size, float learning_rate,
                                     float weight_decay, float mean_decay, float var_decay, float eps, int n_epochs,
                                     void *handle, const char *dump_dir);
/* resnet.cu:1363 */
Class_Metadata *populate_class_info(char *label_filename, char *synset_filename, char *class_size_filename,
                                    int n_classes);
/* resnet.cu:1235 */
void load_new_batch(Train_ResNet *trainer, Class_Metadata *class_metadata, Batch *batch_buffer);
/* resnet.cu:1526: fills forward_buffer->pred and pred_cpu (blocks until pred_cpu is valid) */
void forward_pass(Train_ResNet *trainer);
/* resnet.cu:1777: fills backprop_buffer->param_derivs */
void backwards_pass(Train_ResNet *trainer);
/* resnet.cu:2910: Adam, zero gradients + batch buffers, advance decays, dump every 1000 steps */
void update_parameters(Train_ResNet *trainer);
/* resnet.cu:2755 / 2778 / 2821 */
void dump_trainer(int dump_id, Train_ResNet *trainer, const char *special_dir);
void overwrite_trainer_hyperparams(Train_ResNet *trainer, int dump_id, const char *special_dir);
void overwrite_model_params(Train_ResNet *trainer, int dump_id, const char *special_dir);

/* ---------------- additions (new symbols; nothing above changes) ---------------- */
int mi_device_count(void);
int mi_set_device(int device);       /* call before any init_* ; default device 0 */
const char *mi_last_error(void);     /* "" when no HIP/RCCL error has been recorded */
void mi_device_synchronize(void);    /* cudaDeviceSynchronize of the reference main loop */
void destroy_trainer(Train_ResNet *trainer); /* frees trainer, model, batch and every device buffer */

/* data source selection for load_new_batch (the reference hard-codes /mnt/storage paths, :1275) */
enum { MI_SRC_SHARDS = 0, MI_SRC_BUFFER = 1, MI_SRC_SYNTHETIC = 2, MI_SRC_HOST = 3 };
enum { MI_LAYOUT_NHWC = 0, MI_LAYOUT_NCHW = 1 };
/* shards: <dir>/%03d.images + %03d.labels (build_training_shards.c:150-160 / resnet.cu:1275-1285) */
void mi_batch_source_shards(Batch *b, const char *shard_dir, int layout);
/* overlap the H2D copy of batch t+1 with step t (copy stream, pinned double buffer); shard source only */
void mi_batch_set_prefetch(Batch *b, int on);
/* one dumped batch: images.buffer / labels.buffer (resnet.cu:1301-1311) */
void mi_batch_source_buffer(Batch *b, const char *images_path, const char *labels_path, int layout);
/* seeded synthetic stream kept resident in HBM: images U(-124,152), labels uniform */
void mi_batch_source_synthetic(Batch *b, uint64_t seed_images, uint64_t seed_labels, int n_classes, int pool_batches);
/* caller fills images_float_cpu / correct_classes_cpu itself before each load_new_batch */
void mi_batch_source_host(Batch *b, int layout);
/* 0 = exit(1) on a missing shard/buffer file like fopen failure should (reference leaves it unchecked, :1276) */
int mi_batch_last_status(const Batch *b);

/* options (set after init_trainer, before the first forward_pass) */
void mi_trainer_set_full_store(Train_ResNet *t, int on); /* also keep x-hat / BN-out / pre-ReLU sums (dump parity) */
void mi_trainer_set_dump_root(Train_ResNet *t, const char *root); /* replaces /mnt/storage/.../training_dumps */
void mi_trainer_set_dump_every(Train_ResNet *t, int every);       /* reference: 1000 (:2947); 0 disables */
void mi_trainer_set_input_reset(Train_ResNet *t, int on);
/* weight-gradient scheduling in backwards_pass.  0: everything on one stream, in the reference's order.  1: each layer's
 * weight gradient runs on a second stream next to the following layer's BN backward only (default).  2: weight gradients
 * run free on a low-priority second stream; the derivative tensors they read come from a ring of buffers and are only
 * rewritten after the reader has finished.  Results are bit-identical in all three modes. */
void mi_trainer_set_overlap(Train_ResNet *t, int mode);
float mi_host_loss(Train_ResNet *t, int *n_wrong);                /* resnet.cu:3363-3383 on pred_cpu */

/* raw device access for tests / weight injection (model_params/%03d.buffer semantics, resnet.cu:2845-2874) */
void mi_copy_to_device(void *dst_dev, const void *src_host, size_t bytes);
void mi_copy_to_host(void *dst_host, const void *src_dev, size_t bytes);

/* data parallel (new work): one process per GPU, RCCL all-reduce SUM of the gradient arena */
int mi_dp_unique_id_bytes(void);
int mi_dp_get_unique_id(void *out, int bytes);                        /* rank 0 */
int mi_dp_init(Train_ResNet *t, int rank, int world, const void *unique_id, int bytes);
void mi_dp_set_bucket_bytes(Train_ResNet *t, size_t bytes);
int mi_dp_world(const Train_ResNet *t);
/* option, default off (the reference has no cross-replica BN): batch-norm statistics and the (dbeta, dgamma) sums
 * taken over ALL replicas -- two all-reduces of [C] floats per BN layer in forward, one of [2C] in backward, through a
 * communicator of its own.  unique_id: a second id from mi_dp_get_unique_id (rank 0), broadcast like the first; call after
 * mi_dp_init.  NULL turns it off again. */
int mi_dp_enable_sync_bn(Train_ResNet *t, const void *unique_id, int bytes);

/* per-phase device timing of the last step in ms: [0]=load [1]=forward [2]=backward [3]=update [4]=allreduce wait */
void mi_trainer_last_timings(Train_ResNet *t, float out_ms[5]);

/* optional per-kernel-family timing with HIP events on the launch stream (used by bench.py's roofline):
 * family 0 direct (VALU) conv fwd/dgrad, 1 direct (VALU) conv wgrad, 2 1x1 conv / FC on MFMA, 3 batch norm,
 * 5 3x3 conv on the MFMA implicit GEMM (fwd, dgrad, wgrad; 4 is unused).
 * flops/bytes are the ALGORITHMIC work of the timed launches. */
void mi_prof_enable(int on); /* 0 off, 1 all families, otherwise a bit mask of (1 << family) */
void mi_prof_reset(void);
void mi_prof_get(int family, long *launches, double *ms, double *flops, double *bytes);

/* ---------------- operator layer (prepareAndDo* of resnet.cu:1386-1509), device pointers, NCHW ----------------
 * Exposed so parity tests can drive every kernel through the C-ABI on its own. `stream` NULL = the library's
 * compute stream. All return 0 on success. */
typedef void *mi_stream_t;
void *mi_malloc(size_t bytes);
void mi_free(void *p);
int mi_op_conv_fwd(const float *x, const float *w_kcrs, float *y, int N, int C, int H, int K, int k, int stride);
int mi_op_conv_dgrad(const float *w_kcrs, const float *dy, float *dx, int N, int C, int H, int K, int k, int stride,
                     int to_add);
int mi_op_conv_wgrad(const float *x, const float *dy, float *dw_kcrs, int N, int C, int H, int K, int k, int stride);
int mi_op_bn_fwd(const float *x, const float *gamma, const float *beta, float *means, float *vars, float *y, int N,
                 int C, int H, float eps, int relu);
/* y = relu(BN(x) + residual) fused (addVec + doActivation, resnet.cu:1717-1723) */
int mi_op_bn_fwd_add_relu(const float *x, const float *gamma, const float *beta, const float *residual, float *means,
                          float *vars, float *y, int N, int C, int H, float eps);
/* mask_mode 0 none, 1 own ReLU recomputed from x (activationAndBatchNormDeriv to_activate_deriv),
 * 2 external: dy is gated by mask_src > 0 (doActivationDeriv fused in, resnet.cu:1934) */
int mi_op_bn_bwd(const float *x, const float *gamma, const float *beta, const float *means, const float *vars,
                 const float *dy, const float *mask_src, float *dx, float *dgamma, float *dbeta, int N, int C, int H,
                 float eps, int mask_mode);
/* mask_mode 2 that also writes gated_out = (mask_src > 0 ? dy : 0): what backwards_pass uses for identity blocks, where the
 * gated upstream gradient is needed again as the shortcut addend (doActivationDeriv, resnet.cu:1934, without its own pass) */
int mi_op_bn_bwd_gate(const float *x, const float *gamma, const float *beta, const float *means, const float *vars,
                      const float *dy, const float *mask_src, float *gated_out, float *dx, float *dgamma, float *dbeta, int N,
                      int C, int H, float eps);
int mi_op_maxpool_fwd(const float *x, float *y, int *max_inds, int N, int C, int H, int k, int stride);
int mi_op_maxpool_bwd(const int *max_inds, const float *dy, float *dx, int N, int C, int H, int k, int stride);
int mi_op_avgpool_fwd(const float *x, float *y, int N, int C, int H);
int mi_op_avgpool_bwd(const float *dy, float *dx, int N, int C, int H);
int mi_op_relu_deriv(const float *x, const float *up, float *out, size_t n);
/* out[m x n] = A[m x k] * B[k x n], row-major (matMul, resnet.cu:70-85) and the two transposed forms
 * (prepareAndDoMatMulLeftTranspose / RightTranspose, resnet.cu:1482-1509) */
int mi_op_matmul(const float *A, const float *B, float *out, int m, int k, int n);
int mi_op_matmul_lt(const float *A_kxm, const float *B, float *out, int m, int k, int n);
int mi_op_matmul_rt(const float *A, const float *B_nxk, float *out, int m, int k, int n);
int mi_op_softmax(const float *x, float *out, int N, int L);
int mi_op_ce_deriv(const float *pred, const int *labels, float *d, int N, int L);
int mi_op_adam(float *p, const float *g, float *m, float *v, size_t n, float lr, float wd, float b1, float b2,
               float cur_b1, float cur_b2, float eps, int *nan_flag_dev);
int mi_op_nhwc_to_nchw(const float *in, float *out, int N, int H, int W, int C);
/* device-side seeded fill (splitmix64 counter stream, uniform [lo,hi)) -- synthetic operands for micro-benchmarks */
/* test aid: leaves NaNs in the LDS of every CU (catches kernels that read LDS they did not write) */
int mi_debug_poison_lds(void);
/* host-only (no GPU needed): route and grid the launch planners choose for a convolution.  op 0 fwd, 1 dgrad, 2 wgrad.
 * out[0] 1 = MFMA implicit GEMM / 0 = other kernels, [1] rows per tile, [2] tiles, [3] tiles launched whole, [4] reduction
 * slices per tail tile, [5] k-steps per slice, [6] wgrad splits, [7] workgroups per class or split, [8] k-steps */
int mi_debug_conv_plan(int op, int N, int C, int H, int K, int k, int stride, int out[9]);
int mi_op_fill_uniform(float *out, size_t n, uint64_t seed, float lo, float hi);


/* ---------------- bf16-activation path (BASELINE configs[4]) ----------------
 * Activations and activation gradients stored as bf16 in the same NCHW tensors (the `float *` fields of Activations then
 * point at bf16 data, half the bytes), all arithmetic in fp32, convolutions on v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation; parameters, parameter gradients, Adam state, BN statistics, the stem convolution's own output, the pooled
 * features and the FC / soft-max head stay fp32.  Structure mirrored: resnet_cudnn_nchw.cu:1196-1211 (NCHW tensors,
 * TENSOR_OP_MATH_ALLOW_CONVERSION), storage policy of resnet_cudnn_lowmem.cu:2152-2170. */
enum { MI_DTYPE_F32 = 0, MI_DTYPE_BF16 = 1 };
/* call after init_trainer and before the first load_new_batch / forward_pass: rebuilds the activation buffers at the new
 * element size.  Returns 0, or -1 (mi_last_error says why: a layer shape the bf16 kernels do not tile, full-store on). */
int mi_trainer_set_dtype(Train_ResNet *t, int dtype);
int mi_trainer_get_dtype(const Train_ResNet *t);
/* what backward keeps from forward.  FAST (default): per convolution the raw output and the BN(+ReLU) output, per block the
 * post-ReLU output.  RECOMPUTE_BN: raw convolution outputs, BN statistics and block outputs only; the BN(+ReLU) tensors are
 * re-derived in backward (resnet_clean.cu:2714, 2753, 2812; resnet_cudnn_lowmem.cu:2303-2313) -- bit-identical gradients,
 * fewer stored bytes.  FULL: FAST plus x-hat / BN-out / pre-ReLU sums (= mi_trainer_set_full_store, fp32 only). */
enum { MI_STORE_FAST = 0, MI_STORE_RECOMPUTE_BN = 1, MI_STORE_FULL = 2 };
int mi_trainer_set_store_policy(Train_ResNet *t, int policy);
/* bytes of device memory the trainer holds for forward activations (kept for backward) / for everything */
size_t mi_trainer_activation_bytes(const Train_ResNet *t);
size_t mi_trainer_device_bytes(const Train_ResNet *t);
void mi_clear_error(void);
/* check_errors on demand (resnet.cu:2879-2907): update_parameters no longer blocks to read the NaN / Inf flag of its Adam
 * launch; it is read at the next forward_pass, or here (waits for the device; dumps id 99999999 and exits like the reference
 * when set).  Returns 0 when clean. */
int mi_trainer_check_errors(Train_ResNet *t);
/* the locations[] index the last NaN / Inf report named -- "ERROR: nan or inf found at location: %d" (resnet.cu:2896; the
 * reference walks locations[] from the last to the first, :2952, so the highest offending index is the one it prints); -1 = none.
 * mi_trainer_set_nan_exit(t, 0): a report no longer ends the process (the reference's exit(1), :2899) but comes back through
 * mi_trainer_check_errors / mi_trainer_nan_location -- for tests. */
int mi_trainer_stem_dtype(Train_ResNet *t); /* storage type of activations->init_conv_applied (the stem convolution's own output) and of its gradient: MI_DTYPE_BF16 in the bf16 mode with the matrix-core stem, else MI_DTYPE_F32 */
int mi_trainer_nan_location(const Train_ResNet *t);
void mi_trainer_set_nan_exit(Train_ResNet *t, int on);
/* test aid: the device-side merge of cross-replica batch norm (mi_dp_enable_sync_bn) on R replicas held by ONE process -- the
 * kernels the trainer launches around its collectives, with the all-reduce replaced by a sum over the R supplied buffers.
 * Device pointers [R][C]: means / vars (per-replica in, merged out), dgamma / dbeta (per-replica sums in, gradient-arena values
 * out = sum / R), sums_out [R][2C] (the sums each replica's dx formula sees) or NULL.  Either pair may be NULL. */
int mi_debug_bn_merge(int R, int C, float *means, float *vars, float *dgamma, float *dbeta, float *sums_out);
/* the end-of-epoch bookkeeping of the reference's main() (resnet.cu:3410-3421) */
void mi_trainer_end_epoch(Train_ResNet *t, float epoch_loss, float epoch_n_wrong, float total_images_per_epoch);
/* host-only (no GPU needed): the gradient buckets the data-parallel path cuts for a network -- float offsets [from, to) into
 * the gradient arena in issue order (FC side first).  mi_debug_last_buckets: what the last backwards_pass really issued. */
int mi_debug_dp_plan(const Dims *d, size_t bucket_bytes, size_t *from, size_t *to, int max);
size_t mi_debug_arena_floats(const Dims *d);
int mi_debug_last_buckets(const Train_ResNet *t, size_t *from, size_t *to, int max);

/* the offline shard writer, build_training_shards.c:13-167 with its literal paths as arguments: reads the partition CSV
 * ("CCC,NNNN,RR,CC" per line) and <class_dir>/%08d.buffer (uint8, dim_in x dim_in x 3, B,G,R), crops dim_out x dim_out at the
 * per-image offsets, converts to R,G,B floats minus 103.94 / 116.78 / 123.68, writes <out_dir>/%03d.images (fp32, NCHW like the
 * reference, or NHWC) and %03d.labels (int32).  Returns the number of images written, < 0 on error. */
int mi_build_shard(const char *partition_csv, const char *class_dir, const char *out_dir, int shard_id, int image_dim_in,
                   int image_dim_out, int layout);
/* data parallel: this rank's slice of every global batch of a shard ("each rank reads its slice of the same
 * shard/batch"): global batch g of a shard = images [g*world*N, (g+1)*world*N), rank r takes [.. + r*N, .. + (r+1)*N) */
void mi_batch_set_rank_slice(Batch *b, int rank, int world);

/* typed operator layer: x_dt = storage type of the convolution-side tensors (x, dx), a_dt = of the activation-side tensors
 * (y, residual, dy, mask_src, gated_out).  Supported pairs: (F32,F32), (BF16,BF16), (F32,BF16). */
int mi_op_convert(const void *in, int in_dt, void *out, int out_dt, size_t n);
int mi_bf16_conv_supported(int op, int N, int C, int H, int K, int k, int stride); /* op 0 fwd, 1 dgrad, 2 wgrad */
int mi_bf16_pw_wgrad_supported(int N, int C, int H, int K); /* 1: this 1x1 weight gradient runs on the LDS-DMA kernel (both operands as they lie) */
int mi_op_conv_fwd_bf16(const void *x_bf16, const float *w_kcrs, void *y_bf16, int N, int C, int H, int K, int k, int stride);
int mi_op_conv_dgrad_bf16(const float *w_kcrs, const void *dy_bf16, void *dx_bf16, int N, int C, int H, int K, int k, int stride,
                          int to_add);
int mi_op_conv_wgrad_bf16(const void *x_bf16, const void *dy_bf16, float *dw_kcrs, int N, int C, int H, int K, int k, int stride);
/* prepreAndDoConvolutionDeriv + activationAndBatchNormDeriv as backwards_pass chains them in bf16 storage (resnet.cu:1399-1429,
 * 1455-1480): dgrad of a convolution, then the backward of the batch norm (+ReLU, gate = mask > 0) in front of it; where the launch
 * allows the dgrad does the BN' reduction pass in its epilogue.  gated = (mask > 0 ? dgrad(+addend) : 0), bn_dx = the BN's input
 * gradient.  Image tensors bf16.  Returns < 0 on error, else the number of partial rows the dgrad left (0 = separate pass). */
int mi_op_conv_dgrad_bn_bwd_bf16(const float *w_kcrs, const void *dy, const void *addend, void *gated, int N, int C, int H, int K, int k,
                                 int stride, const void *bn_x, const void *mask, const float *gamma, const float *beta, const float *means,
                                 const float *vars, float eps, void *bn_dx, float *dgamma, float *dbeta);
/* 3x3 convolutions of the bf16 path on channel-last, zero-padded operands (round 3; kernels_cl_bf16.hip): the operand is re-laid once
 * (one plane with a halo of 1, or four parity planes for stride 2), both MFMA operands then go global -> LDS by LDS-DMA.  Same tensors
 * and semantics as mi_op_conv_fwd_bf16 / mi_op_conv_dgrad_bf16 with k = 3; -2: shape not covered (channels % 64) */
int mi_op_conv_fwd_bf16_cl(const void *x_bf16, const float *w_kcrs, void *y_bf16, int N, int C, int H, int K, int stride);
int mi_op_conv_wgrad_bf16_cl(const void *x_bf16, const void *dy_bf16, float *dw_kcrs, int N, int C, int H, int K, int stride); /* C % 128, K % 128, plane % 4 */
int mi_op_bn_fwd_cl_bf16(const void *x_bf16, const float *gamma, const float *beta, const void *residual_bf16, float *means, float *vars, void *y_bf16, void *y_cl, int N, int C, int H, float eps, int par); /* BN (+ residual) + ReLU written twice: NCHW and channel-last (par 0: one zero-padded plane [N][H+2][H+2][C]; par 1: the four parity planes of a stride-2 3x3; interior only; C % 64) */
int mi_op_conv1x1_fwd_bf16_cl(const void *x_bf16, const float *w_kc, void *y_bf16, int N, int C, int H, int K); /* 1x1 forward on the input re-laid dense channel-last (one tap of the channel-last kernel); C, K % 64 */
int mi_op_conv_wgrad_bf16_cl2(const void *x, const void *dy, float *dw, int N, int C, int H, int K, int stride); /* 3x3: BOTH operands as channel-last planes (stride 2: the input's parity planes and the dY planes of the stride-2 dgrad; stride 1: both with a halo of 1); C, K % 128, any plane size */
int mi_op_conv_dgrad_bf16_cl(const float *w_kcrs, const void *dy_bf16, void *dx_bf16, int N, int C, int H, int K, int stride, int to_add); /* stride 2: C % 128, no to_add */
/* the same chain in fp32 storage: dgrads with stride 1 on the MFMA implicit-GEMM route do the reduction in their epilogue.  Image tensors
 * fp32.  Returns < 0 on error, else the number of partial rows the dgrad left (0 = separate pass). */
int mi_op_conv_dgrad_bn_bwd_f32(const float *w_kcrs, const float *dy, const float *addend, float *gated, int N, int C, int H, int K, int k,
                                int stride, const float *bn_x, const float *mask, const float *gamma, const float *beta, const float *means,
                                const float *vars, float eps, float *bn_dx, float *dgamma, float *dbeta);
/* the stem convolution of the bf16 storage mode (7x7 stride 2, 3 -> 64 filters, H a multiple of 32; doConvolution /
 * convolutionDerivWeights, resnet.cu:109-156, 227-281): fp32 tensors in and out, image and weights rounded to bf16 inside,
 * fp32 accumulation on the bf16 matrix cores.  -2: shape not covered (the trainer then keeps the fp32 stem). */
int mi_op_stem_fwd_bf16(const float *x, const float *w_kcrs, float *y, int N, int H);
int mi_op_stem_wgrad_bf16(const float *x, const float *w_kcrs, const float *dy, float *dw_kcrs, int N, int H);
/* the same stem in exact fp32 arithmetic on the fp32 matrix cores (what the fp32 trainer runs unless RESNET_MI_IGEMM=0) */
int mi_op_stem_fwd_f32(const float *x, const float *w_kcrs, float *y, int N, int H);
int mi_op_stem_wgrad_f32(const float *x, const float *w_kcrs, const float *dy, float *dw_kcrs, int N, int H);
/* prepareAndDoConvolution + prepareAndDoBatchNormAndActivate as forward_pass pairs them (resnet.cu:1386-1396, 1431-1453): BN
 * statistics from the convolution's own epilogue where the layer runs on the implicit GEMM.  dt = storage type of x, conv_out, y.
 * Returns < 0 on error, else the number of statistics partial rows the convolution left (0 = separate statistics pass). */
int mi_op_conv_bn_fwd_t(const void *x, const float *w_kcrs, void *conv_out, int dt, const float *gamma, const float *beta,
                        float *means, float *vars, void *y, int N, int C, int H, int K, int k, int stride, float eps, int relu);
int mi_op_bn_fwd_t(const void *x, int x_dt, const float *gamma, const float *beta, const void *residual, float *means, float *vars,
                   void *y, int a_dt, int N, int C, int H, float eps, int relu);
int mi_op_bn_apply_t(const void *x, int x_dt, const float *gamma, const float *beta, const void *residual, const float *means,
                     const float *vars, void *y, int a_dt, int N, int C, int H, float eps, int relu);
int mi_op_bn_bwd_t(const void *x, int x_dt, const float *gamma, const float *beta, const float *means, const float *vars,
                   const void *dy, const void *mask_src, void *gated_out, int a_dt, void *dx, float *dgamma, float *dbeta, int N, int C,
                   int H, float eps, int mask_mode);
int mi_op_maxpool_fwd_t(const void *x, void *y, int dt, int *max_inds, int N, int C, int H, int k, int stride);
int mi_op_maxpool_bwd_t(const int *max_inds, const void *dy, void *dx, int dt, int N, int C, int H, int k, int stride);
int mi_op_avgpool_fwd_t(const void *x, int dt, float *y, int N, int C, int H);
int mi_op_avgpool_bwd_t(const float *dy, void *dx, int dt, int N, int C, int H);

#ifdef __cplusplus
}
#endif
#endif
