/*
 * trainer.c -- host side of the drop-in trainer surface, in C over the thin device header (mi_device.h).
 * Mirrors the reference's L3/L4 layers: init_dimensions / init_resnet / init_trainer (resnet.cu:666-1194),
 * forward_pass (resnet.cu:1526-1775), backwards_pass (resnet.cu:1777-2248 with the spatial-BN fix of
 * resnet_cudnn.cu:2365-2366), update_parameters (resnet.cu:2910-2987).
 *
 * What is deliberately different from the reference (same results, MI355X-first structure):
 *  - one stream-ordered launch sequence, no cudaMalloc/cudaFree or blocking copies inside a step
 *    (the reference mallocs in FC backward, resnet.cu:1484-1508, and leaks in :2080);
 *  - parameters / gradients / Adam moments live in four contiguous arenas with identical offsets, so the
 *    480 Adam launches + 160 memsets + per-tensor D2H NaN scans of resnet.cu:2952-2978 are one launch, one
 *    memset and one 4-byte flag, and the gradient all-reduce runs over contiguous buckets;
 *  - BN stores only what backward needs (conv output, mean, var, activated); x-hat / BN-out / pre-ReLU
 *    sums exist only in full-store mode; BN+residual-add+ReLU is one kernel; ReLU' is fused into BN';
 *  - activation derivatives use six rolling buffers (the policy of resnet_cudnn_lowmem.cu:2152-2170)
 *    instead of a full mirror of the activation tree (resnet.cu:1151).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mi_host.h"

static MiGlobal G;
MiGlobal *mi_global(void) {
    if (!G.ready) {
        G.compute = mid_stream_create();
        G.comm = mid_stream_create();
        G.copy = mid_stream_create();
        G.aux = mid_stream_create_low_priority();
        G.ready = 1;
    }
    return &G;
}

int mi_device_count(void) { return mid_device_count(); }
int mi_set_device(int device) { return mid_set_device(device); }
const char *mi_last_error(void) { return mid_last_error(); }
void mi_device_synchronize(void) { mid_device_sync(); }
/* Any host write into device memory may have been a parameter (weight injection, resume).  mi_copy_to_device knows no trainer,
 * so it advances a process-wide write counter; every trainer remembers the count its re-laid weight copies were made at
 * (MiCtx.host_epoch_seen) and its own "Adam ran since" flag (MiCtx.params_dirty): staleness is per trainer, two trainers in one
 * process cannot clear each other's. */
static unsigned long g_host_write_epoch = 0;
void mi_params_mark_dirty(void) { g_host_write_epoch++; }
void mi_copy_to_device(void *d, const void *s, size_t n) { MiGlobal *g = mi_global(); mid_memcpy_h2d(d, s, n, g->compute); mid_stream_sync(g->compute); g_host_write_epoch++; }
void mi_copy_to_host(void *d, const void *s, size_t n) { MiGlobal *g = mi_global(); mid_memcpy_d2h(d, s, n, g->compute); mid_stream_sync(g->compute); }

void mi_prof_enable(int on) { mid_prof_enable(on); }
void mi_prof_reset(void) { mid_prof_reset(); }
void mi_prof_get(int family, long *launches, double *ms, double *flops, double *bytes) { mid_prof_get(family, launches, ms, flops, bytes); }

MiRng *mi_rng_create(uint64_t seed) {
    MiRng *r = (MiRng *)calloc(1, sizeof(MiRng));
    r->seed = seed;
    return r;
}
void mi_rng_destroy(MiRng *r) { free(r); }

void *mi_ctx_alloc(MiCtx *c, size_t bytes) {
    void *p = mid_malloc(bytes);
    if (!p) { fprintf(stderr, "resnet_mi: device allocation of %zu bytes failed: %s\n", bytes, mid_last_error()); exit(1); }
    if (c) {
        if (c->n_allocs == c->cap_allocs) {
            c->cap_allocs = c->cap_allocs ? c->cap_allocs * 2 : 256;
            c->allocs = (void **)realloc(c->allocs, sizeof(void *) * c->cap_allocs);
            c->alloc_bytes = (size_t *)realloc(c->alloc_bytes, sizeof(size_t) * c->cap_allocs);
        }
        c->alloc_bytes[c->n_allocs] = bytes;
        c->allocs[c->n_allocs++] = p;
        c->dev_bytes += bytes;
        if (c->counting_act) c->act_bytes += bytes;
    }
    return p;
}
/* frees allocs[first ..): everything a rebuild of the activation buffers replaces */
static void ctx_free_from(MiCtx *c, int first) {
    for (int i = first; i < c->n_allocs; i++) { mid_free(c->allocs[i]); c->dev_bytes -= c->alloc_bytes[i]; }
    c->n_allocs = first;
}
/* every live trainer context: a rank that is about to exit on an error tears ALL its communicators down first (gradient
 * buckets and sync-BN), so that its peers' collectives fail instead of hanging */
static MiCtx *g_live = NULL;
static void abort_all_comms(void) {
    for (MiCtx *c = g_live; c; c = c->next_live) {
        if (c->comm) { mid_rccl_comm_abort(c->comm); c->comm = NULL; }
        if (c->sync_bn_comm) { mid_bn_set_sync(NULL, 1, NULL, 0, 0); mid_rccl_comm_abort(c->sync_bn_comm); c->sync_bn_comm = NULL; }
    }
}
/* a launcher that fails leaves unwritten tensors behind: stop like the allocation failure does (the reference's void
 * API gives its caller nothing to poll) */
static void ck(int rc, const char *what) {
    if (rc) { fprintf(stderr, "resnet_mi: %s failed (%d): %s\n", what, rc, mid_last_error()); abort_all_comms(); exit(1); }
}

/* resnet.cu:666-682 */
Dims *init_dimensions(int input, int init_kernel_dim, int init_conv_filters, int init_conv_stride, int init_maxpool_dim,
                      int init_maxpool_stride, int n_conv_blocks, int *is_block_spatial_reduction, int final_depth,
                      int output) {
    Dims *d = (Dims *)malloc(sizeof(Dims));
    d->input = input; d->init_kernel_dim = init_kernel_dim; d->init_conv_filters = init_conv_filters;
    d->init_conv_stride = init_conv_stride; d->init_maxpool_dim = init_maxpool_dim;
    d->init_maxpool_stride = init_maxpool_stride; d->n_conv_blocks = n_conv_blocks;
    d->is_block_spatial_reduction = is_block_spatial_reduction; d->final_depth = final_depth; d->output = output;
    return d;
}

/* ---------------------------------------------------------------------------------------------- */
/* Parameter-shaped structure over one contiguous arena.  Location order = resnet.cu:838-943.       */
#define ARENA_ALIGN 64
static size_t align_up(size_t v) { return (v + ARENA_ALIGN - 1) / ARENA_ALIGN * ARENA_ALIGN; }

typedef struct { float *base; size_t off; float **loc; int *sizes; int n; double *var; } Carver;
static float *carve(Carver *c, int size, double var /* <0: gamma, ==0: zero */) {
    float *p = c->base + c->off;
    c->loc[c->n] = p; c->sizes[c->n] = size; c->var[c->n] = var; c->n++;
    c->off += align_up((size_t)size);
    return p;
}
static BatchNorm *make_bn(Carver *c, int spatial, int depth) {
    BatchNorm *b = (BatchNorm *)malloc(sizeof(BatchNorm));
    b->spatial_dim = spatial; b->depth = depth;
    b->gamma = carve(c, depth, -1.0);
    b->beta = carve(c, depth, 0.0);
    return b;
}
static int count_locations(const Dims *d, size_t *arena) {
    int n = 3, inc = d->init_conv_filters, ex = 4 * inc, red = inc;
    size_t a = align_up((size_t)d->init_kernel_dim * d->init_kernel_dim * inc * 3) + 2 * align_up(inc);
    for (int i = 0; i < d->n_conv_blocks; i++) {
        int stride = 1;
        if (d->is_block_spatial_reduction[i] == 1) { stride = 2; red *= 2; ex *= 2; }
        n += 9;
        a += align_up((size_t)inc * red) + align_up((size_t)red * red * 9) + align_up((size_t)ex * red) + 4 * align_up(red) + 2 * align_up(ex);
        if (inc != ex) { n += 3; a += align_up((size_t)inc * ex * (stride == 2 ? 9 : 1)) + 2 * align_up(ex); }
        inc = ex;
    }
    n += 1;
    a += align_up((size_t)ex * d->output);
    *arena = a;
    return n;
}
size_t mi_params_arena_floats(const Params *p) {
    const int l = p->n_locations - 1;
    return (size_t)(p->locations[l] - p->locations[0]) + align_up((size_t)p->sizes[l]);
}
float *mi_params_arena_base(const Params *p) { return p->locations[0]; }

/* init_model_parameters, resnet.cu:805-949.  gen == NULL -> zero twin (gradients, Adam moments, :1148-1150) */
static Params *build_params(const Dims *d, MiRng *gen, MiCtx *owner) {
    size_t arena_floats;
    const int nloc = count_locations(d, &arena_floats);
    Params *p = (Params *)malloc(sizeof(Params));
    Carver c;
    c.base = (float *)mi_ctx_alloc(owner, arena_floats * sizeof(float));
    c.off = 0; c.n = 0;
    c.loc = (float **)malloc(sizeof(float *) * nloc);
    c.sizes = (int *)malloc(sizeof(int) * nloc);
    c.var = (double *)malloc(sizeof(double) * nloc);
    const int f = d->init_conv_filters, kd = d->init_kernel_dim;
    p->init_conv_layer = carve(&c, kd * kd * f * 3, 2.0 / (7.0 * 7.0 * (3 + f))); /* fan literal 7*7, resnet.cu:831 */
    p->norm_init_conv = make_bn(&c, d->input / d->init_conv_stride, f);
    p->conv_blocks = (ConvBlock **)malloc(sizeof(ConvBlock *) * (d->n_conv_blocks > 0 ? d->n_conv_blocks : 1));
    int inc = f, H = d->input / 4, red = f, ex = 4 * f; /* resnet.cu:857-862 */
    for (int i = 0; i < d->n_conv_blocks; i++) {
        int stride = 1;
        if (d->is_block_spatial_reduction[i] == 1) { stride = 2; red *= 2; ex *= 2; }
        ConvBlock *b = (ConvBlock *)calloc(1, sizeof(ConvBlock));
        b->incoming_filters = inc; b->incoming_spatial_dim = H; b->reduced_depth = red; b->expanded_depth = ex; b->stride = stride;
        b->depth_reduction = carve(&c, inc * red, 2.0 / (double)(inc + red));
        b->norm_depth_reduction = make_bn(&c, H, red);
        b->spatial = carve(&c, red * red * 9, 2.0 / (9.0 * (red + red)));
        b->norm_spatial = make_bn(&c, H / stride, red);
        b->depth_expansion = carve(&c, ex * red, 2.0 / (double)(red + ex));
        b->norm_expansion = make_bn(&c, H / stride, ex);
        if (inc != ex) { /* resnet.cu:770-793: 3x3 stride-2 projection when the block strides, else 1x1 */
            if (stride == 2) b->projection = carve(&c, 9 * inc * ex, 2.0 / (9.0 * (inc + ex)));
            else b->projection = carve(&c, inc * ex, 2.0 / (double)(inc + ex));
            b->norm_projection = make_bn(&c, H / stride, ex);
        }
        p->conv_blocks[i] = b;
        if (stride == 2) H /= 2;
        inc = ex;
    }
    p->fully_connected = carve(&c, ex * d->output, 1e-4); /* resnet.cu:938 */
    p->locations = c.loc; p->sizes = c.sizes; p->n_locations = c.n;

    MiGlobal *g = mi_global();
    mid_memset(c.base, 0, arena_floats * sizeof(float), g->compute);
    if (gen) {
        /* tensor i draws from the stream at offset sum(sizes[0..i)) -- same rule as tests/synth.make_params */
        size_t off = 0, maxsz = 0;
        for (int i = 0; i < c.n; i++) if ((size_t)c.sizes[i] > maxsz) maxsz = c.sizes[i];
        float *host = (float *)malloc(maxsz * sizeof(float));
        for (int i = 0; i < c.n; i++) {
            const size_t sz = c.sizes[i];
            if (c.var[i] > 0) mi_synth_normal(host, sz, gen->seed, gen->counter + off, c.var[i]);
            else if (c.var[i] < 0) for (size_t j = 0; j < sz; j++) host[j] = 1.0f;
            if (c.var[i] != 0) { mid_memcpy_h2d(c.loc[i], host, sz * sizeof(float), g->compute); mid_stream_sync(g->compute); }
            off += sz;
        }
        gen->counter += off;
        free(host);
    }
    mid_stream_sync(g->compute);
    free(c.var);
    return p;
}
static void free_params_host(Params *p, const Dims *d) {
    if (!p) return;
    free(p->norm_init_conv);
    for (int i = 0; i < d->n_conv_blocks; i++) {
        ConvBlock *b = p->conv_blocks[i];
        free(b->norm_depth_reduction); free(b->norm_spatial); free(b->norm_expansion); free(b->norm_projection); free(b);
    }
    free(p->conv_blocks); free(p->locations); free(p->sizes); free(p);
}

/* resnet.cu:951-957 */
ResNet *init_resnet(Dims *dims, MiRng *gen) {
    ResNet *m = (ResNet *)malloc(sizeof(ResNet));
    MiRng tmp = {1234, 0};
    m->dims = dims;
    m->params = build_params(dims, gen ? gen : &tmp, NULL);
    return m;
}

/* ---------------------------------------------------------------------------------------------- */
static Cache_BatchNorm *make_cache(MiCtx *c, int input_size, int feature_size, int with_stats) {
    Cache_BatchNorm *k = (Cache_BatchNorm *)calloc(1, sizeof(Cache_BatchNorm));
    k->input_size = input_size; k->feature_size = feature_size;
    if (with_stats) {
        k->means = (float *)mi_ctx_alloc(c, sizeof(float) * feature_size);
        k->vars = (float *)mi_ctx_alloc(c, sizeof(float) * feature_size);
    }
    return k;
}
static float *falloc(MiCtx *c, size_t n) { return (float *)mi_ctx_alloc(c, n * sizeof(float)); }
/* an activation tensor of n elements in the trainer's storage type (the struct fields stay `float *`, resnet.h).  bf16
 * tensors get MI_GUARD bytes of slack on both sides: the bf16 convolution reads a tap-shifted operand with 16-byte loads,
 * which reach up to (W + 1) elements before the first / past the last pixel of a tensor (those lanes are masked to zero) */
static float *aalloc(MiCtx *c, size_t n) {
    if (c->dtype != MID_BF16) return (float *)mi_ctx_alloc(c, n * 4);
    return (float *)((char *)mi_ctx_alloc(c, n * 2 + 2 * MI_GUARD) + MI_GUARD);
}

/* init_activations, resnet.cu:1057-1113.
 * mode 0: the forward tree.  What it keeps follows c->policy (FAST: raw + BN(+ReLU) output per convolution;
 *         RECOMPUTE_BN: the BN(+ReLU) tensors are two shared scratch buffers, re-derived in backward).
 * mode 1: the derivative tree over six rolling buffers (resnet_cudnn_lowmem.cu:2152-2170).
 * mode 2: the derivative tree with a buffer of its own per tensor, the reference's full mirror (resnet.cu:1151):
 *         FULL policy, so that a dump of activation_derivs/ holds what its file names say. */
static Activations *build_activations(MiCtx *c, const Dims *d, ConvBlock **blocks, int N, float **pool, int mode) {
    Activations *a = (Activations *)calloc(1, sizeof(Activations));
    const int f = d->init_conv_filters, Hs = d->input / d->init_conv_stride, Hp = Hs / d->init_maxpool_stride;
    const size_t stem = (size_t)N * f * Hs * Hs, pl = (size_t)N * f * Hp * Hp;
    const int rc = mode == 0 && c->policy == MI_STORE_RECOMPUTE_BN;
    a->n_conv_blocks = d->n_conv_blocks;
    a->activation_conv_blocks = (Activation_ConvBlock **)calloc(d->n_conv_blocks > 0 ? d->n_conv_blocks : 1, sizeof(void *));
    if (mode == 0) {
        a->init_conv_applied = falloc(c, stem); /* the 7x7 stem keeps fp32 tensors in every storage type */
        a->norm_init_conv = make_cache(c, (int)stem, f, 1);
        a->init_conv_activated = rc ? c->rc_buf[0] : aalloc(c, stem);
        a->max_inds = (int *)mi_ctx_alloc(c, pl * sizeof(int));
        a->init_convblock_input = aalloc(c, pl);
    } else if (mode == 2) {
        a->init_conv_applied = falloc(c, stem);
        a->norm_init_conv = make_cache(c, (int)stem, f, 0);
        a->init_conv_activated = aalloc(c, stem);
        a->init_convblock_input = aalloc(c, pl);
    } else {
        a->init_conv_applied = c->dtype == MID_BF16 ? c->stem_dx : pool[3]; /* B */
        a->norm_init_conv = make_cache(c, (int)stem, f, 0);
        a->init_conv_activated = pool[2]; /* A */
        a->init_convblock_input = pool[0]; /* U0 */
    }
    for (int i = 0; i < d->n_conv_blocks; i++) {
        const ConvBlock *b = blocks[i];
        Activation_ConvBlock *k = (Activation_ConvBlock *)calloc(1, sizeof(Activation_ConvBlock));
        k->incoming_filters = b->incoming_filters; k->incoming_spatial_dim = b->incoming_spatial_dim;
        k->reduced_depth = b->reduced_depth; k->expanded_depth = b->expanded_depth; k->stride = b->stride;
        const int H = b->incoming_spatial_dim, Ho = H / b->stride;
        const size_t rsz = (size_t)N * b->reduced_depth * H * H, ssz = (size_t)N * b->reduced_depth * Ho * Ho,
                     osz = (size_t)N * b->expanded_depth * Ho * Ho;
        const int has_proj = b->projection != NULL;
        k->norm_post_reduced = make_cache(c, (int)rsz, b->reduced_depth, mode == 0);
        k->norm_post_spatial = make_cache(c, (int)ssz, b->reduced_depth, mode == 0);
        k->norm_post_expanded = make_cache(c, (int)osz, b->expanded_depth, mode == 0);
        if (has_proj) k->norm_post_projection = make_cache(c, (int)osz, b->expanded_depth, mode == 0);
        if (mode == 0) {
            k->post_reduced = aalloc(c, rsz); k->post_reduced_activated = rc ? c->rc_buf[0] : aalloc(c, rsz);
            k->post_spatial = aalloc(c, ssz); k->post_spatial_activated = rc ? c->rc_buf[1] : aalloc(c, ssz);
            k->post_expanded = aalloc(c, osz);
            if (has_proj) { k->transformed_residual = aalloc(c, osz); k->post_projection_norm_vals = rc ? c->rc_buf[0] : aalloc(c, osz); }
            k->output_activated = aalloc(c, osz);
            k->output = k->output_activated; /* pre-ReLU sum is not kept unless full-store */
        } else if (mode == 2) {
            k->output_activated = aalloc(c, osz); k->output = aalloc(c, osz);
            k->transformed_residual = has_proj ? aalloc(c, osz) : NULL;
            k->post_expanded = aalloc(c, osz);
            k->post_spatial_activated = aalloc(c, ssz); k->post_spatial = aalloc(c, ssz);
            k->post_reduced_activated = aalloc(c, rsz); k->post_reduced = aalloc(c, rsz);
        } else {
            /* lifetimes inside one block's backward (see backwards_pass): A,B,C,D scratch + alternating U */
            k->output_activated = pool[(i + 1) & 1];
            k->output = pool[5];               /* D: ReLU'-gated upstream, identity blocks only */
            k->transformed_residual = has_proj ? pool[2] : NULL; /* A */
            k->post_expanded = pool[3];         /* B */
            k->post_spatial_activated = pool[4]; /* C */
            k->post_spatial = pool[2];          /* A */
            k->post_reduced_activated = pool[3]; /* B */
            k->post_reduced = pool[4];          /* C */
        }
        a->activation_conv_blocks[i] = k;
    }
    if (mode == 0) {
        a->final_conv_output_pooled = falloc(c, (size_t)N * d->final_depth);
        a->linear_output = falloc(c, (size_t)N * d->output);
    } else a->final_conv_output_pooled = falloc(c, (size_t)N * d->final_depth);
    return a;
}

static size_t max_tensor_elems(const Dims *d, ConvBlock **blocks, int N) {
    const int f = d->init_conv_filters, Hs = d->input / d->init_conv_stride;
    size_t m = (size_t)N * f * Hs * Hs;
    for (int i = 0; i < d->n_conv_blocks; i++) {
        const ConvBlock *b = blocks[i];
        const int H = b->incoming_spatial_dim, Ho = H / b->stride;
        size_t v[3] = {(size_t)N * b->reduced_depth * H * H, (size_t)N * b->expanded_depth * Ho * Ho,
                       (size_t)N * b->incoming_filters * H * H};
        for (int j = 0; j < 3; j++) if (v[j] > m) m = v[j];
    }
    return m;
}

static void size_workspaces(MiCtx *c, const Dims *d, ConvBlock **blocks, int N) {
    size_t wt = 0, part = 0;
    int maxc = d->init_conv_filters;
#define LAYER(C_, H_, K_, k_, s_)                                                             \
    do {                                                                                      \
        size_t a_ = mid_conv_ws_wt_floats(C_, K_, k_), b_ = mid_conv_ws_part_floats(N, C_, H_, K_, k_, s_); \
        if (c->dtype == MID_BF16 && (k_) <= 3) { size_t e_ = mid_bf16_part_floats(N, C_, H_, K_, k_, s_); if (e_ > b_) b_ = e_; } \
        if (c->dtype == MID_BF16 && (k_) == 3 && mid_cl_wgrad_supported(N, C_, H_, K_, s_)) { size_t e_ = mid_cl_wgrad_part_floats(N, C_, H_, K_, s_); if (e_ > b_) b_ = e_; } \
        if (c->dtype == MID_BF16 && (k_) == 3 && mid_cl_wgrad2_supported(N, C_, H_, K_, s_)) { size_t e_ = mid_cl_wgrad2_part_floats(N, C_, H_, K_, s_); if (e_ > b_) b_ = e_; } \
        if (a_ > wt) wt = a_;                                                                 \
        if (b_ > part) part = b_;                                                             \
        if ((K_) > maxc) maxc = (K_);                                                         \
    } while (0)
    LAYER(3, d->input, d->init_conv_filters, d->init_kernel_dim, d->init_conv_stride);
    for (int i = 0; i < d->n_conv_blocks; i++) {
        const ConvBlock *b = blocks[i];
        const int H = b->incoming_spatial_dim;
        LAYER(b->incoming_filters, H, b->reduced_depth, 1, 1);
        LAYER(b->reduced_depth, H, b->reduced_depth, 3, b->stride);
        LAYER(b->reduced_depth, H / b->stride, b->expanded_depth, 1, 1);
        if (b->projection) LAYER(b->incoming_filters, H, b->expanded_depth, b->stride == 2 ? 3 : 1, b->stride);
    }
#undef LAYER
    c->ws.s2d = NULL; c->ws.s2d_bytes = 0; c->ws.s2d_valid = 0;
    free(c->par); c->par = NULL;
    if (c->dtype == MID_BF16) {
        /* stride-2 layers read their input as four parity planes per channel (kernels_igemm_bf16.hip): one copy per such layer,
         * written by the forward pass and read again by the weight gradient (1.1 GB in all at N = 256) */
        c->par = (MiParity *)calloc((size_t)(d->n_conv_blocks > 0 ? d->n_conv_blocks : 1), sizeof(MiParity));
        for (int i = 0; i < d->n_conv_blocks; i++) {
            const ConvBlock *b = blocks[i];
            if (b->stride != 2) {
                /* stride-1 3x3: forward and weight gradient on the channel-last plane the reduction BN writes beside its NCHW output
                 * (RESNET_MI_BF16_CL_S1=0: the NCHW kernels) */
                const int Hs = b->incoming_spatial_dim;
                if (!(getenv("RESNET_MI_BF16_CL_S1") && atoi(getenv("RESNET_MI_BF16_CL_S1")) == 0) && mid_cl_supported(0, N, b->reduced_depth, Hs, b->reduced_depth, 1)) {
                    const size_t by = mid_cl_operand_bytes(0, N, b->reduced_depth, Hs, b->reduced_depth, 1);
                    c->par[i].cl_s1 = mi_ctx_alloc(c, by);
                    mid_memset(c->par[i].cl_s1, 0, by, G.compute);
                    if (!(getenv("RESNET_MI_BF16_CL_S1_DGRAD") && atoi(getenv("RESNET_MI_BF16_CL_S1_DGRAD")) == 0) &&
                        mid_cl_supported(1, N, b->reduced_depth, Hs, b->reduced_depth, 1)) {
                        const size_t dyb = mid_cl_operand_bytes(1, N, b->reduced_depth, Hs, b->reduced_depth, 1);
                        c->par[i].dy1 = mi_ctx_alloc(c, dyb);
                        mid_memset(c->par[i].dy1, 0, dyb, G.compute);
                    }
                }
                continue;
            }
            const size_t H = b->incoming_spatial_dim, e1 = (size_t)N * b->reduced_depth * H * H, e2 = (size_t)N * b->incoming_filters * H * H;
            /* forward and weight gradient on channel-last parity planes (RESNET_MI_BF16_CL_S2=0: the NCHW kernels and their planes) */
            int need_sp = 1, need_pr = b->projection != NULL;
            if (!(getenv("RESNET_MI_BF16_CL_S2") && atoi(getenv("RESNET_MI_BF16_CL_S2")) == 0)) {
                if (mid_cl_supported(0, N, b->reduced_depth, (int)H, b->reduced_depth, 2)) {
                    const size_t by = mid_cl_operand_bytes(0, N, b->reduced_depth, (int)H, b->reduced_depth, 2);
                    c->par[i].cl_spatial = mi_ctx_alloc(c, by);
                    mid_memset(c->par[i].cl_spatial, 0, by, G.compute); /* the halo stays zero: the re-layout writes the interior only */
                    need_sp = !mid_cl_wgrad_supported(N, b->reduced_depth, (int)H, b->reduced_depth, 2) && !mid_cl_wgrad2_supported(N, b->reduced_depth, (int)H, b->reduced_depth, 2);
                }
                if (b->projection && mid_cl_supported(0, N, b->incoming_filters, (int)H, b->expanded_depth, 2)) {
                    const size_t by = mid_cl_operand_bytes(0, N, b->incoming_filters, (int)H, b->expanded_depth, 2);
                    c->par[i].cl_proj = mi_ctx_alloc(c, by);
                    mid_memset(c->par[i].cl_proj, 0, by, G.compute);
                    need_pr = !mid_cl_wgrad_supported(N, b->incoming_filters, (int)H, b->expanded_depth, 2) && !mid_cl_wgrad2_supported(N, b->incoming_filters, (int)H, b->expanded_depth, 2);
                }
            }
            if (need_sp) { c->par[i].spatial_bytes = e1 * 2; c->par[i].spatial = (char *)mi_ctx_alloc(c, e1 * 2 + 2 * MI_GUARD) + MI_GUARD; }
            if (need_pr) { c->par[i].proj_bytes = e2 * 2; c->par[i].proj = (char *)mi_ctx_alloc(c, e2 * 2 + 2 * MI_GUARD) + MI_GUARD; }
            /* the stride-2 dgrads on channel-last dY (RESNET_MI_BF16_CL_DGRAD2=0: the NCHW kernel's four parity classes) */
            if (!(getenv("RESNET_MI_BF16_CL_DGRAD2") && atoi(getenv("RESNET_MI_BF16_CL_DGRAD2")) == 0)) {
                if (mid_cl_dgrad2_supported(N, b->reduced_depth, (int)H, b->reduced_depth)) {
                    const size_t by = mid_cl_dgrad2_operand_bytes(N, b->reduced_depth, (int)H / 2);
                    c->par[i].dye_spatial = mi_ctx_alloc(c, by);
                    mid_memset(c->par[i].dye_spatial, 0, by, G.compute);
                }
                if (b->projection && mid_cl_dgrad2_supported(N, b->incoming_filters, (int)H, b->expanded_depth)) {
                    const size_t by = mid_cl_dgrad2_operand_bytes(N, b->expanded_depth, (int)H / 2);
                    c->par[i].dye_proj = mi_ctx_alloc(c, by);
                    mid_memset(c->par[i].dye_proj, 0, by, G.compute);
                }
            }
        }
    }
    c->ws.wt_floats = wt; c->ws.part_floats = part;
    c->ws.wt = wt ? falloc(c, wt) : NULL;
    c->ws.part = part ? falloc(c, part) : NULL;
    c->bn_ws = falloc(c, mid_bn_ws_floats(maxc));
    {
        size_t pf = 0; /* largest statistics-partials table of any conv + BN unit */
        for (int i = 0; i < d->n_conv_blocks; i++) {
            const ConvBlock *b = blocks[i];
            const int H = b->incoming_spatial_dim, Ho = H / b->stride;
            size_t a = mid_bn_parts_floats(N, b->reduced_depth, H), e = mid_bn_parts_floats(N, b->expanded_depth, Ho);
            if (a > pf) pf = a;
            if (e > pf) pf = e;
        }
        c->bn_parts.floats = pf;
        c->bn_parts.buf = pf ? falloc(c, pf) : NULL;
        c->bn_parts.nparts = 0;
    }
    { /* table of the convolutions whose weights are re-laid once per forward pass (mid_conv_prelayout_all) */
        const int maxn = 4 * d->n_conv_blocks + 1;
        const int prelayout = getenv("RESNET_MI_PRELAYOUT") ? atoi(getenv("RESNET_MI_PRELAYOUT")) : 1; /* 0: each conv re-lays its own */
        free(c->wt_tab);
        c->wt_tab = (mid_wt_entry *)calloc((size_t)maxn, sizeof(mid_wt_entry));
        c->wt_n = 0; c->wt_tiles = 0;
#define WT_LAYER(w_, C_, H_, K_, k_, s_)                                                       \
        do {                                                                                       \
            int nf_ = 0, nd_ = 0;                                                                  \
            if ((w_) && c->dtype == MID_BF16) nf_ = nd_ = 1; /* bf16 k-step tiles, forward and dgrad forms */ \
            else if ((w_) && prelayout) mid_conv_prelayout_needs(N, C_, H_, K_, k_, s_, &nf_, &nd_); \
            if (nf_ || nd_) {                                                                      \
                mid_wt_entry *e_ = &c->wt_tab[c->wt_n++];                                          \
                const size_t n_ = ((size_t)(k_) * (k_) * (C_) * (K_)) / (c->dtype == MID_BF16 ? 2 : 1); \
                e_->w = (w_); e_->K = (K_); e_->C = (C_); e_->T = (k_) * (k_);                     \
                e_->fwd = nf_ ? falloc(c, n_) : NULL; e_->dgrad = nd_ ? falloc(c, n_) : NULL;      \
                e_->tile0 = c->wt_tiles; c->wt_tiles += ((C_) / 32) * ((K_) / 32);                 \
            }                                                                                      \
        } while (0)
        for (int i = 0; i < d->n_conv_blocks; i++) {
            const ConvBlock *b = blocks[i];
            const int H = b->incoming_spatial_dim;
            WT_LAYER(b->depth_reduction, b->incoming_filters, H, b->reduced_depth, 1, 1);
            WT_LAYER(b->spatial, b->reduced_depth, H, b->reduced_depth, 3, b->stride);
            WT_LAYER(b->depth_expansion, b->reduced_depth, H / b->stride, b->expanded_depth, 1, 1);
            WT_LAYER(b->projection, b->incoming_filters, H, b->expanded_depth, b->stride == 2 ? 3 : 1, b->stride);
        }
#undef WT_LAYER
        c->wt_tab_dev = NULL; c->wt_tile_entry_dev = NULL;
        if (c->wt_n) {
            int *te = (int *)malloc((size_t)c->wt_tiles * sizeof(int));
            for (int e = 0; e < c->wt_n; e++) {
                const int nt = (c->wt_tab[e].C / 32) * (c->wt_tab[e].K / 32);
                for (int q = 0; q < nt; q++) te[c->wt_tab[e].tile0 + q] = e;
            }
            c->wt_tab_dev = (mid_wt_entry *)mi_ctx_alloc(c, (size_t)c->wt_n * sizeof(mid_wt_entry));
            c->wt_tile_entry_dev = (int *)mi_ctx_alloc(c, (size_t)c->wt_tiles * sizeof(int));
            mid_memcpy_h2d(c->wt_tab_dev, c->wt_tab, (size_t)c->wt_n * sizeof(mid_wt_entry), G.compute);
            mid_memcpy_h2d(c->wt_tile_entry_dev, te, (size_t)c->wt_tiles * sizeof(int), G.compute);
            mid_stream_sync(G.compute);
            free(te);
        }
    }
}

static MiCtx *ctx_of(Train_ResNet *t) { return (MiCtx *)t->backend_ctx; }
static void free_activations_host(Activations *a);
static void add_full_store_extras(Train_ResNet *t);
/* everything whose size or layout depends on the storage type / store policy: activation trees, derivative buffers,
 * workspaces, re-laid weight tables.  Called by init_trainer and again by mi_trainer_set_dtype / _set_store_policy. */
static void build_buffers(Train_ResNet *t) {
    MiCtx *c = (MiCtx *)t->backend_ctx;
    Dims *d = t->model->dims;
    ConvBlock **blocks = t->model->params->conv_blocks;
    const int N = t->batch_size;
    const size_t maxe = max_tensor_elems(d, blocks, N);
    c->act_bytes = 0;
    c->rc_buf[0] = c->rc_buf[1] = NULL;
    if (c->policy == MI_STORE_RECOMPUTE_BN) { c->rc_buf[0] = aalloc(c, maxe); c->rc_buf[1] = aalloc(c, maxe); }
    c->counting_act = 1;
    t->forward_buffer->activations = build_activations(c, d, blocks, N, NULL, 0);
    c->counting_act = 0;
    c->full_store = 0;
    if (c->policy == MI_STORE_FULL) add_full_store_extras(t);
    const int f = d->init_conv_filters, Hs = d->input / d->init_conv_stride;
    c->stem_dx = c->dtype == MID_BF16 ? falloc(c, (size_t)N * f * Hs * Hs) : NULL;
    c->stem_xp = NULL; c->stem_scratch = NULL; c->stem_xp_bytes = 0; c->stem_scratch_floats = 0; c->stem_bf16 = 0;
    if (c->dtype == MID_F32 && mid_igemm_mode() > 0 && mid_stem_bf16_supported(3, d->input, f, d->init_kernel_dim, d->init_conv_stride) &&
        !(getenv("RESNET_MI_STEM_MFMA") && atoi(getenv("RESNET_MI_STEM_MFMA")) == 0)) {
        /* fp32 storage: the stem in exact fp32 on the matrix cores (kernels_stem_bf16.hip, st32_*) */
        c->stem_xp_bytes = mid_stem_f32_xp_bytes(N, d->input);
        c->stem_scratch_floats = mid_stem_bf16_part_floats(N, d->input);
        c->stem_xp = falloc(c, (c->stem_xp_bytes + 3) / 4);
        c->stem_scratch = falloc(c, c->stem_scratch_floats);
    }
    if (c->dtype == MID_BF16 && mid_stem_bf16_supported(3, d->input, f, d->init_kernel_dim, d->init_conv_stride) &&
        !(getenv("RESNET_MI_BF16_STEM") && atoi(getenv("RESNET_MI_BF16_STEM")) == 0)) {
        /* the stem on the bf16 matrix cores (image and weights rounded to bf16 like every other convolution of this mode) */
        c->stem_xp_bytes = mid_stem_bf16_xp_bytes(N, d->input);
        c->stem_scratch_floats = mid_stem_bf16_part_floats(N, d->input);
        c->stem_xp = aalloc(c, (c->stem_xp_bytes + 1) / 2);                 /* (aalloc counts 2-byte elements in bf16 mode) */
        c->stem_scratch = falloc(c, c->stem_scratch_floats);
        /* its output and that tensor's gradient are stored as bf16 like every other convolution's (they stay in their fp32-sized buffers):
         * 822 MB tensors at N = 256 that the stem BN reads twice forward and three times backward */
        c->stem_bf16 = !(getenv("RESNET_MI_BF16_STEM_TENSORS") && !strcmp(getenv("RESNET_MI_BF16_STEM_TENSORS"), "f32"));
    }
    float *pool[6];
    for (int i = 0; i < 6; i++) pool[i] = aalloc(c, maxe);
    for (int i = 0; i < 6; i++) c->dpool[i] = pool[i];
    for (int i = 0; i < MI_RING; i++) { c->ring_buf[i] = i < 4 ? pool[2 + i] : aalloc(c, maxe); c->ring_busy[i] = 0; }
    c->ring_next = 0;
    t->backprop_buffer->activation_derivs = build_activations(c, d, blocks, N, pool, c->policy == MI_STORE_FULL ? 2 : 1);
    /* ring mode re-points derivative tensors to equal-sized slots: FAST policy and fp32 only (the bf16 path keeps the stem's
     * fp32 gradient in a buffer of its own) */
    if ((c->policy != MI_STORE_FAST || c->dtype != MID_F32) && c->overlap_wgrad > 1) c->overlap_wgrad = 1;
    /* bf16: the weight gradients are no longer bound by the matrix pipe but by memory, like the batch norm they would run
     * next to -- measured 6028 img/s serial against 5973 overlapped; an explicit RESNET_MI_OVERLAP still wins */
    if (c->dtype == MID_BF16 && !getenv("RESNET_MI_OVERLAP") && !c->overlap_set) c->overlap_wgrad = 0;
    size_workspaces(c, d, blocks, N);
    mid_stream_sync(G.compute);
}
static void drop_buffers(Train_ResNet *t) {
    MiCtx *c = (MiCtx *)t->backend_ctx;
    mid_device_sync();
    ctx_free_from(c, c->n_persist);
    free_activations_host(t->forward_buffer->activations);
    free_activations_host(t->backprop_buffer->activation_derivs);
    t->forward_buffer->activations = NULL; t->backprop_buffer->activation_derivs = NULL;
    c->wt_n = 0; c->wt_tiles = 0; c->wt_tab_dev = NULL; c->wt_tile_entry_dev = NULL;
    c->wgrad_pending = 0;
}

/* resnet.cu:1157-1194 */
Train_ResNet *init_trainer(ResNet *model, Batch *cur_batch, int batch_size, float learning_rate, float weight_decay,
                           float mean_decay, float var_decay, float eps, int n_epochs, const char *dump_dir) {
    Train_ResNet *t = (Train_ResNet *)calloc(1, sizeof(Train_ResNet));
    MiCtx *c = (MiCtx *)calloc(1, sizeof(MiCtx));
    Dims *d = model->dims;
    mi_global();
    t->backend_ctx = c;
    t->model = model; t->cur_batch = cur_batch; t->batch_size = batch_size;
    c->dump_every = 1000; /* resnet.cu:2947 */
    c->input_reset = 1;   /* resnet.cu:2981-2982 */
    c->world = 1; c->bucket_bytes = (size_t)32 << 20;
    c->dtype = MID_F32; c->policy = MI_STORE_FAST;
    c->fz_enable = !(getenv("RESNET_MI_BF16_BNFUSE_BWD") && atoi(getenv("RESNET_MI_BF16_BNFUSE_BWD")) == 0);
    /* fp32 storage: which dgrads do it (bits: 1 expansion dgrad -> spatial BN', 2 spatial dgrad -> reduction BN', 4 reduction dgrad -> the
     * expansion BN' of the identity block below); RESNET_MI_F32_BNFUSE_BWD overrides.  Measured at batch 256 (same box, ms/step):
     * none 108.3-109.5, site 4 alone 108.3-108.9, site 1 alone 109.4-110.2, sites 1+2 111.6-112.8, all 111.7-112.2 -- the fp32 epilogue
     * keeps lane = column, so the fused form reads x / mask / addend with 4-byte accesses (four times the memory instructions of the
     * bf16 kernel's row-major drain) and pays for it wherever the separate reduction pass was only 2 tensors; site 4 replaces a
     * 4-tensor pass and breaks even, so it is the default */
    c->fz_bf16 = getenv("RESNET_MI_BF16_BNFUSE_SITES") ? atoi(getenv("RESNET_MI_BF16_BNFUSE_SITES")) : 7;
    c->cl_pre = !(getenv("RESNET_MI_BF16_CL_PRE") && atoi(getenv("RESNET_MI_BF16_CL_PRE")) == 0);
    c->cl_wgrad2 = !(getenv("RESNET_MI_BF16_CL_WGRAD2") && atoi(getenv("RESNET_MI_BF16_CL_WGRAD2")) == 0);
    c->fz_f32 = mid_igemm_mode() >= 2 ? (getenv("RESNET_MI_F32_BNFUSE_BWD") ? atoi(getenv("RESNET_MI_F32_BNFUSE_BWD")) : 4) : 0;
    c->overlap_wgrad = getenv("RESNET_MI_OVERLAP") ? atoi(getenv("RESNET_MI_OVERLAP")) : 1;
    c->ev_bn_done = mid_event_create(); c->ev_wgrad_done = mid_event_create();

    /* persistent device state: survives a change of storage type / store policy */
    Forward_Buffer *fb = (Forward_Buffer *)calloc(1, sizeof(Forward_Buffer));
    fb->pred = falloc(c, (size_t)batch_size * d->output); /* reference over-allocates N^2*output (:1126, hazard h4) */
    fb->pred_cpu = (float *)mid_malloc_host((size_t)batch_size * d->output * sizeof(float));
    t->forward_buffer = fb;
    Backprop_Buffer *bb = (Backprop_Buffer *)calloc(1, sizeof(Backprop_Buffer));
    bb->output_layer_deriv = falloc(c, (size_t)batch_size * d->output);
    bb->param_derivs = build_params(d, NULL, c);
    bb->prev_means = build_params(d, NULL, c);
    bb->prev_vars = build_params(d, NULL, c);
    t->backprop_buffer = bb;
    c->arena_floats = mi_params_arena_floats(model->params);
    c->g_arena = mi_params_arena_base(bb->param_derivs);
    c->m_arena = mi_params_arena_base(bb->prev_means);
    c->v_arena = mi_params_arena_base(bb->prev_vars);
    { /* arena offsets of the tensors, for the Adam kernel's "which location" report */
        const Params *mp = model->params;
        size_t *lo = (size_t *)malloc(sizeof(size_t) * (size_t)(mp->n_locations + 1));
        for (int i = 0; i < mp->n_locations; i++) lo[i] = (size_t)(mp->locations[i] - mp->locations[0]);
        lo[mp->n_locations] = c->arena_floats;
        c->loc_off_dev = (size_t *)mi_ctx_alloc(c, sizeof(size_t) * (size_t)(mp->n_locations + 1));
        mid_memcpy_h2d(c->loc_off_dev, lo, sizeof(size_t) * (size_t)(mp->n_locations + 1), G.compute);
        mid_stream_sync(G.compute);
        free(lo);
        c->n_loc = mp->n_locations;
    }
    c->ev_nan = mid_event_create();
    c->nan_location = -1;
    c->next_live = g_live; g_live = c;
    c->nan_flag_dev = (int *)mi_ctx_alloc(c, sizeof(int));
    c->nan_flag_host = (int *)mid_malloc_host(sizeof(int));
    *c->nan_flag_host = 0;
    mid_memset(c->nan_flag_dev, 0, sizeof(int), G.compute);
    c->n_persist = c->n_allocs;
    for (int i = 0; i < MI_RING; i++) c->ring_ev[i] = mid_event_create();
    c->ev_grads = mid_event_create(); c->ev_reduced = mid_event_create();
    for (int i = 0; i < 6; i++) c->ev_t[i] = mid_event_create();
    for (int i = 0; i < MI_MAX_BUCKETS; i++) c->bk_ev[i] = mid_event_create();
    c->fuse_bn_stats = getenv("RESNET_MI_BNFUSE") ? atoi(getenv("RESNET_MI_BNFUSE")) : 1;

    build_buffers(t);

    t->learning_rate = learning_rate; t->weight_decay = weight_decay;
    t->base_mean_decay = mean_decay; t->base_var_decay = var_decay;
    t->cur_mean_decay = 1; t->cur_var_decay = 1;
    t->eps = eps; t->n_epochs = n_epochs; t->cur_dump_id = -1; t->cur_epoch = 0;
    t->loss_per_epoch = (float *)calloc(n_epochs > 0 ? n_epochs : 1, sizeof(float));
    t->accuracy_per_epoch = (float *)calloc(n_epochs > 0 ? n_epochs : 1, sizeof(float));
    t->init_loaded = 0;
    t->dump_dir = dump_dir;
    mid_stream_sync(G.compute);
    return t;
}
Train_ResNet *init_trainer_cudnn_abi(ResNet *model, Batch *cur_batch, int batch_size, float learning_rate,
                                     float weight_decay, float mean_decay, float var_decay, float eps, int n_epochs,
                                     void *handle, const char *dump_dir) {
    (void)handle;
    return init_trainer(model, cur_batch, batch_size, learning_rate, weight_decay, mean_decay, var_decay, eps, n_epochs, dump_dir);
}

/* FULL policy: x-hat, BN output and pre-ReLU sums as the reference stores them (dump parity; fp32 only) */
static void add_full_store_extras(Train_ResNet *t) {
    MiCtx *c = ctx_of(t);
    Activations *a = t->forward_buffer->activations;
    c->counting_act = 1;
#define EXTRA(cache) do { (cache)->normalized_temp = falloc(c, (cache)->input_size); (cache)->normalized = falloc(c, (cache)->input_size); } while (0)
    EXTRA(a->norm_init_conv);
    for (int i = 0; i < a->n_conv_blocks; i++) {
        Activation_ConvBlock *k = a->activation_conv_blocks[i];
        EXTRA(k->norm_post_reduced); EXTRA(k->norm_post_spatial); EXTRA(k->norm_post_expanded);
        if (k->norm_post_projection) EXTRA(k->norm_post_projection);
        k->post_expanded_norm_vals = falloc(c, k->norm_post_expanded->input_size);
        k->output = falloc(c, k->norm_post_expanded->input_size);
    }
#undef EXTRA
    c->counting_act = 0;
    c->full_store = 1;
}
/* every convolution of the bottleneck blocks must tile for the bf16 kernels (the stem stays on the fp32 path) */
static int bf16_net_supported(const Train_ResNet *t, char *why, size_t whylen) {
    const Dims *d = t->model->dims;
    ConvBlock **blocks = t->model->params->conv_blocks;
    const int N = t->batch_size;
    for (int i = 0; i < d->n_conv_blocks; i++) {
        const ConvBlock *b = blocks[i];
        const int H = b->incoming_spatial_dim;
        const int L[4][5] = {{b->incoming_filters, H, b->reduced_depth, 1, 1},
                             {b->reduced_depth, H, b->reduced_depth, 3, b->stride},
                             {b->reduced_depth, H / b->stride, b->expanded_depth, 1, 1},
                             {b->incoming_filters, H, b->expanded_depth, b->stride == 2 ? 3 : 1, b->stride}};
        for (int j = 0; j < (b->projection ? 4 : 3); j++)
            for (int op = 0; op < 3; op++)
                if (!mid_bf16_supported(op, N, L[j][0], L[j][1], L[j][2], L[j][3], L[j][4])) {
                    snprintf(why, whylen, "block %d conv %d (C=%d H=%d K=%d k=%d s=%d) op %d does not tile for the bf16 kernels", i, j,
                             L[j][0], L[j][1], L[j][2], L[j][3], L[j][4], op);
                    return 0;
                }
    }
    return 1;
}
void mi_record_host_error(const char *what, const char *detail);
int mi_trainer_set_store_policy(Train_ResNet *t, int policy) {
    MiCtx *c = ctx_of(t);
    if (policy < MI_STORE_FAST || policy > MI_STORE_FULL) { mi_record_host_error("mi_trainer_set_store_policy", "unknown policy"); return -1; }
    if (policy == MI_STORE_FULL && c->dtype != MID_F32) { mi_record_host_error("mi_trainer_set_store_policy", "the FULL policy (x-hat / BN-out / pre-ReLU sums) exists in fp32 only"); return -1; }
    if (policy == c->policy) return 0;
    drop_buffers(t);
    c->policy = policy;
    build_buffers(t);
    return 0;
}
void mi_trainer_set_full_store(Train_ResNet *t, int on) { (void)mi_trainer_set_store_policy(t, on ? MI_STORE_FULL : MI_STORE_FAST); }
int mi_trainer_set_dtype(Train_ResNet *t, int dtype) {
    MiCtx *c = ctx_of(t);
    if (dtype != MID_F32 && dtype != MID_BF16) { mi_record_host_error("mi_trainer_set_dtype", "unknown dtype"); return -1; }
    if (dtype == c->dtype) return 0;
    if (dtype == MID_BF16) {
        char why[256];
        if (c->policy == MI_STORE_FULL) { mi_record_host_error("mi_trainer_set_dtype", "the FULL store policy exists in fp32 only"); return -1; }
        if (!bf16_net_supported(t, why, sizeof why)) { mi_record_host_error("mi_trainer_set_dtype", why); return -1; }
    }
    drop_buffers(t);
    c->dtype = dtype;
    build_buffers(t);
    return 0;
}
int mi_trainer_get_dtype(const Train_ResNet *t) { return ((MiCtx *)t->backend_ctx)->dtype; }
size_t mi_trainer_activation_bytes(const Train_ResNet *t) { return ((MiCtx *)t->backend_ctx)->act_bytes; }
size_t mi_trainer_device_bytes(const Train_ResNet *t) {
    const MiCtx *c = (MiCtx *)t->backend_ctx;
    return c->dev_bytes + c->arena_floats * sizeof(float); /* + the parameter arena (owned by the model) */
}
void mi_trainer_set_dump_every(Train_ResNet *t, int every) { ctx_of(t)->dump_every = every; }
void mi_trainer_set_overlap(Train_ResNet *t, int mode) {
    MiCtx *c = ctx_of(t);
    mid_stream_sync(G.aux); mid_stream_sync(G.compute);
    c->overlap_wgrad = mode < 0 ? 0 : mode > 2 ? 2 : mode;
    c->overlap_set = 1;
    if ((c->policy != MI_STORE_FAST || c->dtype != MID_F32) && c->overlap_wgrad > 1) c->overlap_wgrad = 1; /* the ring re-points derivative tensors */
    c->wgrad_pending = 0;
    for (int i = 0; i < MI_RING; i++) c->ring_busy[i] = 0;
    if (c->overlap_wgrad != 2) { /* back to the fixed aliasing of build_activations */
        Activations *da = t->backprop_buffer->activation_derivs;
        float **pool = c->dpool;
        if (c->policy == MI_STORE_FULL) return; /* every derivative tensor has a buffer of its own */
        da->init_conv_applied = c->dtype == MID_BF16 ? c->stem_dx : pool[3]; da->init_conv_activated = pool[2];
        for (int i = 0; i < da->n_conv_blocks; i++) {
            Activation_ConvBlock *k = da->activation_conv_blocks[i];
            k->output = pool[5];
            if (k->transformed_residual) k->transformed_residual = pool[2];
            k->post_expanded = pool[3]; k->post_spatial_activated = pool[4]; k->post_spatial = pool[2];
            k->post_reduced_activated = pool[3]; k->post_reduced = pool[4];
        }
    }
}
void mi_trainer_set_input_reset(Train_ResNet *t, int on) { ctx_of(t)->input_reset = on; }
void mi_trainer_set_dump_root(Train_ResNet *t, const char *root) {
    MiCtx *c = ctx_of(t);
    free(c->dump_root);
    c->dump_root = root ? strdup(root) : NULL;
}
void mi_trainer_last_timings(Train_ResNet *t, float out_ms[5]) {
    MiCtx *c = ctx_of(t);
    mid_stream_sync(G.compute);
    out_ms[0] = 0;
    out_ms[1] = mid_event_elapsed_ms(c->ev_t[0], c->ev_t[1]);
    out_ms[2] = mid_event_elapsed_ms(c->ev_t[2], c->ev_t[3]);
    out_ms[3] = mid_event_elapsed_ms(c->ev_t[4], c->ev_t[5]);
    out_ms[4] = c->last_ms[4];
}

/* ---------------------------------------------------------------------------------------------- */
static const mid_wt_entry *wt_lookup(const MiCtx *c, const float *w) {
    for (int i = 0; i < c->wt_n; i++)
        if (c->wt_tab[i].w == w) return &c->wt_tab[i];
    return NULL;
}

/* every implicit-GEMM layer's weights in the layouts forward and dgrad want, one launch (they hold until the parameters
 * change: update_parameters, or a host write -- mi_copy_to_device / overwrite_model_params set the dirty flag) */
static void relayout_weights(MiCtx *c) {
    if (c->dtype == MID_BF16) ck(mid_conv_prelayout_all_bf16(G.compute, c->wt_tab_dev, c->wt_tile_entry_dev, c->wt_tiles), "weight re-layout (bf16)");
    else ck(mid_conv_prelayout_all(G.compute, c->wt_tab_dev, c->wt_tile_entry_dev, c->wt_tiles), "weight re-layout");
    c->params_dirty = 0;
    c->host_epoch_seen = g_host_write_epoch;
}
static int weights_stale(const MiCtx *c) { return c->params_dirty || c->host_epoch_seen != g_host_write_epoch; }
/* the NaN / Inf flag update_parameters queued a copy of; valid after any later synchronisation of the compute stream
 * (check_errors, resnet.cu:2879-2907: dump id 99999999 and exit; offending gradients are still in the arena, the Adam
 * kernel clears only finite ones) */
void dump_trainer(int dump_id, Train_ResNet *trainer, const char *special_dir);
static void poll_nan_flag(Train_ResNet *t) {
    MiCtx *c = ctx_of(t);
    if (!c->nan_check_pending) return;
    c->nan_check_pending = 0;
    if (*c->nan_flag_host) {
        /* the Adam kernel left (highest offending locations[] index + 1): the tensor check_errors would have named first,
         * walking locations[] from the last to the first (resnet.cu:2896, :2952) */
        printf("ERROR: nan or inf found at location: %d\n", *c->nan_flag_host - 1);
        printf("Dumping data to id=99999999 and exiting...\n");
        c->nan_location = *c->nan_flag_host - 1;
        if (c->dp_pending) mid_stream_sync(G.comm);
        dump_trainer(99999999, t, t->dump_dir);
        if (c->nan_no_exit) { /* test hook: report through mi_trainer_check_errors instead of exiting; the run goes on from a clean flag */
            mid_memset(c->nan_flag_dev, 0, sizeof(int), G.compute);
            mid_stream_sync(G.compute);
            *c->nan_flag_host = 0;
            return;
        }
        abort_all_comms(); /* do not leave the peers waiting in a collective */
        exit(1);
    }
}
/* load_new_batch calls this before it replaces the batch: the flag copy update_parameters queued is waited for (one event, the
 * step is over by then), so that a 99999999 dump holds the OFFENDING step's inputs, activations and dump id -- what
 * resnet.cu:2879-2907 dumps from inside update_parameters */
void mi_trainer_poll_errors(Train_ResNet *t) {
    MiCtx *c = ctx_of(t);
    if (!c || !c->nan_check_pending) return;
    mid_event_sync(c->ev_nan);
    poll_nan_flag(t);
}

/* conv + BN (+ReLU | +residual+ReLU): prepareAndDoConvolution + prepareAndDoBatchNormAndActivate.
 * stem: the 7x7 convolution keeps fp32 input / output in every storage type; only its BN output is an activation tensor */
/* c->cur_par: parity copy of the NEXT stride-2 convolution's input (set by the caller); c->cur_par_valid: where the forward pass
 * records whether it really wrote the planes (it does only on the 16-byte staging route), read back by the weight gradient */
static void set_cur_par(MiCtx *c, void *buf, size_t bytes, int *valid) { c->cur_par = buf; c->cur_par_bytes = bytes; c->cur_par_valid = valid; c->cur_dye = NULL; c->cur_cl = NULL; c->cur_dye_valid = 0; c->cur_cl_ready = 0; }
static void unit_fwd(Train_ResNet *t, const float *in, const float *w, const BatchNorm *bn, Cache_BatchNorm *cache,
                     float *conv_out, float *act_out, const float *residual, int C, int H, int K, int k, int stride,
                     int relu, int stem) {
    MiCtx *c = ctx_of(t);
    c->ws.s2d = stride == 2 ? c->cur_par : NULL; c->ws.s2d_bytes = stride == 2 ? c->cur_par_bytes : 0; c->ws.s2d_valid = 0;
    const int N = t->batch_size, Ho = H / stride;
    const int bf = c->dtype == MID_BF16 && !stem;
    /* the convolution leaves per-tile (count, mean, M2) partials of its output: BN reads the tensor twice, not three times */
    const mid_wt_entry *we = wt_lookup(c, w);
    mid_bn_parts *parts = (c->fuse_bn_stats || bf) ? &c->bn_parts : NULL;
    c->ws.pre_fwd = we ? we->fwd : NULL; /* re-laid at the start of this forward pass */
    if (stem && c->stem_scratch && c->dtype == MID_F32) {
        ck(mid_stem_fwd_f32(G.compute, in, w, conv_out, c->stem_xp, c->stem_xp_bytes, c->stem_scratch, c->stem_scratch_floats, N, H, parts),
           "stem convolution forward (fp32 matrix cores)");
    } else if (stem && c->stem_scratch) {
        parts = &c->bn_parts; /* (the stem's tensors are fp32 here, but its statistics still come from the kernel's accumulators) */
        ck(mid_stem_fwd_bf16(G.compute, in, w, conv_out, c->stem_bf16 ? MID_BF16 : MID_F32, c->stem_xp, c->stem_xp_bytes, c->stem_scratch, c->stem_scratch_floats, N, H, parts),
           "stem convolution forward (bf16 operands)");
    } else if (bf && k == 3 && stride == 1 && c->cur_cl && we && we->fwd) {
        /* the reduction BN wrote this input as a zero-padded channel-last plane beside its NCHW output: no re-layout pass */
        ck(mid_cl_fwd(G.compute, c->cur_cl, we->fwd, conv_out, N, C, H, K, 1, parts), "convolution forward (bf16, channel-last)");
    } else if (bf && k == 3 && stride == 2 && c->cur_cl && we && we->fwd) {
        /* channel-last route: the input re-laid once as four zero-padded parity planes, which the weight gradient reads again */
        if (!c->cur_cl_ready) ck(mid_cl_relayout(G.compute, in, c->cur_cl, N, C, H, 1), "input re-layout (channel-last parity planes)");
        ck(mid_cl_fwd(G.compute, c->cur_cl, we->fwd, conv_out, N, C, H, K, 2, parts), "convolution forward (bf16, channel-last)");
        if (c->cur_par_valid) *c->cur_par_valid = 0;
    } else if (bf) {
        ck(mid_conv_fwd_bf16(G.compute, &c->ws, in, w, conv_out, N, C, H, K, k, stride, parts), "convolution forward (bf16)");
        if (stride == 2 && c->cur_par_valid) *c->cur_par_valid = c->ws.s2d_valid; /* the launch says whether it left the parity planes */
    } else ck(mid_conv_fwd_stats(G.compute, &c->ws, in, w, conv_out, N, C, H, K, k, stride, parts), "convolution forward");
    c->ws.pre_fwd = NULL;
    if (c->bn_cl_out) { mid_bn_set_cl_out(bf ? c->bn_cl_out : NULL, c->bn_cl_H); c->bn_cl_out = NULL; }
    ck(mid_bn_fwd_t(G.compute, c->bn_ws, parts, conv_out, (bf || (stem && c->stem_bf16)) ? MID_BF16 : MID_F32, bn->gamma, bn->beta, residual, cache->means, cache->vars,
                    act_out, c->dtype, cache->normalized_temp, cache->normalized, N, K, Ho * Ho, t->eps, relu), "batch norm forward");
}

/* the batch-norm launchers take their cross-replica setting from one process-wide slot (kernels_bn.hip): every pass binds ITS
 * trainer's (none for a trainer without sync-BN), so two trainers in one process do not inherit each other's */
#define MI_SYNC_BN_TMP_FLOATS (2 * 4096)
static void bind_sync_bn(const MiCtx *c) {
    if (c->sync_bn && c->sync_bn_comm) mid_bn_set_sync(c->sync_bn_comm, c->world, c->sync_bn_tmp, MI_SYNC_BN_TMP_FLOATS, 1);
    else mid_bn_set_sync(NULL, 1, NULL, 0, 0);
}
/* resnet.cu:1526-1775 */
void forward_pass(Train_ResNet *t) {
    MiCtx *c = ctx_of(t);
    const Dims *d = t->model->dims;
    const Params *p = t->model->params;
    Activations *a = t->forward_buffer->activations;
    const int N = t->batch_size, f = d->init_conv_filters;
    mid_event_record(c->ev_t[0], G.compute);
    bind_sync_bn(c);
    relayout_weights(c);
    set_cur_par(c, NULL, 0, NULL);
    unit_fwd(t, t->cur_batch->images, p->init_conv_layer, p->norm_init_conv, a->norm_init_conv, a->init_conv_applied,
             a->init_conv_activated, NULL, 3, d->input, f, d->init_kernel_dim, d->init_conv_stride, 1, 1);
    const int Hs = d->input / d->init_conv_stride;
    ck(mid_maxpool_fwd_t(G.compute, a->init_conv_activated, a->init_convblock_input, c->dtype, a->max_inds, N, f, Hs, d->init_maxpool_dim,
                         d->init_maxpool_stride), "max-pool forward");
    const float *bin = a->init_convblock_input;
    int pr_ready = 0; /* the producing BN apply of the block before has already written this block's projection planes */
    for (int i = 0; i < d->n_conv_blocks; i++) {
        const ConvBlock *b = p->conv_blocks[i];
        Activation_ConvBlock *k = a->activation_conv_blocks[i];
        const int H = b->incoming_spatial_dim, Ho = H / b->stride;
        int sp_ready = 0;
        if (c->par && c->par[i].cl_s1) { c->bn_cl_out = c->par[i].cl_s1; c->bn_cl_H = H; }
        else if (c->par && c->cl_pre && b->stride == 2 && c->par[i].cl_spatial && !(H & 1)) { c->bn_cl_out = c->par[i].cl_spatial; c->bn_cl_H = -H; sp_ready = 1; }
        unit_fwd(t, bin, b->depth_reduction, b->norm_depth_reduction, k->norm_post_reduced, k->post_reduced,
                 k->post_reduced_activated, NULL, b->incoming_filters, H, b->reduced_depth, 1, 1, 1, 0);
        if (c->par) set_cur_par(c, c->par[i].spatial, c->par[i].spatial_bytes, &c->par[i].spatial_valid); else set_cur_par(c, NULL, 0, NULL);
        c->cur_cl = c->par ? (b->stride == 2 ? c->par[i].cl_spatial : c->par[i].cl_s1) : NULL;
        c->cur_cl_ready = sp_ready;
        unit_fwd(t, k->post_reduced_activated, b->spatial, b->norm_spatial, k->norm_post_spatial, k->post_spatial,
                 k->post_spatial_activated, NULL, b->reduced_depth, H, b->reduced_depth, 3, b->stride, 1, 0);
        const float *res = bin;
        if (b->projection) { /* resnet.cu:1685-1704 */
            if (c->par) set_cur_par(c, c->par[i].proj, c->par[i].proj_bytes, &c->par[i].proj_valid); else set_cur_par(c, NULL, 0, NULL);
            c->cur_cl = c->par ? c->par[i].cl_proj : NULL;
            c->cur_cl_ready = pr_ready;
            unit_fwd(t, bin, b->projection, b->norm_projection, k->norm_post_projection, k->transformed_residual,
                     k->post_projection_norm_vals, NULL, b->incoming_filters, H, b->expanded_depth, b->stride == 2 ? 3 : 1,
                     b->stride, 0, 0);
            res = k->post_projection_norm_vals;
        }
        pr_ready = 0;
        if (!c->full_store) { /* BN(expanded) + addVec + doActivation in one kernel (:1670, :1717, :1723) */
            if (c->par && c->cl_pre && i + 1 < d->n_conv_blocks && p->conv_blocks[i + 1]->stride == 2 && p->conv_blocks[i + 1]->projection &&
                c->par[i + 1].cl_proj && !(Ho & 1)) { /* this block's output is the next block's 3x3 stride-2 projection input */
                c->bn_cl_out = c->par[i + 1].cl_proj; c->bn_cl_H = -Ho; pr_ready = 1;
            }
            unit_fwd(t, k->post_spatial_activated, b->depth_expansion, b->norm_expansion, k->norm_post_expanded,
                     k->post_expanded, k->output_activated, res, b->reduced_depth, Ho, b->expanded_depth, 1, 1, 0, 0);
        } else {
            unit_fwd(t, k->post_spatial_activated, b->depth_expansion, b->norm_expansion, k->norm_post_expanded,
                     k->post_expanded, k->post_expanded_norm_vals, NULL, b->reduced_depth, Ho, b->expanded_depth, 1, 1, 0, 0);
            ck(mid_add_relu(G.compute, k->post_expanded_norm_vals, res, k->output, k->output_activated,
                            (size_t)N * b->expanded_depth * Ho * Ho), "add + ReLU");
        }
        bin = k->output_activated;
    }
    const ConvBlock *last = p->conv_blocks[d->n_conv_blocks - 1];
    const int Hl = last->incoming_spatial_dim; /* resnet.cu:1732 */
    ck(mid_avgpool_fwd_t(G.compute, bin, c->dtype, a->final_conv_output_pooled, N, d->final_depth, Hl * Hl), "average pool");
    ck(mid_gemm_nn(G.compute, a->final_conv_output_pooled, p->fully_connected, a->linear_output, N, d->final_depth, d->output), "FC forward");
    ck(mid_softmax(G.compute, a->linear_output, t->forward_buffer->pred, N, d->output), "soft-max");
    mid_event_record(c->ev_t[1], G.compute);
    mid_memcpy_d2h(t->forward_buffer->pred_cpu, t->forward_buffer->pred, (size_t)N * d->output * sizeof(float), G.compute);
    mid_stream_sync(G.compute); /* the reference's blocking cudaMemcpy (:1774) */
    poll_nan_flag(t);           /* the previous update's flag copy has landed by now */
}

/* resnet.cu:3363-3383 */
float mi_host_loss(Train_ResNet *t, int *n_wrong) {
    const int N = t->batch_size, L = t->model->dims->output;
    const float *pred = t->forward_buffer->pred_cpu;
    const int *lab = t->cur_batch->correct_classes_cpu;
    float loss = 0;
    int wrong = 0;
    for (int s = 0; s < N; s++) loss += -1 * logf(pred[(size_t)s * L + lab[s]]);
    for (int s = 0; s < N; s++) {
        const float pc = pred[(size_t)s * L + lab[s]];
        for (int cc = 0; cc < L; cc++)
            if (cc != lab[s] && pred[(size_t)s * L + cc] >= pc) { wrong++; break; }
    }
    if (n_wrong) *n_wrong = wrong;
    return loss;
}

/* BN' (+fused ReLU') then conv' : prepareAndDoActivationAndBatchNormDeriv + prepreAndDoConvolutionDeriv */
/* the aux stream must have finished the previous weight gradient before a dgrad may overwrite the rolling buffer it reads */
static void join_wgrad(MiCtx *c) {
    if (c->wgrad_pending) { mid_stream_wait_event(G.compute, c->ev_wgrad_done); c->wgrad_pending = 0; }
}
/* mode 2: next slot of the derivative ring; the compute stream first waits for the weight gradient (if any) that still
 * reads the slot's previous contents */
static float *ring_take(MiCtx *c, int *slot) {
    const int i = c->ring_next;
    c->ring_next = (i + 1) % MI_RING;
    if (c->ring_busy[i]) { mid_stream_wait_event(G.compute, c->ring_ev[i]); c->ring_busy[i] = 0; }
    if (slot) *slot = i;
    return c->ring_buf[i];
}
/* typed launch helpers: the bottleneck convolutions run on the bf16 kernels in bf16 mode, the stem always on the fp32 path */
static void conv_dgrad_t(Train_ResNet *t, const float *w, const float *dy, float *dx, const float *addend, int C, int H, int K, int k,
                         int stride) {
    MiCtx *c = ctx_of(t);
    const mid_wt_entry *we = wt_lookup(c, w);
    c->ws.pre_dgrad = we ? we->dgrad : NULL;
    if (c->dtype == MID_BF16 && stride == 1 && k == 3 && c->cur_dye && c->cur_dye_valid && we && we->dgrad) {
        /* stride-1 dgrad on the channel-last dY plane (the BN' below then runs its own reduction pass: measured neutral) */
        ck(mid_cl_dgrad(G.compute, c->cur_dye, we->dgrad, dx, addend, t->batch_size, C, H, K), "convolution dgrad (bf16, channel-last)");
        c->fz_req_valid = 0;
        c->ws.pre_dgrad = NULL;
        return;
    }
    if (c->dtype == MID_BF16 && stride == 2 && k == 3 && !addend && c->cur_dye && we && we->dgrad) {
        /* stride-2 dgrad on channel-last dY: re-lay dY (K channels, H/2 x H/2) into the layer's zero-bordered buffer, then both column
         * parities of dx per workgroup by LDS-DMA staged MFMAs (dense stores; 1.5-1.9x the NCHW kernel's four parity classes) */
        if (!c->cur_dye_valid) ck(mid_cl_relayout_end(G.compute, dy, c->cur_dye, t->batch_size, K, H / 2), "dY re-layout (channel-last)");
        c->cur_dye_valid = 1;
        ck(mid_cl_dgrad2(G.compute, c->cur_dye, we->dgrad, dx, t->batch_size, C, H, K), "convolution dgrad (bf16, channel-last, stride 2)");
        c->fz_req_valid = 0;
        c->ws.pre_dgrad = NULL;
        return;
    }
    {
        static int minp = -1; /* diagnostic: RESNET_MI_BNFUSE_MINP = smallest plane (pixels) whose dgrad still carries the BN' reduction */
        if (minp < 0) minp = getenv("RESNET_MI_BNFUSE_MINP") ? atoi(getenv("RESNET_MI_BNFUSE_MINP")) : 0;
        if (c->fz_req_valid && H * H < minp) c->fz_req_valid = 0;
    }
    if (c->dtype == MID_BF16 && c->fz_req_valid) { /* ... and the reduction pass of the BN' its output feeds */
        ck(mid_conv_dgrad_bn_bf16(G.compute, &c->ws, w, dy, dx, addend, t->batch_size, C, H, K, k, stride, &c->fz_req), "convolution dgrad + BN' reduction (bf16)");
        if (c->fz_req.nparts > 0) { c->fz_done = c->fz_req; c->fz_ready = 1; }
    } else if (c->fz_req_valid) { /* fp32 storage: the stride-1 layers on the implicit-GEMM route do the same */
        ck(mid_conv_dgrad_bn_f32(G.compute, &c->ws, w, dy, dx, addend, t->batch_size, C, H, K, k, stride, &c->fz_req), "convolution dgrad + BN' reduction");
        if (c->fz_req.nparts > 0) { c->fz_done = c->fz_req; c->fz_ready = 1; }
    } else if (c->dtype == MID_BF16) ck(mid_conv_dgrad_bf16(G.compute, &c->ws, w, dy, dx, addend, t->batch_size, C, H, K, k, stride), "convolution dgrad (bf16)");
    else ck(mid_conv_dgrad(G.compute, &c->ws, w, dy, dx, addend, t->batch_size, C, H, K, k, stride), "convolution dgrad");
    c->fz_req_valid = 0;
    c->ws.pre_dgrad = NULL;
}
static void conv_wgrad_t(Train_ResNet *t, mid_stream st, const float *x, const float *dy, float *dw, int C, int H, int K, int k, int stride,
                         int stem) {
    MiCtx *c = ctx_of(t);
    /* the forward pass left the parity planes of x in the layer's own buffer: the weight gradient reads them again */
    c->ws.s2d = stride == 2 ? c->cur_par : NULL; c->ws.s2d_bytes = stride == 2 ? c->cur_par_bytes : 0;
    c->ws.s2d_valid = stride == 2 && c->cur_par != NULL && c->cur_par_valid && *c->cur_par_valid;
    if (stem && c->stem_scratch && c->dtype == MID_F32)
        ck(mid_stem_wgrad_f32(st, c->stem_xp, dy, dw, c->stem_scratch, c->stem_scratch_floats, t->batch_size, H), "stem convolution wgrad (fp32 matrix cores)");
    else if (stem && c->stem_scratch) /* the forward pass left the batch as padded bf16 parity planes */
        ck(mid_stem_wgrad_bf16(st, c->stem_xp, dy, c->stem_bf16 ? MID_BF16 : MID_F32, dw, c->stem_scratch, c->stem_scratch_floats, t->batch_size, H), "stem convolution wgrad (bf16 operands)");
    else if (c->dtype == MID_BF16 && !stem && k == 3 && c->cur_cl && c->cur_dye && c->cur_dye_valid && c->cl_wgrad2 &&
             mid_cl_wgrad2_supported(t->batch_size, C, H, K, stride) && (((H / stride) * (H / stride)) % 64 != 0 || !mid_cl_wgrad_supported(t->batch_size, C, H, K, stride)))
        /* both operands channel-last (the dY planes the dgrad has just made): planes that do not fill 64-pixel tiles (784, 196, 49 pixels:
         * all of the benchmark network's stride-2 layers; -0.8 ms per step, most of it the two 7x7 layers the other kernel cannot take) */
        ck(mid_cl_wgrad2(st, c->cur_cl, c->cur_dye, dw, c->ws.part, c->ws.part_floats, t->batch_size, C, H, K, stride), "convolution wgrad (bf16, both operands channel-last)");
    else if (c->dtype == MID_BF16 && !stem && k == 3 && stride == 1 && c->cur_cl && mid_cl_wgrad_supported(t->batch_size, C, H, K, 1))
        ck(mid_cl_wgrad(st, c->cur_cl, dy, dw, c->ws.part, c->ws.part_floats, t->batch_size, C, H, K, 1), "convolution wgrad (bf16, channel-last input)");
    else if (c->dtype == MID_BF16 && !stem && k == 3 && stride == 2 && c->cur_cl && mid_cl_wgrad_supported(t->batch_size, C, H, K, 2))
        ck(mid_cl_wgrad(st, c->cur_cl, dy, dw, c->ws.part, c->ws.part_floats, t->batch_size, C, H, K, 2), "convolution wgrad (bf16, channel-last)");
    else if (c->dtype == MID_BF16 && !stem) ck(mid_conv_wgrad_bf16(st, &c->ws, x, dy, dw, t->batch_size, C, H, K, k, stride), "convolution wgrad (bf16)");
    else ck(mid_conv_wgrad(st, &c->ws, x, dy, dw, t->batch_size, C, H, K, k, stride), "convolution wgrad");
}
/* d_slot: ring slot holding d_conv_out (mode 2), -1 otherwise */
static void unit_bwd(Train_ResNet *t, const float *in, const float *w, const BatchNorm *bn, const Cache_BatchNorm *cache,
                     const BatchNorm *dbn, const float *conv_out, const float *dy, const float *mask_src, int mask_mode,
                     float *gated_out, float *d_conv_out, int d_slot, float *dx, const float *addend, float *dw, int C, int H, int K, int k,
                     int stride, int stem) {
    MiCtx *c = ctx_of(t);
    const int N = t->batch_size, Ho = H / stride;
    const int x_dt = (c->dtype == MID_BF16 && (!stem || c->stem_bf16)) ? MID_BF16 : MID_F32;
    /* BN' of this unit (HBM-bound) runs next to earlier units' weight gradients (FMA-bound, low-priority aux stream);
     * mask_mode 3: ReLU' of the block output fused in, and its product with the upstream gradient kept (gated_out) */
    if (c->fz_ready) { /* the dgrad that produced dy gated it and left the sums: merge, finalize, apply */
        c->fz_ready = 0;
        ck(mid_bn_bwd_parts_t(G.compute, c->bn_ws, &c->fz_done, conv_out, x_dt, bn->gamma, bn->beta, cache->means, cache->vars, dy, c->dtype,
                              d_conv_out, dbn->gamma, dbn->beta, N, K, Ho * Ho, t->eps), "batch norm backward (reduction done by the dgrad)");
    } else
    ck(mid_bn_bwd_t(G.compute, c->bn_ws, conv_out, x_dt, bn->gamma, bn->beta, cache->means, cache->vars, dy, mask_src, gated_out, c->dtype,
                    d_conv_out, dbn->gamma, dbn->beta, N, K, Ho * Ho, t->eps, mask_mode), "batch norm backward");
    if (c->dtype == MID_BF16 && !stem && stride == 1 && k == 3 && c->cur_dye && !c->cur_dye_valid) {
        /* stride 1: the same, one plane with a halo of 1 */
        ck(mid_cl_relayout(G.compute, d_conv_out, c->cur_dye, N, K, Ho, 0), "dY re-layout (channel-last)");
        c->cur_dye_valid = 1;
    }
    if (c->dtype == MID_BF16 && !stem && stride == 2 && k == 3 && c->cur_dye && !c->cur_dye_valid) {
        /* the channel-last copy of d_conv_out that the stride-2 dgrad AND the weight gradient read: made here, before either is
         * launched, so that every weight-gradient schedule (the free-running one starts before the dgrad) runs the same kernels */
        ck(mid_cl_relayout_end(G.compute, d_conv_out, c->cur_dye, N, K, Ho), "dY re-layout (channel-last)");
        c->cur_dye_valid = 1;
    }
    if (c->overlap_wgrad == 2 && d_slot >= 0) {
        /* d_conv_out is final once BN' is: the weight gradient may start now and run for as long as the slot lives */
        mid_event_record(c->ev_bn_done, G.compute);
        mid_stream_wait_event(G.aux, c->ev_bn_done);
        conv_wgrad_t(t, G.aux, in, d_conv_out, dw, C, H, K, k, stride, stem);
        mid_event_record(c->ring_ev[d_slot], G.aux);
        c->ring_busy[d_slot] = 1;
        mid_event_record(c->ev_wgrad_done, G.aux);
        c->wgrad_pending = 1;
        if (dx) conv_dgrad_t(t, w, d_conv_out, dx, addend, C, H, K, k, stride);
        return;
    }
    join_wgrad(c);
    if (dx) conv_dgrad_t(t, w, d_conv_out, dx, addend, C, H, K, k, stride);
    if (c->overlap_wgrad) {
        mid_event_record(c->ev_bn_done, G.compute);
        mid_stream_wait_event(G.aux, c->ev_bn_done);
        conv_wgrad_t(t, G.aux, in, d_conv_out, dw, C, H, K, k, stride, stem);
        mid_event_record(c->ev_wgrad_done, G.aux);
        c->wgrad_pending = 1;
    } else conv_wgrad_t(t, G.compute, in, d_conv_out, dw, C, H, K, k, stride, stem);
}

/* resnet.cu:1777-2248 */
void backwards_pass(Train_ResNet *t) {
    MiCtx *c = ctx_of(t);
    const Dims *d = t->model->dims;
    const Params *p = t->model->params;
    Activations *a = t->forward_buffer->activations;
    Backprop_Buffer *bb = t->backprop_buffer;
    const Params *dp = bb->param_derivs;
    Activations *da = bb->activation_derivs;
    const int N = t->batch_size, L = d->output, D = d->final_depth, nb = d->n_conv_blocks;
    const int recompute = c->policy == MI_STORE_RECOMPUTE_BN;
    mid_event_record(c->ev_t[2], G.compute);
    bind_sync_bn(c);
    if (weights_stale(c)) relayout_weights(c); /* parameters were rewritten from the host after forward_pass */
    c->dp_cursor = c->arena_floats;
    c->n_buckets = 0;
    /* dlogits = softmax - onehot, batch SUM (no 1/N: resnet.cu:1806-1811) */
    ck(mid_ce_deriv(G.compute, t->forward_buffer->pred, t->cur_batch->correct_classes, bb->output_layer_deriv, N, L), "cross-entropy derivative");
    /* FC: dW = pooled^T dlogits (:1823), dpooled = dlogits W^T (:1830) -- no transposed temporaries */
    ck(mid_gemm_tn(G.compute, a->final_conv_output_pooled, bb->output_layer_deriv, dp->fully_connected, D, N, L), "FC wgrad");
    ck(mid_gemm_nt(G.compute, bb->output_layer_deriv, p->fully_connected, da->final_conv_output_pooled, N, L, D), "FC dgrad");
    mi_dp_reduce_ready(t, (size_t)(dp->fully_connected - c->g_arena), 0);
    const ConvBlock *last = p->conv_blocks[nb - 1];
    const int Hl = last->incoming_spatial_dim;
    ck(mid_avgpool_bwd_t(G.compute, da->final_conv_output_pooled, da->activation_conv_blocks[nb - 1]->output_activated, c->dtype, N, D, Hl * Hl),
       "average pool backward");
    const int ring = c->overlap_wgrad == 2;
    for (int i = nb - 1; i >= 0; i--) {
        const ConvBlock *b = p->conv_blocks[i];
        const ConvBlock *db = dp->conv_blocks[i];
        const Activation_ConvBlock *k = a->activation_conv_blocks[i];
        Activation_ConvBlock *dk = da->activation_conv_blocks[i];
        const float *bin = i == 0 ? a->init_convblock_input : a->activation_conv_blocks[i - 1]->output_activated;
        float *dbin = i == 0 ? da->init_convblock_input : da->activation_conv_blocks[i - 1]->output_activated;
        const int H = b->incoming_spatial_dim, Ho = H / b->stride;
        const float *up = dk->output_activated; /* dL/d(block output) */
        const float *exp_dy, *exp_mask, *red_addend;
        int exp_mode, s_proj = -1, s_exp = -1, s_spa = -1, s_red = -1;
        if (ring) { /* this block's derivative tensors: fresh ring slots (resnet_cudnn_lowmem.cu:2152-2170 keeps four) */
            dk->output = ring_take(c, NULL);
            if (b->projection) dk->transformed_residual = ring_take(c, &s_proj);
        }
        /* bf16 + FAST: a dgrad may do the reduction pass of the BN' its output feeds (and gate that output) */
        /* (RECOMPUTE_BN: the gating tensors have just been re-derived when the dgrad runs.)  fp32 storage: not with the FULL policy
         * (its derivative mirror keeps the ungated gradients the dump tree names) */
        const int fz = c->fz_enable && (c->dtype == MID_BF16 || (c->fz_f32 && c->policy != MI_STORE_FULL));
#define FZ_REQ(site_, x_, mask_, means_) do { if (fz && (c->dtype == MID_BF16 ? (c->fz_bf16 & (site_)) : (c->fz_f32 & (site_)))) { c->fz_req.x = (x_); c->fz_req.mask = (mask_); c->fz_req.means = (means_); \
        c->fz_req.buf = c->bn_parts.buf; c->fz_req.floats = c->bn_parts.floats; c->fz_req.nparts = 0; c->fz_req_valid = 1; } } while (0)
        const int up_gated = c->fz_ready; /* the block above's reduction dgrad already gated `up` by this block's output and summed for the expansion BN' */
        if (b->projection) {
            /* ReLU' of the block output (doActivationDeriv, :1934) is fused into the projection BN' as an external mask; that
             * pass also leaves relu'(out) * up in dk->output, which the expansion BN' then reads instead of up + mask */
            if (c->par) set_cur_par(c, c->par[i].proj, c->par[i].proj_bytes, &c->par[i].proj_valid); else set_cur_par(c, NULL, 0, NULL);
            c->cur_dye = c->par ? c->par[i].dye_proj : NULL;
            c->cur_cl = c->par ? c->par[i].cl_proj : NULL;
            unit_bwd(t, bin, b->projection, b->norm_projection, k->norm_post_projection, db->norm_projection,
                     k->transformed_residual, up, k->output_activated, 3, dk->output, dk->transformed_residual, s_proj, dbin, NULL,
                     db->projection, b->incoming_filters, H, b->expanded_depth, b->stride == 2 ? 3 : 1, b->stride, 0);
            exp_dy = dk->output; exp_mask = NULL; exp_mode = 0;
            red_addend = dbin; /* reduce-conv dgrad accumulates onto the projection path (toAdd, :2157) */
        } else {
            /* doActivationDeriv (:1934) rides in the expansion BN' reduce pass, which also leaves relu'(out) * up in dk->output
             * (one pass over the block output less than a separate ReLU' kernel) */
            exp_dy = up; exp_mask = k->output_activated; exp_mode = 3;
            red_addend = up_gated ? up : dk->output; /* identity shortcut: setVal 0 + addVec (:2003-2004) folded into the dgrad epilogue */
        }
        if (ring) { dk->post_expanded = ring_take(c, &s_exp); dk->post_spatial_activated = ring_take(c, NULL); }
        if (recompute) /* the expansion's input, re-derived: relu(BN(post_spatial)) (resnet_clean.cu:2753) */
            ck(mid_bn_apply_t(G.compute, k->post_spatial, c->dtype, b->norm_spatial->gamma, b->norm_spatial->beta, NULL, k->norm_post_spatial->means,
                              k->norm_post_spatial->vars, k->post_spatial_activated, c->dtype, N, b->reduced_depth, Ho * Ho, t->eps, 1), "BN recompute");
        FZ_REQ(1, k->post_spatial, k->post_spatial_activated, k->norm_post_spatial->means); /* expansion dgrad -> spatial BN' */
        unit_bwd(t, k->post_spatial_activated, b->depth_expansion, b->norm_expansion, k->norm_post_expanded,
                 db->norm_expansion, k->post_expanded, exp_dy, exp_mask, exp_mode, dk->output, dk->post_expanded, s_exp,
                 dk->post_spatial_activated, NULL, db->depth_expansion, b->reduced_depth, Ho, b->expanded_depth, 1, 1, 0);
        /* the call resnet.cu:2060-2083 forgot; present in resnet_cudnn.cu:2365-2366 */
        if (ring) { dk->post_spatial = ring_take(c, &s_spa); dk->post_reduced_activated = ring_take(c, NULL); }
        if (recompute) /* the 3x3's input, re-derived: relu(BN(post_reduced)) (resnet_clean.cu:2714) */
            ck(mid_bn_apply_t(G.compute, k->post_reduced, c->dtype, b->norm_depth_reduction->gamma, b->norm_depth_reduction->beta, NULL,
                              k->norm_post_reduced->means, k->norm_post_reduced->vars, k->post_reduced_activated, c->dtype, N, b->reduced_depth,
                              H * H, t->eps, 1), "BN recompute");
        if (c->par) set_cur_par(c, c->par[i].spatial, c->par[i].spatial_bytes, &c->par[i].spatial_valid); else set_cur_par(c, NULL, 0, NULL);
        c->cur_dye = c->par ? (b->stride == 2 ? c->par[i].dye_spatial : c->par[i].dy1) : NULL;
        c->cur_cl = c->par ? (b->stride == 2 ? c->par[i].cl_spatial : c->par[i].cl_s1) : NULL;
        FZ_REQ(2, k->post_reduced, k->post_reduced_activated, k->norm_post_reduced->means); /* spatial dgrad -> reduction BN' */
        unit_bwd(t, k->post_reduced_activated, b->spatial, b->norm_spatial, k->norm_post_spatial, db->norm_spatial,
                 k->post_spatial, dk->post_spatial_activated, NULL, 1, NULL, dk->post_spatial, s_spa, dk->post_reduced_activated, NULL,
                 db->spatial, b->reduced_depth, H, b->reduced_depth, 3, b->stride, 0);
        if (ring) dk->post_reduced = ring_take(c, &s_red);
        if (i > 0 && !p->conv_blocks[i - 1]->projection) { /* reduction dgrad -> the expansion BN' of the identity block below */
            const Activation_ConvBlock *kb = a->activation_conv_blocks[i - 1];
            FZ_REQ(4, kb->post_expanded, kb->output_activated, kb->norm_post_expanded->means);
        }
        unit_bwd(t, bin, b->depth_reduction, b->norm_depth_reduction, k->norm_post_reduced, db->norm_depth_reduction,
                 k->post_reduced, dk->post_reduced_activated, NULL, 1, NULL, dk->post_reduced, s_red, dbin, red_addend,
                 db->depth_reduction, b->incoming_filters, H, b->reduced_depth, 1, 1, 0);
        mi_dp_reduce_ready(t, (size_t)(db->depth_reduction - c->g_arena), 0);
#undef FZ_REQ
    }
    c->fz_ready = 0; c->fz_req_valid = 0;
    const int Hs = d->input / d->init_conv_stride;
    int s_stem = -1;
    if (ring) { da->init_conv_activated = ring_take(c, NULL); da->init_conv_applied = ring_take(c, &s_stem); }
    ck(mid_maxpool_bwd_t(G.compute, a->max_inds, da->init_convblock_input, da->init_conv_activated, c->dtype, N, d->init_conv_filters, Hs,
                         d->init_maxpool_dim, d->init_maxpool_stride), "max-pool backward");
    set_cur_par(c, NULL, 0, NULL); /* the stem strides too, but has its own padded planes (stem_xp) */
    unit_bwd(t, t->cur_batch->images, p->init_conv_layer, p->norm_init_conv, a->norm_init_conv, dp->norm_init_conv,
             a->init_conv_applied, da->init_conv_activated, NULL, 1, NULL, da->init_conv_applied, s_stem, NULL, NULL,
             dp->init_conv_layer, 3, d->input, d->init_conv_filters, d->init_kernel_dim, d->init_conv_stride, 1);
    mi_dp_reduce_ready(t, 0, 1);
    mid_event_record(c->ev_t[3], G.compute);
}

/* Where the data-parallel path may cut the gradient arena: after the FC layer, after each block (walking backwards) and at
 * the end of backward.  A bucket is emitted at a cut as soon as at least bucket_bytes are pending.  Pure arithmetic on the
 * arena offsets (the carve order of build_params), shared by backwards_pass and mi_debug_dp_plan. */
int mi_dp_plan_buckets(const Dims *d, size_t bucket_bytes, size_t *from, size_t *to, int max) {
    size_t arena;
    (void)count_locations(d, &arena);
    /* offsets of each block's first tensor and of the FC tensor */
    size_t *boff = (size_t *)malloc(sizeof(size_t) * (size_t)(d->n_conv_blocks + 1));
    int inc = d->init_conv_filters, ex = 4 * inc, red = inc;
    size_t off = align_up((size_t)d->init_kernel_dim * d->init_kernel_dim * inc * 3) + 2 * align_up(inc);
    for (int i = 0; i < d->n_conv_blocks; i++) {
        int stride = 1;
        if (d->is_block_spatial_reduction[i] == 1) { stride = 2; red *= 2; ex *= 2; }
        boff[i] = off;
        off += align_up((size_t)inc * red) + align_up((size_t)red * red * 9) + align_up((size_t)ex * red) + 4 * align_up(red) + 2 * align_up(ex);
        if (inc != ex) off += align_up((size_t)inc * ex * (stride == 2 ? 9 : 1)) + 2 * align_up(ex);
        inc = ex;
    }
    const size_t fc_off = off;
    int n = 0;
    size_t cursor = arena;
    /* at most MI_MAX_BUCKETS buckets (one event each): the last slot is kept for the forced final cut, so a network with more
     * cut points than slots gets a larger last bucket, never a lost range */
#define CUT(from_, force_)                                                                      \
    do {                                                                                        \
        if ((from_) < cursor && ((force_) || ((cursor - (from_)) * sizeof(float) >= bucket_bytes && n < MI_MAX_BUCKETS - 1))) { \
            if (n < max) { from[n] = (from_); to[n] = cursor; }                                  \
            n++; cursor = (from_);                                                              \
        }                                                                                       \
    } while (0)
    CUT(fc_off, 0);
    for (int i = d->n_conv_blocks - 1; i >= 0; i--) CUT(boff[i], 0);
    CUT((size_t)0, 1);
#undef CUT
    free(boff);
    return n;
}

/* data parallel: hand the finished tail [from, cursor) of the gradient arena to RCCL on the comm stream as soon
 * as it is at least one bucket (or `force`).  Gradients become ready FC-first, i.e. from the arena's end
 * (the order update_parameters walks, resnet.cu:2952). */
void mi_dp_reduce_ready(Train_ResNet *t, size_t from, int force) {
    MiCtx *c = ctx_of(t);
    const int pending = c->wgrad_pending;
    if (force) join_wgrad(c); /* end of backward: every weight gradient is on the compute stream's timeline */
    if (!c->comm) return;
    if (from >= c->dp_cursor) return;
    const size_t n = c->dp_cursor - from;
    if (!force && n * sizeof(float) < c->bucket_bytes) return;
    if (!force && c->n_buckets >= MI_MAX_BUCKETS - 1) return; /* the last event slot is kept for the forced final cut (mi_dp_plan_buckets) */
    mid_event_record(c->ev_grads, G.compute);
    mid_stream_wait_event(G.comm, c->ev_grads);
    /* the bucket also holds weight gradients from the aux stream: the comm stream waits for the latest of them itself,
     * the compute stream does not stall */
    if (pending) mid_stream_wait_event(G.comm, c->ev_wgrad_done);
    ck(mid_rccl_allreduce_sum(c->comm, c->g_arena + from, c->dp_cursor - from, G.comm), "RCCL all-reduce");
    const int b = c->n_buckets++; /* < MI_MAX_BUCKETS by the rule above */
    c->bk_from[b] = from; c->bk_to[b] = c->dp_cursor;
    mid_event_record(c->bk_ev[b], G.comm);
    c->dp_cursor = from;
    c->dp_pending = 1;
}

/* resnet.cu:2910-2987.  No host synchronisation in here: the NaN / Inf flag is copied back asynchronously and read at the
 * next synchronisation point (forward_pass's pred copy), and with a communicator Adam runs bucket by bucket, each launch
 * waiting only for its own bucket's all-reduce -- the FC ... stage-3 updates run while the stem-side buckets are still on
 * the wire, and the host is free to queue the next load_new_batch / forward_pass behind them. */
void update_parameters(Train_ResNet *t) {
    MiCtx *c = ctx_of(t);
    const Params *p = t->model->params;
    float *p_arena = mi_params_arena_base(p);
    const float cur_b1 = t->cur_mean_decay * t->base_mean_decay; /* decays advance BEFORE use (:2920-2921) */
    const float cur_b2 = t->cur_var_decay * t->base_var_decay;
    if (c->dump_every > 0 && t->cur_dump_id % c->dump_every == 0) {
        if (c->dp_pending) mid_stream_sync(G.comm); /* the dump reads the gradient arena: not while RCCL is reducing it */
        dump_trainer(t->cur_dump_id, t, t->dump_dir);
    }
    mid_event_record(c->ev_t[4], G.compute);
    if (c->dp_pending && c->n_buckets > 0) {
        for (int b = 0; b < c->n_buckets; b++) {
            mid_stream_wait_event(G.compute, c->bk_ev[b]);
            const size_t o = c->bk_from[b], n = c->bk_to[b] - c->bk_from[b];
            ck(mid_adam(G.compute, p_arena + o, c->g_arena + o, c->m_arena + o, c->v_arena + o, n, t->learning_rate, t->weight_decay,
                        t->base_mean_decay, t->base_var_decay, cur_b1, cur_b2, t->eps, c->nan_flag_dev, 1, c->loc_off_dev, c->n_loc, o), "Adam");
        }
        c->dp_pending = 0; c->n_buckets = 0;
    } else {
        /* one launch over the whole arena; it also clears the gradients (:2972-2978) */
        ck(mid_adam(G.compute, p_arena, c->g_arena, c->m_arena, c->v_arena, c->arena_floats, t->learning_rate, t->weight_decay,
                    t->base_mean_decay, t->base_var_decay, cur_b1, cur_b2, t->eps, c->nan_flag_dev, 1, c->loc_off_dev, c->n_loc, 0), "Adam");
    }
    if (c->input_reset) { /* :2981-2982 */
        mid_memset(t->cur_batch->images, 0, (size_t)t->batch_size * t->cur_batch->image_size * sizeof(float), G.compute);
        mid_memset(t->cur_batch->correct_classes, 0, (size_t)t->batch_size * sizeof(int), G.compute);
    }
    mid_event_record(c->ev_t[5], G.compute);
    mid_memcpy_d2h(c->nan_flag_host, c->nan_flag_dev, sizeof(int), G.compute);
    mid_event_record(c->ev_nan, G.compute);
    c->nan_check_pending = 1;
    c->params_dirty = 1; /* the re-laid weight copies are stale until the next forward_pass */
    t->cur_mean_decay = cur_b1;
    t->cur_var_decay = cur_b2;
}
/* end of an epoch as the reference's main() does it (resnet.cu:3410-3421): loss_per_epoch = the epoch's SUMMED loss,
 * accuracy_per_epoch = fraction right, data source rewound to the first shard, cur_epoch advanced */
void mi_trainer_end_epoch(Train_ResNet *t, float epoch_loss, float epoch_n_wrong, float total_images_per_epoch) {
    if (t->cur_epoch >= 0 && t->cur_epoch < (t->n_epochs > 0 ? t->n_epochs : 1)) {
        t->loss_per_epoch[t->cur_epoch] = epoch_loss;
        t->accuracy_per_epoch[t->cur_epoch] = (total_images_per_epoch - epoch_n_wrong) / total_images_per_epoch;
    }
    t->cur_batch->cur_shard_id = -1;
    t->cur_batch->cur_batch_in_shard = -1;
    t->cur_epoch += 1;
}
/* check_errors on demand (resnet.cu:2879-2907): waits for the device and reads the flag of the last update */
int mi_trainer_check_errors(Train_ResNet *t) {
    MiCtx *c = ctx_of(t);
    mid_stream_sync(G.compute);
    const int bad = c->nan_check_pending && *c->nan_flag_host;
    poll_nan_flag(t);
    return bad;
}
/* the locations[] index the last NaN / Inf report named (-1: none); mi_trainer_set_nan_exit(t, 0) makes the report return through
 * mi_trainer_check_errors instead of exit(1) (tests) */
/* storage type of the stem convolution's own output and of that tensor's gradient (MI_DTYPE_*): bf16 in the bf16 mode when the stem runs
 * on the matrix cores, fp32 otherwise (fp32 mode; VALU stem; RESNET_MI_BF16_STEM_TENSORS=f32) */
int mi_trainer_stem_dtype(Train_ResNet *t) { const MiCtx *c = ctx_of(t); return c->dtype == MID_BF16 && c->stem_bf16 ? MID_BF16 : MID_F32; }

int mi_trainer_nan_location(const Train_ResNet *t) { return ((const MiCtx *)t->backend_ctx)->nan_location; }
void mi_trainer_set_nan_exit(Train_ResNet *t, int on) {
    MiCtx *c = ctx_of(t);
    c->nan_no_exit = !on;
    if (on) return;
    /* a run that goes on after a report starts from a clean flag */
    mid_memset(c->nan_flag_dev, 0, sizeof(int), G.compute);
    mid_stream_sync(G.compute);
    *c->nan_flag_host = 0;
}

/* ---------------------------------------------------------------------------------------------- */
static void free_activations_host(Activations *a) {
    if (!a) return;
    free(a->norm_init_conv);
    for (int i = 0; i < a->n_conv_blocks; i++) {
        Activation_ConvBlock *k = a->activation_conv_blocks[i];
        free(k->norm_post_reduced); free(k->norm_post_spatial); free(k->norm_post_expanded); free(k->norm_post_projection);
        free(k);
    }
    free(a->activation_conv_blocks);
    free(a);
}
void destroy_trainer(Train_ResNet *t) {
    if (!t) return;
    MiCtx *c = ctx_of(t);
    mid_device_sync();
    const Dims *d = t->model->dims;
    for (MiCtx **pp = &g_live; *pp; pp = &(*pp)->next_live) if (*pp == c) { *pp = c->next_live; break; }
    mid_event_destroy(c->ev_nan);
    if (c->comm) mid_rccl_comm_destroy(c->comm);
    if (c->sync_bn_comm) { mid_bn_set_sync(NULL, 1, NULL, 0, 0); mid_rccl_comm_destroy(c->sync_bn_comm); mid_free(c->sync_bn_tmp); }
    for (int i = 0; i < c->n_allocs; i++) mid_free(c->allocs[i]);
    free(c->allocs); free(c->alloc_bytes);
    for (int i = 0; i < MI_MAX_BUCKETS; i++) mid_event_destroy(c->bk_ev[i]);
    mid_free_host(t->forward_buffer->pred_cpu);
    mid_free_host(c->nan_flag_host);
    mid_event_destroy(c->ev_grads); mid_event_destroy(c->ev_reduced);
    free(c->wt_tab);
    mid_event_destroy(c->ev_bn_done); mid_event_destroy(c->ev_wgrad_done);
    for (int i = 0; i < MI_RING; i++) mid_event_destroy(c->ring_ev[i]);
    for (int i = 0; i < 6; i++) mid_event_destroy(c->ev_t[i]);
    free_activations_host(t->forward_buffer->activations);
    free_activations_host(t->backprop_buffer->activation_derivs);
    free_params_host(t->backprop_buffer->param_derivs, d);
    free_params_host(t->backprop_buffer->prev_means, d);
    free_params_host(t->backprop_buffer->prev_vars, d);
    free(t->forward_buffer); free(t->backprop_buffer);
    mid_free(mi_params_arena_base(t->model->params));
    free_params_host(t->model->params, d);
    mi_batch_ext_free(t->cur_batch);
    free(t->model->dims); free(t->model);
    free(t->loss_per_epoch); free(t->accuracy_per_epoch);
    free(c->dump_root); free(c->par); free(c);
    free(t);
}

/* ---------------------------------------------------------------------------------------------- */
int mi_dp_unique_id_bytes(void) { return mid_rccl_unique_id_bytes(); }
int mi_dp_get_unique_id(void *out, int bytes) { return mid_rccl_get_unique_id(out, bytes); }
int mi_dp_init(Train_ResNet *t, int rank, int world, const void *unique_id, int bytes) {
    MiCtx *c = ctx_of(t);
    if (world < 1 || rank < 0 || rank >= world) return -1;
    if (world == 1 && !unique_id) { c->world = 1; return 0; }
    /* world == 1 with an id builds a one-rank communicator: the whole bucket/stream/event path runs (self-test) */
    c->comm = mid_rccl_comm_init(rank, world, unique_id, bytes);
    if (!c->comm) return -1;
    c->rank = rank; c->world = world;
    return 0;
}
void mi_dp_set_bucket_bytes(Train_ResNet *t, size_t bytes) { ctx_of(t)->bucket_bytes = bytes; }
/* Cross-replica batch norm ("offer sync-BN as an option, default off" -- the reference has none): statistics and
 * the (dbeta, dgamma) sums of every BN layer are all-reduced over the replicas, through a communicator of its own (the
 * gradient buckets' collectives run concurrently on the comm stream).  unique_id: a SECOND id from mi_dp_get_unique_id,
 * broadcast like the first.  With it, DP-N equals one replica at batch N x 256 up to summation order. */
int mi_dp_enable_sync_bn(Train_ResNet *t, const void *unique_id, int bytes) {
    MiCtx *c = ctx_of(t);
    if (!unique_id) { c->sync_bn = 0; mid_bn_set_sync(NULL, 1, NULL, 0, 0); return 0; }
    if (c->sync_bn_comm) { c->sync_bn = 1; return 0; }
    c->sync_bn_comm = mid_rccl_comm_init(c->rank, c->world, unique_id, bytes);
    if (!c->sync_bn_comm) return -1;
    const size_t nf = MI_SYNC_BN_TMP_FLOATS;
    c->sync_bn_tmp = (float *)mid_malloc(nf * sizeof(float));
    mid_bn_set_sync(c->sync_bn_comm, c->world, c->sync_bn_tmp, nf, 1);
    c->sync_bn = 1;
    return 0;
}
int mi_dp_world(const Train_ResNet *t) { return ((MiCtx *)t->backend_ctx)->world; }

/* host-only view of the bucket plan (no GPU touched): the cuts backwards_pass will make for this network */
int mi_debug_dp_plan(const Dims *d, size_t bucket_bytes, size_t *from, size_t *to, int max) { return mi_dp_plan_buckets(d, bucket_bytes, from, to, max); }
size_t mi_debug_arena_floats(const Dims *d) { size_t a; (void)count_locations(d, &a); return a; }
/* the buckets the last backwards_pass actually handed to RCCL (valid until update_parameters) */
int mi_debug_last_buckets(const Train_ResNet *t, size_t *from, size_t *to, int max) {
    const MiCtx *c = (const MiCtx *)t->backend_ctx;
    for (int i = 0; i < c->n_buckets && i < max; i++) { from[i] = c->bk_from[i]; to[i] = c->bk_to[i]; }
    return c->n_buckets;
}
