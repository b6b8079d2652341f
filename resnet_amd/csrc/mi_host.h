/* mi_host.h -- internal host-side state shared by trainer.c / loader.c / dump.c / ops.c (plain C). */
#ifndef MI_HOST_H
#define MI_HOST_H
#include "mi_device.h"
#include "resnet_mi.h"

struct MiRng { uint64_t seed; uint64_t counter; };

/* splitmix64 counter streams (synth.c) -- identical to tests/synth.py */
uint64_t mi_splitmix64_at(uint64_t seed, uint64_t i);
void mi_synth_uniform(float *out, size_t n, uint64_t seed, uint64_t offset, float lo, float hi);
void mi_synth_normal(float *out, size_t n, uint64_t seed, uint64_t offset, double var);
void mi_synth_labels(int *out, size_t n, uint64_t seed, uint64_t offset, int n_classes);

/* process-wide device state: one compute stream (everything the reference put on the default stream),
 * one communication stream (RCCL), one copy stream (H2D of the next batch) */
typedef struct {
    int ready;
    mid_stream compute, comm, copy, aux; /* aux: weight-gradient kernels, concurrent with the next layer's BN' */
} MiGlobal;
MiGlobal *mi_global(void);

/* data source attached to a Batch (the reference hard-codes its shard path, resnet.cu:1275) */
typedef struct BatchExt {
    Batch *batch;
    int source, layout, status;
    int rank, world; /* data parallel: this rank's slice of every global batch of a shard */
    char *shard_dir, *images_path, *labels_path;
    uint64_t seed_images, seed_labels;
    int n_classes, pool_batches, pool_next;
    float *pool_images; /* device, NCHW, pool_batches * n_images * image_size */
    int *pool_labels;   /* device */
    int *pool_labels_host;
    float *stage_dev;   /* device staging for NHWC -> NCHW */
    /* shard prefetch: batch t+1 is copied to the device on the copy stream while step t computes */
    int prefetch, have_next, next_shard_id, next_batch_in_shard;
    float *images_next, *stage_next, *pinned_next;
    int *labels_next, *labels_next_host;
    mid_event ev_next, ev_compute; /* copy done / compute stream's position when the target buffers were handed to the copy stream */
    uint64_t synth_step;
    struct BatchExt *next;
} BatchExt;
BatchExt *mi_batch_ext(Batch *b);
void mi_batch_ext_free(Batch *b);

/* parity copies of a striding block's conv inputs; *_valid: the last forward pass wrote the planes (the weight gradient may read them) */
typedef struct {
    void *spatial, *proj; size_t spatial_bytes, proj_bytes; int spatial_valid, proj_valid;
    /* the stride-2 dgrads' output gradient re-laid channel-last with a zero row / column at the far end (kernels_cl_bf16.hip); the
     * halo is zeroed once, when the buffers are made; NULL = that layer's dgrad stays on the NCHW kernel */
    void *dye_spatial, *dye_proj;
    void *dy1;                   /* stride-1 blocks: the 3x3's output gradient as a zero-padded channel-last plane (re-laid once per backward pass: dgrad and weight gradient read it) */
    void *cl_s1;                 /* stride-1 blocks: the 3x3's input as ONE zero-padded channel-last plane, written by the reduction BN's apply itself */
    void *cl_spatial, *cl_proj;  /* channel-last parity planes of the layer's input: forward + weight gradient (kernels_cl_bf16.hip) */
} MiParity;
typedef struct MiCtx {
    mid_workspace ws;
    float *bn_ws;
    mid_bn_parts bn_parts; /* statistics partials a forward convolution leaves for its batch norm */
    /* weights re-laid for the implicit-GEMM kernel once per forward pass (one launch): host copy of the table for the
     * per-call lookup by weight pointer, device copies for the kernel */
    mid_wt_entry *wt_tab;
    int wt_n, wt_tiles;
    mid_wt_entry *wt_tab_dev;
    int *wt_tile_entry_dev;
    int fuse_bn_stats;
    int *nan_flag_dev, *nan_flag_host;
    int nan_check_pending;       /* update_parameters queued a copy of the flag; read it at the next host sync point */
    mid_event ev_nan;            /* recorded behind that copy */
    size_t *loc_off_dev; int n_loc; /* arena offsets of locations[] (+ the arena's end) for the Adam kernel's report */
    int nan_location, nan_no_exit;  /* last reported locations[] index (-1 none); test hook: report without exit(1) */
    struct MiCtx *next_live;     /* registry of live contexts (communicator teardown on a fatal error) */
    int full_store, dump_every, input_reset;
    int dtype;                   /* MID_F32 | MID_BF16: storage type of activations and activation gradients */
    int policy;                  /* MI_STORE_FAST | MI_STORE_RECOMPUTE_BN | MI_STORE_FULL */
    int n_persist;               /* allocs[0 .. n_persist) survive a rebuild of the activation buffers */
    size_t act_bytes, dev_bytes; /* forward activations kept for backward / every tracked allocation */
    size_t *alloc_bytes;
    float *rc_buf[2];            /* RECOMPUTE_BN: scratch for the BN(+ReLU) tensors (forward: consumed at once; backward: re-derived) */
    /* bf16: the reduction pass of a unit's BN' done by the dgrad that produces its dy (mid_conv_dgrad_bn_bf16).  backwards_pass
     * fills fz_req before the unit whose dgrad should do it; the unit's dgrad moves it to fz_done (nparts > 0) for the next unit_bwd */
    mid_bn_bwd_parts fz_req, fz_done;
    int cl_pre;                  /* bf16: the stride-2 layers' parity planes are written by the producing BN apply too (RESNET_MI_BF16_CL_PRE=0: by a re-layout pass) */
    int cur_cl_ready;            /* forward: cur_cl already holds this step's planes (written by the producer) */
    void *bn_cl_out; int bn_cl_H; /* forward: the next unit's BN apply also writes this channel-last plane (stride-1 3x3 input), or NULL */
    int stem_bf16;               /* bf16 mode: the stem convolution's output and its gradient are bf16 tensors too (RESNET_MI_BF16_STEM_TENSORS=f32: fp32 as in round 2) */
    int cl_wgrad2;               /* bf16: stride-2 weight gradients with both operands channel-last where the plane does not fill 64-pixel tiles (RESNET_MI_BF16_CL_WGRAD2=0: off) */
    int fz_bf16;                 /* bf16: which dgrads carry a BN' reduction (sites 1 | 2 | 4 as fz_f32; RESNET_MI_BF16_BNFUSE_SITES, default all) */
    int fz_req_valid, fz_ready, fz_enable, fz_f32; /* fz_f32: the fp32 dgrads do it too (RESNET_MI_F32_BNFUSE_BWD, default on) */
    float *stem_dx;              /* bf16 mode: the stem convolution's output gradient stays fp32 */
    void *stem_xp; size_t stem_xp_bytes;           /* bf16 mode: the batch as zero-padded bf16 parity planes (kernels_stem_bf16.hip) */
    float *stem_scratch; size_t stem_scratch_floats; /*            its wave partials + re-laid weights; NULL = the fp32 stem kernels */
    int counting_act;
    int overlap_set;             /* mi_trainer_set_overlap was called: keep the caller's mode */
    MiParity *par;               /* bf16: per block, NULL buffers for blocks that do not stride */
    int params_dirty;            /* update_parameters ran since the last weight re-layout */
    unsigned long host_epoch_seen; /* the process-wide host-write count (mi_copy_to_device) that re-layout was made at */
    void *cur_par; size_t cur_par_bytes; int *cur_par_valid; /* parity buffer of the stride-2 convolution about to be launched */
    void *cur_cl;                /* ... its channel-last parity planes (the forward pass fills them, the weight gradient reads them), or NULL */
    int cur_dye_valid;           /* the stride-2 dgrad of this layer has filled cur_dye (this backward pass): the weight gradient may read it */
    void *cur_dye;               /* ... and the channel-last buffer for its output gradient (stride-2 dgrad), or NULL */
    char *dump_root;
    /* every device allocation of this trainer (freed by destroy_trainer) */
    void **allocs;
    int n_allocs, cap_allocs;
    /* contiguous parameter-shaped arenas with identical offsets: params, grads, m, v */
    size_t arena_floats;
    float *g_arena, *m_arena, *v_arena;
    /* data parallel */
    void *comm;
    int rank, world;
    size_t bucket_bytes;
    size_t dp_cursor; /* floats: gradients [dp_cursor, arena_floats) already handed to RCCL */
    mid_event ev_grads, ev_reduced;
    int dp_pending;
    /* buckets handed to RCCL during the last backwards_pass, in issue order (FC first): Adam of bucket b waits only for
     * bucket b's event, so the early buckets update while the late ones are still on the wire */
#define MI_MAX_BUCKETS 64
    size_t bk_from[MI_MAX_BUCKETS], bk_to[MI_MAX_BUCKETS];
    mid_event bk_ev[MI_MAX_BUCKETS];
    int n_buckets;
    int sync_bn;                 /* cross-replica batch-norm statistics (default off: the reference has none) */
    void *sync_bn_comm;          /* its own communicator: BN collectives run on the compute stream, the buckets' on the comm stream */
    float *sync_bn_tmp;
    /* weight-gradient overlap: wgrad(L) runs on the aux stream next to BN'(L-1); joined before the next dgrad */
    int overlap_wgrad, wgrad_pending; /* 0 serial, 1 wgrad next to the following BN' only, 2 free-running (ring of buffers) */
    mid_event ev_bn_done, ev_wgrad_done;
    /* mode 2: derivative tensors of the backward chain come from a ring; a slot remembers the aux-stream weight gradient
     * that still reads it, and the compute stream waits for exactly that kernel before the slot is written again */
#define MI_RING 10
    float *ring_buf[MI_RING];
    mid_event ring_ev[MI_RING];
    int ring_busy[MI_RING], ring_next;
    float *dpool[6]; /* the fixed U0,U1,A,B,C,D buffers of modes 0/1 */
    /* timing */
    mid_event ev_t[6];
    float last_ms[5];
} MiCtx;

#define MI_GUARD 256 /* bytes of slack in front of and behind every tensor the bf16 kernels read (see aalloc, mi_malloc) */
void *mi_ctx_alloc(MiCtx *c, size_t bytes);
void mi_params_mark_dirty(void);
/* buckets the data-parallel path cuts for a network (host-only arithmetic shared with backwards_pass): fills
 * from[] / to[] (float offsets into the gradient arena, issue order) and returns their number */
int mi_dp_plan_buckets(const Dims *d, size_t bucket_bytes, size_t *from, size_t *to, int max);
size_t mi_params_arena_floats(const Params *p);
float *mi_params_arena_base(const Params *p);
void mi_dp_reduce_ready(Train_ResNet *t, size_t from_float_offset, int force);
void mi_trainer_poll_errors(Train_ResNet *t); /* load_new_batch: wait for and read the NaN / Inf flag of the last update */

#endif
