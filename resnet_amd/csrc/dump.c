/*
 * dump.c -- checkpoint dump / resume in the reference's on-disk format (resnet.cu:2250-2875):
 *   <root>/<dir>/<%08d>/{model_params,gradients,means,vars}/%03d.buffer   raw fp32 per locations[] index
 *   <root>/<dir>/<%08d>/activations|activation_derivs/...                 raw fp32 trees (int32 max_inds)
 *   trainer_metadata.txt, trainer_checkpoint.txt (6 lines, :2743-2750)
 * Image-shaped tensors are written NHWC like the reference's (the device layout is NCHW), so the files are
 * interchangeable with dumps of the reference.  Unlike the reference, directories are created here
 * (it needs build_dirs_for_dumping.ipynb) and a dump that cannot be written is reported and skipped
 * instead of dereferencing a NULL FILE*.  Tensors that the fast path does not keep are skipped
 * (x-hat, BN-out, pre-ReLU sums: MI_STORE_FULL), and so is the image-shaped part of activation_derivs/ unless the policy
 * is MI_STORE_FULL: outside it the derivative tensors share a few rolling buffers (resnet_cudnn_lowmem.cu:2152-2170) and
 * would not hold what their file names say.  bf16 tensors are widened to fp32 on the way out: the files are fp32 always.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include "mi_host.h"

static const char *root_of(Train_ResNet *t) {
    MiCtx *c = (MiCtx *)t->backend_ctx;
    return c->dump_root ? c->dump_root : "/mnt/storage/data/vision/imagenet/training_dumps";
}
static int mkdir_p(const char *path) {
    char *tmp = strdup(path);
    for (char *p = tmp + 1; *p; p++)
        if (*p == '/') { *p = 0; if (mkdir(tmp, 0777) && errno != EEXIST) { free(tmp); return -1; } *p = '/'; }
    int rc = mkdir(tmp, 0777);
    free(tmp);
    return (rc && errno != EEXIST) ? -1 : 0;
}
static int write_dev(const char *dir, const char *name, const void *dev, size_t elems) {
    if (!dev) return 0;
    char *path = NULL;
    if (mkdir_p(dir) || asprintf(&path, "%s/%s", dir, name) < 0) return -1;
    void *host = malloc(elems * 4);
    mi_copy_to_host(host, dev, elems * 4);
    FILE *f = fopen(path, "wb");
    int rc = -1;
    if (f) { rc = fwrite(host, 4, elems, f) == elems ? 0 : -1; fclose(f); }
    free(host); free(path);
    return rc;
}
/* NCHW device tensor (fp32, or bf16 when dt says so) -> NHWC fp32 file */
static float *g_widen = NULL; /* second scratch of the current dump: bf16 -> fp32 before the transposition */
static int write_img_t(const char *dir, const char *name, const float *dev, int N, int C, int H, float *scratch, int dt) {
    if (!dev) return 0;
    if (dt == MID_BF16) {
        mid_bf16_to_f32(mi_global()->compute, dev, g_widen, (size_t)N * C * H * H);
        dev = g_widen;
    }
    mid_nchw_to_nhwc(mi_global()->compute, dev, scratch, N, C, H, H);
    return write_dev(dir, name, scratch, (size_t)N * C * H * H);
}
static int write_img(const char *dir, const char *name, const float *dev, int N, int C, int H, float *scratch) {
    return write_img_t(dir, name, dev, N, C, H, scratch, MID_F32);
}

/* resnet.cu:2250-2317 */
static int dump_parameters(int dump_id, Train_ResNet *t, const char *special_dir) {
    const Params *sets[4] = {t->model->params, t->backprop_buffer->param_derivs, t->backprop_buffer->prev_means, t->backprop_buffer->prev_vars};
    const char *names[4] = {"model_params", "gradients", "means", "vars"};
    int rc = 0;
    for (int s = 0; s < 4; s++) {
        char *dir = NULL;
        if (asprintf(&dir, "%s/%s/%08d/%s", root_of(t), special_dir, dump_id, names[s]) < 0) return -1;
        for (int i = sets[s]->n_locations - 1; i >= 0 && !rc; i--) {
            char nm[32];
            snprintf(nm, sizeof nm, "%03d.buffer", i);
            rc = write_dev(dir, nm, sets[s]->locations[i], sets[s]->sizes[i]);
        }
        free(dir);
    }
    return rc;
}
static void dump_cache(const char *dir, const Cache_BatchNorm *k) {
    if (!k) return;
    write_dev(dir, "means.buffer", k->means, k->feature_size);
    write_dev(dir, "vars.buffer", k->vars, k->feature_size);
}
/* resnet.cu:2321-2680 */
static void dump_activations(int dump_id, Train_ResNet *t, Activations *a, int is_deriv, const char *special_dir) {
    const Dims *d = t->model->dims;
    const int N = t->batch_size, f = d->init_conv_filters, Hs = d->input / d->init_conv_stride, Hp = Hs / d->init_maxpool_stride;
    char *base = NULL, *dir = NULL;
    if (asprintf(&base, "%s/%s/%08d/%s", root_of(t), special_dir, dump_id, is_deriv ? "activation_derivs" : "activations") < 0) return;
    size_t maxe = (size_t)N * f * Hs * Hs;
    if ((size_t)N * t->cur_batch->image_size > maxe) maxe = (size_t)N * t->cur_batch->image_size;
    const MiCtx *ctx = (const MiCtx *)t->backend_ctx;
    const int adt = ctx->dtype;                                   /* storage type of activation tensors */
    const int imgs = !is_deriv || ctx->policy == MI_STORE_FULL;    /* see the header: aliased derivative tensors are not written */
    const int rc = !is_deriv && ctx->policy == MI_STORE_RECOMPUTE_BN; /* the BN(+ReLU) tensors are shared scratch then */
    float *scratch = (float *)mid_malloc(maxe * sizeof(float));
    g_widen = adt == MID_BF16 ? (float *)mid_malloc(maxe * sizeof(float)) : NULL;
    if (!is_deriv) {
        write_img(base, "input.buffer", t->cur_batch->images, N, 3, d->input, scratch);
        write_dev(base, "max_inds.buffer", a->max_inds, (size_t)N * f * Hp * Hp);
        write_dev(base, "correct_classes.buffer", t->cur_batch->correct_classes, N);
        write_dev(base, "softmax.buffer", t->forward_buffer->pred, (size_t)N * d->output);
    } else {
        write_dev(base, "fc_output.buffer", t->backprop_buffer->output_layer_deriv, (size_t)N * d->output);
    }
    if (imgs) {
        write_img_t(base, "init_conv_applied.buffer", a->init_conv_applied, N, f, Hs, scratch, adt == MID_BF16 && ctx->stem_bf16 ? MID_BF16 : MID_F32);
        if (!rc) write_img_t(base, "init_conv_activated.buffer", a->init_conv_activated, N, f, Hs, scratch, adt);
        write_img_t(base, "init_convblock_input.buffer", a->init_convblock_input, N, f, Hp, scratch, adt);
    }
    if (asprintf(&dir, "%s/batch_norms/init", base) >= 0) { dump_cache(dir, a->norm_init_conv); free(dir); }
    for (int i = 0; i < a->n_conv_blocks; i++) {
        const Activation_ConvBlock *k = a->activation_conv_blocks[i];
        const int H = k->incoming_spatial_dim, Ho = H / k->stride, R = k->reduced_depth, X = k->expanded_depth;
        if (asprintf(&dir, "%s/conv_blocks/%02d", base, i) < 0) break;
        if (imgs) {
            write_img_t(dir, "reduction_applied.buffer", k->post_reduced, N, R, H, scratch, adt);
            if (!rc) write_img_t(dir, "reduction_activated.buffer", k->post_reduced_activated, N, R, H, scratch, adt);
            write_img_t(dir, "spatial_applied.buffer", k->post_spatial, N, R, Ho, scratch, adt);
            if (!rc) write_img_t(dir, "spatial_activated.buffer", k->post_spatial_activated, N, R, Ho, scratch, adt);
            write_img_t(dir, "expanded_applied.buffer", k->post_expanded, N, X, Ho, scratch, adt);
            write_img(dir, "expanded_post_norm.buffer", k->post_expanded_norm_vals, N, X, Ho, scratch); /* FULL policy: fp32 */
            write_img_t(dir, "transformed_residual.buffer", k->transformed_residual, N, X, Ho, scratch, adt);
            if (k->output != k->output_activated) write_img_t(dir, "combined_output.buffer", k->output, N, X, Ho, scratch, is_deriv ? adt : MID_F32);
            write_img_t(dir, "output_activated.buffer", k->output_activated, N, X, Ho, scratch, adt);
        }
        free(dir);
        const char *bn[4] = {"reduced", "spatial", "expanded", "projected"};
        const Cache_BatchNorm *kc[4] = {k->norm_post_reduced, k->norm_post_spatial, k->norm_post_expanded, k->norm_post_projection};
        for (int j = 0; j < 4; j++)
            if (kc[j] && asprintf(&dir, "%s/batch_norms/%02d/%s", base, i, bn[j]) >= 0) { dump_cache(dir, kc[j]); free(dir); }
    }
    write_dev(base, "final_avg_pool.buffer", a->final_conv_output_pooled, (size_t)N * d->final_depth);
    if (!is_deriv) write_dev(base, "fc_output.buffer", a->linear_output, (size_t)N * d->output);
    mid_free(scratch);
    mid_free(g_widen); g_widen = NULL;
    free(base);
}
/* resnet.cu:2682-2753 */
static void dump_meta_and_checkpoint(int dump_id, Train_ResNet *t, const char *special_dir) {
    char *dir = NULL, *path = NULL;
    if (asprintf(&dir, "%s/%s/%08d", root_of(t), special_dir, dump_id) < 0 || mkdir_p(dir)) { free(dir); return; }
    if (asprintf(&path, "%s/trainer_metadata.txt", dir) >= 0) {
        FILE *fp = fopen(path, "w");
        if (fp) {
            fprintf(fp, "%d\n%d\n%d\n%d\n", t->batch_size, t->cur_batch->image_size, t->cur_batch->image_dim, t->cur_batch->shard_n_images);
            fprintf(fp, "%f\n%f\n%f\n%f\n%f\n%f\n%f\n", t->learning_rate, t->weight_decay, t->base_mean_decay, t->base_var_decay,
                    t->cur_mean_decay, t->cur_var_decay, t->eps);
            fprintf(fp, "%d\n%d\n%d\n", t->n_epochs, t->cur_dump_id, t->cur_epoch);
            for (int i = 0; i < t->cur_epoch; i++) fprintf(fp, i ? ",%f" : "%f", t->loss_per_epoch[i]);
            fprintf(fp, "\n");
            for (int i = 0; i < t->cur_epoch; i++) fprintf(fp, i ? ",%f" : "%f", t->accuracy_per_epoch[i]);
            fprintf(fp, "\n");
            fclose(fp);
        }
        free(path);
    }
    if (asprintf(&path, "%s/trainer_checkpoint.txt", dir) >= 0) {
        FILE *fp = fopen(path, "w");
        if (fp) {
            fprintf(fp, "%d\n%d\n%f\n%f\n%d\n%d\n", t->cur_batch->cur_shard_id, t->cur_batch->cur_batch_in_shard,
                    t->cur_mean_decay, t->cur_var_decay, t->cur_dump_id, t->cur_epoch);
            fclose(fp);
        }
        free(path);
    }
    free(dir);
}

/* resnet.cu:2755-2772 */
void dump_trainer(int dump_id, Train_ResNet *t, const char *special_dir) {
    if (!special_dir) special_dir = "default";
    if (dump_parameters(dump_id, t, special_dir)) {
        fprintf(stderr, "resnet_mi: cannot write dump %d under %s/%s (skipped)\n", dump_id, root_of(t), special_dir);
        return;
    }
    dump_activations(dump_id, t, t->forward_buffer->activations, 0, special_dir);
    dump_activations(dump_id, t, t->backprop_buffer->activation_derivs, 1, special_dir);
    dump_meta_and_checkpoint(dump_id, t, special_dir);
}

/* resnet.cu:2778-2817 */
void overwrite_trainer_hyperparams(Train_ResNet *t, int dump_id, const char *special_dir) {
    char *path = NULL;
    if (asprintf(&path, "%s/%s/%08d/trainer_checkpoint.txt", root_of(t), special_dir, dump_id) < 0) return;
    FILE *fp = fopen(path, "r");
    free(path);
    if (!fp) { fprintf(stderr, "resnet_mi: no checkpoint %d\n", dump_id); return; }
    int a, b, e, f;
    float c, d;
    if (fscanf(fp, "%d %d %f %f %d %d", &a, &b, &c, &d, &e, &f) == 6) {
        t->cur_batch->cur_shard_id = a; t->cur_batch->cur_batch_in_shard = b;
        t->cur_mean_decay = c; t->cur_var_decay = d; t->cur_dump_id = e; t->cur_epoch = f;
        t->init_loaded = 1;
    }
    fclose(fp);
}
/* resnet.cu:2821-2875 */
void overwrite_model_params(Train_ResNet *t, int dump_id, const char *special_dir) {
    const Params *sets[3] = {t->model->params, t->backprop_buffer->prev_means, t->backprop_buffer->prev_vars};
    const char *names[3] = {"model_params", "means", "vars"};
    for (int i = sets[0]->n_locations - 1; i >= 0; i--) {
        const size_t sz = sets[0]->sizes[i];
        float *host = (float *)malloc(sz * sizeof(float));
        for (int s = 0; s < 3; s++) {
            char *path = NULL;
            if (asprintf(&path, "%s/%s/%08d/%s/%03d.buffer", root_of(t), special_dir, dump_id, names[s], i) < 0) continue;
            FILE *fp = fopen(path, "rb");
            if (fp) {
                if (fread(host, sizeof(float), sz, fp) == sz) mi_copy_to_device(sets[s]->locations[i], host, sz * sizeof(float)); /* marks the re-laid weights dirty */
                fclose(fp);
            } else fprintf(stderr, "resnet_mi: missing %s\n", path);
            free(path);
        }
        free(host);
    }
}
