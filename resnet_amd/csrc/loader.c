/*
 * loader.c -- Batch state and load_new_batch (resnet.cu:1196-1325) plus class metadata (resnet.cu:1328-1381).
 * The reference keeps a whole shard in host RAM, memcpy's one batch into pinned memory and does a blocking
 * H2D copy; shard files are raw fp32 images + int32 labels (build_training_shards.c:150-160), NHWC in the
 * legacy directory and NCHW under nchw/.  Same behaviour here, with the data source selectable (shards,
 * a dumped images.buffer/labels.buffer pair, caller-filled host buffers, or a seeded synthetic pool that
 * stays resident in HBM) because the reference's /mnt/storage paths are literals.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mi_host.h"

static BatchExt *g_ext = NULL;
BatchExt *mi_batch_ext(Batch *b) {
    for (BatchExt *e = g_ext; e; e = e->next) if (e->batch == b) return e;
    BatchExt *e = (BatchExt *)calloc(1, sizeof(BatchExt));
    e->batch = b; e->source = MI_SRC_SHARDS; e->layout = MI_LAYOUT_NCHW; e->world = 1;
    e->shard_dir = strdup("/mnt/storage/data/vision/imagenet/2012/train_data_shards"); /* resnet.cu:1275 */
    e->next = g_ext; g_ext = e;
    return e;
}
void mi_batch_ext_free(Batch *b) {
    if (!b) return;
    BatchExt **pp = &g_ext;
    while (*pp && (*pp)->batch != b) pp = &(*pp)->next;
    if (*pp) {
        BatchExt *e = *pp;
        *pp = e->next;
        free(e->shard_dir); free(e->images_path); free(e->labels_path);
        mid_free(e->pool_images); mid_free(e->pool_labels); mid_free(e->stage_dev); free(e->pool_labels_host);
        mid_free(e->images_next); mid_free(e->stage_next); mid_free(e->labels_next);
        mid_free_host(e->pinned_next); mid_free_host(e->labels_next_host);
        if (e->ev_next) mid_event_destroy(e->ev_next);
        if (e->ev_compute) mid_event_destroy(e->ev_compute);
        free(e);
    }
    mid_free_host(b->images_float_cpu); mid_free_host(b->correct_classes_cpu);
    mid_free(b->images); mid_free(b->correct_classes);
    free(b->full_shard_images); free(b->full_shard_correct_classes);
    free(b);
}

/* resnet.cu:1196-1231.  The shard-sized host buffers are allocated at the first shard load. */
Batch *init_general_batch(int n_images, int image_size, int image_dim, int shard_n_images) {
    Batch *b = (Batch *)calloc(1, sizeof(Batch));
    b->n_images = n_images; b->image_size = image_size; b->image_dim = image_dim;
    b->images_float_cpu = (float *)mid_malloc_host((size_t)n_images * image_size * sizeof(float));
    b->images = (float *)mid_malloc((size_t)n_images * image_size * sizeof(float));
    b->correct_classes_cpu = (int *)mid_malloc_host((size_t)n_images * sizeof(int));
    b->correct_classes = (int *)mid_malloc((size_t)n_images * sizeof(int));
    b->cur_shard_id = -1; b->cur_batch_in_shard = -1; b->shard_n_images = shard_n_images;
    mi_batch_ext(b);
    return b;
}

static void set_str(char **dst, const char *s) { free(*dst); *dst = s ? strdup(s) : NULL; }
void mi_batch_source_shards(Batch *b, const char *dir, int layout) {
    BatchExt *e = mi_batch_ext(b);
    e->source = MI_SRC_SHARDS; e->layout = layout; set_str(&e->shard_dir, dir);
    e->have_next = 0;
}
/* double-buffered H2D: while step t runs, batch t+1 of the resident shard goes pinned -> device on the copy stream
 * (the reference copies synchronously at the top of every step, resnet.cu:1315-1316) */
void mi_batch_set_prefetch(Batch *b, int on) {
    BatchExt *e = mi_batch_ext(b);
    e->prefetch = on; e->have_next = 0;
    if (on && !e->images_next) {
        const size_t bytes = (size_t)b->n_images * b->image_size * sizeof(float);
        e->images_next = (float *)mid_malloc(bytes);
        e->stage_next = (float *)mid_malloc(bytes);
        e->pinned_next = (float *)mid_malloc_host(bytes);
        e->labels_next = (int *)mid_malloc((size_t)b->n_images * sizeof(int));
        e->labels_next_host = (int *)mid_malloc_host((size_t)b->n_images * sizeof(int));
        e->ev_next = mid_event_create();
        e->ev_compute = mid_event_create();
    }
}
/* enqueue batch (shard resident in host RAM, index bi) on the copy stream into the *_next buffers */
static void prefetch_enqueue(Batch *b, BatchExt *e, int bi) {
    MiGlobal *g = mi_global();
    const int N = b->n_images;
    const size_t px = (size_t)N * b->image_size, bytes = px * sizeof(float);
    memcpy(e->pinned_next, b->full_shard_images + (size_t)bi * px, bytes);
    memcpy(e->labels_next_host, b->full_shard_correct_classes + (size_t)bi * N, (size_t)N * sizeof(int));
    /* images_next / labels_next were the PREVIOUS step's batch until the swap a moment ago: that step's backward (stem weight
     * gradient) and update_parameters' input_reset memsets may still be queued on the compute stream against them, and
     * update_parameters no longer synchronises the host.  The copy stream therefore waits for everything the compute stream
     * holds right now before it overwrites the buffers (an event, not a host sync: the caller goes on queueing). */
    mid_event_record(e->ev_compute, g->compute);
    mid_stream_wait_event(g->copy, e->ev_compute);
    if (e->layout == MI_LAYOUT_NHWC) {
        mid_memcpy_h2d(e->stage_next, e->pinned_next, bytes, g->copy);
        mid_nhwc_to_nchw(g->copy, e->stage_next, e->images_next, N, b->image_dim, b->image_dim, b->image_size / (b->image_dim * b->image_dim));
    } else mid_memcpy_h2d(e->images_next, e->pinned_next, bytes, g->copy);
    mid_memcpy_h2d(e->labels_next, e->labels_next_host, (size_t)N * sizeof(int), g->copy);
    mid_event_record(e->ev_next, g->copy);
    e->have_next = 1; e->next_shard_id = b->cur_shard_id; e->next_batch_in_shard = bi;
}
void mi_batch_source_buffer(Batch *b, const char *images_path, const char *labels_path, int layout) {
    BatchExt *e = mi_batch_ext(b);
    e->source = MI_SRC_BUFFER; e->layout = layout; set_str(&e->images_path, images_path); set_str(&e->labels_path, labels_path);
    e->pool_next = 0; /* "not read yet" */
}
void mi_batch_source_host(Batch *b, int layout) {
    BatchExt *e = mi_batch_ext(b);
    e->source = MI_SRC_HOST; e->layout = layout;
}
/* data parallel: global batch g of a shard = images [g*world*N, (g+1)*world*N); rank r reads the r-th N of it.  Every rank
 * advances cur_batch_in_shard by one per step and rolls to the next shard at the same step. */
void mi_batch_set_rank_slice(Batch *b, int rank, int world) {
    BatchExt *e = mi_batch_ext(b);
    if (world < 1) world = 1;
    if (rank < 0 || rank >= world) rank = 0;
    e->rank = rank; e->world = world; e->have_next = 0;
}
int mi_batch_last_status(const Batch *b) { return mi_batch_ext((Batch *)b)->status; }

/* Synthetic pool: batch j of the pool = stream elements [j*n, (j+1)*n) of the two seeds, generated on the
 * device (NCHW order) and kept in HBM; load_new_batch cycles through it with a D2D copy. */
void mi_batch_source_synthetic(Batch *b, uint64_t seed_images, uint64_t seed_labels, int n_classes, int pool_batches) {
    BatchExt *e = mi_batch_ext(b);
    MiGlobal *g = mi_global();
    if (pool_batches < 1) pool_batches = 1;
    e->source = MI_SRC_SYNTHETIC; e->layout = MI_LAYOUT_NCHW;
    e->seed_images = seed_images; e->seed_labels = seed_labels; e->n_classes = n_classes;
    mid_free(e->pool_images); mid_free(e->pool_labels); free(e->pool_labels_host);
    const size_t per = (size_t)b->n_images * b->image_size;
    e->pool_batches = pool_batches; e->pool_next = 0;
    e->pool_images = (float *)mid_malloc(per * pool_batches * sizeof(float));
    e->pool_labels = (int *)mid_malloc((size_t)b->n_images * pool_batches * sizeof(int));
    e->pool_labels_host = (int *)malloc((size_t)b->n_images * pool_batches * sizeof(int));
    mid_fill_uniform(g->compute, e->pool_images, per * pool_batches, seed_images, 0, -124.0f, 152.0f);
    mi_synth_labels(e->pool_labels_host, (size_t)b->n_images * pool_batches, seed_labels, 0, n_classes);
    mid_memcpy_h2d(e->pool_labels, e->pool_labels_host, (size_t)b->n_images * pool_batches * sizeof(int), g->compute);
    mid_stream_sync(g->compute);
}

static size_t read_file(const char *path, void *dst, size_t elem, size_t count) {
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    const size_t n = fread(dst, elem, count, f);
    fclose(f);
    return n;
}

static void upload(Batch *b, BatchExt *e) {
    MiGlobal *g = mi_global();
    const size_t bytes = (size_t)b->n_images * b->image_size * sizeof(float);
    if (e->layout == MI_LAYOUT_NHWC) {
        if (!e->stage_dev) e->stage_dev = (float *)mid_malloc(bytes);
        mid_memcpy_h2d(e->stage_dev, b->images_float_cpu, bytes, g->compute);
        mid_nhwc_to_nchw(g->compute, e->stage_dev, b->images, b->n_images, b->image_dim, b->image_dim,
                         b->image_size / (b->image_dim * b->image_dim));
    } else {
        mid_memcpy_h2d(b->images, b->images_float_cpu, bytes, g->compute);
    }
    mid_memcpy_h2d(b->correct_classes, b->correct_classes_cpu, (size_t)b->n_images * sizeof(int), g->compute);
    mid_stream_sync(g->compute); /* the reference's cudaMemcpy is blocking (:1315-1316) */
}

/* resnet.cu:1235-1325 */
void load_new_batch(Train_ResNet *trainer, Class_Metadata *class_metadata, Batch *b) {
    (void)class_metadata;
    mi_trainer_poll_errors(trainer); /* check_errors of the step that just ended, while its batch and activations are still there */
    BatchExt *e = mi_batch_ext(b);
    MiGlobal *g = mi_global();
    const int N = b->n_images;
    const size_t total_pixels = (size_t)N * b->image_size;
    e->status = 0;
    if (e->source == MI_SRC_SHARDS) {
        const int W = e->world, R = e->rank;
        /* later variants skip a ragged tail instead of assuming divisibility (resnet_cudnn_lowmem.cu:1293-1297) */
        if (trainer->init_loaded || b->cur_shard_id == -1 || (b->cur_batch_in_shard + 1) * W * N > b->shard_n_images) {
            if (!trainer->init_loaded) b->cur_shard_id += 1;
            if (!b->full_shard_images) {
                b->full_shard_images = (float *)malloc((size_t)b->shard_n_images * b->image_size * sizeof(float));
                b->full_shard_correct_classes = (int *)malloc((size_t)b->shard_n_images * sizeof(int));
            }
            char *pi = NULL, *pl = NULL;
            if (asprintf(&pi, "%s/%03d.images", e->shard_dir, b->cur_shard_id) < 0 || asprintf(&pl, "%s/%03d.labels", e->shard_dir, b->cur_shard_id) < 0) exit(1);
            const size_t ni = read_file(pi, b->full_shard_images, sizeof(float), (size_t)b->shard_n_images * b->image_size);
            const size_t nl = read_file(pl, b->full_shard_correct_classes, sizeof(int), b->shard_n_images);
            if (ni != (size_t)b->shard_n_images * b->image_size || nl != (size_t)b->shard_n_images) {
                fprintf(stderr, "resnet_mi: cannot read shard %s (%zu of %zu floats)\n", pi, ni, (size_t)b->shard_n_images * b->image_size);
                e->status = -1;
            }
            free(pi); free(pl);
            if (!trainer->init_loaded) b->cur_batch_in_shard = 0;
            trainer->init_loaded = 0;
        }
        if (e->status == 0) {
            const int bi = b->cur_batch_in_shard * W + R; /* this rank's batch of the shard */
            if (e->prefetch && e->have_next && e->next_shard_id == b->cur_shard_id && e->next_batch_in_shard == bi) {
                /* batch already on the device: order the compute stream after the copy and swap buffers */
                mid_stream_wait_event(g->compute, e->ev_next);
                mid_event_sync(e->ev_next); /* the pinned staging buffer is rewritten below */
                float *ti = b->images; b->images = e->images_next; e->images_next = ti;
                int *tl = b->correct_classes; b->correct_classes = e->labels_next; e->labels_next = tl;
                memcpy(b->correct_classes_cpu, e->labels_next_host, (size_t)N * sizeof(int));
                e->have_next = 0;
            } else {
                memcpy(b->images_float_cpu, b->full_shard_images + (size_t)bi * total_pixels, total_pixels * sizeof(float));
                memcpy(b->correct_classes_cpu, b->full_shard_correct_classes + (size_t)bi * N, (size_t)N * sizeof(int));
                upload(b, e);
            }
            if (e->prefetch && (b->cur_batch_in_shard + 2) * W * N <= b->shard_n_images) {
                /* the swapped-out buffer may still be read (stem weight gradient) and cleared (input_reset) by the step that
                 * just ended: prefetch_enqueue orders the copy stream behind the compute stream before writing into it */
                prefetch_enqueue(b, e, (b->cur_batch_in_shard + 1) * W + R);
            }
        }
    } else if (e->source == MI_SRC_BUFFER) {
        if (!e->pool_next) {
            const size_t ni = read_file(e->images_path, b->images_float_cpu, sizeof(float), total_pixels);
            const size_t nl = read_file(e->labels_path, b->correct_classes_cpu, sizeof(int), N);
            if (ni != total_pixels || nl != (size_t)N) { fprintf(stderr, "resnet_mi: cannot read %s / %s\n", e->images_path, e->labels_path); e->status = -1; }
            e->pool_next = 1;
        }
        if (e->status == 0) upload(b, e);
    } else if (e->source == MI_SRC_HOST) {
        upload(b, e);
    } else { /* synthetic, resident in HBM */
        const int j = e->pool_next;
        mid_memcpy_d2d(b->images, e->pool_images + (size_t)j * total_pixels, total_pixels * sizeof(float), g->compute);
        mid_memcpy_d2d(b->correct_classes, e->pool_labels + (size_t)j * N, (size_t)N * sizeof(int), g->compute);
        memcpy(b->correct_classes_cpu, e->pool_labels_host + (size_t)j * N, (size_t)N * sizeof(int));
        e->pool_next = (j + 1) % e->pool_batches;
    }
    b->cur_batch_in_shard += 1;
    trainer->cur_dump_id += 1;
}

/* resnet.cu:1331-1381 */
static void text_file_to_buffer(void *buffer, const char *filename, int as_int) {
    FILE *fp = fopen(filename, "r");
    if (!fp) exit(EXIT_FAILURE); /* resnet.cu:1341-1342 */
    char *line = NULL;
    size_t len = 0;
    int cnt = 0;
    while (getline(&line, &len, fp) != -1) {
        if (as_int) ((int *)buffer)[cnt] = atoi(line);
        else ((char **)buffer)[cnt] = strdup(line);
        cnt++;
    }
    fclose(fp);
    free(line);
}
Class_Metadata *populate_class_info(char *label_filename, char *synset_filename, char *class_size_filename, int n_classes) {
    Class_Metadata *c = (Class_Metadata *)malloc(sizeof(Class_Metadata));
    c->labels = (char **)calloc(n_classes, sizeof(char *));
    c->synsets = (char **)calloc(n_classes, sizeof(char *));
    c->counts = (int *)calloc(n_classes, sizeof(int));
    text_file_to_buffer(c->labels, label_filename, 0);
    text_file_to_buffer(c->synsets, synset_filename, 0);
    text_file_to_buffer(c->counts, class_size_filename, 1);
    c->n_classes = n_classes;
    return c;
}
