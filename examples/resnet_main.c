/*
 * resnet_main.c -- the reference's driver (main() of resnet.cu:3222-3429) rebuilt on libresnet_mi.so: same call
 * sequence, same per-iteration printout (resnet.cu:3386) and avg_loss_log.txt (resnet.cu:3388), with the literals of
 * resnet.cu:3245-3299 exposed as command-line options and the data source selectable (the reference hard-codes
 * /mnt/storage paths).  Host loss/accuracy loop copied in spirit from resnet.cu:3363-3383 (it is the caller's code).
 *
 *   gcc -O2 -Iinclude examples/resnet_main.c -Lresnet_amd -lresnet_mi -lm -Wl,-rpath,$PWD/resnet_amd -o ResNetMI
 *   ./ResNetMI --iters 20 --batch 64                       synthetic data, reference-defined ResNet-50
 *   ./ResNetMI --shards /data/train_data_shards/nchw --layout nchw --shard-images 32768 --batch 256
 *   ./ResNetMI --labels-file id_to_label_mapping.txt --synsets-file id_to_synset_mapping.txt --counts-file id_to_img_count_mapping.txt
 *              the class metadata of resnet.cu:3236-3242: iterations per epoch = ceil(sum of the class counts / batch) (:3309)
 *              unless --iters says otherwise
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "resnet_mi.h"

static const char *opt(int argc, char **argv, const char *name, const char *def) {
    for (int i = 1; i + 1 < argc; i++) if (!strcmp(argv[i], name)) return argv[i + 1];
    return def;
}

int main(int argc, char **argv) {
    const int N_CLASSES = atoi(opt(argc, argv, "--classes", "1000"));
    const int INPUT_DIM = atoi(opt(argc, argv, "--input", "224"));
    const int N_CONV_BLOCKS = atoi(opt(argc, argv, "--blocks", "16"));
    const int BATCH_SIZE = atoi(opt(argc, argv, "--batch", "32"));          /* resnet.cu:3279 */
    int iters = atoi(opt(argc, argv, "--iters", "-1"));                      /* iterations per epoch; default: from the class counts (below), else 10 */
    const int N_EPOCHS = atoi(opt(argc, argv, "--epochs", "1"));              /* resnet.cu:3293 (40) */
    const float LEARNING_RATE = (float)atof(opt(argc, argv, "--lr", "0.0001")); /* resnet.cu:3286-3291 */
    const float WEIGHT_DECAY = (float)atof(opt(argc, argv, "--wd", "0"));
    const float EPS = (float)atof(opt(argc, argv, "--eps", "0.0000001"));
    const int SHARD_N_IMAGES = atoi(opt(argc, argv, "--shard-images", "32768"));
    const char *shards = opt(argc, argv, "--shards", NULL);
    const char *layout = opt(argc, argv, "--layout", "nchw");
    const char *dump_root = opt(argc, argv, "--dump-root", NULL);
    const char *loss_log = opt(argc, argv, "--loss-log", "avg_loss_log.txt");
    const int resume_id = atoi(opt(argc, argv, "--resume", "-1"));           /* LOAD_FROM_DUMP_ID, resnet.cu:3299 */

    /* GETTING CLASS METADATA (resnet.cu:3236-3242): total_images = sum of the per-class image counts */
    char *labels_file = (char *)opt(argc, argv, "--labels-file", NULL), *synsets_file = (char *)opt(argc, argv, "--synsets-file", NULL),
         *counts_file = (char *)opt(argc, argv, "--counts-file", NULL);
    Class_Metadata *class_metadata = NULL;
    int total_images = 0;
    if (labels_file && synsets_file && counts_file) {
        class_metadata = populate_class_info(labels_file, synsets_file, counts_file, N_CLASSES);
        for (int i = 0; i < N_CLASSES; i++) total_images += class_metadata->counts[i];
        printf("class metadata: %d classes, %d images\n", N_CLASSES, total_images);
    }
    if (iters < 0) iters = class_metadata ? (int)ceil((float)total_images / BATCH_SIZE) : 10; /* resnet.cu:3309 */

    if (mi_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 1; }
    int *reductions = (int *)calloc(N_CONV_BLOCKS > 0 ? N_CONV_BLOCKS : 1, sizeof(int));
    int final_depth = 256;
    if (N_CONV_BLOCKS == 16) { reductions[3] = reductions[7] = reductions[13] = 1; final_depth = 2048; } /* :3255-3258 */
    Dims *dims = init_dimensions(INPUT_DIM, 7, 64, 2, 3, 2, N_CONV_BLOCKS, reductions, final_depth, N_CLASSES);
    MiRng *gen = mi_rng_create(1234ULL);                                      /* :3264-3267 */
    ResNet *model = init_resnet(dims, gen);
    Batch *batch = init_general_batch(BATCH_SIZE, INPUT_DIM * INPUT_DIM * 3, INPUT_DIM, SHARD_N_IMAGES);
    if (shards) mi_batch_source_shards(batch, shards, !strcmp(layout, "nhwc") ? MI_LAYOUT_NHWC : MI_LAYOUT_NCHW);
    else mi_batch_source_synthetic(batch, 1234, 1235, N_CLASSES, 4);
    if (shards) mi_batch_set_prefetch(batch, 1);
    Train_ResNet *trainer = init_trainer(model, batch, BATCH_SIZE, LEARNING_RATE, WEIGHT_DECAY, 0.9f, 0.999f, EPS, N_EPOCHS, "my_custom");
    if (dump_root) mi_trainer_set_dump_root(trainer, dump_root); else mi_trainer_set_dump_every(trainer, 0);
    if (resume_id != -1) { overwrite_trainer_hyperparams(trainer, resume_id, "my_custom"); overwrite_model_params(trainer, resume_id, "my_custom"); }

    FILE *loss_file = fopen(loss_log, "w");
    printf("iterations per epoch: %d\n", iters);
    /* the epoch loop of resnet.cu:3327-3421, including the restart position after a resume (:3324-3325) */
    const int iterations_per_epoch = iters;
    const float total_images_per_epoch = (float)BATCH_SIZE * iterations_per_epoch;
    int cur_iter_in_epoch = (trainer->cur_dump_id + 1) % iterations_per_epoch;
    int stop = 0;
    for (int epoch = trainer->cur_epoch; epoch < N_EPOCHS && !stop; epoch++) {
        float epoch_loss = 0, epoch_n_wrong = 0;
        for (int iter = cur_iter_in_epoch; iter < iterations_per_epoch; iter++) {
            load_new_batch(trainer, class_metadata, trainer->cur_batch);
            if (mi_batch_last_status(trainer->cur_batch)) { fprintf(stderr, "data source exhausted\n"); stop = 1; break; }
            forward_pass(trainer);
            const float *pred = trainer->forward_buffer->pred_cpu;
            const int *correct = trainer->cur_batch->correct_classes_cpu;
            float batch_loss = 0, batch_n_wrong = 0;
            for (int s = 0; s < BATCH_SIZE; s++) batch_loss += -1 * logf(pred[s * N_CLASSES + correct[s]]);
            for (int s = 0; s < BATCH_SIZE; s++) {
                const float v = pred[s * N_CLASSES + correct[s]];
                for (int c = 0; c < N_CLASSES; c++)
                    if (c != correct[s] && pred[s * N_CLASSES + c] >= v) { batch_n_wrong++; break; }
            }
            epoch_loss += batch_loss; epoch_n_wrong += batch_n_wrong;
            const float avg = batch_loss / BATCH_SIZE, acc = 100 * ((float)BATCH_SIZE - batch_n_wrong) / (float)BATCH_SIZE;
            printf("\nEpoch: %d, Batch: %d ----- Avg. Loss: %.4f, Accuracy: %.2f%%\n\n", epoch, iter, avg, acc);
            if (loss_file) { fprintf(loss_file, "%.4f\n", avg); fflush(loss_file); }
            backwards_pass(trainer);
            update_parameters(trainer);
            if (mi_last_error()[0]) { fprintf(stderr, "device error: %s\n", mi_last_error()); return 2; }
        }
        if (stop) break;
        /* resnet.cu:3410-3421: per-epoch loss (a SUM over the epoch) and accuracy, rewind the data source */
        mi_trainer_end_epoch(trainer, epoch_loss, epoch_n_wrong, total_images_per_epoch);
        cur_iter_in_epoch = 0;
    }
    mi_trainer_check_errors(trainer);
    if (dump_root) dump_trainer(77777777, trainer, trainer->dump_dir);          /* :3424-3425 */
    if (loss_file) fclose(loss_file);
    destroy_trainer(trainer);
    free(reductions);
    mi_rng_destroy(gen);
    return 0;
}
