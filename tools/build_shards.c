/* build_shards -- command-line front-end of mi_build_shard (the reference's BuildShards target, Makefile:30-31, with its
 * literal paths as arguments):  build_shards <partition_dir> <class_dir> <out_dir> <n_shards> [dim_in dim_out nchw|nhwc] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "resnet_mi.h"
int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s <partition_dir> <class_dir> <out_dir> <n_shards> [dim_in dim_out nchw|nhwc]\n", argv[0]); return 2; }
    const int n = atoi(argv[4]), din = argc > 5 ? atoi(argv[5]) : 256, dout = argc > 6 ? atoi(argv[6]) : 224;
    const int layout = (argc > 7 && !strcmp(argv[7], "nhwc")) ? MI_LAYOUT_NHWC : MI_LAYOUT_NCHW;
    for (int s = 0; s < n; s++) {
        char csv[4096];
        snprintf(csv, sizeof csv, "%s/%03d_images.csv", argv[1], s); /* build_training_shards.c:28 */
        printf("Building Shard #%d\n", s);
        const int rc = mi_build_shard(csv, argv[2], argv[3], s, din, dout, layout);
        if (rc < 0) return 1;
        printf("  %d images\n", rc);
    }
    return 0;
}
