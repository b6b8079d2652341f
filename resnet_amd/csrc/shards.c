/*
 * shards.c -- the offline shard writer of the reference's data path (build_training_shards.c:13-167) behind the C-ABI,
 * with the reference's literal /mnt/storage paths turned into arguments.
 *
 *   partition CSV   one image per line, fixed columns "CCC,NNNN,RR,CC": class id, image number inside the class file, row
 *                   and column offset of the crop (build_training_shards.c:41-50 parses by column position)
 *   class files     <class_dir>/%08d.buffer: raw uint8, image_dim_in x image_dim_in x 3 per image, pixels interleaved B,G,R
 *   crop            image_dim_out rows of image_dim_out pixels starting at (row_off, col_off)            (:88-96)
 *   float + swap    B,G,R bytes -> R,G,B floats with 103.94 / 116.78 / 123.68 subtracted (the subtraction is done in double
 *                   and rounded once, as the reference's `((float) byte) - 123.68` does)                 (:115-129)
 *   layout          NHWC -> NCHW                                                                         (:132-144)
 *   files           <out_dir>/%03d.images (fp32) and %03d.labels (int32)                                 (:150-160)
 * Differences: every fopen is checked, the staging buffers are sized by the rows actually present (the reference mallocs a
 * full 32768-image shard of bytes up front, :20), and layout NHWC can be kept for the legacy loader directory.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "resnet_mi.h"

static int field(const char *line, int off, int len) {
    char tmp[16];
    memcpy(tmp, line + off, (size_t)len);
    tmp[len] = 0;
    return atoi(tmp);
}

int mi_build_shard(const char *partition_csv, const char *class_dir, const char *out_dir, int shard_id, int image_dim_in,
                   int image_dim_out, int layout) {
    const long channels = 3;
    const size_t size_in = (size_t)image_dim_in * image_dim_in * channels, size_out = (size_t)image_dim_out * image_dim_out * channels;
    FILE *fp = fopen(partition_csv, "r");
    if (!fp) { fprintf(stderr, "mi_build_shard: cannot open %s\n", partition_csv); return -1; }
    int cap = 1024, cnt = 0;
    int *cls = (int *)malloc(sizeof(int) * cap), *num = (int *)malloc(sizeof(int) * cap), *ro = (int *)malloc(sizeof(int) * cap),
        *co = (int *)malloc(sizeof(int) * cap);
    char *line = NULL;
    size_t len = 0;
    ssize_t got;
    while ((got = getline(&line, &len, fp)) != -1) {
        if (got < 14) continue; /* the reference would read past a short line; skip it */
        if (cnt == cap) {
            cap *= 2;
            cls = (int *)realloc(cls, sizeof(int) * cap); num = (int *)realloc(num, sizeof(int) * cap);
            ro = (int *)realloc(ro, sizeof(int) * cap); co = (int *)realloc(co, sizeof(int) * cap);
        }
        cls[cnt] = field(line, 0, 3); num[cnt] = field(line, 4, 4); ro[cnt] = field(line, 9, 2); co[cnt] = field(line, 12, 2);
        cnt++;
    }
    free(line);
    fclose(fp);

    int rc = 0;
    uint8_t *bytes = (uint8_t *)malloc(size_out * (size_t)(cnt > 0 ? cnt : 1));
    const size_t row_bytes = (size_t)image_dim_out * channels;
    for (int i = 0; i < cnt && !rc; i++) {
        char path[4096];
        snprintf(path, sizeof path, "%s/%08d.buffer", class_dir, cls[i]);
        FILE *f = fopen(path, "rb");
        if (!f) { fprintf(stderr, "mi_build_shard: cannot open class file %s\n", path); rc = -2; break; }
        if (ro[i] + image_dim_out > image_dim_in || co[i] + image_dim_out > image_dim_in) { fclose(f); rc = -3; break; }
        for (int r = 0; r < image_dim_out; r++) {
            const long off = ((long)(ro[i] + r) * image_dim_in + co[i]) * channels;
            if (fseek(f, (long)num[i] * (long)size_in + off, SEEK_SET) ||
                fread(bytes + (size_t)i * size_out + (size_t)r * row_bytes, 1, row_bytes, f) != row_bytes) { rc = -4; break; }
        }
        fclose(f);
    }
    if (!rc) {
        const size_t total = size_out * (size_t)cnt;
        float *out = (float *)malloc(sizeof(float) * (total > 0 ? total : 1));
        const size_t plane = (size_t)image_dim_out * image_dim_out;
        static const double mean_of_src[3] = {123.68, 116.78, 103.94}; /* subtracted from source byte 0 (B), 1 (G), 2 (R) */
        for (size_t px = 0; px < total; px++) {
            const size_t n = px / size_out, within = px % size_out, pix = within / 3;
            const int src_c = (int)(within % 3), dst_c = 2 - src_c; /* B,G,R -> position 2,1,0 */
            const float v = (float)((double)(float)bytes[px] - mean_of_src[src_c]);
            if (layout == MI_LAYOUT_NCHW) out[n * size_out + (size_t)dst_c * plane + pix] = v;
            else out[n * size_out + pix * 3 + (size_t)dst_c] = v;
        }
        char path[4096];
        snprintf(path, sizeof path, "%s/%03d.images", out_dir, shard_id);
        FILE *fi = fopen(path, "wb");
        if (!fi || fwrite(out, sizeof(float), total, fi) != total) rc = -5;
        if (fi) fclose(fi);
        snprintf(path, sizeof path, "%s/%03d.labels", out_dir, shard_id);
        FILE *fl = fopen(path, "wb");
        if (!fl || fwrite(cls, sizeof(int), (size_t)cnt, fl) != (size_t)cnt) rc = -5;
        if (fl) fclose(fl);
        free(out);
    }
    free(bytes); free(cls); free(num); free(ro); free(co);
    return rc ? rc : cnt;
}
