/* mnt_redirect.c -- TEST INFRASTRUCTURE (like everything under oracle/).
 * The reference's shard builder is the one part of the data path that is plain C (build_training_shards.c, gcc target
 * BuildShards, Makefile:30-31), so oracle/Makefile compiles it UNMODIFIED from /root/reference into oracle/_ref/.  Its file
 * names are string literals under /mnt/storage (:28, :78, :150, :156), a path this container may not touch: this preloaded
 * shim rewrites that prefix to $MI_REF_ROOT at fopen time, creates missing output directories (the reference expects them to
 * exist), and hands an empty file to a read of a partition file that does not exist (the reference loops over 40 shards
 * without checking fopen).  Nothing else of the program is touched. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
static FILE *(*real_fopen)(const char *, const char *);
static void mkdirs(char *path) {
    for (char *p = path + 1; *p; p++)
        if (*p == '/') { *p = 0; mkdir(path, 0777); *p = '/'; }
}
FILE *fopen(const char *path, const char *mode) {
    if (!real_fopen) real_fopen = (FILE * (*)(const char *, const char *)) dlsym(RTLD_NEXT, "fopen");
    const char *root = getenv("MI_REF_ROOT");
    static const char pre[] = "/mnt/storage/";
    if (root && !strncmp(path, pre, sizeof pre - 1)) {
        char buf[4096];
        snprintf(buf, sizeof buf, "%s/%s", root, path + sizeof pre - 1);
        if (mode[0] == 'w') mkdirs(buf);
        FILE *f = real_fopen(buf, mode);
        if (!f && mode[0] == 'r' && strstr(path, "_images.csv")) f = real_fopen("/dev/null", "r");
        return f;
    }
    return real_fopen(path, mode);
}
