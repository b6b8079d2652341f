/* synth.c -- seeded synthetic streams: splitmix64 in counter mode so any element of any
 * stream can be produced independently (host here, device in kernels_misc.hip, numpy in tests/synth.py).
 * Stands in for curandGenerateNormal (resnet.cu:52-55) for weight init and for the ImageNet shards for
 * benchmarking; it does NOT reproduce curand's XORWOW stream. */
#include <math.h>
#include "mi_host.h"

uint64_t mi_splitmix64_at(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double u01(uint64_t seed, uint64_t i) { return (double)(mi_splitmix64_at(seed, i) >> 11) * (1.0 / 9007199254740992.0); }

void mi_synth_uniform(float *out, size_t n, uint64_t seed, uint64_t offset, float lo, float hi) {
    for (size_t i = 0; i < n; i++) out[i] = (float)((double)lo + ((double)hi - (double)lo) * u01(seed, offset + i));
}
/* Box-Muller, cosine branch, on the pair (u[2i], u[2i+1]) */
void mi_synth_normal(float *out, size_t n, uint64_t seed, uint64_t offset, double var) {
    const double sd = sqrt(var), two_pi = 2.0 * 3.14159265358979323846;
    for (size_t i = 0; i < n; i++) {
        const double u1 = 1.0 - u01(seed, 2 * (offset + i)), u2 = u01(seed, 2 * (offset + i) + 1);
        out[i] = (float)(sqrt(-2.0 * log(u1)) * cos(two_pi * u2) * sd);
    }
}
void mi_synth_labels(int *out, size_t n, uint64_t seed, uint64_t offset, int n_classes) {
    for (size_t i = 0; i < n; i++) out[i] = (int)(mi_splitmix64_at(seed, offset + i) % (uint64_t)n_classes);
}
