// runtime.hip -- HIP runtime behind mi_device.h (memory, streams, events) and the RCCL binding.
// Replaces the cudaMalloc/cudaMemcpy/cudaDeviceSynchronize calls scattered through resnet.cu
// (e.g. :693-704, :1315-1316, :3342) with stream-ordered equivalents.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "mi_device.h"
#include "mi_common.hpp"

static char g_err[512] = "";
void mi_record_error(const char *what, const char *detail) {
    if (g_err[0] == 0) snprintf(g_err, sizeof g_err, "%s: %s", what, detail);
}
// ---- launch trace (diagnostic, off unless RESNET_MI_TRACE=1) ----
#include <signal.h>
#include <unistd.h>
#define MI_TRACE_N 96
static const char *g_trace[MI_TRACE_N];
static unsigned g_trace_n = 0;
static int g_trace_on = -1;
static void (*g_trace_prev)(int) = SIG_DFL; /* e.g. Python's faulthandler: runs after the dump */
static void mi_trace_dump(int sig) {
    const char hdr[] = "\nresnet_mi: aborted; last kernel launches, oldest first:\n";
    if (write(2, hdr, sizeof hdr - 1) < 0) {}
    const unsigned n = g_trace_n < MI_TRACE_N ? g_trace_n : MI_TRACE_N;
    for (unsigned i = 0; i < n; i++) {
        const char *s = g_trace[(g_trace_n - n + i) % MI_TRACE_N];
        if (s && (write(2, "  ", 2) < 0 || write(2, s, strlen(s)) < 0 || write(2, "\n", 1) < 0)) {}
    }
    signal(sig, g_trace_prev == SIG_IGN || g_trace_prev == SIG_ERR ? SIG_DFL : g_trace_prev);
    raise(sig);
}
void mi_trace_launch(const char *name) {
    if (g_trace_on < 0) {
        const char *e = getenv("RESNET_MI_TRACE");
        g_trace_on = e && atoi(e) ? 1 : 0;
        if (g_trace_on && atoi(e) > 1) fprintf(stderr, "resnet_mi: launch trace on\n");
    }
    if (!g_trace_on) return;
    g_trace[g_trace_n++ % MI_TRACE_N] = name;
    /* others (Python's faulthandler, test runners) install SIGABRT handlers of their own later on: stay in front of them */
    struct sigaction cur;
    if (sigaction(SIGABRT, NULL, &cur) == 0 && cur.sa_handler != mi_trace_dump) {
        g_trace_prev = (cur.sa_flags & SA_SIGINFO) ? SIG_DFL : cur.sa_handler;
        signal(SIGABRT, mi_trace_dump);
    }
}
#define HIPCHK(x)                                                          \
    do {                                                                   \
        hipError_t e_ = (x);                                               \
        if (e_ != hipSuccess) mi_record_error(#x, hipGetErrorString(e_)); \
    } while (0)

extern "C" {
const char *mid_last_error(void) { return g_err; }
void mid_clear_error(void) { g_err[0] = 0; }
void mi_record_host_error(const char *what, const char *detail) { mi_record_error(what, detail); }
int mid_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int mid_set_device(int dev) {
    hipError_t e = hipSetDevice(dev);
    if (e != hipSuccess) { mi_record_error("hipSetDevice", hipGetErrorString(e)); return -1; }
    return 0;
}
void *mid_malloc(size_t bytes) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 4);
    if (e != hipSuccess) { mi_record_error("hipMalloc", hipGetErrorString(e)); return nullptr; }
    return p;
}
void mid_free(void *p) { if (p) HIPCHK(hipFree(p)); }
void *mid_malloc_host(size_t bytes) {
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 4, hipHostMallocDefault);
    if (e != hipSuccess) { mi_record_error("hipHostMalloc", hipGetErrorString(e)); return nullptr; }
    return p;
}
void mid_free_host(void *p) { if (p) HIPCHK(hipHostFree(p)); }
void mid_memcpy_h2d(void *d, const void *s, size_t n, mid_stream st) { HIPCHK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, (hipStream_t)st)); }
void mid_memcpy_d2h(void *d, const void *s, size_t n, mid_stream st) { HIPCHK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, (hipStream_t)st)); }
void mid_memcpy_d2d(void *d, const void *s, size_t n, mid_stream st) { HIPCHK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, (hipStream_t)st)); }
void mid_memset(void *d, int b, size_t n, mid_stream st) { HIPCHK(hipMemsetAsync(d, b, n, (hipStream_t)st)); }
mid_stream mid_stream_create(void) {
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return (mid_stream)s;
}
/* a stream whose workgroups are dispatched only when the normal-priority streams leave capacity */
mid_stream mid_stream_create_low_priority(void) {
    int lo = 0, hi = 0;
    hipStream_t s = nullptr;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; }
    HIPCHK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo)); /* numerically greatest = lowest priority */
    return (mid_stream)s;
}
void mid_stream_destroy(mid_stream s) { if (s) HIPCHK(hipStreamDestroy((hipStream_t)s)); }
void mid_stream_sync(mid_stream s) { HIPCHK(hipStreamSynchronize((hipStream_t)s)); }
void mid_device_sync(void) { HIPCHK(hipDeviceSynchronize()); }
mid_event mid_event_create(void) {
    hipEvent_t e = nullptr;
    HIPCHK(hipEventCreate(&e));
    return (mid_event)e;
}
void mid_event_destroy(mid_event e) { if (e) HIPCHK(hipEventDestroy((hipEvent_t)e)); }
void mid_event_record(mid_event e, mid_stream s) { HIPCHK(hipEventRecord((hipEvent_t)e, (hipStream_t)s)); }
void mid_stream_wait_event(mid_stream s, mid_event e) { HIPCHK(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0)); }
void mid_event_sync(mid_event e) { HIPCHK(hipEventSynchronize((hipEvent_t)e)); }
float mid_event_elapsed_ms(mid_event a, mid_event b) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b));
    return ms;
}

} // extern "C"

// ---------------- per-family kernel timing ----------------
#define PROF_MAX 8192
static struct {
    int on, n, open;
    hipEvent_t a[PROF_MAX], b[PROF_MAX];
    int fam[PROF_MAX];
    long launches[MI_FAM_COUNT];
    double ms[MI_FAM_COUNT], flops[MI_FAM_COUNT], bytes[MI_FAM_COUNT];
    int created;
} P;
static void prof_resolve(void) {
    for (int i = 0; i < P.n; i++) {
        float ms = 0;
        if (hipEventSynchronize(P.b[i]) == hipSuccess && hipEventElapsedTime(&ms, P.a[i], P.b[i]) == hipSuccess) P.ms[P.fam[i]] += ms;
    }
    P.n = 0;
}
void mi_prof_begin(hipStream_t st, int fam, double flops, double bytes) {
    if (!(P.on & (1 << fam))) return;
    if (P.n == PROF_MAX) prof_resolve();
    const int i = P.n;
    if (i >= P.created) { HIPCHK(hipEventCreate(&P.a[i])); HIPCHK(hipEventCreate(&P.b[i])); P.created = i + 1; }
    P.fam[i] = fam; P.launches[fam]++; P.flops[fam] += flops; P.bytes[fam] += bytes;
    HIPCHK(hipEventRecord(P.a[i], st));
    P.open = 1;
}
void mi_prof_end(hipStream_t st) {
    if (!P.on || !P.open) return;
    HIPCHK(hipEventRecord(P.b[P.n], st));
    P.n++; P.open = 0;
}
extern "C" {
void mid_prof_enable(int on) { P.on = on == 1 ? 0xff : on; } /* 1 = all families, else a bit mask (1 << family) */
void mid_prof_reset(void) {
    prof_resolve();
    for (int f = 0; f < MI_FAM_COUNT; f++) { P.launches[f] = 0; P.ms[f] = P.flops[f] = P.bytes[f] = 0; }
}
void mid_prof_get(int family, long *launches, double *ms, double *flops, double *bytes) {
    prof_resolve();
    if (family < 0 || family >= MI_FAM_COUNT) return;
    if (launches) *launches = P.launches[family];
    if (ms) *ms = P.ms[family];
    if (flops) *flops = P.flops[family];
    if (bytes) *bytes = P.bytes[family];
}
}

extern "C" {
// ---------------- RCCL through dlopen: no link-time dependency, one copy per process ----------------
typedef struct { char internal[128]; } rcclUniqueId;
typedef int (*fn_getuid)(rcclUniqueId *);
typedef int (*fn_cominit)(void **, int, rcclUniqueId, int);
typedef int (*fn_allreduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_destroy)(void *);
typedef const char *(*fn_errstr)(int);
static struct { void *h; fn_getuid getuid; fn_cominit init; fn_allreduce allreduce; fn_destroy destroy; fn_destroy abort; fn_errstr errstr; } R;
static int rccl_load(void) {
    if (R.h) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int i = 0; i < 3 && !R.h; i++) R.h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!R.h) { mi_record_error("dlopen(librccl)", dlerror()); return -1; }
    R.getuid = (fn_getuid)dlsym(R.h, "ncclGetUniqueId");
    R.init = (fn_cominit)dlsym(R.h, "ncclCommInitRank");
    R.allreduce = (fn_allreduce)dlsym(R.h, "ncclAllReduce");
    R.destroy = (fn_destroy)dlsym(R.h, "ncclCommDestroy");
    R.abort = (fn_destroy)dlsym(R.h, "ncclCommAbort");
    R.errstr = (fn_errstr)dlsym(R.h, "ncclGetErrorString");
    if (!R.getuid || !R.init || !R.allreduce || !R.destroy) { mi_record_error("dlsym(rccl)", "missing symbol"); return -1; }
    return 0;
}
#define RCCLCHK(x, what)                                                               \
    do {                                                                               \
        int r_ = (x);                                                                  \
        if (r_ != 0) { mi_record_error(what, R.errstr ? R.errstr(r_) : "rccl error"); return -1; } \
    } while (0)
int mid_rccl_unique_id_bytes(void) { return (int)sizeof(rcclUniqueId); }
int mid_rccl_get_unique_id(void *out, int bytes) {
    if (bytes < (int)sizeof(rcclUniqueId) || rccl_load()) return -1;
    rcclUniqueId id;
    RCCLCHK(R.getuid(&id), "ncclGetUniqueId");
    memcpy(out, &id, sizeof id);
    return 0;
}
void *mid_rccl_comm_init(int rank, int world, const void *uid, int bytes) {
    if (bytes < (int)sizeof(rcclUniqueId) || rccl_load()) return nullptr;
    rcclUniqueId id;
    memcpy(&id, uid, sizeof id);
    void *comm = nullptr;
    int r = R.init(&comm, world, id, rank);
    if (r != 0) { mi_record_error("ncclCommInitRank", R.errstr ? R.errstr(r) : "rccl error"); return nullptr; }
    return comm;
}
int mid_rccl_allreduce_sum(void *comm, float *buf, size_t count, mid_stream s) {
    /* ncclFloat32 = 7, ncclSum = 0; in place */
    RCCLCHK(R.allreduce(buf, buf, count, 7, 0, comm, (hipStream_t)s), "ncclAllReduce");
    return 0;
}
void mid_rccl_comm_destroy(void *comm) { if (comm && R.destroy) R.destroy(comm); }
/* a rank that is about to exit on an error: tears the communicator down so that its peers' collectives fail instead of hanging */
void mid_rccl_comm_abort(void *comm) { if (comm && R.abort) R.abort(comm); }
}
