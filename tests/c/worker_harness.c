/*
 * worker_harness.c -- ONE IMP worker process as the reference runs them (docs/02 - Configuration.md:18 worker_processes N;
 * module.c:100-107 OnEnvStart after the fork; module.c:298 / bridge.c:302 RunJob: one request at a time, synchronously):
 *     JPEG file in  ->  resize=224,0  ->  JPEG file out (quality 86)
 * either IN PROCESS (the worker links libimpgpu.so and owns a device context of its own: impgpu_image_decode_jpeg,
 * impgpu_resize, impgpu_image_encode_jpeg) or THROUGH THE BROKER (the worker never touches HIP: impgpu_client_run over
 * include/impgpu_broker.h).  tools/worker_scaling.py starts N of these, releases them together and adds up the lines.
 *
 *   worker_harness <pool.bin> <seconds> <id> <dir> direct|broker[:name] [answers.bin]
 * pool.bin: u32 count, then per file u32 size + bytes (bench.jpeg_pool).  answers.bin: the same layout, the file each
 * request must produce (written by the test with the oracle).  Every file is requested once before the clock starts (all
 * sizes warm), then the worker writes <dir>/ready.<id> and waits for <dir>/go.  Prints one JSON line.
 */
#define _POSIX_C_SOURCE 200809L
#include <impgpu.h>
#include <impgpu_broker.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef struct { unsigned count; unsigned char** blobs; size_t* sizes; } pool_t;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int load(const char* path, pool_t* p) {
    FILE* f = fopen(path, "rb");
    unsigned i;
    if (!f || fread(&p->count, 4, 1, f) != 1 || p->count == 0) return -1;
    p->blobs = (unsigned char**)malloc(sizeof(unsigned char*) * p->count);
    p->sizes = (size_t*)malloc(sizeof(size_t) * p->count);
    for (i = 0; i < p->count; i++) {
        unsigned sz = 0;
        if (fread(&sz, 4, 1, f) != 1) return -1;
        p->sizes[i] = sz;
        p->blobs[i] = (unsigned char*)malloc(sz ? sz : 1);
        if (fread(p->blobs[i], 1, sz, f) != sz) return -1;
    }
    fclose(f);
    return 0;
}

static int cmp_float(const void* a, const void* b) {
    const float x = *(const float*)a, y = *(const float*)b;
    return x < y ? -1 : x > y;
}

static impgpu_client* g_client;
static unsigned char* g_out;
static size_t g_cap;
static impgpu_config g_cfg;
static long g_batch_sum;

/* one request; the answer's bytes in *data / *len */
static int request(int broker, const unsigned char* blob, size_t size, const unsigned char** data, size_t* len) {
    if (broker) {
        impgpu_client_request r;
        impgpu_client_answer a;
        impgpu_job job;
        int rc;
        memset(&r, 0, sizeof r);
        memset(&job, 0, sizeof job);
        job.resize = "224,0";
        r.in_kind = IMPB_IN_FILE; r.input = blob; r.input_bytes = size;
        r.job = &job; r.config = &g_cfg; r.out_kind = IMPB_OUT_JPEG; r.quality = 86;
        rc = impgpu_client_run(g_client, &r, &a);
        if (rc != IMP_OK) { fprintf(stderr, "client: %s\n", impgpu_client_last_error()); return rc; }
        if (a.code != IMP_OK) { fprintf(stderr, "broker answered %d at step %d: %s\n", a.code, a.step, a.error); return a.code > 0 ? a.code : IMP_ERROR_DECODE_FAILED; }
        *data = a.data; *len = a.bytes;
        g_batch_sum += a.batch_size;
        return IMP_OK;
    } else {
        impgpu_image* im = NULL;
        int rc = impgpu_image_decode_jpeg(blob, size, &im);
        if (rc == IMP_OK) rc = impgpu_resize(&im, "224,0", &g_cfg, 0);
        if (rc == IMP_OK) rc = impgpu_image_encode_jpeg(im, 86, g_out, g_cap, len);
        impgpu_image_release(&im);
        *data = g_out;
        g_batch_sum += 1;
        if (rc != IMP_OK) fprintf(stderr, "request failed: %d %s\n", rc, impgpu_last_error());
        return rc;
    }
}

int main(int argc, char** argv) {
    pool_t pool, want;
    double seconds, t0, t1;
    int id, broker, have_want = 0;
    char path[512];
    float* lat;
    long cap_lat = 4000000, n = 0, bad = 0, i;
    unsigned k;
    struct stat st;
    if (argc < 6) { fprintf(stderr, "usage: %s pool.bin seconds id dir direct|broker[:name] [answers.bin]\n", argv[0]); return 2; }
    memset(&pool, 0, sizeof pool); memset(&want, 0, sizeof want);
    if (load(argv[1], &pool)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    seconds = atof(argv[2]);
    id = atoi(argv[3]);
    broker = !strncmp(argv[5], "broker", 6);
    if (argc > 6) { if (load(argv[6], &want) || want.count != pool.count) { fprintf(stderr, "cannot read %s\n", argv[6]); return 2; } have_want = 1; }
    memset(&g_cfg, 0, sizeof g_cfg);
    g_cfg.max_target_w = 2000; g_cfg.max_target_h = 2000; g_cfg.max_filters_count = 5;
    if (broker) {
        const char* name = argv[5][6] == ':' ? argv[5] + 7 : NULL;
        if (impgpu_client_attach(name, &g_client) != IMP_OK) { fprintf(stderr, "attach: %s\n", impgpu_client_last_error()); return 3; }
    } else {
        setenv("IMPGPU_JPEG_HUFF", "device", 1);                 /* every file through the device's entropy stage, like the broker's batches */
        if (impgpu_env_start(-1) != IMP_OK) { fprintf(stderr, "impgpu_env_start: %s\n", impgpu_last_error()); return 3; }
        (void)impgpu_env_bind_thread();
        g_cap = impgpu_jpeg_encode_bound(2000, 2000, 3);
        g_out = (unsigned char*)malloc(g_cap);
    }
    lat = (float*)malloc(sizeof(float) * (size_t)cap_lat);
    /* every size once before the clock (pools, staging, tables) -- and checked, so a wrong answer cannot hide in the warm-up */
    for (k = 0; k < pool.count; k++) {
        const unsigned char* data = NULL; size_t len = 0;
        if (request(broker, pool.blobs[k], pool.sizes[k], &data, &len) != IMP_OK) return 4;
        if (have_want && (len != want.sizes[k] || memcmp(data, want.blobs[k], len))) bad++;
    }
    g_batch_sum = 0;
    snprintf(path, sizeof path, "%s/ready.%d", argv[4], id);
    { FILE* f = fopen(path, "w"); if (f) fclose(f); }
    snprintf(path, sizeof path, "%s/go", argv[4]);
    while (stat(path, &st) != 0) { struct timespec nap = {0, 500000}; nanosleep(&nap, NULL); }
    t0 = now_s();
    t1 = t0;
    while (t1 - t0 < seconds) {
        const unsigned f = (unsigned)((unsigned long)(id * 13 + n * 7) % pool.count);
        const unsigned char* data = NULL; size_t len = 0;
        const double a = t1;
        if (request(broker, pool.blobs[f], pool.sizes[f], &data, &len) != IMP_OK) return 4;
        if (have_want && (len != want.sizes[f] || memcmp(data, want.blobs[f], len))) bad++;
        t1 = now_s();
        if (n < cap_lat) lat[n] = (float)(1e6 * (t1 - a));
        n++;
    }
    {
        const long m = n < cap_lat ? n : cap_lat;
        unsigned long long cnt[4] = {0, 0, 0, 0};
        double mean = 0;
        qsort(lat, (size_t)m, sizeof(float), cmp_float);
        for (i = 0; i < m; i++) mean += lat[i];
        if (!broker) impgpu_jpeg_counters(cnt, 4);
        printf("{\"worker\": %d, \"mode\": \"%s\", \"requests\": %ld, \"seconds\": %.6f, \"mismatches\": %ld, \"checked\": %s, "
               "\"p50_us\": %.1f, \"p95_us\": %.1f, \"p99_us\": %.1f, \"mean_us\": %.1f, \"mean_batch\": %.2f, \"chain_timeouts\": %llu, \"refused\": %llu}\n",
               id, broker ? "broker" : "direct", n, t1 - t0, bad, have_want ? "true" : "false",
               m ? lat[m / 2] : 0.0, m ? lat[(long)(0.95 * (double)(m - 1))] : 0.0, m ? lat[(long)(0.99 * (double)(m - 1))] : 0.0, m ? mean / (double)m : 0.0,
               n ? (double)g_batch_sum / (double)n : 0.0, cnt[2], cnt[1]);
    }
    if (broker) impgpu_client_detach(&g_client);
    else impgpu_env_destroy();
    return bad ? 5 : 0;
}
