/*
 * stream_harness.c -- BASELINE configs[4] with the requests arriving as JPEG files, driven from C99 threads over
 * include/impgpu.h the way a pool of nginx workers would drive it (docs/02 - Configuration.md:18, worker_processes N):
 * every thread takes `batch` files off its share of the request list and runs, per batch,
 *     impgpu_batch_decode_jpeg            cvDecodeImage, bridge.c:545-552
 *     impgpu_batch_resize_mixed           Resize(), bridge.c:588-604, "resize=224,0"
 *     impgpu_batch_encode_jpeg            cvEncodeImage(".jpg"), bridge.c:704      (quality > 0)
 *  or impgpu_batch_download               the raw thumbnails                          (quality = 0)
 * bench.py --stream --jpeg --native starts it: a Python thread pool measures the interpreter's lock as much as the device.
 *
 *   stream_harness <pool.bin> <requests> <threads> <batch> <quality> [warmup_requests] [ahead]
 * ahead = 1: a thread begins the decode of its next batch (impgpu_batch_decode_jpeg_begin) before it finishes the current one
 * (_finish), so its own unstuffing copy overlaps the device's work on the batch before.
 * pool.bin: u32 count, then per file u32 size + bytes.  Prints one JSON line.
 */
#define _POSIX_C_SOURCE 200809L
#include <impgpu.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    int id, nthreads, batch, quality, ahead;
    long first, count;                   /* this thread's requests: first, first + nthreads, ... */
    unsigned char** blobs;
    size_t* sizes;
    int nfiles;
    int rc, bound;
    long done;
    double file_bytes, answer_bytes;
} worker_t;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* worker(void* arg) {
    worker_t* w = (worker_t*)arg;
    const int B = w->batch;
    impgpu_config cfg;
    const unsigned char** blobs = (const unsigned char**)malloc(sizeof(*blobs) * (size_t)B);
    size_t* sizes = (size_t*)malloc(sizeof(size_t) * (size_t)B);
    impgpu_image** imgs = (impgpu_image**)calloc((size_t)B, sizeof(*imgs));
    impgpu_image** outs = (impgpu_image**)calloc((size_t)B, sizeof(*outs));
    int* codes = (int*)malloc(sizeof(int) * (size_t)B);
    impgpu_resize_item* items = (impgpu_resize_item*)malloc(sizeof(*items) * (size_t)B);
    const size_t cap = impgpu_jpeg_encode_bound(224, 224, 3), raw = 224u * 224u * 4u * 4u;
    unsigned char* host = (unsigned char*)impgpu_host_alloc((w->quality ? cap : raw) * (size_t)B);
    unsigned char** datas = (unsigned char**)malloc(sizeof(*datas) * (size_t)B);
    size_t *caps = (size_t*)malloc(sizeof(size_t) * (size_t)B), *lens = (size_t*)malloc(sizeof(size_t) * (size_t)B);
    int* steps = (int*)malloc(sizeof(int) * (size_t)B);
    long at = 0;
    memset(&cfg, 0, sizeof cfg);
    cfg.max_target_w = 2000; cfg.max_target_h = 2000; cfg.max_filters_count = 5;
    w->bound = impgpu_env_bind_thread() == IMP_OK;
    if (!host) { w->rc = IMP_ERROR_MALLOC_FAILED; return NULL; }
    impgpu_jpeg_batch* pending = NULL;                           /* ahead: the batch begun one round early, and its files */
    const unsigned char** blobs2 = (const unsigned char**)malloc(sizeof(*blobs2) * (size_t)B);
    size_t* sizes2 = (size_t*)malloc(sizeof(size_t) * (size_t)B);
    int npending = 0;
    while ((at < w->count || pending) && w->rc == IMP_OK) {
        int n = 0, k, rc = IMP_OK;
        if (w->ahead) {
            /* this round's batch was begun last round (or is begun now, the first time); the next one is begun before the wait */
            const unsigned char** tb; size_t* ts;
            if (!pending) {
                for (; npending < B && at < w->count; npending++, at++) {
                    const long req = w->first + at * w->nthreads;
                    blobs2[npending] = w->blobs[req % w->nfiles]; sizes2[npending] = w->sizes[req % w->nfiles];
                    w->file_bytes += (double)sizes2[npending];
                }
                rc = impgpu_batch_decode_jpeg_begin(blobs2, sizes2, npending, &pending);
            }
            tb = blobs; blobs = blobs2; blobs2 = tb;                /* `blobs` / `sizes` now name the batch in `pending` */
            ts = sizes; sizes = sizes2; sizes2 = ts;
            n = npending; npending = 0;
            {
                impgpu_jpeg_batch* mine = pending;
                pending = NULL;
                if (rc == IMP_OK && at < w->count) {
                    for (; npending < B && at < w->count; npending++, at++) {
                        const long req = w->first + at * w->nthreads;
                        blobs2[npending] = w->blobs[req % w->nfiles]; sizes2[npending] = w->sizes[req % w->nfiles];
                        w->file_bytes += (double)sizes2[npending];
                    }
                    rc = impgpu_batch_decode_jpeg_begin(blobs2, sizes2, npending, &pending);
                }
                if (mine) { const int rf = impgpu_batch_decode_jpeg_finish(&mine, imgs, codes); if (rc == IMP_OK) rc = rf; }
            }
        } else {
        for (; n < B && at < w->count; n++, at++) {
            const long req = w->first + at * w->nthreads;
            blobs[n] = w->blobs[req % w->nfiles];
            sizes[n] = w->sizes[req % w->nfiles];
            w->file_bytes += (double)sizes[n];
        }
        rc = impgpu_batch_decode_jpeg(blobs, sizes, n, imgs, codes);
        }
        for (k = 0; k < n && rc == IMP_OK; k++) rc = codes[k];
        for (k = 0; k < n && rc == IMP_OK; k++) {
            int ow = 0, oh = 0, ip = 0;
            const int sw = impgpu_image_width(imgs[k]), sh = impgpu_image_height(imgs[k]);
            rc = impgpu_resize_geometry(sw, sh, "224,0", &cfg, 0, &ow, &oh, &ip);
            if (rc == IMP_OK) rc = impgpu_image_create(ow, oh, 3, &outs[k]);
            if (rc != IMP_OK) break;
            items[k].src = impgpu_image_device_ptr(imgs[k]); items[k].src_width = sw; items[k].src_height = sh; items[k].src_step = impgpu_image_step(imgs[k]);
            items[k].dst = impgpu_image_device_ptr(outs[k]); items[k].dst_width = ow; items[k].dst_height = oh; items[k].dst_step = impgpu_image_step(outs[k]);
        }
        if (rc == IMP_OK) rc = impgpu_batch_resize_mixed(items, n, 3, 0, NULL);
        if (rc == IMP_OK && w->quality) {
            for (k = 0; k < n; k++) { datas[k] = host + cap * (size_t)k; caps[k] = cap; }
            rc = impgpu_batch_encode_jpeg((const impgpu_image* const*)outs, n, w->quality, datas, caps, lens, codes);
            for (k = 0; k < n && rc == IMP_OK; k++) { rc = codes[k]; w->answer_bytes += (double)lens[k]; }
        } else if (rc == IMP_OK) {
            for (k = 0; k < n; k++) {
                datas[k] = host + raw * (size_t)k;
                steps[k] = impgpu_image_step(outs[k]);
                w->answer_bytes += (double)steps[k] * impgpu_image_height(outs[k]);
            }
            rc = impgpu_batch_download((const impgpu_image* const*)outs, n, datas, steps);
        }
        for (k = 0; k < n; k++) { impgpu_image_release(&imgs[k]); impgpu_image_release(&outs[k]); }
        if (rc != IMP_OK) w->rc = rc;
        w->done += n;
    }
    impgpu_host_free(host);
    if (pending) { impgpu_batch_decode_jpeg_finish(&pending, imgs, codes); for (int k2 = 0; k2 < B; k2++) impgpu_image_release(&imgs[k2]); }
    free(blobs2); free(sizes2);
    free(blobs); free(sizes); free(imgs); free(outs); free(codes); free(items); free(datas); free(caps); free(lens); free(steps);
    return NULL;
}

static int g_all_bound = 1;

static int run(worker_t* proto, long requests, int nthreads, double* seconds, double* file_bytes, double* answer_bytes) {
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
    worker_t* ws = (worker_t*)malloc(sizeof(worker_t) * (size_t)nthreads);
    int i, rc = IMP_OK;
    const double t0 = now_s();
    for (i = 0; i < nthreads; i++) {
        ws[i] = *proto;
        ws[i].id = i;
        ws[i].first = i;
        ws[i].count = requests / nthreads + (i < requests % nthreads ? 1 : 0);
        pthread_create(&th[i], NULL, worker, &ws[i]);
    }
    for (i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    *seconds = now_s() - t0;
    *file_bytes = *answer_bytes = 0;
    for (i = 0; i < nthreads; i++) {
        if (ws[i].rc != IMP_OK) rc = ws[i].rc;
        if (!ws[i].bound) g_all_bound = 0;
        *file_bytes += ws[i].file_bytes;
        *answer_bytes += ws[i].answer_bytes;
    }
    free(th); free(ws);
    return rc;
}

int main(int argc, char** argv) {
    worker_t proto;
    FILE* f;
    unsigned count = 0, i;
    long requests, warm;
    double s = 0, fb = 0, ab = 0;
    int rc;
    if (argc < 6) { fprintf(stderr, "usage: %s pool.bin requests threads batch quality [warmup]\n", argv[0]); return 2; }
    memset(&proto, 0, sizeof proto);
    requests = strtol(argv[2], NULL, 10);
    proto.nthreads = (int)strtol(argv[3], NULL, 10);
    proto.batch = (int)strtol(argv[4], NULL, 10);
    proto.quality = (int)strtol(argv[5], NULL, 10);
    warm = argc > 6 ? strtol(argv[6], NULL, 10) : 0;
    proto.ahead = argc > 7 ? (int)strtol(argv[7], NULL, 10) : 0;
    if (requests < 1 || proto.nthreads < 1 || proto.batch < 1 || proto.batch > 256) return 2;
    f = fopen(argv[1], "rb");
    if (!f || fread(&count, 4, 1, f) != 1 || count == 0) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    proto.nfiles = (int)count;
    proto.blobs = (unsigned char**)malloc(sizeof(unsigned char*) * count);
    proto.sizes = (size_t*)malloc(sizeof(size_t) * count);
    for (i = 0; i < count; i++) {
        unsigned sz = 0;
        if (fread(&sz, 4, 1, f) != 1) return 2;
        proto.sizes[i] = sz;
        proto.blobs[i] = (unsigned char*)malloc(sz ? sz : 1);
        if (fread(proto.blobs[i], 1, sz, f) != sz) return 2;
    }
    fclose(f);
    if (impgpu_env_start(-1) != IMP_OK) { fprintf(stderr, "impgpu_env_start: %s\n", impgpu_last_error()); return 3; }
    if (warm > 0) {
        rc = run(&proto, warm, proto.nthreads, &s, &fb, &ab);
        if (rc != IMP_OK) { fprintf(stderr, "warm-up failed: %d %s\n", rc, impgpu_last_error()); return 4; }
    }
    rc = run(&proto, requests, proto.nthreads, &s, &fb, &ab);
    if (rc != IMP_OK) { fprintf(stderr, "stream failed: %d %s\n", rc, impgpu_last_error()); return 4; }
    printf("{\"requests\": %ld, \"seconds\": %.6f, \"requests_per_s\": %.1f, \"threads\": %d, \"batch\": %d, \"quality\": %d, "
           "\"file_bytes\": %.0f, \"answer_bytes\": %.0f, \"ahead\": %d, \"numa_node\": %d, \"threads_bound\": %s}\n",
           requests, s, (double)requests / s, proto.nthreads, proto.batch, proto.quality, fb, ab, proto.ahead, impgpu_env_numa_node(), g_all_bound ? "true" : "false");
    impgpu_env_destroy();
    return 0;
}
