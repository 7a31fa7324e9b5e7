/*
 * latency_harness.c -- one request at a time from C, as an nginx worker that links libimpgpu.so runs them (RunJob,
 * bridge.c:302): JPEG file in -> resize=224,0 -> JPEG file out, in the two forms the library offers:
 *   two waits   impgpu_image_decode_jpeg (waits for the verdict) -> impgpu_resize -> impgpu_image_encode_jpeg (waits)
 *   one wait    impgpu_batch_decode_jpeg_prepared_begin -> impgpu_batch_decode_jpeg_pending (the frame, ahead of its verdict)
 *               -> impgpu_resize -> impgpu_batch_encode_jpeg_begin -> ..._decode_jpeg_finish -> ..._encode_jpeg_finish
 *   latency_harness <file.jpg> [repetitions]
 * Prints the two medians and p95s in microseconds and whether the two answers are the same file.
 */
#define _POSIX_C_SOURCE 200809L
#include <impgpu.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_us(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return 1e6 * (double)ts.tv_sec + 1e-3 * (double)ts.tv_nsec; }
static int cmp(const void* a, const void* b) { const double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : x > y; }

static unsigned char* g_blob; static size_t g_size;
static unsigned char g_out[2][1 << 20]; static size_t g_len[2];
static impgpu_config g_cfg;

static int two_waits(void) {
    impgpu_image* im = NULL;
    int rc = impgpu_image_decode_jpeg(g_blob, g_size, &im);
    if (rc) return rc;
    rc = impgpu_resize(&im, "224,0", &g_cfg, 0);
    if (!rc) rc = impgpu_image_encode_jpeg(im, 86, g_out[0], sizeof g_out[0], &g_len[0]);
    impgpu_image_release(&im);
    return rc;
}

static int one_wait(void) {
    impgpu_jpeg_prepared f; memset(&f, 0, sizeof f);
    f.head = g_blob; f.head_size = g_size;
    impgpu_jpeg_batch* b = NULL;
    int rc = impgpu_batch_decode_jpeg_prepared_begin(&f, 1, &b);
    if (rc) return rc;
    impgpu_image* im = NULL; impgpu_image* none = NULL; int code = 0;
    rc = impgpu_batch_decode_jpeg_pending(b, &im);
    if (rc || !im) { (void)impgpu_batch_decode_jpeg_finish(&b, &none, &code); if (none) impgpu_image_release(&none); return rc ? rc : 1000; }
    int rc_ops = impgpu_resize(&im, "224,0", &g_cfg, 0);
    impgpu_jpeg_encode* e = NULL;
    const impgpu_image* cim = im;
    int rc_enc = rc_ops ? rc_ops : impgpu_batch_encode_jpeg_begin(&cim, 1, 86, &e);
    /* the ONE wait: for the answer, the last thing on the stream; the verdict (in front of it) is there by then */
    if (!rc_enc) {
        unsigned char* outs[1] = {g_out[1]}; size_t caps[1] = {sizeof g_out[1]}; int ecode = 0;
        rc_enc = impgpu_batch_encode_jpeg_finish(&e, outs, caps, &g_len[1], &ecode);
        if (!rc_enc) rc_enc = ecode;
    }
    rc = impgpu_batch_decode_jpeg_finish(&b, &none, &code);
    impgpu_image_release(&im);
    return rc ? rc : code ? code : rc_enc;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* fp = fopen(argv[1], "rb");
    if (!fp) return 3;
    fseek(fp, 0, SEEK_END); g_size = (size_t)ftell(fp); fseek(fp, 0, SEEK_SET);
    g_blob = (unsigned char*)malloc(g_size);
    if (fread(g_blob, 1, g_size, fp) != g_size) return 3;
    fclose(fp);
    const int reps = argc > 2 ? atoi(argv[2]) : 200;
    if (impgpu_env_start(0) != IMP_OK) { fprintf(stderr, "%s\n", impgpu_last_error()); return 4; }
    memset(&g_cfg, 0, sizeof g_cfg);
    double* t = (double*)malloc(sizeof(double) * (size_t)reps);
    double med[2], p95[2];
    for (int form = 0; form < 2; form++) {
        for (int i = 0; i < 10; i++) { const int rc = form ? one_wait() : two_waits(); if (rc) { fprintf(stderr, "form %d: code %d (%s)\n", form, rc, impgpu_last_error()); return 5; } }
        for (int i = 0; i < reps; i++) { const double t0 = now_us(); (void)(form ? one_wait() : two_waits()); t[i] = now_us() - t0; }
        qsort(t, (size_t)reps, sizeof(double), cmp);
        med[form] = t[reps / 2]; p95[form] = t[(int)(reps * 0.95)];
    }
    const int same = g_len[0] == g_len[1] && !memcmp(g_out[0], g_out[1], g_len[0]);
    printf("{\"file_bytes\": %zu, \"answer_bytes\": %zu, \"two_waits_us\": %.1f, \"two_waits_p95_us\": %.1f, \"one_wait_us\": %.1f, \"one_wait_p95_us\": %.1f, \"same_answer\": %s}\n",
           g_size, g_len[0], med[0], p95[0], med[1], p95[1], same ? "true" : "false");
    impgpu_env_destroy();
    return same ? 0 : 6;
}
