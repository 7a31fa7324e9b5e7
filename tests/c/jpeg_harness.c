/*
 * jpeg_harness.c -- a request the way it arrives and leaves, from plain C99 over include/impgpu.h: the JPEG file is decoded
 * on the device (what cvDecodeImage does at bridge.c:545-552), the operator segment runs (bridge.c:574-656), and the answer
 * is written as a JPEG file on the device (what cvEncodeImage does at bridge.c:703-709, with the quality rule of
 * bridge.c:474-486: quality= or JPEG_QUALITY_DEFAULT, required.h:76).  tests/test_c_harness.py compares the file it writes
 * with the oracle's, byte for byte.
 *
 *   jpeg_harness <in.jpg> <uri> <extension> <out.jpg>
 * prints one line:  code=<IMP_*> step=<IMP_STEP_*> w=<w> h=<h> c=<c> bytes=<n>
 */
#include <impgpu.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define JPEG_QUALITY_DEFAULT 86

int main(int argc, char** argv) {
    impgpu_config cfg;
    impgpu_request* req = NULL;
    impgpu_image* image = NULL;
    unsigned char *blob = NULL, *out = NULL;
    size_t size = 0, cap = 0, len = 0;
    int code, step = IMP_STEP_START, quality = JPEG_QUALITY_DEFAULT;
    FILE* f;

    if (argc != 5) { fprintf(stderr, "usage: %s in.jpg uri ext out.jpg\n", argv[0]); return 2; }
    f = fopen(argv[1], "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fseek(f, 0, SEEK_END); size = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    blob = (unsigned char*)malloc(size ? size : 1);
    if (!blob || fread(blob, 1, size, f) != size) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);

    if (impgpu_env_start(-1) != IMP_OK) { fprintf(stderr, "impgpu_env_start: %s\n", impgpu_last_error()); return 3; }
    memset(&cfg, 0, sizeof cfg);
    cfg.max_target_w = 2000; cfg.max_target_h = 2000; cfg.max_filters_count = 5; cfg.allow_experiments = 1;

    code = impgpu_parse_request(argv[2], argv[3], &cfg, &req);
    if (code == IMP_OK) {
        const char* q = impgpu_request_quality(req);
        if (q) quality = (int)strtol(q, NULL, 10);              /* bridge.c:478-480 */
        if (quality < 0 || quality > 100) code = IMP_ERROR_INVALID_ARGS;   /* bridge.c:481-486 */
    }
    if (code == IMP_OK) { step = IMP_STEP_DECODE; code = impgpu_image_decode_jpeg(blob, size, &image); }
    if (code == IMP_OK) code = impgpu_run_ops(&image, impgpu_request_job(req), &cfg, &step);
    if (code == IMP_OK) {
        step = IMP_STEP_ENCODE;
        cap = impgpu_jpeg_encode_bound(impgpu_image_width(image), impgpu_image_height(image), impgpu_image_channels(image));
        out = (unsigned char*)malloc(cap ? cap : 1);
        code = out ? impgpu_image_encode_jpeg(image, quality, out, cap, &len) : IMP_ERROR_MALLOC_FAILED;
    }
    if (code == IMP_OK) {
        f = fopen(argv[4], "wb");
        if (!f || fwrite(out, 1, len, f) != len || fclose(f) != 0) { fprintf(stderr, "cannot write %s\n", argv[4]); return 2; }
    }
    printf("code=%d step=%d w=%d h=%d c=%d bytes=%lu\n", code, step, impgpu_image_width(image), impgpu_image_height(image),
           impgpu_image_channels(image), (unsigned long)len);
    impgpu_image_release(&image);
    impgpu_request_free(&req);
    impgpu_env_destroy();
    free(blob); free(out);
    return 0;
}
