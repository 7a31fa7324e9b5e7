/*
 * mixed_harness.c -- impgpu_batch_resize_mixed driven from plain C99: a worker that holds several decoded frames of
 * different sizes (the per-frame Resize() loop of bridge.c:588-604 over whatever arrived) hands them over in one call.
 *
 *   mixed_harness <channels> <simple> <n>  then n times:  <frame.raw> <w> <h> <dw> <dh> <out.raw>
 * Frames are uploaded with the frame's own 4-byte padded pitch (cvCreateImage's widthStep), resized on the device with the
 * interpolation Resize() picks per frame (bridge.c:188-192) and written back tightly packed.  Prints  code=<IMP_*>.
 */
#include <impgpu.h>
#include <stdio.h>
#include <stdlib.h>

static unsigned char* read_file(const char* path, size_t want) {
    FILE* f = fopen(path, "rb");
    unsigned char* buf;
    if (!f) return NULL;
    buf = (unsigned char*)malloc(want ? want : 1);
    if (buf && fread(buf, 1, want, f) != want) { free(buf); buf = NULL; }
    fclose(f);
    return buf;
}

int main(int argc, char** argv) {
    int c, simple, n, i, code;
    impgpu_image** src;
    impgpu_image** dst;
    impgpu_resize_item* items;
    if (argc < 4) { fprintf(stderr, "usage: %s channels simple n {frame.raw w h dw dh out.raw}...\n", argv[0]); return 2; }
    c = atoi(argv[1]); simple = atoi(argv[2]); n = atoi(argv[3]);
    if (n < 0 || argc != 4 + 6 * n) { fprintf(stderr, "expected %d frame records\n", n); return 2; }
    code = impgpu_env_start(-1);
    if (code != IMP_OK) { printf("code=%d\n", code); return 0; }
    src = (impgpu_image**)calloc((size_t)n + 1, sizeof *src);
    dst = (impgpu_image**)calloc((size_t)n + 1, sizeof *dst);
    items = (impgpu_resize_item*)calloc((size_t)n + 1, sizeof *items);
    for (i = 0; i < n && code == IMP_OK; i++) {
        char** r = argv + 4 + 6 * i;
        const int w = atoi(r[1]), h = atoi(r[2]), dw = atoi(r[3]), dh = atoi(r[4]);
        unsigned char* frame = read_file(r[0], (size_t)w * h * c);
        if (!frame) { fprintf(stderr, "cannot read %s\n", r[0]); return 2; }
        code = impgpu_image_upload(frame, w, h, c, w * c, &src[i]);
        free(frame);
        if (code == IMP_OK) code = impgpu_image_create(dw, dh, c, &dst[i]);
        if (code != IMP_OK) break;
        items[i].src = impgpu_image_device_ptr(src[i]);
        items[i].src_width = w; items[i].src_height = h; items[i].src_step = impgpu_image_step(src[i]);
        items[i].dst = impgpu_image_device_ptr(dst[i]);
        items[i].dst_width = dw; items[i].dst_height = dh; items[i].dst_step = impgpu_image_step(dst[i]);
    }
    if (code == IMP_OK) code = impgpu_batch_resize_mixed(items, n, c, simple, NULL);
    for (i = 0; i < n && code == IMP_OK; i++) {
        char** r = argv + 4 + 6 * i;
        const int dw = atoi(r[3]), dh = atoi(r[4]);
        unsigned char* out = (unsigned char*)malloc((size_t)dw * dh * c + 1);
        FILE* f;
        code = impgpu_image_download(dst[i], out, dw * c);
        if (code != IMP_OK) { free(out); break; }
        f = fopen(r[5], "wb");
        if (!f || fwrite(out, 1, (size_t)dw * dh * c, f) != (size_t)dw * dh * c) { fprintf(stderr, "cannot write %s\n", r[5]); return 2; }
        fclose(f);
        free(out);
    }
    for (i = 0; i < n; i++) {
        if (src[i]) impgpu_image_release(&src[i]);
        if (dst[i]) impgpu_image_release(&dst[i]);
    }
    impgpu_env_destroy();
    printf("code=%d\n", code);
    return 0;
}
