/*
 * runjob_harness.c -- a plain C99 caller of include/impgpu.h, the way the reference's own RunJob (bridge.c:302-724)
 * would call it: OnEnvStart -> [PrepareWatermark] -> parse the request line -> upload the decoded frame ->
 * impgpu_run_ops (the operator segment bridge.c:574-656) -> the json exit's brightness (bridge.c:659-666, :283-300) or
 * the text exit (bridge.c:668-677) or download for the encoder (bridge.c:680-710) -> OnEnvDestroy.
 *
 * No Python, no C++, no torch: this is the language the reference is written in.  tests/test_c_harness.py feeds it raw
 * frames and compares every byte it writes with the oracle.
 *
 *   runjob_harness <frame.raw> <w> <h> <channels> <uri> <extension> <out.raw> [<overlay.raw> <ow> <oh> <oc> <gx> <gy> <offx> <offy> <opacity>]
 * prints one line:  code=<IMP_*> step=<IMP_STEP_*> mime=<IMP_MIME_*> w=<w> h=<h> c=<c> bytes=<n> [brightness=<percent>]
 */
#include <impgpu.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned char* read_file(const char* path, size_t want) {
    FILE* f = fopen(path, "rb");
    unsigned char* buf;
    if (!f) return NULL;
    buf = (unsigned char*)malloc(want ? want : 1);
    if (buf && fread(buf, 1, want, f) != want) { free(buf); buf = NULL; }
    fclose(f);
    return buf;
}

static int write_file(const char* path, const unsigned char* data, size_t n) {
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    if (n && fwrite(data, 1, n, f) != n) { fclose(f); return -1; }
    return fclose(f);
}

int main(int argc, char** argv) {
    impgpu_config cfg;
    impgpu_request* req = NULL;
    impgpu_image* image = NULL;
    unsigned char *frame = NULL, *overlay = NULL, *out = NULL;
    int w, h, c, code, step = IMP_STEP_START, mime = 0, ow = 0, oh = 0, oc = 0;
    long nbytes = 0;
    float brightness = -1.f;

    if (argc != 8 && argc != 17) {
        fprintf(stderr, "usage: %s frame.raw w h c uri ext out.raw [overlay.raw ow oh oc gx gy offx offy opacity]\n", argv[0]);
        return 2;
    }
    w = atoi(argv[2]); h = atoi(argv[3]); c = atoi(argv[4]);
    frame = read_file(argv[1], (size_t)w * h * c);
    if (!frame) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }

    if (impgpu_env_start(-1) != IMP_OK) {                       /* OnEnvStart, bridge.c:10 */
        fprintf(stderr, "impgpu_env_start: %s\n", impgpu_last_error());
        return 3;
    }
    {   /* the test's way of asking for an injected device fault: HARNESS_FAULT=<step>[:<n>] -> impgpu_fault_arm (the
         * library itself never reads the environment for this) */
        const char* fault = getenv("HARNESS_FAULT");
        if (fault && *fault) {
            char* end = NULL;
            long step = strtol(fault, &end, 10);
            impgpu_fault_arm((int)step, (end && *end == ':') ? strtol(end + 1, NULL, 10) : 1);
        }
    }
    memset(&cfg, 0, sizeof cfg);                                /* OnConfigMerge defaults, module.c:168-187 */
    cfg.max_target_w = 2000; cfg.max_target_h = 2000;
    cfg.max_filters_count = 5; cfg.allow_experiments = 1;
    cfg.watermark_opacity = 100; cfg.watermark_gravity_x = 'l'; cfg.watermark_gravity_y = 't';
    if (argc == 17) {                                           /* PrepareWatermark, bridge.c:199-237 */
        int wmw = atoi(argv[9]), wmh = atoi(argv[10]), wmc = atoi(argv[11]);
        overlay = read_file(argv[8], (size_t)wmw * wmh * wmc);
        if (!overlay) { fprintf(stderr, "cannot read %s\n", argv[8]); return 2; }
        code = impgpu_prepare_watermark(&cfg, overlay, wmw, wmh, wmc, wmw * wmc);
        if (code) { printf("code=%d step=%d\n", code, IMP_STEP_START); return 0; }
        cfg.watermark_gravity_x = argv[12][0]; cfg.watermark_gravity_y = argv[13][0];
        cfg.watermark_offset_x = atoi(argv[14]); cfg.watermark_offset_y = atoi(argv[15]);
        cfg.watermark_opacity = atoi(argv[16]);
    }

    code = impgpu_parse_request(argv[5], argv[6], &cfg, &req);  /* bridge.c:304-372, :413-466 */
    mime = impgpu_request_mime(req);
    if (code == IMP_OK) {
        step = IMP_STEP_DECODE;                                 /* the decoded frame arrives here, bridge.c:541-572 */
        code = impgpu_image_upload(frame, w, h, c, w * c, &image);
    }
    if (code == IMP_OK) code = impgpu_run_ops(&image, impgpu_request_job(req), &cfg, &step);   /* bridge.c:574-656 */
    if (code == IMP_OK) {
        ow = impgpu_image_width(image); oh = impgpu_image_height(image); oc = impgpu_image_channels(image);
        if (mime == -3) {                                       /* IMP_MIME_JSON: Info(), bridge.c:283-300 */
            code = impgpu_calc_perceived_brightness(image, &brightness);
        } else if (mime == -5) {                                /* IMP_MIME_TEXT: ASCII(), bridge.c:668-677 */
            const char* q = impgpu_request_quality(req);
            long cap = (long)(ow + 1) * oh - 1;
            out = (unsigned char*)malloc((size_t)cap + 1);
            code = impgpu_ascii(image, q ? q : "", out, cap, &nbytes);
        } else {                                                /* the encoder's input, bridge.c:703-704 */
            step = IMP_STEP_ENCODE;
            nbytes = (long)ow * oh * oc;
            out = (unsigned char*)malloc((size_t)nbytes);
            code = impgpu_image_download(image, out, ow * oc);
        }
    }
    if (code == IMP_OK && out && write_file(argv[7], out, (size_t)nbytes) != 0) { fprintf(stderr, "cannot write %s\n", argv[7]); return 2; }
    printf("code=%d step=%d mime=%d w=%d h=%d c=%d bytes=%ld", code, step, mime, ow, oh, oc, nbytes);
    if (brightness >= 0) printf(" brightness=%d", (int)round(brightness * 100));            /* bridge.c:296 */
    printf("\n");

    impgpu_image_release(&image);                               /* finalize:, bridge.c:714-723 */
    impgpu_request_free(&req);
    if (cfg.watermark) impgpu_image_release(&cfg.watermark);
    impgpu_env_destroy();                                       /* OnEnvDestroy, bridge.c:14 */
    free(frame); free(overlay); free(out);
    return 0;
}
