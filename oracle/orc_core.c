/*
 * orc_core.c -- images, Crop, Resize argument handling, flips/rotations, chain order.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see imp_oracle.h).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "imp_oracle.h"
#include "orc_internal.h"

/* ---- image container: cvCreateImage's layout (4-byte aligned rows) ---- */

orc_image* orc_image_create(int width, int height, int channels) {
    if (width <= 0 || height <= 0 || (channels != 1 && channels != 3 && channels != 4)) return NULL;
    orc_image* img = (orc_image*)malloc(sizeof(orc_image));
    img->width = width; img->height = height; img->channels = channels;
    img->step = (width * channels + 3) & ~3;
    img->data = (unsigned char*)calloc((size_t)img->step * height, 1);
    return img;
}

orc_image* orc_image_from(const unsigned char* data, int width, int height, int channels, int step) {
    orc_image* img = orc_image_create(width, height, channels);
    if (!img) return NULL;
    for (int y = 0; y < height; y++)
        memcpy(img->data + (size_t)y * img->step, data + (size_t)y * step, (size_t)width * channels);
    return img;
}

orc_image* orc_image_clone(const orc_image* src) {
    return orc_image_from(src->data, src->width, src->height, src->channels, src->step);
}

void orc_image_free(orc_image* img) {
    if (!img) return;
    free(img->data);
    free(img);
}

unsigned char* orc_image_data(orc_image* img) { return img->data; }
int orc_image_width(const orc_image* img) { return img->width; }
int orc_image_height(const orc_image* img) { return img->height; }
int orc_image_channels(const orc_image* img) { return img->channels; }
int orc_image_step(const orc_image* img) { return img->step; }

/* ---- Crop: bridge.c:18-141 ---- */

/* One gravity token -> window origin along an axis of length `full` for a window of
 * `win` (bridge.c:81-96 for X with "l"/"r", :108-123 for Y with "t"/"b"). */
static int axis_origin(const char* token, const char* lo, const char* hi,
                       size_t full, unsigned int win, int* origin) {
    if (token == NULL) return ORC_ERROR_INVALID_ARGS; /* reference would strcmp(NULL): defined as invalid */
    if (strcmp(token, lo) == 0) { *origin = 0; return ORC_OK; }
    if (strcmp(token, hi) == 0) { *origin = (int)(full - win); return ORC_OK; }
    if (strcmp(token, "c") == 0) { *origin = (int)round((full - win) / 2.0); return ORC_OK; }
    char* mode;
    unsigned int px = (unsigned int)strtol(token, &mode, 10);
    if (strcmp("px", mode) != 0) return ORC_ERROR_INVALID_ARGS;
    *origin = (int)px;
    return ORC_OK;
}

int orc_crop_geometry(int icol, int irow, const char* _args, const char* _gravity,
                      int* ox, int* oy, int* ow, int* oh) {
    size_t col = (size_t)icol, row = (size_t)irow;
    char* args = strdup(_args ? _args : "");
    char* gravity = _gravity ? strdup(_gravity) : NULL;
    int rc = ORC_ERROR_INVALID_ARGS;
    char *next = NULL, *gnext = NULL;

    char* token = strtok_r(args, ",", &next);
    char* wmode; unsigned int ww = (unsigned int)strtol(token ? token : "", &wmode, 10);
    token = strtok_r(NULL, ",", &next);
    char* hmode; unsigned int wh = (unsigned int)strtol(token ? token : "", &hmode, 10);

    int respect = 0;                                   /* bridge.c:37-45 */
    if (gravity != NULL) {
        if (strlen(gravity) > 2) respect = 1; else goto done;
    }

    if (*wmode == 0 && *hmode == 0) {                  /* ratio mode, bridge.c:47-57 */
        if (ww == 0 || wh == 0) goto done;             /* reference: inf/nan -> 0 or INT_MIN -> rejected at :65 */
        float px = (float)col;
        float py = px / ww * wh;
        if (py > row) { py = (float)row; px = py / wh * ww; }
        ww = (unsigned int)(int)round(px);
        wh = (unsigned int)(int)round(py);
    } else if (strcmp(wmode, "px") == 0 && strcmp(hmode, "px") == 0) {
        /* absolute */
    } else goto done;

    if (ww == 0 || ww > col || wh == 0 || wh > row) goto done;   /* bridge.c:65-68 */

    int wx, wy;
    if (respect) token = strtok_r(gravity, ",", &gnext);
    else { token = strtok_r(NULL, ",", &next); if (!token) token = "c"; }
    if (axis_origin(token, "l", "r", col, ww, &wx)) goto done;

    if (respect) token = strtok_r(NULL, ",", &gnext);
    else { token = strtok_r(NULL, ",", &next); if (!token) token = "t"; }
    if (axis_origin(token, "t", "b", row, wh, &wy)) goto done;

    if (wx + (int)ww > icol || wy + (int)wh > irow) goto done;   /* bridge.c:125-128 */
    /* A negative Npx origin passes :125 in the reference and then trips cvSetImageROI /
     * cvCopy inside OpenCV; defined here as INVALID_ARGS. */
    if (wx < 0 || wy < 0) goto done;

    *ox = wx; *oy = wy; *ow = (int)ww; *oh = (int)wh;
    rc = ORC_OK;
done:
    free(args);
    free(gravity);
    return rc;
}

int orc_crop(orc_image** pointer, const char* args, const char* gravity) {
    orc_image* img = *pointer;
    int x, y, w, h;
    int rc = orc_crop_geometry(img->width, img->height, args, gravity, &x, &y, &w, &h);
    if (rc) return rc;
    orc_image* out = orc_image_create(w, h, img->channels);      /* bridge.c:130-137 */
    for (int r = 0; r < h; r++)
        memcpy(out->data + (size_t)r * out->step,
               img->data + (size_t)(y + r) * img->step + (size_t)x * img->channels,
               (size_t)w * img->channels);
    orc_image_free(img);
    *pointer = out;
    return ORC_OK;
}

/* ---- Resize: bridge.c:143-197 ---- */

int orc_resize_geometry(int icol, int irow, const char* _args, unsigned max_w, unsigned max_h,
                        int simple, int* ow, int* oh, int* interpolation) {
    size_t col = (size_t)icol, row = (size_t)irow;
    char* args = strdup(_args ? _args : "");
    char* next = NULL;
    char* token = strtok_r(args, ",", &next);
    char* m;
    unsigned int width = (unsigned int)strtol(token ? token : "", &m, 10);
    token = strtok_r(NULL, ",", &next);
    unsigned int height = (unsigned int)strtol(token ? token : "", &m, 10);

    if (width == 0 && height == 0) { free(args); return ORC_ERROR_INVALID_ARGS; }
    if (width == 0)  width  = (unsigned int)(int)round((float)height / row * col);   /* :167-169 */
    if (height == 0) height = (unsigned int)(int)round((float)width / col * row);    /* :171-173 */

    char* opt = strtok_r(NULL, ",", &next);
    int up = opt && strcmp(opt, "up") == 0;
    if (!up) {                                                                        /* :178-181 */
        width  = (unsigned int)fmin(width, col);
        height = (unsigned int)fmin(height, row);
    }
    free(args);

    /* :183-187 -- note the reference compares width (not height) against H; kept. */
    if ((max_w > 0 && width > max_w) || (max_h > 0 && width > max_h)) return ORC_ERROR_TOO_BIG_TARGET;
    /* cvCreateImage would raise on a zero-sized target: defined as INVALID_ARGS. */
    if (width == 0 || height == 0 || width > 0x7fff0000u || height > 0x7fff0000u) return ORC_ERROR_INVALID_ARGS;

    *ow = (int)width; *oh = (int)height;
    *interpolation = simple ? ORC_INTER_NN
                   : (width > col || height > row) ? ORC_INTER_CUBIC : ORC_INTER_AREA;  /* :190 */
    return ORC_OK;
}

int orc_resize(orc_image** pointer, const char* args, unsigned max_w, unsigned max_h, int simple) {
    orc_image* img = *pointer;
    int w, h, interp;
    int rc = orc_resize_geometry(img->width, img->height, args, max_w, max_h, simple, &w, &h, &interp);
    if (rc) return rc;
    orc_image* out = orc_image_create(w, h, img->channels);
    rc = orc_cv_resize(img, out, interp);
    if (rc) { orc_image_free(out); return rc; }
    orc_image_free(img);
    *pointer = out;
    return ORC_OK;
}

/* ---- cvFlip / cvTranspose compositions used by Flip and Rotate ---- */

/* mode as cvFlip: 0 = around x-axis (vertical), >0 = around y-axis (horizontal), <0 = both. */
orc_image* orc_cv_flip(const orc_image* src, int mode) {
    orc_image* dst = orc_image_create(src->width, src->height, src->channels);
    int c = src->channels;
    for (int y = 0; y < src->height; y++) {
        int sy = (mode <= 0) ? src->height - 1 - y : y;
        const unsigned char* s = src->data + (size_t)sy * src->step;
        unsigned char* d = dst->data + (size_t)y * dst->step;
        for (int x = 0; x < src->width; x++) {
            int sx = (mode != 0) ? src->width - 1 - x : x;
            memcpy(d + x * c, s + sx * c, (size_t)c);
        }
    }
    return dst;
}

orc_image* orc_cv_transpose(const orc_image* src) {
    orc_image* dst = orc_image_create(src->height, src->width, src->channels);
    int c = src->channels;
    for (int y = 0; y < dst->height; y++)
        for (int x = 0; x < dst->width; x++)
            memcpy(dst->data + (size_t)y * dst->step + x * c,
                   src->data + (size_t)x * src->step + y * c, (size_t)c);
    return dst;
}

/* bridge.c:613-618 (cvCvtColor GRAY2BGR) */
int orc_gray2bgr(orc_image** pointer) {
    orc_image* img = *pointer;
    if (img->channels != 1) return ORC_OK;
    orc_image* out = orc_image_create(img->width, img->height, 3);
    for (int y = 0; y < img->height; y++)
        for (int x = 0; x < img->width; x++) {
            unsigned char v = img->data[(size_t)y * img->step + x];
            unsigned char* d = out->data + (size_t)y * out->step + x * 3;
            d[0] = d[1] = d[2] = v;
        }
    orc_image_free(img);
    *pointer = out;
    return ORC_OK;
}

/* ---- RunJob's operator segment: bridge.c:574-656 ---- */

int orc_run_chain(orc_image** pointer, const orc_chain* ch, int* step) {
    int rc;
    *step = ORC_STEP_CROP;
    if (ch->crop) { rc = orc_crop(pointer, ch->crop, ch->gravity); if (rc) return rc; }
    *step = ORC_STEP_RESIZE;
    if (ch->resize) { rc = orc_resize(pointer, ch->resize, ch->max_w, ch->max_h, ch->simple); if (rc) return rc; }
    *step = ORC_STEP_FILTERING;
    orc_gray2bgr(pointer);
    for (int i = 0; i < ch->filter_count; i++) {
        rc = orc_filter(pointer, ch->filters[i], ch->allow_experiments);
        if (rc) return rc;
    }
    *step = ORC_STEP_WATERMARK;
    if (ch->overlay) {
        rc = orc_watermark(*pointer, ch->overlay, ch->gravity_x, ch->gravity_y,
                           ch->offset_x, ch->offset_y, ch->opacity);
        if (rc) return rc;
    }
    if (ch->flatten && (*pointer)->channels == 4) orc_blend_with_paper(*pointer);
    *step = ORC_STEP_INFO;
    return ORC_OK;
}

/* ---- codec-adjacent repacks: advancedio.c:65-101 (IplToFI32 / IplToFI24) and :310-318 (LoadSingle) ---- */

/* FreeImage bitmaps are bottom-up with 4-byte aligned rows. bpp = 32: B,G,R,A (A = 255 for 3-channel sources);
 * bpp = 24: B,G,R (alpha dropped). out must hold pitch * height bytes. */
int orc_ipl_to_fi(const orc_image* img, int bpp, unsigned char* out, int pitch) {
    if ((bpp != 24 && bpp != 32) || img->channels < 3) return ORC_ERROR_INVALID_ARGS;
    const int bc = bpp / 8;
    for (int y = 0; y < img->height; y++) {
        const int row = img->height - 1 - y;
        for (int x = 0; x < img->width; x++) {
            const unsigned char* s = img->data + (size_t)row * img->step + (size_t)x * img->channels;
            unsigned char* d = out + (size_t)y * pitch + (size_t)x * bc;
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
            if (bc == 4) d[3] = img->channels == 4 ? s[3] : 255;
        }
    }
    return ORC_OK;
}

/* LoadSingle: 32-bit bottom-up FreeImage bits (pitch = 4 * w there) -> 4-channel top-down image */
orc_image* orc_fi32_to_ipl(const unsigned char* bits, int width, int height, int pitch) {
    orc_image* img = orc_image_create(width, height, 4);
    if (!img) return NULL;
    for (int y = 0; y < height; y++)
        memcpy(img->data + (size_t)(height - 1 - y) * img->step, bits + (size_t)y * pitch, (size_t)width * 4);
    return img;
}

/* ---- advancedio.c:103-262 LoadGIF: the per-pixel compositing loop (:204-247) ----
 * Defined where the reference is undefined (the product defines the same):
 *  - `x > left + w` lets x == left + w through, which reads row[w]: one byte past the frame's row.  Mirrored as the
 *    same linear read (the pitch padding, or the next scanline's first byte); past the end of the page it is the key.
 *  - an index < 0 (key -1 on a pixel outside the frame, or a never-written master entry) would index the palette
 *    out of bounds: colour 0,0,0.  `master` starts at 0 (ngx_palloc does not clear it).
 *  - page < -1: INVALID_ARGS (the reference walks every page and then indexes Frames[page] below the array).
 * A page request (page != -1) forces the destructive walk and a page past the last one means page 0
 * (advancedio.c:111-116): `isdestructive = 1; if (page > framecount - 1) page = 0;`. */
int orc_gif_compose(const orc_gif_page* pages, int count, int destructive, int page, orc_image** frames) {
    if (!pages || !frames || count <= 0 || page < -1) return ORC_ERROR_INVALID_ARGS;
    if (page != -1) {                                             /* :111-116 */
        destructive = 1;
        if (page > count - 1) page = 0;
    }
    const int cw = pages[0].width, ch = pages[0].height;          /* :133-136 canvas = first page */
    if (cw <= 0 || ch <= 0) return ORC_ERROR_INVALID_ARGS;
    int* master = destructive ? (int*)calloc((size_t)cw * ch, sizeof(int)) : NULL;   /* :195-200 */
    if (destructive && !master) return ORC_ERROR_MALLOC_FAILED;
    const int last = page >= 0 ? page : count - 1;
    orc_image* kept = NULL;
    for (int f = 0; f <= last; f++) {
        const orc_gif_page* p = &pages[f];
        const int w = p->width, h = p->height, left = p->left, top = p->top, key = p->transparency_key;
        orc_image* img = orc_image_create(cw, ch, 4);
        if (!img) { free(master); return ORC_ERROR_MALLOC_FAILED; }
        for (int y = 0; y < ch; y++) {
            const int rowidx = h + top - y - 1;                   /* :206 */
            for (int x = 0; x < cw; x++) {
                int coloridx;
                if (rowidx < 0 || x < left || y < top || x > left + w || y > top + h) {   /* :213 */
                    coloridx = key;
                } else {
                    const long long o = (long long)rowidx * p->pitch + (x - left);
                    coloridx = o < (long long)p->pitch * h ? p->indices[o] : key;
                }
                if (destructive) {                                /* :219-240 */
                    const size_t offset = (size_t)y * cw + x;
                    if (p->dispose == 2) {                        /* GIF_DISPOSAL_BACKGROUND */
                        if (coloridx == key) coloridx = 0;
                        else master[offset] = coloridx;
                    } else {
                        if (coloridx == key && f > 0) coloridx = master[offset];
                        else master[offset] = coloridx;
                    }
                }
                unsigned char* d = img->data + (size_t)y * img->step + (size_t)x * 4;
                if (coloridx >= 0 && coloridx < 256) {            /* :242-246 RGBQUAD = B,G,R,reserved */
                    d[0] = p->palette[4 * coloridx]; d[1] = p->palette[4 * coloridx + 1]; d[2] = p->palette[4 * coloridx + 2];
                } else {
                    d[0] = d[1] = d[2] = 0;
                }
                d[3] = coloridx == key ? 0 : 255;
            }
        }
        if (page < 0) frames[f] = img;
        else if (f == page) kept = img;
        else orc_image_free(img);
    }
    free(master);
    if (page >= 0) frames[0] = kept;
    return ORC_OK;
}
