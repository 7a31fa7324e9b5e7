/*
 * orc_filters.c -- restatement of filters.c (filter-* dispatch, the 14 default filters,
 * blends, brightness, ASCII) and of helpers.c's in-place HSV conversions.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see imp_oracle.h).
 *
 * Loops here run row-major; the reference's x-outer order only matters
 * for CalcPerceivedBrightness, whose float accumulator is order-sensitive, and that
 * one keeps the reference's order.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "imp_oracle.h"
#include "orc_internal.h"

#define PX(img, x, y) ((img)->data + (size_t)(y) * (img)->step + (size_t)(x) * (img)->channels)

/* ---- helpers.c:70-107 ---- */
static inline void px_rgb2hsv(unsigned char* p) {
    int b = p[0], g = p[1], r = p[2];
    int mn = b < g ? (b < r ? b : r) : (g < r ? g : r);
    int mx = b > g ? (b > r ? b : r) : (g > r ? g : r);
    int delta = mx - mn, h = 0, s = 0, v = mx;
    if (v != 0) s = 255 * delta / v;
    if (s != 0) {
        if (mx == r)      h = 30 * (g - b) / delta;
        else if (mx == g) h = 60 + 30 * (b - r) / delta;
        else              h = 120 + 30 * (r - g) / delta;
    }
    if (h < 0) h += 180;
    p[0] = orc_byte(h); p[1] = orc_byte(s); p[2] = orc_byte(v);
}

/* ---- helpers.c:109-176 ---- */
static inline void px_hsv2rgb(unsigned char* px) {
    float h = (float)(px[0] * 2), s = px[1], v = px[2];
    int r, g, b;
    if (s == 0) {
        r = g = b = orc_trunc(v);
    } else {
        s /= 255;
        h /= 60;
        int i = (int)floor(h);
        float f = h - i;
        int p = orc_trunc(v * (1 - s));
        int q = orc_trunc(v * (1 - s * f));
        int t = orc_trunc(v * (1 - s * (1 - f)));
        int vi = orc_trunc(v);
        switch (i) {
            case 0: r = vi; g = t;  b = p;  break;
            case 1: r = q;  g = vi; b = p;  break;
            case 2: r = p;  g = vi; b = t;  break;
            case 3: r = p;  g = q;  b = vi; break;
            case 4: r = t;  g = p;  b = vi; break;
            default: r = vi; g = p; b = q;  break;
        }
    }
    px[0] = orc_byte(b); px[1] = orc_byte(g); px[2] = orc_byte(r);
}

void orc_rgb2hsv(orc_image* img) {
    for (int y = 0; y < img->height; y++)
        for (int x = 0; x < img->width; x++) px_rgb2hsv(PX(img, x, y));
}
void orc_hsv2rgb(orc_image* img) {
    for (int y = 0; y < img->height; y++)
        for (int x = 0; x < img->width; x++) px_hsv2rgb(PX(img, x, y));
}

/* ---- filters.c:524-547 ---- */
static void modulate_hsv(orc_image* img, const int* hsv) {
    orc_rgb2hsv(img);
    for (int y = 0; y < img->height; y++)
        for (int x = 0; x < img->width; x++) {
            unsigned char* p = PX(img, x, y);
            if (hsv[0] != 0) {
                int hue = p[0] + hsv[0];
                if (hue > 180) hue -= 180;
                p[0] = orc_byte(hue);
            }
            for (int c = 1; c < 3; c++) {
                int cval = p[c];
                cval = orc_trunc(fmin(cval * hsv[c] / 100.0, 255));
                p[c] = orc_byte(cval);
            }
        }
    orc_hsv2rgb(img);
}

/* ---- filters.c:608-616 ---- */
static void add_color(orc_image* img, const int* rgb, float alpha) {
    float beta = 1 - alpha;
    for (int y = 0; y < img->height; y++)
        for (int x = 0; x < img->width; x++) {
            unsigned char* p = PX(img, x, y);
            for (int c = 0; c < 3; c++) p[c] = orc_store((beta * p[c]) + (rgb[2 - c] * alpha));
        }
}

/* ---- filters.c:549-570 ---- */
static void apply_gamma(orc_image* img, float gamma) {
    float inverse = 1 / gamma;
    unsigned char lut[256];
    for (int i = 0; i < 256; i++) lut[i] = orc_byte(orc_trunc(pow(i / 255.0, inverse) * 255.0));
    for (int y = 0; y < img->height; y++)
        for (int x = 0; x < img->width; x++) {
            unsigned char* p = PX(img, x, y);
            for (int c = 0; c < img->channels; c++) p[c] = lut[p[c]];
        }
}

/* ---- filters.c:595-605 ---- */
static void brightness_contrast(orc_image* img, float br, float ct) {
    int nc = img->channels < 3 ? img->channels : 3;
    for (int y = 0; y < img->height; y++)
        for (int x = 0; x < img->width; x++) {
            unsigned char* p = PX(img, x, y);
            for (int c = 0; c < nc; c++) {
                int val = p[c];
                val = orc_trunc((ct * val) + (br * 255));
                val = orc_trunc(fmax(fmin(val, 255), 0));
                p[c] = orc_byte(val);
            }
        }
}

/* ---- the callbacks ---- */

static int f_flip(orc_image** pp, char* args) {                /* filters.c:72-109 */
    if (strlen(args) != 2) return ORC_ERROR_INVALID_ARGS;
    int hz = 0, vt = 0;
    if (args[0] == '1') hz = 1; else if (args[0] != '0') return ORC_ERROR_INVALID_ARGS;
    if (args[1] == '1') vt = 1; else if (args[1] != '0') return ORC_ERROR_INVALID_ARGS;
    if (!hz && !vt) return ORC_OK;
    orc_image* out = orc_cv_flip(*pp, hz && vt ? -1 : hz ? 1 : 0);
    orc_image_free(*pp);
    *pp = out;
    return ORC_OK;
}

static int f_rotate(orc_image** pp, char* args) {              /* filters.c:111-133 */
    int amount = (int)strtol(args, NULL, 10);
    if (amount == 90 || amount == 270) {
        orc_image* t = orc_cv_transpose(*pp);
        orc_image* out = orc_cv_flip(t, 270 - amount);
        orc_image_free(t);
        orc_image_free(*pp);
        *pp = out;
        return ORC_OK;
    }
    if (amount == 180) {
        orc_image* out = orc_cv_flip(*pp, -1);
        orc_image_free(*pp);
        *pp = out;
        return ORC_OK;
    }
    return ORC_ERROR_INVALID_ARGS;
}

static int f_modulate(orc_image** pp, char* args) {            /* filters.c:135-158 */
    int params[3];
    char* ctx = NULL;
    for (int i = 0; i < 3; i++) {
        char* tok = strtok_r(args, ",", &ctx);
        if (!tok) return ORC_ERROR_INVALID_ARGS;
        args = NULL;
        params[i] = (int)strtol(tok, NULL, 10);
    }
    if (params[0] < 0 || params[0] > 180) return ORC_ERROR_INVALID_ARGS;
    if (params[2] <= 0) return ORC_ERROR_INVALID_ARGS;
    modulate_hsv(*pp, params);
    return ORC_OK;
}

static int hex_pair(const char* s) { char t[3] = {s[0], s[1], 0}; return (int)strtol(t, NULL, 16); }

static int f_colorize(orc_image** pp, char* args) {            /* filters.c:160-190 */
    char* ctx = NULL;
    char* color = strtok_r(args, ",", &ctx);
    if (!color || strlen(color) != 6) return ORC_ERROR_INVALID_ARGS;
    int rgb[3];
    for (int i = 0; i < 3; i++) rgb[i] = hex_pair(color + 2 * i);
    char* op = strtok_r(NULL, ",", &ctx);
    float opacity = op ? strtof(op, NULL) : 0.5f;
    if (opacity < 0 || opacity > 1) return ORC_ERROR_INVALID_ARGS;
    add_color(*pp, rgb, opacity);
    return ORC_OK;
}

static int f_blur(orc_image** pp, char* args) {                /* filters.c:192-207 */
    char* ctx = NULL;
    char* arg = strtok_r(args, ",", &ctx);
    if (!arg) return ORC_ERROR_INVALID_ARGS;
    float sigma = strtof(arg, NULL);
    if (sigma < 0) return ORC_ERROR_INVALID_ARGS;
    return orc_cv_smooth_gaussian(*pp, sigma);
}

static int f_gamma(orc_image** pp, char* args) {               /* filters.c:209-212 */
    apply_gamma(*pp, strtof(args, NULL));
    return ORC_OK;
}

static int f_contrast(orc_image** pp, char* args) {            /* filters.c:214-221 */
    float v = strtof(args, NULL);
    if (v <= 0) return ORC_ERROR_INVALID_ARGS;
    brightness_contrast(*pp, 0, v);
    return ORC_OK;
}

static int f_gradmap(orc_image** pp, char* args) {             /* filters.c:223-286, 572-593 */
    unsigned char colors[8][3];
    int n = 0;
    char* ctx = NULL;
    char* cur;
    while ((cur = strtok_r(args, ",", &ctx))) {
        args = NULL;
        if (strlen(cur) != 6) return ORC_ERROR_INVALID_ARGS;
        if (n >= 8) return ORC_ERROR_INVALID_ARGS;   /* reference overflows its 8-slot array: defined invalid */
        for (int i = 0; i < 3; i++) colors[n][i] = (unsigned char)hex_pair(cur + 2 * i);
        n++;
    }
    if (n < 2) return ORC_ERROR_INVALID_ARGS;        /* reference reads an unfilled LUT: defined invalid */
    unsigned char lut[768];
    int segments = n - 1;
    float inner = 256 / (float)segments;
    int ptr = 0;
    for (int c = 0; c < segments; c++)
        for (int i = 0; i < (int)inner; i++) {
            float step = i / inner;
            for (int j = 0; j < 3; j++)
                lut[ptr++] = orc_store(round(colors[c][j] + step * (colors[c + 1][j] - colors[c][j])));
        }
    /* entries the reference leaves uninitialised when 256 % segments != 0: defined = last colour */
    while (ptr < 768) { lut[ptr] = colors[n - 1][ptr % 3]; ptr++; }
    orc_image* img = *pp;
    for (int y = 0; y < img->height; y++)
        for (int x = 0; x < img->width; x++) {
            unsigned char* p = PX(img, x, y);
            int off = ((p[2] + p[1] + p[0]) / 3) * 3;
            p[2] = lut[off]; p[1] = lut[off + 1]; p[0] = lut[off + 2];
        }
    return ORC_OK;
}

/* helpers.c:46-66 */
static float dist_f(int ax, int ay, int bx, int by) {
    return (float)sqrt(pow((float)(ax - bx), 2) + pow((float)(ay - by), 2));
}

static int f_vignette(orc_image** pp, char* args) {            /* filters.c:295-323, 693-703 */
    char* ctx = NULL;
    char* a = strtok_r(args, ",", &ctx);
    float intensity = a == NULL ? 0.5f : strtof(a, NULL);
    char* r = strtok_r(NULL, ",", &ctx);
    float radius = r == NULL ? 1.0f : strtof(r, NULL);
    orc_image* img = *pp;
    int w = img->width, h = img->height, cx = w / 2, cy = h / 2;
    float maxdis = 0;
    int corners[4][2] = {{0, 0}, {w, 0}, {0, h}, {w, h}};
    for (int i = 0; i < 4; i++) { float d = dist_f(corners[i][0], corners[i][1], cx, cy); if (maxdis < d) maxdis = d; }
    float maxrad = radius * maxdis;
    orc_rgb2hsv(img);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float d = dist_f(cx, cy, x, y);
            float raw = d / maxrad * intensity;
            float mask = (float)pow(cos(raw), 4);
            unsigned char* p = PX(img, x, y);
            float source = p[2];
            p[2] = orc_store(source * mask);
        }
    orc_hsv2rgb(img);
    return ORC_OK;
}

static int f_gotham(orc_image** pp, char* args) {              /* filters.c:325-333 */
    (void)args;
    int hsv[] = {120, 5, 100};
    modulate_hsv(*pp, hsv);
    int rgb[] = {17, 27, 93};
    add_color(*pp, rgb, (float)0.15);
    apply_gamma(*pp, (float)0.3);
    brightness_contrast(*pp, (float)-0.07, (float)1.5);
    return ORC_OK;
}

static int f_lomo(orc_image** pp, char* args) {                /* filters.c:335-346 */
    (void)args;
    orc_image* img = *pp;
    for (int y = 0; y < img->height; y++)
        for (int x = 0; x < img->width; x++) {
            unsigned char* p = PX(img, x, y);
            for (int c = 1; c < 3; c++) {
                float val = p[c];
                val = (float)fmax(fmin(val * 1.5 - 50, 255), 0);
                p[c] = orc_store(val);
            }
        }
    return ORC_OK;
}

static int f_kelvin(orc_image** pp, char* args) {              /* filters.c:348-354 */
    (void)args;
    int hsv[] = {120, 50, 100};
    modulate_hsv(*pp, hsv);
    int rgb[] = {255, 153, 0};
    add_color(*pp, rgb, (float)0.5);
    return ORC_OK;
}

static int f_rainbow(orc_image** pp, char* args) {             /* filters.c:356-403 */
    int sat = 255;
    if (strcmp(args, "mid") == 0) sat = 190;
    else if (strcmp(args, "pale") == 0) sat = 120;
    else if (strcmp(args, "full") != 0) return ORC_ERROR_INVALID_ARGS;
    orc_image* img = *pp;
    orc_rgb2hsv(img);
    for (int y = 0; y < img->height; y++)
        for (int x = 0; x < img->width; x++) {
            unsigned char* p = PX(img, x, y);
            int hue = p[0] * 2, light = p[2], saturation = sat;
            if (light < 20) { light = 0; saturation = 0; }
            else if (light > 254) saturation = 0;
            else if (hue <= 10 || hue > 340) hue = 0;
            else if (hue >= 10 && hue < 35) hue = 30;
            else if (hue >= 35 && hue < 68) hue = 60;
            else if (hue >= 68 && hue < 150) hue = 120;
            else if (hue >= 150 && hue < 200) hue = 195;
            else if (hue >= 200 && hue < 250) hue = 225;
            else hue = 285;
            p[0] = orc_store(hue / 2.0);
            p[1] = orc_byte(saturation);
            p[2] = orc_byte(light);
        }
    orc_hsv2rgb(img);
    return ORC_OK;
}

static int f_scanline(orc_image** pp, char* args) {            /* filters.c:405-455 */
    char* ctx = NULL;
    char* a = strtok_r(args, ",", &ctx);
    if (!a) return ORC_ERROR_INVALID_ARGS;   /* unreachable through Filter (:52-56) */
    float intensity = strtof(a, NULL);
    if (intensity < 0 || intensity > 1) return ORC_ERROR_INVALID_ARGS;
    char* o = strtok_r(NULL, ",", &ctx);
    float opacity = o == NULL ? 0 : strtof(o, NULL);
    if (opacity < 0 || opacity > 1) return ORC_ERROR_INVALID_ARGS;
    char* f = strtok_r(NULL, ",", &ctx);
    int freq = f == NULL ? 1 : (int)strtol(f, NULL, 10);
    if (freq < 1) return ORC_ERROR_INVALID_ARGS;
    char* wd = strtok_r(NULL, ",", &ctx);
    int width = wd == NULL ? 1 : (int)strtol(wd, NULL, 10);
    if (width < 1) return ORC_ERROR_INVALID_ARGS;
    orc_image* img = *pp;
    orc_rgb2hsv(img);
    int skipped = 0, drawed = 0;
    for (int y = 0; y < img->height; y++) {
        if (skipped == freq) {
            if (drawed == width) skipped = drawed = 0;
            else {
                for (int x = 0; x < img->width; x++) {
                    unsigned char* p = PX(img, x, y);
                    p[1] = orc_store(255 * opacity);
                    p[2] = orc_store(255 * intensity);
                }
                drawed++;
            }
        } else skipped++;
    }
    orc_hsv2rgb(img);
    return ORC_OK;
}

/* ---- filters.c:5-28, 43-70 ---- */
static const struct {
    const char* name;
    int (*fn)(orc_image**, char*);
    int experimental;
} filter_map[] = {
    {"flip", f_flip, 0},         {"rotate", f_rotate, 0},     {"modulate", f_modulate, 0},
    {"colorize", f_colorize, 0}, {"blur", f_blur, 0},         {"gamma", f_gamma, 0},
    {"contrast", f_contrast, 0}, {"gradmap", f_gradmap, 0},   {"vignette", f_vignette, 1},
    {"gotham", f_gotham, 1},     {"lomo", f_lomo, 1},         {"kelvin", f_kelvin, 1},
    {"rainbow", f_rainbow, 1},   {"scanline", f_scanline, 1},
};

int orc_filter(orc_image** pointer, const char* _request, int allow_experiments) {
    char* request = strdup(_request);
    char* ctx = NULL;
    char* type = strtok_r(request, "=", &ctx);
    if (!type) { free(request); return ORC_ERROR_NO_SUCH_FILTER; }
    char* args = strtok_r(NULL, "=", &ctx);
    if (!args) { free(request); return ORC_ERROR_INVALID_ARGS; }
    for (size_t i = 0; i < sizeof(filter_map) / sizeof(filter_map[0]); i++) {
        if (strcmp(type, filter_map[i].name) == 0 && (allow_experiments || !filter_map[i].experimental)) {
            int rc = filter_map[i].fn(pointer, args);
            free(request);
            return rc;
        }
    }
    free(request);
    return ORC_ERROR_NO_SUCH_FILTER;
}

/* ---- Watermark: bridge.c:239-281 + AlphaBlendOver filters.c:619-662 ---- */
int orc_watermark(orc_image* img, const orc_image* ov, char gx, char gy, int offx, int offy, int opacity_pct) {
    int basew = img->width, baseh = img->height, overw = ov->width, overh = ov->height;
    int left, top;
    if (gx == 'c') left = (basew - overw) / 2 + offx;
    else if (gx == 'r') left = basew - overw - offx;
    else left = offx;
    if (gy == 'c') top = (baseh - overh) / 2 + offy;
    else if (gy == 'b') top = baseh - overh - offy;
    else top = offy;

    /* cvSetImageROI (OpenCV 2.4.9 core/array.cpp): assertion, then clip to the image.
     * The assertion aborts the reference; defined here as INVALID_ARGS. */
    int rw = overw, rh = overh;
    if (!(left < basew && top < baseh && left + rw >= (rw > 0) && top + rh >= (rh > 0)))
        return ORC_ERROR_INVALID_ARGS;
    int rx = left < 0 ? 0 : left, ry = top < 0 ? 0 : top;

    float opacity = (float)(opacity_pct / 100.0);
    float alpha = 1 - opacity;
    int maxrow = (int)fmin(overh, baseh - ry);
    int maxcol = (int)fmin(overw, basew - rx);
    for (int row = 0; row < maxrow; row++)
        for (int col = 0; col < maxcol; col++) {
            unsigned char* d = PX(img, col + rx, row + ry);
            const unsigned char* s = PX(ov, col, row);
            int dB = d[0], dG = d[1], dR = d[2];
            float dA = img->channels == 4 ? (float)(d[3] / 255.0) : 1;
            int sB = s[0], sG = s[1], sR = s[2];
            float sA = ov->channels == 4 ? (float)(s[3] / 255.0) : 1;
            sA = (float)fmax(sA - alpha, 0);
            float tA = sA + dA * (1 - sA);
            int tB, tG, tR;
            if (tA == 0) tB = tG = tR = 0;
            else {
                tB = orc_trunc((sB * sA + dB * dA * (1 - sA)) / tA);
                tG = orc_trunc((sG * sA + dG * dA * (1 - sA)) / tA);
                tR = orc_trunc((sR * sA + dR * dA * (1 - sA)) / tA);
            }
            d[0] = orc_byte(tB); d[1] = orc_byte(tG); d[2] = orc_byte(tR);
            if (img->channels == 4) d[3] = orc_store(tA * 255);
        }
    return ORC_OK;
}

/* ---- filters.c:666-687 ---- */
void orc_blend_with_paper(orc_image* img) {
    if (img->channels != 4) return;
    for (int y = 0; y < img->height; y++)
        for (int x = 0; x < img->width; x++) {
            unsigned char* p = PX(img, x, y);
            int a = p[3];
            int diff = 255 - a;
            float prod = (float)(a / 255.0);
            int tb = orc_trunc(diff + (p[0] * prod));
            int tg = orc_trunc(diff + (p[1] * prod));
            int tr = orc_trunc(diff + (p[2] * prod));
            p[0] = orc_byte(tb); p[1] = orc_byte(tg); p[2] = orc_byte(tr); p[3] = 255;
        }
}

/* ---- filters.c:707-729: float accumulator, x-outer / y-inner order (order matters) ---- */
float orc_calc_perceived_brightness(const orc_image* img) {
    float sum = 0;
    if (img->channels == 1) {
        for (int x = 0; x < img->width; x++)
            for (int y = 0; y < img->height; y++) sum += *PX(img, x, y);
    } else {
        for (int x = 0; x < img->width; x++)
            for (int y = 0; y < img->height; y++) {
                const unsigned char* p = PX(img, x, y);
                int r = p[2], g = p[1], b = p[0];
                sum += sqrt(r * r * 0.241 + g * g * 0.691 + b * b * 0.068);
            }
    }
    return (float)(sum / (img->width * img->height) / 255.0);
}

/* ---- filters.c:486-522 ---- */
long orc_ascii(orc_image* img, const char* args, unsigned char* out) {
    static const unsigned char wide[] = "$@B%8&WM#*oahkbdpqwmZO0QLCJUYXzcvunxrjft/\\|()1{}[]?-_+~<>i!lI;:,\"^`'. ";
    static const unsigned char narrow[] = "@%8#*+=-:. ";
    const unsigned char* table = strcmp(args, "wide") == 0 ? wide : narrow;
    int tablelen = (int)strlen((const char*)table);
    float factor = (float)(256.0 / tablelen);
    int w = img->width, h = img->height;
    long buflen = (long)(w + 1) * h - 1;
    orc_rgb2hsv(img);
    for (int y = 0; y < h; y++) {
        long ro = (long)y * (w + 1);
        for (int x = 0; x < w; x++) {
            int density = (int)floor(PX(img, x, y)[2] / factor);
            out[ro + x] = table[density];
        }
        if (ro > 0) out[ro - 1] = '\n';
    }
    return buflen;
}
