/*
 * orc_resize.c -- restatement of OpenCV 2.4.9 cv::resize for CV_8U, 1/3/4 channels
 * (called from the reference at bridge.c:191 as cvResize(image, resized, filter)).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED: OpenCV 2.4.9 (pinned by the reference's
 * docs/01 - Installation.md:25 and config:5) is absent from /root/reference and from
 * this image.  The algorithm below follows modules/imgproc/src/imgwarp.cpp of that
 * release as built for x86-64:
 *   - geometry: scale = 1./((double)dst/src)
 *   - NN:       resizeNN
 *   - LINEAR / CUBIC / LANCZOS4: resizeGeneric_ with 11-bit fixed-point coefficient
 *     tables (INTER_RESIZE_COEF_BITS), int32 horizontal pass, and per-mode vertical pass:
 *       linear   VResizeLinear<uchar,int,short>            (two >>, +2, >>2)
 *       cubic    VResizeCubicVec_32s8u (SSE2: float) for the first width*cn & ~7
 *                elements of a row, FixedPtCast<int,uchar,22> for the remainder
 *       lanczos4 FixedPtCast<int,uchar,22>
 *   - AREA:     resizeAreaFast_ for integer scales (2x2: (a+b+c+d+2)>>2), resizeArea_
 *     (float accumulation over DecimateAlpha tables) otherwise.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "imp_oracle.h"
#include "orc_internal.h"

#define COEF_BITS  11
#define COEF_SCALE (1 << COEF_BITS)

static int g_simd = 1;
void orc_set_cv_simd(int simd) { g_simd = simd; }

static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

/* imgwarp.cpp interpolateCubic (A = -0.75), float arithmetic */
static void cubic_coeffs(float x, float* c) {
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

/* imgwarp.cpp interpolateLanczos4 */
static void lanczos4_coeffs(float x, float* c) {
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45},
                                   {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    if (x < FLT_EPSILON) {
        for (int i = 0; i < 8; i++) c[i] = 0;
        c[3] = 1;
        return;
    }
    float sum = 0;
    double y0 = -(x + 3) * 3.1415926535897932384626433832795 * 0.25, s0 = sin(y0), c0 = cos(y0);
    for (int i = 0; i < 8; i++) {
        double y = -(x + 3 - i) * 3.1415926535897932384626433832795 * 0.25;
        c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        sum += c[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++) c[i] *= sum;
}

/* One axis of the coefficient tables of cv::resize's generic branch.
 * ofs[d] = first-tap-centre source index (may be out of range; taps are clamped when read),
 * coef[d*ksize + k] = fixed-point weights. is_x selects the x-only edge rule. */
static void build_axis(int ssize, int dsize, double scale, int interp, int is_x, int* ofs, short* coef) {
    int ksize = interp == ORC_INTER_LINEAR ? 2 : interp == ORC_INTER_CUBIC ? 4 : 8;
    float cbuf[8];
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floor(f);
        f -= s;
        /* imgwarp.cpp (2.4.9) cv::resize, xofs loop: `if( sx < 0 ) fx = 0, sx = 0;` and
         * `if( sx >= ssize.width-1 ) fx = 0, sx = ssize.width-1;` apply to LINEAR, CUBIC and LANCZOS4 alike
         * (3.x exempted CUBIC / LANCZOS4 later); the yofs loop has no such rule. */
        if (is_x) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        ofs[d] = s;
        if (interp == ORC_INTER_CUBIC) cubic_coeffs(f, cbuf);
        else if (interp == ORC_INTER_LANCZOS4) lanczos4_coeffs(f, cbuf);
        else { cbuf[0] = 1.f - f; cbuf[1] = f; }
        for (int k = 0; k < ksize; k++)
            coef[d * ksize + k] = sat_short(orc_cvround(cbuf[k] * COEF_SCALE));
    }
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

static int resize_generic(const orc_image* src, orc_image* dst, int interp, double scale_x, double scale_y) {
    int cn = src->channels, sw = src->width, sh = src->height, dw = dst->width, dh = dst->height;
    int ksize = interp == ORC_INTER_LINEAR ? 2 : interp == ORC_INTER_CUBIC ? 4 : 8;
    int ksize2 = ksize / 2;
    int* xofs = (int*)malloc(sizeof(int) * dw);
    int* yofs = (int*)malloc(sizeof(int) * dh);
    short* alpha = (short*)malloc(sizeof(short) * dw * ksize);
    short* beta = (short*)malloc(sizeof(short) * dh * ksize);
    build_axis(sw, dw, scale_x, interp, 1, xofs, alpha);
    build_axis(sh, dh, scale_y, interp, 0, yofs, beta);

    int roww = dw * cn;
    /* ring of horizontally-resampled rows keyed by source row, as resizeGeneric_Invoker keeps */
    int* rowbuf = (int*)malloc(sizeof(int) * (size_t)roww * ksize);
    int rowsy[8];
    for (int k = 0; k < ksize; k++) rowsy[k] = -1;
    const int* rows[8];
    int vec_end = g_simd ? (roww & ~7) : 0;   /* VResizeCubicVec_32s8u covers x <= width-8 in steps of 8 */

    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy];
        for (int k = 0; k < ksize; k++) {
            int sy = clampi(sy0 - ksize2 + 1 + k, 0, sh - 1);
            int slot = -1;
            for (int j = 0; j < ksize; j++) if (rowsy[j] == sy) { slot = j; break; }
            if (slot < 0) {
                /* take a slot not needed by this dy */
                for (int j = 0; j < ksize && slot < 0; j++) {
                    int needed = 0;
                    for (int kk = 0; kk < ksize; kk++)
                        if (rowsy[j] == clampi(sy0 - ksize2 + 1 + kk, 0, sh - 1)) needed = 1;
                    if (!needed || rowsy[j] < 0) slot = j;
                }
                rowsy[slot] = sy;
                const unsigned char* S = src->data + (size_t)sy * src->step;
                int* D = rowbuf + (size_t)slot * roww;
                for (int dx = 0; dx < dw; dx++) {
                    const short* a = alpha + dx * ksize;
                    int sx0 = xofs[dx] - ksize2 + 1;
                    for (int c = 0; c < cn; c++) {
                        int v = 0;
                        for (int k2 = 0; k2 < ksize; k2++)
                            v += S[clampi(sx0 + k2, 0, sw - 1) * cn + c] * a[k2];
                        D[dx * cn + c] = v;
                    }
                }
            }
            rows[k] = rowbuf + (size_t)slot * roww;
        }
        const short* b = beta + dy * ksize;
        unsigned char* D = dst->data + (size_t)dy * dst->step;
        if (interp == ORC_INTER_LINEAR) {
            for (int x = 0; x < roww; x++)
                D[x] = (unsigned char)((((b[0] * (rows[0][x] >> 4)) >> 16) + ((b[1] * (rows[1][x] >> 4)) >> 16) + 2) >> 2);
        } else if (interp == ORC_INTER_CUBIC) {
            const float scale = 1.f / (COEF_SCALE * COEF_SCALE);
            float b0 = b[0] * scale, b1 = b[1] * scale, b2 = b[2] * scale, b3 = b[3] * scale;
            int x = 0;
            for (; x < vec_end; x++) {
                float s = (float)rows[0][x] * b0;
                float f = (float)rows[1][x] * b1;
                s = s + f;
                f = (float)rows[2][x] * b2;
                s = s + f;
                f = (float)rows[3][x] * b3;
                s = s + f;
                D[x] = orc_sat_u8(orc_cvround(s));
            }
            for (; x < roww; x++) {
                int v = rows[0][x] * b[0] + rows[1][x] * b[1] + rows[2][x] * b[2] + rows[3][x] * b[3];
                D[x] = orc_sat_u8((v + (1 << 21)) >> 22);
            }
        } else {
            for (int x = 0; x < roww; x++) {
                unsigned int v = 0;   /* int32 wrap-around as the x86 build */
                for (int k = 0; k < 8; k++) v += (unsigned int)(rows[k][x] * b[k]);
                D[x] = orc_sat_u8(((int)(v + (1u << 21))) >> 22);
            }
        }
    }
    free(rowbuf); free(xofs); free(yofs); free(alpha); free(beta);
    return ORC_OK;
}

static void resize_nn(const orc_image* src, orc_image* dst, double scale_x, double scale_y) {
    int cn = src->channels;
    for (int y = 0; y < dst->height; y++) {
        int sy = (int)floor(y * scale_y);
        if (sy > src->height - 1) sy = src->height - 1;
        const unsigned char* S = src->data + (size_t)sy * src->step;
        unsigned char* D = dst->data + (size_t)y * dst->step;
        for (int x = 0; x < dst->width; x++) {
            int sx = (int)floor(x * scale_x);
            if (sx > src->width - 1) sx = src->width - 1;
            memcpy(D + x * cn, S + sx * cn, (size_t)cn);
        }
    }
}

/* resizeAreaFast_: integer scale factors */
static void resize_area_fast(const orc_image* src, orc_image* dst, int isx, int isy) {
    int cn = src->channels;
    int area = isx * isy;
    float scale = 1.f / area;
    for (int dy = 0; dy < dst->height; dy++) {
        unsigned char* D = dst->data + (size_t)dy * dst->step;
        for (int dx = 0; dx < dst->width; dx++)
            for (int c = 0; c < cn; c++) {
                int sum = 0;
                for (int ky = 0; ky < isy; ky++) {
                    const unsigned char* S = src->data + (size_t)(dy * isy + ky) * src->step;
                    for (int kx = 0; kx < isx; kx++) sum += S[(dx * isx + kx) * cn + c];
                }
                if (isx == 2 && isy == 2) D[dx * cn + c] = (unsigned char)((sum + 2) >> 2);
                else D[dx * cn + c] = orc_sat_u8(orc_cvround(sum * scale));
            }
    }
}

typedef struct { int si, di; float alpha; } decimate_alpha;

/* imgwarp.cpp computeResizeAreaTab */
static int area_tab(int ssize, int dsize, double scale, decimate_alpha* tab) {
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale;
        double fsx2 = fsx1 + scale;
        double cell = fmin(scale, ssize - fsx1);
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        if (sx1 - fsx1 > 1e-3) { tab[k].di = dx; tab[k].si = sx1 - 1; tab[k++].alpha = (float)((sx1 - fsx1) / cell); }
        for (int sx = sx1; sx < sx2; sx++) { tab[k].di = dx; tab[k].si = sx; tab[k++].alpha = (float)(1.0 / cell); }
        if (fsx2 - sx2 > 1e-3) { tab[k].di = dx; tab[k].si = sx2; tab[k++].alpha = (float)(fmin(fmin(fsx2 - sx2, 1.), cell) / cell); }
    }
    return k;
}

/* resizeArea_<uchar,float> */
static void resize_area(const orc_image* src, orc_image* dst, double scale_x, double scale_y) {
    int cn = src->channels, dw = dst->width, dh = dst->height;
    decimate_alpha* xtab = (decimate_alpha*)malloc(sizeof(decimate_alpha) * src->width * 2);
    decimate_alpha* ytab = (decimate_alpha*)malloc(sizeof(decimate_alpha) * src->height * 2);
    int xn = area_tab(src->width, dw, scale_x, xtab);
    int yn = area_tab(src->height, dh, scale_y, ytab);
    int roww = dw * cn;
    float* buf = (float*)malloc(sizeof(float) * roww);
    float* sum = (float*)malloc(sizeof(float) * roww);
    int prev_dy = ytab[0].di;
    for (int x = 0; x < roww; x++) sum[x] = 0;
    for (int j = 0; j < yn; j++) {
        float beta = ytab[j].alpha;
        int dy = ytab[j].di, sy = ytab[j].si;
        const unsigned char* S = src->data + (size_t)sy * src->step;
        for (int x = 0; x < roww; x++) buf[x] = 0;
        for (int k = 0; k < xn; k++) {
            int dxn = xtab[k].di * cn, sxn = xtab[k].si * cn;
            float a = xtab[k].alpha;
            for (int c = 0; c < cn; c++) buf[dxn + c] = buf[dxn + c] + S[sxn + c] * a;
        }
        if (dy != prev_dy) {
            unsigned char* D = dst->data + (size_t)prev_dy * dst->step;
            for (int x = 0; x < roww; x++) { D[x] = orc_sat_u8(orc_cvround(sum[x])); sum[x] = beta * buf[x]; }
            prev_dy = dy;
        } else {
            for (int x = 0; x < roww; x++) sum[x] = sum[x] + beta * buf[x];
        }
    }
    {
        unsigned char* D = dst->data + (size_t)prev_dy * dst->step;
        for (int x = 0; x < roww; x++) D[x] = orc_sat_u8(orc_cvround(sum[x]));
    }
    free(xtab); free(ytab); free(buf); free(sum);
}

int orc_cv_resize(const orc_image* src, orc_image* dst, int interp) {
    if (src->channels != dst->channels) return ORC_ERROR_INVALID_ARGS;
    double inv_x = (double)dst->width / src->width, inv_y = (double)dst->height / src->height;
    double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
    if (interp == ORC_INTER_NN) { resize_nn(src, dst, scale_x, scale_y); return ORC_OK; }
    if (interp == ORC_INTER_AREA) {
        /* the reference only requests AREA when neither axis grows (bridge.c:190) */
        if (!(scale_x >= 1 && scale_y >= 1)) return ORC_ERROR_INVALID_ARGS;
        int isx = orc_cvround(scale_x), isy = orc_cvround(scale_y);
        if (fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON)
            resize_area_fast(src, dst, isx, isy);
        else
            resize_area(src, dst, scale_x, scale_y);
        return ORC_OK;
    }
    if (interp != ORC_INTER_LINEAR && interp != ORC_INTER_CUBIC && interp != ORC_INTER_LANCZOS4)
        return ORC_ERROR_INVALID_ARGS;
    return resize_generic(src, dst, interp, scale_x, scale_y);
}
