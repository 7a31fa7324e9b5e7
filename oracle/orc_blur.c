/*
 * orc_blur.c -- restatement of OpenCV 2.4.9 cvSmooth(src, dst, CV_GAUSSIAN, 0, 0, sigma, 0)
 * for CV_8U images, as called in place by the reference's Blur (filters.c:204).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED: OpenCV 2.4.9 is absent here.  Follows
 * modules/imgproc/src/smooth.cpp (cvSmooth -> GaussianBlur -> createGaussianFilter ->
 * getGaussianKernel) and filter.cpp (createSeparableLinearFilter's 8-bit fixed-point
 * branch: kernels * 256 -> int32; row pass int32; column pass SymmColumnVec_32s8u, which
 * on x86-64 works in float for the first width*cn & ~3 elements of a row, with
 * FixedPtCastEx<int,uchar>(16) for the remainder), BORDER_REPLICATE.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "imp_oracle.h"
#include "orc_internal.h"

int orc_gaussian_ksize(double sigma) {
    /* createGaussianFilter: ksize = cvRound(sigma*3*2 + 1) | 1 for CV_8U */
    if (!(sigma > 0)) return 0;
    return orc_cvround(sigma * 3 * 2 + 1) | 1;
}

/* getGaussianKernel(n, sigma, CV_32F) then convertTo(CV_32S, 256) */
static void gaussian_kernel_fixed(int n, double sigma, int* ik) {
    float* cf = (float*)malloc(sizeof(float) * n);
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) {
        cf[i] = (float)(cf[i] * sum);
        ik[i] = orc_cvround(cf[i] * 256.f);
    }
    free(cf);
}

int orc_cv_smooth_gaussian(orc_image* img, double sigma) {
    if (sigma == 0) return ORC_OK;   /* reference: ksize 0 -> OpenCV assertion; defined as no-op */
    int kx = orc_gaussian_ksize(sigma), ky = kx;
    /* GaussianBlur: a 1-pixel-wide/high image forces that axis' ksize to 1 */
    if (img->height == 1) ky = 1;
    if (img->width == 1) kx = 1;
    if (kx == 1 && ky == 1) return ORC_OK;  /* src.copyTo(dst) */

    int* ikx = (int*)malloc(sizeof(int) * kx);
    int* iky = (int*)malloc(sizeof(int) * ky);
    gaussian_kernel_fixed(kx, sigma, ikx);
    gaussian_kernel_fixed(ky, sigma, iky);

    int w = img->width, h = img->height, cn = img->channels, roww = w * cn;
    int rx = kx / 2, ry = ky / 2;
    int* tmp = (int*)malloc(sizeof(int) * (size_t)roww * h);
    for (int y = 0; y < h; y++) {
        const unsigned char* S = img->data + (size_t)y * img->step;
        int* T = tmp + (size_t)y * roww;
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                int s = 0;
                for (int k = 0; k < kx; k++) {
                    int sx = x + k - rx;
                    sx = sx < 0 ? 0 : sx > w - 1 ? w - 1 : sx;
                    s += S[sx * cn + c] * ikx[k];
                }
                T[x * cn + c] = s;
            }
    }
    float* fky = (float*)malloc(sizeof(float) * (ry + 1));
    for (int k = 0; k <= ry; k++) fky[k] = (float)(iky[ry + k] * (1. / 65536));
    int vec_end = roww & ~3;
    for (int y = 0; y < h; y++) {
        unsigned char* D = img->data + (size_t)y * img->step;
        for (int i = 0; i < roww; i++) {
            const int* c0 = tmp + (size_t)y * roww + i;
            if (i < vec_end) {
                float s = (float)(*c0) * fky[0] + 0.f;
                for (int k = 1; k <= ry; k++) {
                    int ya = y + k > h - 1 ? h - 1 : y + k, yb = y - k < 0 ? 0 : y - k;
                    int x0 = tmp[(size_t)ya * roww + i] + tmp[(size_t)yb * roww + i];
                    s = s + (float)x0 * fky[k];
                }
                D[i] = orc_sat_u8(orc_cvround(s));
            } else {
                int s = iky[ry] * (*c0);
                for (int k = 1; k <= ry; k++) {
                    int ya = y + k > h - 1 ? h - 1 : y + k, yb = y - k < 0 ? 0 : y - k;
                    s += iky[ry + k] * (tmp[(size_t)ya * roww + i] + tmp[(size_t)yb * roww + i]);
                }
                D[i] = orc_sat_u8((s + (1 << 15)) >> 16);
            }
        }
    }
    free(fky); free(tmp); free(ikx); free(iky);
    return ORC_OK;
}
