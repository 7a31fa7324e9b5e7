// imp_tables.cpp -- host-side coefficient tables for the resamplers and the Gaussian.
//
// These are the parts of cvResize / cvSmooth (reference call sites bridge.c:191 and
// filters.c:204) that OpenCV 2.4.9 evaluates once per call in double / float and libm
// (imgwarp.cpp: coefficient loops of cv::resize, interpolateCubic, interpolateLanczos4,
// computeResizeAreaTab; smooth.cpp: getGaussianKernel).  They are a few KB per geometry,
// so they are built here in the same arithmetic, cached per geometry and uploaded once;
// the kernels only consume the fixed-point / float entries.  Built with
// -ffp-contract=off: the float expressions below must round after every operation.
#include <cfloat>
#include <cmath>
#include "imp_internal.h"

namespace imp {

static inline int round_half_even(double v) { return (int)std::lrint(v); }   // cvRound
static inline short clamp_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

static void weights_cubic(float x, float* w) {
    const float A = -0.75f;
    w[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    w[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    w[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    w[3] = 1.f - w[0] - w[1] - w[2];
}

static void weights_lanczos4(float x, float* w) {
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45},
                                    {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    static const double pi = 3.1415926535897932384626433832795;
    if (x < FLT_EPSILON) {
        for (int i = 0; i < 8; i++) w[i] = 0;
        w[3] = 1;
        return;
    }
    float sum = 0;
    double y0 = -(x + 3) * pi * 0.25, s0 = std::sin(y0), c0 = std::cos(y0);
    for (int i = 0; i < 8; i++) {
        double y = -(x + 3 - i) * pi * 0.25;
        w[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        sum += w[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++) w[i] *= sum;
}

void build_tap_axis(int ssize, int dsize, double scale, int interp, bool is_x, TapAxis* out) {
    const int ksize = interp == IMP_INTER_LINEAR ? 2 : (interp == IMP_INTER_CUBIC ? 4 : 8);
    out->ksize = ksize;
    out->ofs.resize(dsize);
    out->coef.resize((size_t)dsize * ksize);
    float w[8];
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (is_x) {     // cv::resize's xofs loop, OpenCV 2.4.9: unconditional for every generic mode (the CUBIC /
            if (s < 0) { f = 0; s = 0; }                    // LANCZOS4 exemption only arrived in 3.x), x axis only:
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }   // an enlargement's edge columns copy src[0] / src[w-1]
        }
        out->ofs[d] = s;
        if (interp == IMP_INTER_CUBIC) weights_cubic(f, w);
        else if (interp == IMP_INTER_LANCZOS4) weights_lanczos4(f, w);
        else { w[0] = 1.f - f; w[1] = f; }
        for (int k = 0; k < ksize; k++)
            out->coef[(size_t)d * ksize + k] = clamp_short(round_half_even(w[k] * 2048.f));
    }
}

void build_area_axis(int ssize, int dsize, double scale, AreaAxis* out) {
    out->start.assign(dsize, 0);
    out->count.assign(dsize, 0);
    out->aoff.assign(dsize, 0);
    out->alpha.clear();
    out->max_count = 0;
    for (int d = 0; d < dsize; d++) {
        double f1 = d * scale, f2 = f1 + scale;
        double cell = std::fmin(scale, ssize - f1);
        int s1 = (int)std::ceil(f1), s2 = (int)std::floor(f2);
        if (s2 > ssize - 1) s2 = ssize - 1;
        if (s1 > s2) s1 = s2;
        int first = -1, n = 0;
        out->aoff[d] = (int)out->alpha.size();
        if (s1 - f1 > 1e-3) { first = s1 - 1; out->alpha.push_back((float)((s1 - f1) / cell)); n++; }
        for (int s = s1; s < s2; s++) { if (first < 0) first = s; out->alpha.push_back((float)(1.0 / cell)); n++; }
        if (f2 - s2 > 1e-3) {
            if (first < 0) first = s2;
            out->alpha.push_back((float)(std::fmin(std::fmin(f2 - s2, 1.), cell) / cell));
            n++;
        }
        out->start[d] = first < 0 ? 0 : first;
        out->count[d] = n;
        if (n > out->max_count) out->max_count = n;
    }
}

// The widest cell of an axis (AreaAxis::max_count without building the axis): what picks the window class of the
// kernels that compute their weights themselves.
int area_max_count(int ssize, int dsize, double scale) {
    // called for every frame of a mixed-geometry batch, sometimes by two planners in a row: the last answer is kept, and
    // ceil / floor are spelt with integer conversions (the operands are non-negative and below 2^31, so the values are
    // the same): this loop was most of impgpu_batch_resize_mixed's host time
    static thread_local struct { int ssize, dsize; double scale; int most; } last = {0, 0, 0.0, 0};
    if (last.ssize == ssize && last.dsize == dsize && last.scale == scale && dsize > 0) return last.most;
    int most = 0;
    // a cell of length `scale` touches at most floor(scale) + 2 source pixels (partial, whole ..., partial): the walk ends
    // at the first cell that does -- after about 1 / frac(scale) cells (6.1 -> 0.3 ms per 4096 frames of the mixed stream)
    const int hi = scale < 2147483000.0 ? (int)scale + 2 : 0x7fffffff;
    for (int d = 0; d < dsize && most < hi; d++) {
        const double f1 = d * scale, f2 = f1 + scale;
        const int t1 = (int)f1;
        int s1 = t1 + ((double)t1 < f1), s2 = (int)f2;
        if (s2 > ssize - 1) s2 = ssize - 1;
        if (s1 > s2) s1 = s2;
        const int n = (s1 - f1 > 1e-3) + (s2 - s1) + (f2 - s2 > 1e-3);
        if (n > most) most = n;
    }
    last = {ssize, dsize, scale, most};
    return most;
}

int gaussian_ksize(double sigma) {
    if (!(sigma > 0)) return 0;
    return round_half_even(sigma * 3 * 2 + 1) | 1;
}

void gaussian_kernel_fixed(int n, double sigma, std::vector<int>* ik) {
    std::vector<float> cf(n);
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        cf[i] = (float)std::exp(scale2X * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    ik->resize(n);
    for (int i = 0; i < n; i++) {
        cf[i] = (float)(cf[i] * sum);
        (*ik)[i] = round_half_even(cf[i] * 256.f);
    }
}

}  // namespace imp
