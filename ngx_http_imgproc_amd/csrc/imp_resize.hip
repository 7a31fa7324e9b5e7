// imp_resize.hip -- cvResize (reference call site bridge.c:189-191) as gfx950 kernels.
//
// OpenCV 2.4.9 semantics for CV_8U, 1/3/4 interleaved channels (imgwarp.cpp), x86-64 build:
//   NN        nearest source index  min(floor(d * scale), size-1)
//   LINEAR    2 taps,  11-bit fixed-point weights, VResizeLinear's >>4 / >>16 / +2 >>2
//   CUBIC     4 taps,  11-bit weights; vertical pass in float (VResizeCubicVec_32s8u) for the
//             first (dw*cn & ~7) elements of a row, (v + 2^21) >> 22 for the rest
//   LANCZOS4  8 taps,  (v + 2^21) >> 22
//   AREA      integer scales: box sum ((a+b+c+d+2)>>2 for 2x2, else round(sum * 1.f/area));
//             otherwise float accumulation over per-axis (source run, weight) tables
//
// All of these are HBM-bound byte shuffles (no contraction, no MFMA).  Layout: one thread
// per destination pixel, 256-thread blocks over the flattened (dy, dx) index so a wave's 64
// lanes walk 64 neighbouring dx of one destination row: their tap windows fall in the same
// few source rows and every 64-byte sector a wave touches is consumed by neighbouring lanes
// of the same instruction.  blockIdx.y is the frame of the batch.  For BGRA the KS taps of a
// row are one (4-byte aligned) 8/16/32-byte vector load per lane.  The per-geometry weight
// tables (a few KB, built on the host in imp_tables.cpp) stay L1/L2 resident.
#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
#include "imp_internal.h"

namespace imp {

struct RArgs {
    const uint8_t* src; long long src_stride; int sstep, sw, sh;
    uint8_t* dst; long long dst_stride; int dstep, dw, dh;
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// ------------------------------------------------------------------ LINEAR / CUBIC / LANCZOS4
enum { M_LINEAR = 0, M_CUBIC = 1, M_LANCZOS = 2 };

template <int KS, int CN, int MODE>
__global__ __launch_bounds__(256) void k_resize_taps(RArgs a, const int* __restrict__ xofs,
                                                     const short* __restrict__ xco,
                                                     const int* __restrict__ yofs,
                                                     const short* __restrict__ yco, int vec_end) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.dw * a.dh) return;
    const int dy = idx / a.dw, dx = idx - dy * a.dw;
    const uint8_t* S = a.src + (long long)blockIdx.y * a.src_stride;
    uint8_t* D = a.dst + (long long)blockIdx.y * a.dst_stride + (size_t)dy * a.dstep + (size_t)dx * CN;

    const int sx0 = xofs[dx] - (KS / 2 - 1);
    const int sy0 = yofs[dy] - (KS / 2 - 1);
    int ax[KS], by[KS];
#pragma unroll
    for (int k = 0; k < KS; k++) { ax[k] = xco[dx * KS + k]; by[k] = yco[dy * KS + k]; }

    int hs[KS][CN];     // horizontal pass, one int32 per tap row and channel
    const bool interior = sx0 >= 0 && sx0 + KS <= a.sw;
    if (CN == 4 && interior) {
        uint32_t px[KS][KS];
#pragma unroll
        for (int r = 0; r < KS; r++) {
            const int sy = clampi(sy0 + r, 0, a.sh - 1);
            const uint8_t* row = S + (size_t)sy * a.sstep + (size_t)sx0 * 4;
            __builtin_memcpy(px[r], __builtin_assume_aligned(row, 4), KS * 4);
        }
#pragma unroll
        for (int r = 0; r < KS; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                int v = 0;
#pragma unroll
                for (int k = 0; k < KS; k++) v += (int)((px[r][k] >> (8 * c)) & 0xff) * ax[k];
                hs[r][c] = v;
            }
    } else {
        int sxk[KS];
#pragma unroll
        for (int k = 0; k < KS; k++) sxk[k] = clampi(sx0 + k, 0, a.sw - 1) * CN;
#pragma unroll
        for (int r = 0; r < KS; r++) {
            const int sy = clampi(sy0 + r, 0, a.sh - 1);
            const uint8_t* row = S + (size_t)sy * a.sstep;
#pragma unroll
            for (int c = 0; c < CN; c++) {
                int v = 0;
#pragma unroll
                for (int k = 0; k < KS; k++) v += (int)row[sxk[k] + c] * ax[k];
                hs[r][c] = v;
            }
        }
    }

    int out[CN];
#pragma unroll
    for (int c = 0; c < CN; c++) {
        if (MODE == M_LINEAR) {
            out[c] = (uint8_t)((((by[0] * (hs[0][c] >> 4)) >> 16) + ((by[1] * (hs[1][c] >> 4)) >> 16) + 2) >> 2);
        } else if (MODE == M_CUBIC) {
            if (dx * CN + c < vec_end) {
                const float sc = 1.f / (2048.f * 2048.f);
                float s = __fmul_rn(__int2float_rn(hs[0][c]), __fmul_rn((float)by[0], sc));
                s = __fadd_rn(s, __fmul_rn(__int2float_rn(hs[1][c]), __fmul_rn((float)by[1], sc)));
                s = __fadd_rn(s, __fmul_rn(__int2float_rn(hs[2][c]), __fmul_rn((float)by[2], sc)));
                s = __fadd_rn(s, __fmul_rn(__int2float_rn(hs[3][c]), __fmul_rn((float)by[3], sc)));
                out[c] = sat_u8(__float2int_rn(s));
            } else {
                int v = hs[0][c] * by[0] + hs[1][c] * by[1] + hs[2][c] * by[2] + hs[3][c] * by[3];
                out[c] = sat_u8((v + (1 << 21)) >> 22);
            }
        } else {
            uint32_t v = 0;     // int32 wrap-around like the CPU build
#pragma unroll
            for (int k = 0; k < KS; k++) v += (uint32_t)(hs[k][c] * by[k]);
            out[c] = sat_u8(((int)(v + (1u << 21))) >> 22);
        }
    }
    if (CN == 4) {
        *(uint32_t*)D = (uint32_t)out[0] | ((uint32_t)out[1] << 8) | ((uint32_t)out[2] << 16) | ((uint32_t)out[3] << 24);
    } else {
#pragma unroll
        for (int c = 0; c < CN; c++) D[c] = (uint8_t)out[c];
    }
}

// ------------------------------------------------------------------ NN
template <int CN>
__global__ __launch_bounds__(256) void k_resize_nn(RArgs a, double scale_x, double scale_y) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.dw * a.dh) return;
    const int dy = idx / a.dw, dx = idx - dy * a.dw;
    int sx = (int)floor(dx * scale_x), sy = (int)floor(dy * scale_y);
    sx = sx > a.sw - 1 ? a.sw - 1 : sx;
    sy = sy > a.sh - 1 ? a.sh - 1 : sy;
    const uint8_t* s = a.src + (long long)blockIdx.y * a.src_stride + (size_t)sy * a.sstep + (size_t)sx * CN;
    uint8_t* d = a.dst + (long long)blockIdx.y * a.dst_stride + (size_t)dy * a.dstep + (size_t)dx * CN;
    if (CN == 4) *(uint32_t*)d = *(const uint32_t*)s;
    else {
#pragma unroll
        for (int c = 0; c < CN; c++) d[c] = s[c];
    }
}

// ------------------------------------------------------------------ AREA, integer scales
template <int CN>
__global__ __launch_bounds__(256) void k_resize_area_int(RArgs a, int isx, int isy) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.dw * a.dh) return;
    const int dy = idx / a.dw, dx = idx - dy * a.dw;
    const uint8_t* S = a.src + (long long)blockIdx.y * a.src_stride + (size_t)(dy * isy) * a.sstep + (size_t)(dx * isx) * CN;
    int sum[CN];
#pragma unroll
    for (int c = 0; c < CN; c++) sum[c] = 0;
    for (int ky = 0; ky < isy; ky++) {
        const uint8_t* row = S + (size_t)ky * a.sstep;
        for (int kx = 0; kx < isx; kx++) {
            if (CN == 4) {
                uint32_t p = *(const uint32_t*)(row + kx * 4);
                sum[0] += p & 0xff; sum[1] += (p >> 8) & 0xff; sum[2] += (p >> 16) & 0xff; sum[3] += p >> 24;
            } else {
#pragma unroll
                for (int c = 0; c < CN; c++) sum[c] += row[kx * CN + c];
            }
        }
    }
    uint8_t* d = a.dst + (long long)blockIdx.y * a.dst_stride + (size_t)dy * a.dstep + (size_t)dx * CN;
    int out[CN];
    if (isx == 2 && isy == 2) {
#pragma unroll
        for (int c = 0; c < CN; c++) out[c] = (sum[c] + 2) >> 2;
    } else {
        const float scale = 1.f / (float)(isx * isy);
#pragma unroll
        for (int c = 0; c < CN; c++) out[c] = sat_u8(__float2int_rn(__fmul_rn((float)sum[c], scale)));
    }
    if (CN == 4) *(uint32_t*)d = (uint32_t)out[0] | ((uint32_t)out[1] << 8) | ((uint32_t)out[2] << 16) | ((uint32_t)out[3] << 24);
    else {
#pragma unroll
        for (int c = 0; c < CN; c++) d[c] = (uint8_t)out[c];
    }
}

// ------------------------------------------------------------------ AREA, general (float tables)
struct AreaDev {
    const int *xstart, *xcount, *xaoff; const float* xalpha;
    const int *ystart, *ycount, *yaoff; const float* yalpha;
};

template <int CN>
__global__ __launch_bounds__(256) void k_resize_area(RArgs a, AreaDev t) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.dw * a.dh) return;
    const int dy = idx / a.dw, dx = idx - dy * a.dw;
    const uint8_t* S = a.src + (long long)blockIdx.y * a.src_stride;
    const int xs = t.xstart[dx], nx = t.xcount[dx];
    const float* xa = t.xalpha + t.xaoff[dx];
    const int ys = t.ystart[dy], ny = t.ycount[dy];
    const float* ya = t.yalpha + t.yaoff[dy];
    float sum[CN];
#pragma unroll
    for (int c = 0; c < CN; c++) sum[c] = 0.f;
    for (int j = 0; j < ny; j++) {
        const uint8_t* row = S + (size_t)(ys + j) * a.sstep + (size_t)xs * CN;
        float buf[CN];
#pragma unroll
        for (int c = 0; c < CN; c++) buf[c] = 0.f;
        for (int k = 0; k < nx; k++) {
            const float al = xa[k];
            if (CN == 4) {
                const uint32_t p = *(const uint32_t*)(row + k * 4);
                buf[0] = __fadd_rn(buf[0], __fmul_rn((float)(p & 0xff), al));
                buf[1] = __fadd_rn(buf[1], __fmul_rn((float)((p >> 8) & 0xff), al));
                buf[2] = __fadd_rn(buf[2], __fmul_rn((float)((p >> 16) & 0xff), al));
                buf[3] = __fadd_rn(buf[3], __fmul_rn((float)(p >> 24), al));
            } else {
#pragma unroll
                for (int c = 0; c < CN; c++) buf[c] = __fadd_rn(buf[c], __fmul_rn((float)row[k * CN + c], al));
            }
        }
        const float be = ya[j];
#pragma unroll
        for (int c = 0; c < CN; c++)
            sum[c] = (j == 0) ? __fmul_rn(be, buf[c]) : __fadd_rn(sum[c], __fmul_rn(be, buf[c]));
    }
    uint8_t* d = a.dst + (long long)blockIdx.y * a.dst_stride + (size_t)dy * a.dstep + (size_t)dx * CN;
    int out[CN];
#pragma unroll
    for (int c = 0; c < CN; c++) out[c] = sat_u8(__float2int_rn(sum[c]));
    if (CN == 4) *(uint32_t*)d = (uint32_t)out[0] | ((uint32_t)out[1] << 8) | ((uint32_t)out[2] << 16) | ((uint32_t)out[3] << 24);
    else {
#pragma unroll
        for (int c = 0; c < CN; c++) d[c] = (uint8_t)out[c];
    }
}

// ------------------------------------------------------------------ per-geometry table cache
struct TableSet {
    void* blob = nullptr;     // one device allocation
    const int *xofs = nullptr, *yofs = nullptr;
    const short *xco = nullptr, *yco = nullptr;
    AreaDev area{};
};
using Key = std::tuple<int, int, int, int, int>;
static std::map<Key, TableSet> g_tables;
static std::mutex g_tables_mu;

template <class T>
static size_t put(std::vector<uint8_t>& blob, const std::vector<T>& v) {
    while (blob.size() % 16) blob.push_back(0);
    size_t off = blob.size();
    const uint8_t* p = (const uint8_t*)v.data();
    blob.insert(blob.end(), p, p + v.size() * sizeof(T));
    return off;
}

static int get_tables(int interp, int sw, int sh, int dw, int dh, double scale_x, double scale_y, TableSet* out) {
    std::lock_guard<std::mutex> lk(g_tables_mu);
    Key key{interp, sw, sh, dw, dh};
    auto it = g_tables.find(key);
    if (it != g_tables.end()) { *out = it->second; return IMP_OK; }
    if (g_tables.size() >= 512) {   // bound the cache: drop everything once nothing can still be reading it
        IMP_HIP(hipDeviceSynchronize());
        for (auto& kv : g_tables) (void)hipFree(kv.second.blob);
        g_tables.clear();
    }
    std::vector<uint8_t> blob;
    TableSet ts;
    size_t o[8] = {0};
    if (interp == IMP_INTER_AREA) {
        AreaAxis ax, ay;
        build_area_axis(sw, dw, scale_x, &ax);
        build_area_axis(sh, dh, scale_y, &ay);
        o[0] = put(blob, ax.start); o[1] = put(blob, ax.count); o[2] = put(blob, ax.aoff); o[3] = put(blob, ax.alpha);
        o[4] = put(blob, ay.start); o[5] = put(blob, ay.count); o[6] = put(blob, ay.aoff); o[7] = put(blob, ay.alpha);
    } else {
        TapAxis tx, ty;
        build_tap_axis(sw, dw, scale_x, interp, true, &tx);
        build_tap_axis(sh, dh, scale_y, interp, false, &ty);
        o[0] = put(blob, tx.ofs); o[1] = put(blob, tx.coef); o[2] = put(blob, ty.ofs); o[3] = put(blob, ty.coef);
    }
    uint8_t* dev = nullptr;
    IMP_HIP(hipMalloc((void**)&dev, blob.size()));
    hipError_t e = hipMemcpy(dev, blob.data(), blob.size(), hipMemcpyHostToDevice);   // blocking: visible to every stream
    if (e != hipSuccess) { set_error("hipMemcpy(tables)", e); (void)hipFree(dev); return IMP_ERROR_DEVICE; }
    ts.blob = dev;
    if (interp == IMP_INTER_AREA) {
        ts.area.xstart = (const int*)(dev + o[0]); ts.area.xcount = (const int*)(dev + o[1]);
        ts.area.xaoff = (const int*)(dev + o[2]);  ts.area.xalpha = (const float*)(dev + o[3]);
        ts.area.ystart = (const int*)(dev + o[4]); ts.area.ycount = (const int*)(dev + o[5]);
        ts.area.yaoff = (const int*)(dev + o[6]);  ts.area.yalpha = (const float*)(dev + o[7]);
    } else {
        ts.xofs = (const int*)(dev + o[0]); ts.xco = (const short*)(dev + o[1]);
        ts.yofs = (const int*)(dev + o[2]); ts.yco = (const short*)(dev + o[3]);
    }
    g_tables[key] = ts;
    *out = ts;
    return IMP_OK;
}

// ------------------------------------------------------------------ launcher
template <int CN>
static int launch_cn(const RArgs& a, int count, int interp, double scale_x, double scale_y, hipStream_t s) {
    const dim3 block(256), grid((unsigned)(((long long)a.dw * a.dh + 255) / 256), (unsigned)count);
    if (interp == IMP_INTER_NN) {
        hipLaunchKernelGGL((k_resize_nn<CN>), grid, block, 0, s, a, scale_x, scale_y);
    } else if (interp == IMP_INTER_AREA) {
        const int isx = (int)std::lrint(scale_x), isy = (int)std::lrint(scale_y);
        if (std::fabs(scale_x - isx) < 2.220446049250313e-16 && std::fabs(scale_y - isy) < 2.220446049250313e-16) {
            hipLaunchKernelGGL((k_resize_area_int<CN>), grid, block, 0, s, a, isx, isy);
        } else {
            TableSet ts;
            if (int rc = get_tables(interp, a.sw, a.sh, a.dw, a.dh, scale_x, scale_y, &ts)) return rc;
            hipLaunchKernelGGL((k_resize_area<CN>), grid, block, 0, s, a, ts.area);
        }
    } else {
        TableSet ts;
        if (int rc = get_tables(interp, a.sw, a.sh, a.dw, a.dh, scale_x, scale_y, &ts)) return rc;
        if (interp == IMP_INTER_LINEAR)
            hipLaunchKernelGGL((k_resize_taps<2, CN, M_LINEAR>), grid, block, 0, s, a, ts.xofs, ts.xco, ts.yofs, ts.yco, 0);
        else if (interp == IMP_INTER_CUBIC)
            hipLaunchKernelGGL((k_resize_taps<4, CN, M_CUBIC>), grid, block, 0, s, a, ts.xofs, ts.xco, ts.yofs, ts.yco,
                               (a.dw * CN) & ~7);
        else
            hipLaunchKernelGGL((k_resize_taps<8, CN, M_LANCZOS>), grid, block, 0, s, a, ts.xofs, ts.xco, ts.yofs, ts.yco, 0);
    }
    IMP_HIP(hipGetLastError());
    return IMP_OK;
}

int launch_cv_resize(const Frames& f, int interp, hipStream_t s) {
    if (f.count <= 0) return IMP_OK;
    if (f.count > 65535) return IMP_ERROR_INVALID_ARGS;
    if (interp < IMP_INTER_NN || interp > IMP_INTER_LANCZOS4) return IMP_ERROR_INVALID_ARGS;
    const View& v = f.v;
    if (v.w <= 0 || v.h <= 0 || f.dw <= 0 || f.dh <= 0) return IMP_ERROR_INVALID_ARGS;
    // cv::resize: scale = 1 / ((double)dsize / ssize)
    const double scale_x = 1. / ((double)f.dw / v.w), scale_y = 1. / ((double)f.dh / v.h);
    // the reference requests AREA only when neither axis grows (bridge.c:190)
    if (interp == IMP_INTER_AREA && !(scale_x >= 1 && scale_y >= 1)) return IMP_ERROR_INVALID_ARGS;
    if (v.c == 4 && (((uintptr_t)f.src | (uintptr_t)f.dst | (uintptr_t)v.step | (uintptr_t)f.dstep |
                      (uintptr_t)f.src_stride | (uintptr_t)f.dst_stride) & 3))
        return IMP_ERROR_INVALID_ARGS;    // BGRA rows must be 4-byte aligned (cvCreateImage guarantees it)
    RArgs a{f.src, f.src_stride, v.step, v.w, v.h, f.dst, f.dst_stride, f.dstep, f.dw, f.dh};
    switch (v.c) {
        case 1: return launch_cn<1>(a, f.count, interp, scale_x, scale_y, s);
        case 3: return launch_cn<3>(a, f.count, interp, scale_x, scale_y, s);
        case 4: return launch_cn<4>(a, f.count, interp, scale_x, scale_y, s);
    }
    return IMP_ERROR_INVALID_ARGS;
}

}  // namespace imp
