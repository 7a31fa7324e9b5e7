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
// clamp(v >> sh) for the fixed-point casts.  The empty asm keeps the shift and the clamp apart:
// hipcc 7.2 otherwise fuses two of them plus the byte packing into v_ashr_pk_u8_i32, and on
// MI355X that instruction leaves bits 31:16 of its destination unchanged while the compiler
// assumes they are zero (seen as garbage OR-ed into channels 2 and 3 of the Lanczos output).
__device__ __forceinline__ int shr_sat_u8(int v, int sh) {
    int t = v >> sh;
    asm volatile("" : "+v"(t));
    return sat_u8(t);
}

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
                int v = __mul24(hs[0][c], by[0]) + __mul24(hs[1][c], by[1]) + __mul24(hs[2][c], by[2]) + __mul24(hs[3][c], by[3]);
                out[c] = shr_sat_u8(v + (1 << 21), 22);
            }
        } else {
            uint32_t v = 0;     // int32 wrap-around like the CPU build
#pragma unroll
            for (int k = 0; k < KS; k++) v += (uint32_t)__mul24(hs[k][c], by[k]);   // |hs| < 2^23 (weights <= 2048 each)
            out[c] = shr_sat_u8((int)(v + (1u << 21)), 22);
        }
    }
    if (CN == 4) {
        *(uint32_t*)D = (uint32_t)out[0] | ((uint32_t)out[1] << 8) | ((uint32_t)out[2] << 16) | ((uint32_t)out[3] << 24);
    } else {
#pragma unroll
        for (int c = 0; c < CN; c++) D[c] = (uint8_t)out[c];
    }
}

// ------------------------------------------------------------------ LDS-tiled separable variant
// For scale factors <= 2 per axis (4K -> 1080p Lanczos, every enlargement) neighbouring
// destination pixels share most of their taps: the gather kernel above would pull each source
// pixel through L1 up to KS*KS/scale^2 times.  Here a 256-thread block owns a 64 x 16 destination
// tile: (0) its source footprint (<= 136 x 40 BGRA pixels, edge-replicated) is staged in LDS with
// coalesced dword loads, (1) the horizontal pass runs once per (source row, destination column)
// out of LDS into an int32x4 LDS plane, (2) the vertical pass reads that plane 16 bytes per lane,
// conflict-free, and writes coalesced dwords.  Every source byte leaves HBM once.
#define TL_TW 64
#define TL_TH 16
#define TL_SXW 136
#define TL_SYH 40

template <int KS, int MODE>
__global__ __launch_bounds__(256) void k_resize_tiled(RArgs a, const int* __restrict__ xofs,
                                                      const short* __restrict__ xco,
                                                      const int* __restrict__ yofs,
                                                      const short* __restrict__ yco, int vec_end) {
    __shared__ uint32_t s_src[TL_SYH * TL_SXW];
    __shared__ __attribute__((aligned(16))) int4 s_hs[TL_SYH * TL_TW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tx0 = blockIdx.x * TL_TW, ty0 = blockIdx.y * TL_TH;
    const int txn = min(TL_TW, a.dw - tx0), tyn = min(TL_TH, a.dh - ty0);
    const uint8_t* S = a.src + (long long)blockIdx.z * a.src_stride;
    const int xlo = xofs[tx0] - (KS / 2 - 1), ylo = yofs[ty0] - (KS / 2 - 1);
    const int sxw = xofs[tx0 + txn - 1] + KS / 2 - xlo + 1;
    const int syh = yofs[ty0 + tyn - 1] + KS / 2 - ylo + 1;
    if (sxw > TL_SXW || syh > TL_SYH) return;   // cannot happen for scales <= 2 (launcher's condition); keeps LDS indexing safe

    // (0) source footprint -> LDS, replicate borders by clamping the coordinates
    // wave wv takes rows wv, wv+4, ...; a lane takes columns lane, lane+64, lane+128.  Four rows
    // (12 dword loads per lane) are issued before the first LDS store so their latencies overlap.
    {
        int sxc[3];
#pragma unroll
        for (int q = 0; q < 3; q++) sxc[q] = clampi(xlo + lane + 64 * q, 0, a.sw - 1) * 4;
        for (int r0 = wv; r0 < syh; r0 += 16) {
            uint32_t v[4][3];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = r0 + 4 * u;
                const uint8_t* row = S + (size_t)clampi(ylo + r, 0, a.sh - 1) * a.sstep;
#pragma unroll
                for (int q = 0; q < 3; q++)
                    v[u][q] = (r < syh && lane + 64 * q < sxw) ? *(const uint32_t*)(row + sxc[q]) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = r0 + 4 * u;
#pragma unroll
                for (int q = 0; q < 3; q++)
                    if (r < syh && lane + 64 * q < sxw) s_src[r * TL_SXW + lane + 64 * q] = v[u][q];
            }
        }
    }
    __syncthreads();

    // (1) horizontal pass: lane = destination column of the tile, waves stride over source rows
    if (lane < txn) {
        int ax[KS];
#pragma unroll
        for (int k = 0; k < KS; k++) ax[k] = xco[(tx0 + lane) * KS + k];
        const int sxl = xofs[tx0 + lane] - (KS / 2 - 1) - xlo;
        for (int r = wv; r < syh; r += 4) {
            const uint32_t* row = s_src + r * TL_SXW + sxl;
            int h0 = 0, h1 = 0, h2 = 0, h3 = 0;
#pragma unroll
            for (int k = 0; k < KS; k++) {
                const uint32_t p = row[k];
                h0 += (int)(p & 0xff) * ax[k]; h1 += (int)((p >> 8) & 0xff) * ax[k];
                h2 += (int)((p >> 16) & 0xff) * ax[k]; h3 += (int)(p >> 24) * ax[k];
            }
            s_hs[r * TL_TW + lane] = make_int4(h0, h1, h2, h3);
        }
    }
    __syncthreads();

    // (2) vertical pass
    if (lane < txn) {
        const int dx = tx0 + lane;
        for (int yl = wv; yl < tyn; yl += 4) {
            const int dy = ty0 + yl;
            const int syl = yofs[dy] - (KS / 2 - 1) - ylo;
            int by[KS];
#pragma unroll
            for (int k = 0; k < KS; k++) by[k] = yco[dy * KS + k];
            int4 h[KS];
#pragma unroll
            for (int k = 0; k < KS; k++) h[k] = s_hs[(syl + k) * TL_TW + lane];
            int out[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                int hc[KS];
#pragma unroll
                for (int k = 0; k < KS; k++) hc[k] = c == 0 ? h[k].x : (c == 1 ? h[k].y : (c == 2 ? h[k].z : h[k].w));
                if (MODE == M_LINEAR) {
                    out[c] = (uint8_t)((((by[0] * (hc[0] >> 4)) >> 16) + ((by[1] * (hc[1] >> 4)) >> 16) + 2) >> 2);
                } else if (MODE == M_CUBIC) {
                    if (dx * 4 + c < vec_end) {
                        const float sc = 1.f / (2048.f * 2048.f);
                        float s = __fmul_rn(__int2float_rn(hc[0]), __fmul_rn((float)by[0], sc));
                        s = __fadd_rn(s, __fmul_rn(__int2float_rn(hc[1]), __fmul_rn((float)by[1], sc)));
                        s = __fadd_rn(s, __fmul_rn(__int2float_rn(hc[2]), __fmul_rn((float)by[2], sc)));
                        s = __fadd_rn(s, __fmul_rn(__int2float_rn(hc[3]), __fmul_rn((float)by[3], sc)));
                        out[c] = sat_u8(__float2int_rn(s));
                    } else {
                        int v = __mul24(hc[0], by[0]) + __mul24(hc[1], by[1]) + __mul24(hc[2], by[2]) + __mul24(hc[3], by[3]);
                        out[c] = shr_sat_u8(v + (1 << 21), 22);
                    }
                } else {
                    uint32_t v = 0;
#pragma unroll
                    for (int k = 0; k < KS; k++) v += (uint32_t)__mul24(hc[k], by[k]);   // |hs| < 2^23: 24-bit multiply is exact mod 2^32
                    out[c] = shr_sat_u8((int)(v + (1u << 21)), 22);
                }
            }
            *(uint32_t*)(a.dst + (long long)blockIdx.z * a.dst_stride + (size_t)dy * a.dstep + (size_t)dx * 4) =
                (uint32_t)out[0] | ((uint32_t)out[1] << 8) | ((uint32_t)out[2] << 16) | ((uint32_t)out[3] << 24);
        }
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
    const float* xalpha_pad;    // [dw][4*nv] weights of each run, zero-padded (nv = 0: not built)
    int nv;
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

// BGRA with source runs of at most 4*NV pixels (scale_x < 4*NV - 1): every lane pulls its whole
// run of a source row with NV 16-byte loads instead of one dword per tap.  Weights come from the
// zero-padded table; adding pixel * 0.f leaves the (non-negative) partial sum unchanged, so the
// float sequence is still exactly resizeArea_'s.
template <int NV>
__global__ __launch_bounds__(256) void k_resize_area_v4(RArgs a, AreaDev t) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.dw * a.dh) return;
    const int dy = idx / a.dw, dx = idx - dy * a.dw;
    const uint8_t* S = a.src + (long long)blockIdx.y * a.src_stride;
    const int xs = t.xstart[dx];
    float al[NV * 4];
    __builtin_memcpy(al, __builtin_assume_aligned(t.xalpha_pad + (size_t)dx * (NV * 4), 16), NV * 16);
    const int ys = t.ystart[dy], ny = t.ycount[dy];
    const float* ya = t.yalpha + t.yaoff[dy];
    const bool whole = xs + NV * 4 <= a.sw;     // the padded run stays inside the row
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int j = 0; j < ny; j++) {
        const uint8_t* row = S + (size_t)(ys + j) * a.sstep + (size_t)xs * 4;
        uint32_t px[NV * 4];
        if (whole) {
            __builtin_memcpy(px, __builtin_assume_aligned(row, 4), NV * 16);
        } else {
#pragma unroll
            for (int k = 0; k < NV * 4; k++) px[k] = (xs + k < a.sw) ? *(const uint32_t*)(row + k * 4) : 0u;
        }
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
#pragma unroll
        for (int k = 0; k < NV * 4; k++) {
            b0 = __fadd_rn(b0, __fmul_rn((float)(px[k] & 0xff), al[k]));
            b1 = __fadd_rn(b1, __fmul_rn((float)((px[k] >> 8) & 0xff), al[k]));
            b2 = __fadd_rn(b2, __fmul_rn((float)((px[k] >> 16) & 0xff), al[k]));
            b3 = __fadd_rn(b3, __fmul_rn((float)(px[k] >> 24), al[k]));
        }
        const float be = ya[j];
        if (j == 0) { s0 = __fmul_rn(be, b0); s1 = __fmul_rn(be, b1); s2 = __fmul_rn(be, b2); s3 = __fmul_rn(be, b3); }
        else {
            s0 = __fadd_rn(s0, __fmul_rn(be, b0)); s1 = __fadd_rn(s1, __fmul_rn(be, b1));
            s2 = __fadd_rn(s2, __fmul_rn(be, b2)); s3 = __fadd_rn(s3, __fmul_rn(be, b3));
        }
    }
    uint8_t* d = a.dst + (long long)blockIdx.y * a.dst_stride + (size_t)dy * a.dstep + (size_t)dx * 4;
    *(uint32_t*)d = (uint32_t)sat_u8(__float2int_rn(s0)) | ((uint32_t)sat_u8(__float2int_rn(s1)) << 8) |
                    ((uint32_t)sat_u8(__float2int_rn(s2)) << 16) | ((uint32_t)sat_u8(__float2int_rn(s3)) << 24);
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
    size_t o[9] = {0};
    int nv = 0;
    if (interp == IMP_INTER_AREA) {
        AreaAxis ax, ay;
        build_area_axis(sw, dw, scale_x, &ax);
        build_area_axis(sh, dh, scale_y, &ay);
        o[0] = put(blob, ax.start); o[1] = put(blob, ax.count); o[2] = put(blob, ax.aoff); o[3] = put(blob, ax.alpha);
        o[4] = put(blob, ay.start); o[5] = put(blob, ay.count); o[6] = put(blob, ay.aoff); o[7] = put(blob, ay.alpha);
        if (ax.max_count <= 16) {        // zero-padded per-destination weight rows for k_resize_area_v4
            nv = (ax.max_count + 3) / 4;
            std::vector<float> pad((size_t)dw * nv * 4, 0.f);
            for (int d = 0; d < dw; d++)
                for (int k = 0; k < ax.count[d]; k++) pad[(size_t)d * nv * 4 + k] = ax.alpha[ax.aoff[d] + k];
            o[8] = put(blob, pad);
        }
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
        ts.area.xalpha_pad = nv ? (const float*)(dev + o[8]) : nullptr;
        ts.area.nv = nv;
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
            if (CN == 4 && ts.area.nv == 1) hipLaunchKernelGGL((k_resize_area_v4<1>), grid, block, 0, s, a, ts.area);
            else if (CN == 4 && ts.area.nv == 2) hipLaunchKernelGGL((k_resize_area_v4<2>), grid, block, 0, s, a, ts.area);
            else if (CN == 4 && ts.area.nv == 3) hipLaunchKernelGGL((k_resize_area_v4<3>), grid, block, 0, s, a, ts.area);
            else if (CN == 4 && ts.area.nv == 4) hipLaunchKernelGGL((k_resize_area_v4<4>), grid, block, 0, s, a, ts.area);
            else hipLaunchKernelGGL((k_resize_area<CN>), grid, block, 0, s, a, ts.area);
        }
    } else {
        TableSet ts;
        if (int rc = get_tables(interp, a.sw, a.sh, a.dw, a.dh, scale_x, scale_y, &ts)) return rc;
        // both scales <= 2: neighbouring outputs share taps -> LDS-tiled separable kernel (BGRA)
        if (CN == 4 && scale_x <= 2.0 && scale_y <= 2.0) {
            const dim3 tgrid((a.dw + TL_TW - 1) / TL_TW, (a.dh + TL_TH - 1) / TL_TH, (unsigned)count);
            if (tgrid.y > 65535) return IMP_ERROR_INVALID_ARGS;
            if (interp == IMP_INTER_LINEAR)
                hipLaunchKernelGGL((k_resize_tiled<2, M_LINEAR>), tgrid, block, 0, s, a, ts.xofs, ts.xco, ts.yofs, ts.yco, 0);
            else if (interp == IMP_INTER_CUBIC)
                hipLaunchKernelGGL((k_resize_tiled<4, M_CUBIC>), tgrid, block, 0, s, a, ts.xofs, ts.xco, ts.yofs, ts.yco, (a.dw * 4) & ~7);
            else
                hipLaunchKernelGGL((k_resize_tiled<8, M_LANCZOS>), tgrid, block, 0, s, a, ts.xofs, ts.xco, ts.yofs, ts.yco, 0);
        } else if (interp == IMP_INTER_LINEAR)
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
