// imp_pixel.hip -- the per-pixel operators of filters.c / helpers.c as gfx950 kernels:
//   * the fused pixel program: RGB2HSV / HSV2RGB (helpers.c:70-176), per-channel tables
//     (ModulateHSV, AlphaBlendAddColor, ApplyGamma, BrightnessContrast, Lomo), Gradmap,
//     Vignette, Rainbow, Scanline -- any run of pointwise filters is ONE read + ONE write of
//     the frame, where the reference makes up to six read-modify-write sweeps (Gotham);
//   * AlphaBlendOver (filters.c:619-662) for Watermark, BlendWithPaper (filters.c:666-687);
//   * CalcPerceivedBrightness (filters.c:707-729), ASCII (filters.c:486-522).
//
// Arithmetic mirrors the reference's C exactly: integer HSV with C division (toward zero),
// float sector maths with truncation, low-byte stores into `char`.  Built with
// -ffp-contract=off and explicit __f*_rn so no multiply-add is fused; IEEE float and
// double division / sqrt on gfx950 are correctly rounded, as on the CPU.
// All kernels are HBM-bound streams: coalesced dword (BGRA) accesses, tables in LDS.
#include <cmath>
#include <cstring>
#include "imp_internal.h"

namespace imp {

#define IMP_MAX_STAGES 24
#define IMP_MAX_TABLE_BYTES 12288

struct ProgDev {
    int n;
    int table_bytes;
    Stage st[IMP_MAX_STAGES];
};

// exact floor(n / d) for 0 <= n < 2^24, 1 <= d < 2^24 without the 32-bit divide expansion
__device__ __forceinline__ int udiv_small(int n, int d) {
    int q = (int)(__fmul_rn((float)n, __builtin_amdgcn_rcpf((float)d)));   // +-1 at most; fixed up below
    int r = n - q * d;
    q += (r >= d);
    q -= (r < 0);
    return q;
}
// C99 signed division (toward zero) for |n| < 2^24
__device__ __forceinline__ int sdiv_small(int n, int d) {
    int q = udiv_small(n < 0 ? -n : n, d);
    return n < 0 ? -q : q;
}

// helpers.c:70-107 on one pixel; b,g,r in -> h,s,v out (same registers)
__device__ __forceinline__ void px_rgb2hsv(int& c0, int& c1, int& c2) {
    const int b = c0, g = c1, r = c2;
    const int mn = min(b, min(g, r)), mx = max(b, max(g, r));
    const int delta = mx - mn;
    int h = 0, s = 0;
    const int v = mx;
    if (v != 0) s = udiv_small(255 * delta, v);
    if (s != 0) {
        if (mx == r) h = sdiv_small(30 * (g - b), delta);
        else if (mx == g) h = 60 + sdiv_small(30 * (b - r), delta);
        else h = 120 + sdiv_small(30 * (r - g), delta);
    }
    if (h < 0) h += 180;
    c0 = h & 0xff; c1 = s & 0xff; c2 = v;
}

// helpers.c:109-176 on one pixel; h,s,v in -> b,g,r out
__device__ __forceinline__ void px_hsv2rgb(int& c0, int& c1, int& c2) {
    float h = (float)(c0 * 2), s = (float)c1;
    const float v = (float)c2;
    int r, g, b;
    if (c1 == 0) {
        r = g = b = c2;
    } else {
        s = __fdiv_rn(s, 255.f);
        h = __fdiv_rn(h, 60.f);
        const int i = (int)floorf(h);
        const float f = __fsub_rn(h, (float)i);
        const int p = (int)__fmul_rn(v, __fsub_rn(1.f, s));
        const int q = (int)__fmul_rn(v, __fsub_rn(1.f, __fmul_rn(s, f)));
        const int t = (int)__fmul_rn(v, __fsub_rn(1.f, __fmul_rn(s, __fsub_rn(1.f, f))));
        const int vi = c2;
        switch (i) {
            case 0: r = vi; g = t; b = p; break;
            case 1: r = q; g = vi; b = p; break;
            case 2: r = p; g = vi; b = t; break;
            case 3: r = p; g = q; b = vi; break;
            case 4: r = t; g = p; b = vi; break;
            default: r = vi; g = p; b = q; break;
        }
    }
    c0 = b & 0xff; c1 = g & 0xff; c2 = r & 0xff;
}

// (char)float as x86-64 gcc does it: truncate, NaN / out of range -> INT_MIN, keep the low byte
__device__ __forceinline__ int store_f(float v) {
    int i = (v > -2147483904.f && v < 2147483648.f) ? (int)v : (int)0x80000000;
    return i & 0xff;
}

#define PIX_PER_THREAD 8

// one stage on one pixel held in registers (c3 = alpha, untouched unless a table covers it)
// VIG = false compiles the vignette stage out: its double-precision sqrt / cos inline to ~2000 instructions and
// 130 VGPRs (3 waves per SIMD), which every other program would pay for (host picks the variant per program)
template <int CN, bool VIG = true>
__device__ __forceinline__ void apply_stage(const Stage& st, int& c0, int& c1, int& c2, int& c3, int x, int y, const uint8_t* lut) {
    {
        switch (st.kind) {
            case ST_LUT4: {
                const uint8_t* t = lut + st.lut_off;
                c0 = t[c0];
                if (CN >= 3) { c1 = t[256 + c1]; c2 = t[512 + c2]; }
                if (CN == 4) c3 = t[768 + c3];
            } break;
            case ST_RGB2HSV: px_rgb2hsv(c0, c1, c2); break;
            case ST_HSV2RGB: px_hsv2rgb(c0, c1, c2); break;
            case ST_GRADMAP: {                      // filters.c:264-276 (table is R,G,B order)
                const uint8_t* t = lut + st.lut_off;
                const int off = ((c2 + c1 + c0) / 3) * 3;
                c2 = t[off]; c1 = t[off + 1]; c0 = t[off + 2];
            } break;
            case ST_VIGNETTE: if constexpr (VIG) {  // filters.c:312-317 with the mask of :693-703 inline
                const double ddx = (double)(st.i0 - x), ddy = (double)(st.i1 - y);
                const float dist = (float)sqrt(ddx * ddx + ddy * ddy);
                const float raw = __fmul_rn(__fdiv_rn(dist, st.f0), st.f1);
                const double cs = cos((double)raw);
                const double c2d = cs * cs;
                const float mask = (float)(c2d * c2d);
                c2 = store_f(__fmul_rn((float)c2, mask));
            } break;
            case ST_RAINBOW: {                      // filters.c:371-397
                int hue = c0 * 2, light = c2, sat = st.i0;
                if (light < 20) { light = 0; sat = 0; }
                else if (light > 254) sat = 0;
                else if (hue <= 10 || hue > 340) hue = 0;
                else if (hue < 35) hue = 30;
                else if (hue < 68) hue = 60;
                else if (hue < 150) hue = 120;
                else if (hue < 200) hue = 195;
                else if (hue < 250) hue = 225;
                else hue = 285;
                c0 = (hue >> 1) & 0xff;             // (char)(hue / 2.0): truncation
                c1 = sat; c2 = light;
            } break;
            case ST_SCANLINE: {                     // filters.c:434-451: period freq+width+1
                const int ph = y % (st.i0 + st.i1 + 1);
                if (ph >= st.i0 && ph < st.i0 + st.i1) { c1 = st.i2; c2 = st.i3; }
            } break;
        }
    }
}

// all stages of the program on one pixel
template <int CN, bool VIG = true>
__device__ __forceinline__ void run_stages(int& c0, int& c1, int& c2, int& c3, int x, int y, const ProgDev& prog, const uint8_t* lut) {
    for (int si = 0; si < prog.n; si++) apply_stage<CN, VIG>(prog.st[si], c0, c1, c2, c3, x, y, lut);
}

template <int CN, bool VIG>
__global__ __launch_bounds__(256) void k_pixel_program(uint8_t* base, long long stride, int w, int h, int step,
                                                       ProgDev prog, const uint8_t* __restrict__ tables) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[IMP_MAX_TABLE_BYTES];
    for (int i = threadIdx.x * 4; i < prog.table_bytes; i += 256 * 4)
        *(uint32_t*)(lut + i) = *(const uint32_t*)(tables + i);
    __syncthreads();
    uint8_t* img = base + (long long)blockIdx.y * stride;
    const long long npix = (long long)w * h;
    const long long first = (long long)blockIdx.x * (256 * PIX_PER_THREAD) + threadIdx.x;
#pragma unroll 1
    for (int it = 0; it < PIX_PER_THREAD; it++) {
        const long long idx = first + (long long)it * 256;
        if (idx >= npix) break;
        const int y = (int)(idx / w), x = (int)(idx - (long long)y * w);
        uint8_t* p = img + (size_t)y * step + (size_t)x * CN;
        int c0, c1, c2, c3 = 255;
        if (CN == 4) {
            const uint32_t u = *(const uint32_t*)p;
            c0 = u & 0xff; c1 = (u >> 8) & 0xff; c2 = (u >> 16) & 0xff; c3 = u >> 24;
        } else if (CN == 3) {
            c0 = p[0]; c1 = p[1]; c2 = p[2];
        } else {
            c0 = p[0]; c1 = c2 = 0;
        }
        run_stages<CN, VIG>(c0, c1, c2, c3, x, y, prog, lut);
        if (CN == 4) *(uint32_t*)p = (uint32_t)c0 | ((uint32_t)c1 << 8) | ((uint32_t)c2 << 16) | ((uint32_t)c3 << 24);
        else if (CN == 3) { p[0] = (uint8_t)c0; p[1] = (uint8_t)c1; p[2] = (uint8_t)c2; }
        else p[0] = (uint8_t)c0;
    }
}

// BGRA frames whose rows are contiguous (step == 4*w) and 16-byte aligned: the frame is one linear run of pixels,
// a lane moves four of them per 16-byte load / store (the coalescing sweet spot) and only derives (x, y) when a
// stage of the program needs coordinates.
// the same program on the four pixels of one group, stage by stage: the stage dispatch (wave-uniform branches) is
// paid once per group and the table reads of a LUT stage are in flight together instead of behind each pixel's own
// wait.  Pixel k sits at (x0 + k, y0) carried into the next row at x == w.  CN = 3: c[k][3] is a dummy.
template <int CN, bool VIG>
__device__ __forceinline__ void run_stages_x4(int (&c)[4][4], int x0, int y0, int w, const ProgDev& prog, const uint8_t* lut) {
    for (int si = 0; si < prog.n; si++) {
        const Stage& st = prog.st[si];
        switch (st.kind) {
            case ST_LUT4: {
                const uint8_t* t = lut + st.lut_off;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    c[k][0] = t[c[k][0]]; c[k][1] = t[256 + c[k][1]]; c[k][2] = t[512 + c[k][2]];
                    if (CN == 4) c[k][3] = t[768 + c[k][3]];
                }
            } break;
            case ST_GRADMAP: {
                const uint8_t* t = lut + st.lut_off;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int off = ((c[k][2] + c[k][1] + c[k][0]) / 3) * 3;
                    c[k][2] = t[off]; c[k][1] = t[off + 1]; c[k][0] = t[off + 2];
                }
            } break;
            case ST_RGB2HSV:
#pragma unroll
                for (int k = 0; k < 4; k++) px_rgb2hsv(c[k][0], c[k][1], c[k][2]);
                break;
            case ST_HSV2RGB:
#pragma unroll
                for (int k = 0; k < 4; k++) px_hsv2rgb(c[k][0], c[k][1], c[k][2]);
                break;
            default: {                               // coordinate-dependent stages
                int x = x0, y = y0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    apply_stage<CN, VIG>(st, c[k][0], c[k][1], c[k][2], c[k][3], x, y, lut);
                    if (++x == w) { x = 0; y++; }
                }
            } break;
        }
    }
}

// Frames whose rows are contiguous (step == CN * w) are one linear run of pixels; a lane moves four of them per
// group: 16 bytes (BGRA, one dwordx4) or 12 bytes (BGR -- what every JPEG decodes to -- one dwordx3 at a 4-byte
// aligned address; every (pixel, channel) sits at a compile-time byte of the three dwords).
// LUT_ONLY: the program is a single per-channel table (gamma, contrast, colorize and whatever the host composed into
// one table): no stage loop at all, every lookup of the thread's groups is independent.
template <int CN, int PV4_GROUPS, bool LUT_ONLY, bool VIG>   // groups per thread: 4 for big batches (bytes in flight), 1 for a single frame
__global__ __launch_bounds__(256) void k_pixel_program_v4(uint8_t* base, long long stride, int w, long long npix,
                                                          ProgDev prog, const uint8_t* __restrict__ tables, int need_xy) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[IMP_MAX_TABLE_BYTES];
    for (int i = threadIdx.x * 4; i < prog.table_bytes; i += 256 * 4)
        *(uint32_t*)(lut + i) = *(const uint32_t*)(tables + i);
    __syncthreads();
    uint32_t* img = (uint32_t*)(base + (long long)blockIdx.y * stride);
    const long long ngroups = npix >> 2;                        // npix % 4 == 0 (launcher)
    const long long first = (long long)blockIdx.x * (256 * PV4_GROUPS) + threadIdx.x;
    uint32_t v[PV4_GROUPS][CN];
#pragma unroll
    for (int it = 0; it < PV4_GROUPS; it++) {
        const long long gi = first + (long long)it * 256;
        if (gi < ngroups) __builtin_memcpy(v[it], __builtin_assume_aligned(img + gi * CN, CN == 4 ? 16 : 4), CN * 4);
    }
#pragma unroll
    for (int it = 0; it < PV4_GROUPS; it++) {
        const long long gi = first + (long long)it * 256;
        if (gi >= ngroups) break;
        int x = 0, y = 0;
        if (need_xy) { const long long pix = gi * 4; y = (int)(pix / w); x = (int)(pix - (long long)y * w); }
        int c[4][4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int ch = 0; ch < 4; ch++) {
                const int o = k * CN + ch;                      // byte index inside the group
                c[k][ch] = ch < CN ? (int)((v[it][o >> 2] >> (8 * (o & 3))) & 0xff) : 255;
            }
        }
        if constexpr (LUT_ONLY) {
            const uint8_t* t = lut + prog.st[0].lut_off;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                c[k][0] = t[c[k][0]]; c[k][1] = t[256 + c[k][1]]; c[k][2] = t[512 + c[k][2]];
                if (CN == 4) c[k][3] = t[768 + c[k][3]];
            }
        } else {
            run_stages_x4<CN, VIG>(c, x, y, w, prog, lut);
        }
        uint32_t o4[CN];
#pragma unroll
        for (int d = 0; d < CN; d++) o4[d] = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int ch = 0; ch < CN; ch++) {
                const int o = k * CN + ch;
                o4[o >> 2] |= (uint32_t)(c[k][ch] & 0xff) << (8 * (o & 3));
            }
        }
        __builtin_memcpy(__builtin_assume_aligned(img + gi * CN, CN == 4 ? 16 : 4), o4, CN * 4);
    }
}

int launch_pixel_program(uint8_t* d, long long stride, int w, int h, int c, int step, int count,
                         const PixelProgram& prog, hipStream_t s) {
    if (prog.empty() || count <= 0) return IMP_OK;
    if (count > 65535) return IMP_ERROR_INVALID_ARGS;
    if (prog.stages.size() > IMP_MAX_STAGES || prog.tables.size() > IMP_MAX_TABLE_BYTES) {
        // a long run of pointwise filters (a raised imgproc_max_filters): the reference just runs them one after another,
        // so cut the program into launches that fit.  Exact anywhere: every stage already rounds to 8 bits per channel,
        // which is all a frame in HBM holds between two launches.
        PixelProgram part;
        for (const Stage& st : prog.stages) {
            const size_t tb = st.kind == ST_LUT4 ? 1024 : (st.kind == ST_GRADMAP ? 768 : 0);
            if (part.stages.size() + 1 > IMP_MAX_STAGES || part.tables.size() + tb > IMP_MAX_TABLE_BYTES) {
                if (int rc = launch_pixel_program(d, stride, w, h, c, step, count, part, s)) return rc;
                part.clear();
            }
            Stage cp = st;
            if (tb) {
                cp.lut_off = (int)part.tables.size();
                part.tables.insert(part.tables.end(), prog.tables.begin() + st.lut_off, prog.tables.begin() + st.lut_off + tb);
            }
            part.stages.push_back(cp);
        }
        return launch_pixel_program(d, stride, w, h, c, step, count, part, s);
    }
    if (c == 4 && (((uintptr_t)d | (uintptr_t)step | (uintptr_t)stride) & 3)) return IMP_ERROR_INVALID_ARGS;
    ProgDev pd{};
    pd.n = (int)prog.stages.size();
    for (int i = 0; i < pd.n; i++) pd.st[i] = prog.stages[i];
    std::vector<uint8_t> tb = prog.tables;
    while (tb.size() % 4) tb.push_back(0);
    if (tb.empty()) tb.resize(4, 0);
    pd.table_bytes = (int)prog.tables.size();
    pd.table_bytes = (pd.table_bytes + 3) & ~3;
    void* dev_tables = nullptr;
    if (int rc = upload_small(tb.data(), tb.size(), &dev_tables, s)) return rc;
    hipError_t e;
    const long long npix = (long long)w * h;
    const dim3 grid((unsigned)((npix + 256 * PIX_PER_THREAD - 1) / (256 * PIX_PER_THREAD)), (unsigned)count), block(256);
    bool need_xy = false;
    for (const Stage& st : prog.stages) need_xy = need_xy || st.kind == ST_VIGNETTE || st.kind == ST_SCANLINE;
    const bool lut_only = prog.stages.size() == 1 && prog.stages[0].kind == ST_LUT4;
    bool has_vig = false;
    for (const Stage& st : prog.stages) has_vig = has_vig || st.kind == ST_VIGNETTE;
    const bool vec4 = c == 4 && step == 4 * w && (npix & 3) == 0 && !(((uintptr_t)d | (uintptr_t)stride) & 15);
    const bool vec3 = c == 3 && step == 3 * w && (npix & 3) == 0 && !(((uintptr_t)d | (uintptr_t)stride) & 3);
    if (vec4 || vec3) {
        const bool big = npix * count >= (16LL << 20);
        const int groups = big ? 4 : 1;                        // a single frame or a small album: one group per thread keeps every CU busy
        const dim3 vgrid((unsigned)(((npix >> 2) + 256 * groups - 1) / (256 * groups)), (unsigned)count);
        const uint8_t* tb_dev = (const uint8_t*)dev_tables;
#define IMP_PV4(CN_, G_)                                                                                                 \
    do {                                                                                                                 \
        if (lut_only) hipLaunchKernelGGL((k_pixel_program_v4<CN_, G_, true, false>), vgrid, block, 0, s, d, stride, w, npix, pd, tb_dev, 0); \
        else if (has_vig) hipLaunchKernelGGL((k_pixel_program_v4<CN_, G_, false, true>), vgrid, block, 0, s, d, stride, w, npix, pd, tb_dev, need_xy ? 1 : 0); \
        else hipLaunchKernelGGL((k_pixel_program_v4<CN_, G_, false, false>), vgrid, block, 0, s, d, stride, w, npix, pd, tb_dev, need_xy ? 1 : 0); \
    } while (0)
        if (vec4 && big) IMP_PV4(4, 4);
        else if (vec4) IMP_PV4(4, 1);
        else if (big) IMP_PV4(3, 4);
        else IMP_PV4(3, 1);
#undef IMP_PV4
    } else if (c == 4 && has_vig) hipLaunchKernelGGL((k_pixel_program<4, true>), grid, block, 0, s, d, stride, w, h, step, pd, (const uint8_t*)dev_tables);
    else if (c == 4) hipLaunchKernelGGL((k_pixel_program<4, false>), grid, block, 0, s, d, stride, w, h, step, pd, (const uint8_t*)dev_tables);
    else if (c == 3 && has_vig) hipLaunchKernelGGL((k_pixel_program<3, true>), grid, block, 0, s, d, stride, w, h, step, pd, (const uint8_t*)dev_tables);
    else if (c == 3) hipLaunchKernelGGL((k_pixel_program<3, false>), grid, block, 0, s, d, stride, w, h, step, pd, (const uint8_t*)dev_tables);
    else hipLaunchKernelGGL((k_pixel_program<1, false>), grid, block, 0, s, d, stride, w, h, step, pd, (const uint8_t*)dev_tables);
    e = hipGetLastError();
    // the table buffer goes back to the pool: at once in lane-stream order, or behind an event of a caller's stream
    dev_free_on(dev_tables, s);
    if (e != hipSuccess) { set_error("k_pixel_program", e); return IMP_ERROR_DEVICE; }
    return IMP_OK;
}

// ------------------------------------------------------------------ AlphaBlendOver, filters.c:619-662
template <int DC, int SC>
__global__ __launch_bounds__(256) void k_blend_over(uint8_t* base, long long stride, int step,
                                                    const uint8_t* __restrict__ ov, int ostep,
                                                    int rx, int ry, int maxcol, int maxrow, float alpha) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= maxcol * maxrow) return;
    const int row = idx / maxcol, col = idx - row * maxcol;
    uint8_t* d = base + (long long)blockIdx.y * stride + (size_t)(row + ry) * step + (size_t)(col + rx) * DC;
    const uint8_t* sp = ov + (size_t)row * ostep + (size_t)col * SC;
    if (DC == 4 && SC == 4 && !(((uintptr_t)d | (uintptr_t)sp) & 3)) {     // the common pair: one dword in, one dword out
        *(uint32_t*)d = blend_over_bgra(*(const uint32_t*)d, *(const uint32_t*)sp, alpha);
        return;
    }
    const int dB = d[0], dG = d[1], dR = d[2];
    const float dA = DC == 4 ? (float)((double)d[3] / 255.0) : 1.f;
    const int sB = sp[0], sG = sp[1], sR = sp[2];
    float sA = SC == 4 ? (float)((double)sp[3] / 255.0) : 1.f;
    sA = (float)fmax((double)__fsub_rn(sA, alpha), 0.0);
    const float inv = __fsub_rn(1.f, sA);
    const float tA = __fadd_rn(sA, __fmul_rn(dA, inv));
    int tB = 0, tG = 0, tR = 0;
    if (tA != 0.f) {
        tB = (int)__fdiv_rn(__fadd_rn(__fmul_rn((float)sB, sA), __fmul_rn(__fmul_rn((float)dB, dA), inv)), tA);
        tG = (int)__fdiv_rn(__fadd_rn(__fmul_rn((float)sG, sA), __fmul_rn(__fmul_rn((float)dG, dA), inv)), tA);
        tR = (int)__fdiv_rn(__fadd_rn(__fmul_rn((float)sR, sA), __fmul_rn(__fmul_rn((float)dR, dA), inv)), tA);
    }
    d[0] = (uint8_t)tB; d[1] = (uint8_t)tG; d[2] = (uint8_t)tR;
    if (DC == 4) d[3] = (uint8_t)store_f(__fmul_rn(tA, 255.f));
}

int launch_blend_over(uint8_t* d, long long stride, int w, int h, int c, int step, int count,
                      const impgpu_image* ov, int rx, int ry, int maxcol, int maxrow, float alpha, hipStream_t s) {
    if (count <= 0 || maxcol <= 0 || maxrow <= 0) return IMP_OK;
    if (count > 65535 || c < 3 || ov->c < 3) return IMP_ERROR_INVALID_ARGS;
    (void)w; (void)h;
    const dim3 grid((unsigned)(((long long)maxcol * maxrow + 255) / 256), (unsigned)count), block(256);
    if (c == 4 && ov->c == 4) hipLaunchKernelGGL((k_blend_over<4, 4>), grid, block, 0, s, d, stride, step, ov->d, ov->step, rx, ry, maxcol, maxrow, alpha);
    else if (c == 4) hipLaunchKernelGGL((k_blend_over<4, 3>), grid, block, 0, s, d, stride, step, ov->d, ov->step, rx, ry, maxcol, maxrow, alpha);
    else if (ov->c == 4) hipLaunchKernelGGL((k_blend_over<3, 4>), grid, block, 0, s, d, stride, step, ov->d, ov->step, rx, ry, maxcol, maxrow, alpha);
    else hipLaunchKernelGGL((k_blend_over<3, 3>), grid, block, 0, s, d, stride, step, ov->d, ov->step, rx, ry, maxcol, maxrow, alpha);
    IMP_HIP(hipGetLastError());
    return IMP_OK;
}

// ------------------------------------------------------------------ BlendWithPaper, filters.c:666-687
__global__ __launch_bounds__(256) void k_blend_paper(uint8_t* base, long long stride, int w, int h, int step) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)w * h) return;
    const int y = (int)(idx / w), x = (int)(idx - (long long)y * w);
    uint32_t* p = (uint32_t*)(base + (long long)blockIdx.y * stride + (size_t)y * step + (size_t)x * 4);
    const uint32_t u = *p;
    const int a = u >> 24;
    const int diff = 255 - a;
    const float prod = (float)((double)a / 255.0);
    const int tb = (int)__fadd_rn((float)diff, __fmul_rn((float)(u & 0xff), prod));
    const int tg = (int)__fadd_rn((float)diff, __fmul_rn((float)((u >> 8) & 0xff), prod));
    const int tr = (int)__fadd_rn((float)diff, __fmul_rn((float)((u >> 16) & 0xff), prod));
    *p = (uint32_t)(tb & 0xff) | ((uint32_t)(tg & 0xff) << 8) | ((uint32_t)(tr & 0xff) << 16) | 0xff000000u;
}

int launch_blend_paper(uint8_t* d, long long stride, int w, int h, int step, int count, hipStream_t s) {
    if (count <= 0) return IMP_OK;
    if (count > 65535 || (((uintptr_t)d | (uintptr_t)step | (uintptr_t)stride) & 3)) return IMP_ERROR_INVALID_ARGS;
    hipLaunchKernelGGL(k_blend_paper, dim3((unsigned)(((long long)w * h + 255) / 256), (unsigned)count), dim3(256), 0, s,
                       d, stride, w, h, step);
    IMP_HIP(hipGetLastError());
    return IMP_OK;
}

// ------------------------------------------------------------------ CalcPerceivedBrightness, filters.c:707-729
// The reference adds one double term per pixel into a FLOAT accumulator, walking x-outer /
// y-inner: sum = (float)((double)sum + term).  The running float sum rounds at every step (by up
// to 16 once it passes 2^28 on a 1080p frame), so the result is not the true mean -- a flat
// 1080p frame of value 100 yields 0.3815, not 0.3922 -- and a tree reduction would differ from it.
// It is reproduced bit for bit without a serial walk.  While the sum stays inside one binade
// [2^e, 2^(e+1)) it is a multiple of u = 2^(e-23), and adding t = k*u + r moves it by k*u, plus u
// when r is above u/2, plus "round to even" when the double sum lands exactly on the midpoint --
// which, because (double)sum + t is itself rounded to 53 bits first, happens precisely when
// |r - u/2| <= 2^(e-53) (the midpoint is a double; rounding is monotone).  So inside a binade each
// term is a function of ONE bit of state, the parity of the sum's mantissa: parity -> (steps, parity').
// Such functions compose associatively, and they do not depend on where in the binade the sum stands.
//   k_brightness_fold (the whole GPU, ONE pass over the frame): a wave takes 1024 consecutive terms of the visiting order,
//     16 per lane -- the pixels come through LDS, read along the rows, so the column-major visiting order costs no
//     strided traffic, and the terms are never written out -- and folds them for every binade from 2^20 up that the
//     frame's sum can reach (at most 255 per pixel): one ParityFn per wave and binade.
//   k_brightness_walk (one block of four waves): carries the accumulator through the frame.  In a binade with summaries it
//     scans them 1024 at a time to the 1024-term chunk in which the sum leaves the binade and replays only that chunk,
//     a term per thread (recomputed from the pixels); below 2^20, where a few thousand terms live, it replays every
//     chunk.  The one term that crosses into the next binade is added with the literal float/double sequence.
// Round 2 had a 16.6 MB plane of double terms, a (summarize, replay) launch pair per binade -- fourteen launches -- and
// 0.48 ms at 1080p; this is two launches.
// A ParityFn is two words: for each parity of the incoming mantissa, (steps of u added) << 1 | parity out.  Steps saturate
// at 2^30 - 1 -- a binade has at most 2^23 of them, so a saturated count always means "left the binade", which is all that is
// ever asked of a count that large (its parity bit is then meaningless and never used).
struct ParityFn {
    uint32_t a0, a1;
};
#define PF_SAT 0x7fffffffu
__device__ __forceinline__ ParityFn pf_identity() { return ParityFn{0u, 1u}; }
__device__ __forceinline__ ParityFn pf_make(uint32_t inc0, int b0, uint32_t inc1, int b1) {
    return ParityFn{(min(inc0, PF_SAT >> 1) << 1) | (uint32_t)b0, (min(inc1, PF_SAT >> 1) << 1) | (uint32_t)b1};
}
__device__ __forceinline__ uint32_t pf_steps(const ParityFn& f, int parity) { return (parity ? f.a1 : f.a0) >> 1; }
__device__ __forceinline__ int pf_parity(const ParityFn& f, int parity) { return (int)((parity ? f.a1 : f.a0) & 1u); }
// `first` applied before `second`
__device__ __forceinline__ ParityFn pf_compose(const ParityFn& first, const ParityFn& second) {
    ParityFn r;
    r.a0 = min((first.a0 & ~1u) + ((first.a0 & 1u) ? second.a1 : second.a0), PF_SAT);      // both operands <= PF_SAT: no wrap
    r.a1 = min((first.a1 & ~1u) + ((first.a1 & 1u) ? second.a1 : second.a0), PF_SAT);
    return r;
}
// inclusive scan over the 64 lanes of a wave (lane l: f_0 .. f_l composed in order), all on the DPP network: shifts inside the
// rows of 16, then lane 15 / lane 31 broadcast into the rows above.  A lane without a source gets the identity.
template <int CTRL, int ROWS>
__device__ __forceinline__ ParityFn pf_dpp(const ParityFn& f) {
    ParityFn r;
    r.a0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f.a0, CTRL, ROWS, 0xf, false);
    r.a1 = (uint32_t)__builtin_amdgcn_update_dpp(1, (int)f.a1, CTRL, ROWS, 0xf, false);
    return r;
}
__device__ __forceinline__ ParityFn pf_wave_scan(ParityFn f) {
    f = pf_compose(pf_dpp<0x111, 0xf>(f), f);                   // row_shr:1
    f = pf_compose(pf_dpp<0x112, 0xf>(f), f);                   // row_shr:2
    f = pf_compose(pf_dpp<0x114, 0xf>(f), f);                   // row_shr:4
    f = pf_compose(pf_dpp<0x118, 0xf>(f), f);                   // row_shr:8
    f = pf_compose(pf_dpp<0x142, 0xa>(f), f);                   // row_bcast:15 into rows 1 and 3
    f = pf_compose(pf_dpp<0x143, 0xc>(f), f);                   // row_bcast:31 into rows 2 and 3
    return f;
}

#define BR_EPT 16                     // terms per lane
#define BR_WCHUNK (64 * BR_EPT)       // terms per wave = the granularity of the summaries
#define BR_FOLD_WAVES 8               // waves per block of k_brightness_fold
#define BR_E0 20                      // first binade with summaries (below it: a few thousand terms, replayed)

struct BrRegime {                     // everything a term's classification needs inside one binade
    double u, invu, eps, mid;
};
__device__ __forceinline__ double br_pow2(int k) { return __longlong_as_double((long long)(1023 + k) << 52); }   // 2^k, |k| < 1000
__device__ __forceinline__ BrRegime br_regime(int e) {
    return BrRegime{br_pow2(e - 23), br_pow2(23 - e), br_pow2(e - 53), br_pow2(e - 24)};
}
// one term as a function of the mantissa parity (see the block comment above)
__device__ __forceinline__ ParityFn br_classify(double t, const BrRegime& g) {
    const double kd = fmin(floor(t * g.invu), 1073741823.0);      // (a term far above the binade: saturated steps)
    const uint32_t k = (uint32_t)kd;
    const double diff = (t - kd * g.u) - g.mid;                   // exact: both products are exact, the differences are small
    if (fabs(diff) <= g.eps) {                                    // the double sum lands on the midpoint: ties-to-even
        const uint32_t kp = k & 1u;
        return pf_make(k + kp, 0, k + (kp ^ 1u), 0);
    }
    const uint32_t r = k + (diff > 0.0 ? 1u : 0u);
    return pf_make(r, (int)(r & 1u), r, (int)((r & 1u) ^ 1u));
}
// the reference's term of one pixel (filters.c:715-722): packed B | G << 8 | R << 16, or the gray value
template <int CN>
__device__ __forceinline__ double br_term_of(uint32_t px) {
    if (CN == 1) return (double)(px & 0xff);
    const int b = px & 0xff, g = (px >> 8) & 0xff, r = (px >> 16) & 0xff;
    return sqrt((double)(r * r) * 0.241 + (double)(g * g) * 0.691 + (double)(b * b) * 0.068);
}
template <int CN>
__device__ __forceinline__ uint32_t br_load_px(const uint8_t* p) {
    if (CN == 1) return p[0];
    if (CN == 4) return *(const uint32_t*)p;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}

template <int CN>
__global__ __launch_bounds__(64 * BR_FOLD_WAVES) void k_brightness_fold(const uint8_t* __restrict__ src, int w, int h, int step, long long n,
                                                                         int ne, long long nchunks, ParityFn* __restrict__ summ) {
    constexpr int BT = BR_WCHUNK * BR_FOLD_WAVES;                 // terms per block
    __shared__ uint32_t s_px[BT + BT / 16];                       // term j of the block at j + j / 16: a lane's 16 terms are contiguous, lanes 17 words apart
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long long i0 = (long long)blockIdx.x * BT;
    // the block's terms are the pixels of a few whole columns (visiting order: x outer, y inner): read them along the rows
    const int x0 = (int)(i0 / h);
    const long long ilast = min(n, i0 + BT) - 1;
    const int ncols = (int)(ilast / h) - x0 + 1;
    const int items = ncols * h;
    for (int k = t; k < items; k += 64 * BR_FOLD_WAVES) {
        const int y = k / ncols, xc = k - y * ncols;
        const long long i = (long long)(x0 + xc) * h + y;
        if (i >= i0 && i <= ilast) {
            const int j = (int)(i - i0);
            s_px[j + (j >> 4)] = br_load_px<CN>(src + (size_t)y * step + (size_t)(x0 + xc) * CN);
        }
    }
    __syncthreads();
    double td[BR_EPT];
    const long long mine = i0 + (long long)t * BR_EPT;
#pragma unroll
    for (int j = 0; j < BR_EPT; j++) td[j] = br_term_of<CN>(s_px[t * 17 + j]);
    const long long chunk = (long long)blockIdx.x * BR_FOLD_WAVES + wv;
    for (int e = 0; e < ne; e++) {
        const BrRegime g = br_regime(BR_E0 + e);
        ParityFn f = pf_identity();
#pragma unroll
        for (int j = 0; j < BR_EPT; j++)
            if (mine + j < n) f = pf_compose(f, br_classify(td[j], g));
        f = pf_wave_scan(f);
        if (lane == 63 && chunk < nchunks) summ[(long long)e * nchunks + chunk] = f;
    }
}

// inclusive scan of one ParityFn per thread over a block of BR_WALK_WAVES waves; s_part: that many entries of shared scratch
#define BR_WALK_WAVES 4
#define BR_HEAD 64                                                  // terms k_brightness_walk adds one by one before it starts scanning
#define BR_WALK_EPT (BR_WCHUNK / (64 * BR_WALK_WAVES))            // 4 consecutive terms (or chunk summaries) per thread
__device__ __forceinline__ ParityFn pf_block_scan(ParityFn f, ParityFn* s_part) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    f = pf_wave_scan(f);
    __syncthreads();                                              // (s_part may still be read from the previous scan)
    if (lane == 63) s_part[wv] = f;
    __syncthreads();
    ParityFn pre = pf_identity();
    for (int k = 0; k < wv; k++) pre = pf_compose(pre, s_part[k]);
    return pf_compose(pre, f);
}

// One block of four waves carries the accumulator through the frame (every thread holds the same sum and position).  A step
// handles 1024 entries, four consecutive ones per thread: the thread folds its four, the block scans 256 functions -- the scan
// is all shuffles, and four waves of them cost a quarter of what sixteen did.
template <int CN>
__global__ __launch_bounds__(64 * BR_WALK_WAVES) void k_brightness_walk(const uint8_t* __restrict__ src, int w, int h, int step, long long n, int ne,
                                                                         long long nchunks, const ParityFn* __restrict__ summ, float* __restrict__ out) {
    constexpr int NT = 64 * BR_WALK_WAVES, EPT = BR_WALK_EPT;
    __shared__ ParityFn s_part[BR_WALK_WAVES];
    __shared__ uint32_t s_tot[NT];
    __shared__ double s_term[BR_WCHUNK];
    __shared__ int s_first[BR_WALK_WAVES], s_par[NT], s_res_taken;
    __shared__ uint32_t s_res_steps;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float sum = 0.f;
    long long pos = 0;
    auto first_of = [&](bool mine) -> int {                       // lowest thread whose flag is set (NT = none); one barrier
        const unsigned long long bal = __ballot(mine);
        if (lane == 0) s_first[wv] = bal ? wv * 64 + (int)__builtin_ctzll(bal) : NT;
        __syncthreads();
        int first = NT;
#pragma unroll
        for (int k = 0; k < BR_WALK_WAVES; k++) first = min(first, s_first[k]);
        return first;
    };
    // The first terms are added with the literal sequence by every thread for itself: a dozen binades go by in the first few
    // dozen terms (the sum passes 2^13 within ~64 of them), and a block-wide pass per crossing would cost twenty times this.
    {
        const int head = (int)min(n, (long long)BR_HEAD);
        if (tid < head) {
            const int x = tid / h, y = tid - x * h;
            s_term[tid] = br_term_of<CN>(br_load_px<CN>(src + (size_t)y * step + (size_t)x * CN));
        }
        __syncthreads();
        for (int i = 0; i < head; i++) sum = (float)__dadd_rn((double)sum, s_term[i]);
        pos = head;
        __syncthreads();                                          // (s_term is the replay's chunk buffer next)
    }
    while (pos < n) {
        {
            const unsigned sbits = __float_as_uint(sum);
            const int e = (int)((sbits >> 23) & 0xff) - 127;
            if (sum != 0.f && e >= BR_E0 && e < BR_E0 + ne && (pos % BR_WCHUNK) == 0) {
                // whole chunks: scan their summaries for this binade, 1024 at a time, up to the chunk in which the sum leaves it
                const uint32_t limit = 0x800000u - (sbits & 0x7fffffu);             // steps of u until the binade ends
                const int parity = (int)(sbits & 1u);
                const long long c0 = pos / BR_WCHUNK + (long long)tid * EPT;
                ParityFn fk[EPT], f = pf_identity();
#pragma unroll
                for (int k = 0; k < EPT; k++) {
                    fk[k] = c0 + k < nchunks ? summ[(long long)(e - BR_E0) * nchunks + c0 + k] : pf_identity();
                    f = pf_compose(f, fk[k]);
                }
                const ParityFn incl = pf_block_scan(f, s_part);
                const uint32_t tot = pf_steps(incl, parity);
                s_tot[tid] = tot;
                s_par[tid] = pf_parity(incl, parity);             // the accumulator's parity behind this thread's chunks
                const int first = first_of(tot >= limit);         // the thread whose four chunks hold the leaving one
                if (tid == min(first, NT - 1)) {
                    // everything before that thread's chunks is consumed; of its own chunks, those before the leaving one
                    uint32_t steps = tid > 0 && first < NT ? s_tot[tid - 1] : (first < NT ? 0u : s_tot[NT - 1]);
                    int taken = first < NT ? tid * EPT : NT * EPT;
                    if (first < NT) {
                        int par = tid > 0 ? s_par[tid - 1] : parity;
#pragma unroll
                        for (int k = 0; k < EPT; k++) {
                            const uint32_t inc = pf_steps(fk[k], par);
                            if (steps + inc >= limit) break;
                            steps += inc;
                            par = pf_parity(fk[k], par);
                            taken++;
                        }
                    }
                    s_res_steps = steps;
                    s_res_taken = taken;
                }
                __syncthreads();
                if (s_res_taken > 0) {
                    sum = __uint_as_float(sbits + s_res_steps);
                    pos = min(n, pos + (long long)s_res_taken * BR_WCHUNK);
                }
                __syncthreads();
                if (first == NT || pos >= n) continue;
                // the chunk at pos leaves the binade: replay it below (the accumulator is still in binade e)
            }
        }
        // ---- term-by-term replay of the rest of the chunk pos stands in.  The chunk's 1024 terms are computed from the pixels
        // once (four consecutive ones per thread) and stay in LDS while the accumulator works its way through the chunk: every
        // pass below ends at the term that leaves the current binade (added with the literal float/double sequence) or at the
        // chunk's end, and only a new chunk costs loads.
        const long long cbase = pos / BR_WCHUNK * BR_WCHUNK;
        const long long nlim = min(n, cbase + BR_WCHUNK);
        double tk[EPT];
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const long long i = cbase + tid * EPT + k;
            double t = 0.0;
            if (i < nlim) {
                const int x = (int)((unsigned)i / (unsigned)h), y = (int)i - x * h;      // n <= 2^30 (launch_brightness)
                t = br_term_of<CN>(br_load_px<CN>(src + (size_t)y * step + (size_t)x * CN));
            }
            tk[k] = t;
            s_term[tid * EPT + k] = t;
        }
        while (pos < nlim) {
            const unsigned sb = __float_as_uint(sum);
            const bool zero = sum == 0.f;
            const int e2 = (int)((sb >> 23) & 0xff) - 127;
            const uint32_t lim2 = 0x800000u - (sb & 0x7fffffu);
            const int parity = (int)(sb & 1u);
            const int o = (int)(pos - cbase);                     // terms of the chunk already consumed
            const BrRegime g = br_regime(e2);
            ParityFn f = pf_identity();
            bool nonzero = false;
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                const int j = tid * EPT + k;
                if (j >= o && cbase + j < nlim) {
                    if (zero) nonzero |= tk[k] != 0.0;
                    else f = pf_compose(f, br_classify(tk[k], g));
                }
            }
            const ParityFn incl = pf_block_scan(f, s_part);
            const uint32_t tot = pf_steps(incl, parity);          // steps added by everything up to and including this thread
            s_tot[tid] = tot;
            const int first = first_of(zero ? nonzero : (tot >= lim2));
            if (first == NT) {                                    // everything left in the chunk stays inside the binade
                if (!zero) sum = __uint_as_float(sb + s_tot[NT - 1]);
                pos = nlim;
            } else {
                // that thread's terms hold the one that leaves the binade (or the first non-zero one): the literal sequence
                float ns = sum;
                if (!zero && first > 0) ns = __uint_as_float(sb + s_tot[first - 1]);
#pragma unroll
                for (int k = 0; k < EPT; k++) {
                    const int j = first * EPT + k;
                    if (j >= o && cbase + j < nlim) ns = (float)__dadd_rn((double)ns, s_term[j]);
                }
                sum = ns;
                pos = min(nlim, cbase + (long long)(first + 1) * EPT);
            }
            __syncthreads();
        }
    }
    if (tid == 0) *out = sum;
}

int launch_brightness(const View& v, float* host_result, hipStream_t s) {
    const long long n = (long long)v.w * v.h;
    if (!view_fits(v.w, v.h, v.c, v.step)) return IMP_ERROR_INVALID_ARGS;       // (n <= 2^30: the kernels divide in 32 bits)
    const long long nchunks = (n + BR_WCHUNK - 1) / BR_WCHUNK;
    // binades with summaries: from 2^20 up to the last one the sum can reach (a term is at most 255; u / 2 = 256 in binade
    // 32, so it never leaves that one)
    int last = 0;
    while (last < 32 && (double)n * 255.0 >= ldexp(1.0, last + 1)) last++;
    const int ne = last >= BR_E0 ? last - BR_E0 + 1 : 0;
    void *out = nullptr, *summ = nullptr;
    if (int rc = dev_alloc(sizeof(float), &out)) return rc;
    if (ne) if (int rc = dev_alloc((size_t)ne * (size_t)nchunks * sizeof(ParityFn), &summ)) { dev_free(out); return rc; }
    if (ne) {
        const dim3 grid((unsigned)((nchunks + BR_FOLD_WAVES - 1) / BR_FOLD_WAVES)), block(64 * BR_FOLD_WAVES);
        if (v.c == 1) hipLaunchKernelGGL((k_brightness_fold<1>), grid, block, 0, s, v.d, v.w, v.h, v.step, n, ne, nchunks, (ParityFn*)summ);
        else if (v.c == 3) hipLaunchKernelGGL((k_brightness_fold<3>), grid, block, 0, s, v.d, v.w, v.h, v.step, n, ne, nchunks, (ParityFn*)summ);
        else hipLaunchKernelGGL((k_brightness_fold<4>), grid, block, 0, s, v.d, v.w, v.h, v.step, n, ne, nchunks, (ParityFn*)summ);
    }
    if (v.c == 1) hipLaunchKernelGGL((k_brightness_walk<1>), dim3(1), dim3(64 * BR_WALK_WAVES), 0, s, v.d, v.w, v.h, v.step, n, ne, nchunks, (const ParityFn*)summ, (float*)out);
    else if (v.c == 3) hipLaunchKernelGGL((k_brightness_walk<3>), dim3(1), dim3(64 * BR_WALK_WAVES), 0, s, v.d, v.w, v.h, v.step, n, ne, nchunks, (const ParityFn*)summ, (float*)out);
    else hipLaunchKernelGGL((k_brightness_walk<4>), dim3(1), dim3(64 * BR_WALK_WAVES), 0, s, v.d, v.w, v.h, v.step, n, ne, nchunks, (const ParityFn*)summ, (float*)out);
    hipError_t e = hipGetLastError();
    uint32_t* box = lane_mailbox();
    if (e == hipSuccess && !box) e = hipErrorNotInitialized;
    if (e == hipSuccess) e = hipMemcpyAsync(box, out, sizeof(float), hipMemcpyDeviceToHost, s);
    dev_free(out);
    dev_free(summ);
    if (e != hipSuccess) { set_error("brightness", e); return IMP_ERROR_DEVICE; }
    if (int rc = lane_wait()) return rc;
    float sum;
    std::memcpy(&sum, box, sizeof sum);
    // filters.c:728: float / int -> float, then / 255.0 in double, returned as float
    const float mean = sum / (float)(v.w * v.h);
    *host_result = (float)((double)mean / 255.0);
    return IMP_OK;
}

// ------------------------------------------------------------------ ASCII, filters.c:486-522
template <int CN>
__global__ __launch_bounds__(256) void k_ascii(uint8_t* img, int w, int h, int step, const uint8_t* __restrict__ table,
                                               float factor, uint8_t* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)w * h) return;
    const int y = (int)(idx / w), x = (int)(idx - (long long)y * w);
    uint8_t* p = img + (size_t)y * step + (size_t)x * CN;
    int c0 = p[0], c1 = p[1], c2 = p[2];
    px_rgb2hsv(c0, c1, c2);                     // the reference converts the frame in place and leaves it so
    p[0] = (uint8_t)c0; p[1] = (uint8_t)c1; p[2] = (uint8_t)c2;
    const long long ro = (long long)y * (w + 1);
    out[ro + x] = table[(int)floorf(__fdiv_rn((float)c2, factor))];
    if (x == 0 && ro > 0) out[ro - 1] = '\n';
}

int launch_ascii(uint8_t* d, int w, int h, int c, int step, const uint8_t* table, int tablelen, float factor,
                 uint8_t* dev_out, hipStream_t s) {
    (void)tablelen;
    const dim3 grid((unsigned)(((long long)w * h + 255) / 256)), block(256);
    if (c == 4) hipLaunchKernelGGL((k_ascii<4>), grid, block, 0, s, d, w, h, step, table, factor, dev_out);
    else if (c == 3) hipLaunchKernelGGL((k_ascii<3>), grid, block, 0, s, d, w, h, step, table, factor, dev_out);
    else return IMP_ERROR_INVALID_ARGS;
    IMP_HIP(hipGetLastError());
    return IMP_OK;
}

}  // namespace imp
