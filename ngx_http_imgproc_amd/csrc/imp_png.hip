// imp_png.hip -- impgpu_image_decode_png: the reference's cvDecodeImage(&rawencoded, -1) (bridge.c:545-552) for a PNG
// blob (SIG_PNG, bridge.c:376-378), i.e. OpenCV 2.4's PngDecoder over libpng -- a third-party dependency that is not under
// /root/reference; what is restated here is the published format (PNG specification, 2nd edition: chunk layout section 5,
// filtering section 9, zlib stream section 10) and what OpenCV asks libpng for with the "unchanged" flag: 8-bit gray stays
// one channel, RGB becomes BGR, RGBA becomes BGRA, a tRNS chunk is not expanded.
//
// ROUND 4, A BOUNDED EXPERIMENT: the zlib stream is inflated ON THE HOST (imp_png.cpp / imp_inflate.cpp:
// there is no device inflate here) into the pinned staging buffer, the filtered scanlines cross the link as they are -- in
// slices of 128 rows, WHILE the inflate is still running -- and k_png_unfilter undoes the five scanline filters and swaps R/B on
// the device.  Unfiltering is a recurrence along x (Sub, Average, Paeth read the pixel to the left) and along y (Up, Average,
// Paeth read the row above), so a slice is ONE workgroup that walks it as a wavefront: lane l of a wave owns row 64 * band + l
// and works on group (S - l) of 4 pixels at macro step S, one group behind the lane above it, whose finished group arrives by a
// wave_shr DPP move.  The rows between bands (lane 63 of one wave -> lane 0 of the next) travel through LDS with a progress
// counter per band; the row above a slice's first row is read from the frame, where the slice before left it.
//
// Takes: bit depth 8, colour types 0 / 2 / 6, not interlaced, width <= 4096, height <= 16384.  Everything else is
// IMP_ERROR_UNSUPPORTED (decode with cvDecodeImage as before); a damaged file is IMP_ERROR_DECODE_FAILED.
#include <chrono>
#include <cstring>
#include "imp_internal.h"
#include "imp_png.h"

namespace imp {

constexpr int PNG_WAVES = 8;             // waves of the workgroup = bands in flight = edge rows held in LDS
// (PNG_MAX_W = 4096: 8 edge rows x ceil(w / 4) groups x 16 bytes = 128 KB of the CU's 160 KB; PNG_MAX_H = 16384: one progress
// word per band of 64 rows -- imp_png.h)
constexpr int PNG_RAW_SLACK = 64;        // the word stream of the last row reads a few bytes past its end

struct PngJob {
    const uint8_t* raw;                  // h rows of (1 filter byte + w * bpp bytes), as inflate delivered them
    uint8_t* dst;
    const uint8_t* prev;                 // the finished row above row 0 (a slice of an image: the slice before wrote it), or null
    int w, h, step;
};

// one pixel (BPP bytes in the low bytes of a word): Recon(x) = Filt(x) + predictor, PNG specification 9.2 - 9.4
template <int BPP>
__device__ __forceinline__ uint32_t png_recon(int type, uint32_t f, uint32_t a, uint32_t b, uint32_t c) {
    uint32_t out = 0;
#pragma unroll
    for (int ch = 0; ch < BPP; ch++) {
        const int fa = (a >> (8 * ch)) & 255, fb = (b >> (8 * ch)) & 255, fc = (c >> (8 * ch)) & 255;
        const int p = fa + fb - fc;
        const int pa = abs(p - fa), pb = abs(p - fb), pc = abs(p - fc);
        const int paeth = (pa <= pb && pa <= pc) ? fa : (pb <= pc ? fb : fc);          // ties: a, then b (9.4)
        const int pred = type == 0 ? 0 : type == 1 ? fa : type == 2 ? fb : type == 3 ? ((fa + fb) >> 1) : paeth;
        out |= ((((f >> (8 * ch)) & 255) + (uint32_t)pred) & 255u) << (8 * ch);
    }
    return out;
}

// The same for a pixel of three or four channels, two channels per instruction: the bytes are spread to 16-bit lanes
// (c0 | c1 << 16, c2 | c3 << 16) and every step of the predictors is a packed 16-bit operation (v_pk_sub_i16, v_pk_max_i16,
// v_pk_min_u16, v_pk_mad_u16 ...).  Paeth without compares: with m = min(pa, pb, pc), nea = min(pa - m, 1) is 0 exactly where
// a is the choice, neb = min(pb - m, 1) likewise for b, and pred = a + nea * ((b + neb * (c - b)) - a) -- ties go to a, then b,
// as 9.4 asks.  The filter type is a lane's constant for a whole row: four masks pick the predictor.
typedef short png_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short png_u2 __attribute__((ext_vector_type(2)));
struct PngTypeMasks { uint32_t sub, up, avg, paeth; };
__device__ __forceinline__ PngTypeMasks png_type_masks(int type) {
    return PngTypeMasks{type == 1 ? ~0u : 0u, type == 2 ? ~0u : 0u, type == 3 ? ~0u : 0u, type == 4 ? ~0u : 0u};
}
__device__ __forceinline__ uint32_t png_recon_pair(const PngTypeMasks& M, uint32_t f, uint32_t a, uint32_t b, uint32_t c) {
    const png_s2 A = __builtin_bit_cast(png_s2, a), B = __builtin_bit_cast(png_s2, b), C = __builtin_bit_cast(png_s2, c);
    const png_s2 d1 = B - C, d2 = A - C, d3 = d1 + d2;
    const png_s2 zero = {0, 0};
    const png_u2 pa = __builtin_bit_cast(png_u2, __builtin_elementwise_max(d1, zero - d1));
    const png_u2 pb = __builtin_bit_cast(png_u2, __builtin_elementwise_max(d2, zero - d2));
    const png_u2 pc = __builtin_bit_cast(png_u2, __builtin_elementwise_max(d3, zero - d3));
    const png_u2 m = __builtin_elementwise_min(pa, __builtin_elementwise_min(pb, pc));
    const png_u2 one = {1, 1};
    const png_u2 nea = __builtin_elementwise_min((png_u2)(pa - m), one), neb = __builtin_elementwise_min((png_u2)(pb - m), one);
    const png_u2 Au = __builtin_bit_cast(png_u2, a), Bu = __builtin_bit_cast(png_u2, b), Cu = __builtin_bit_cast(png_u2, c);
    const png_u2 u = Bu + neb * (png_u2)(Cu - Bu);
    const png_u2 paeth = Au + nea * (png_u2)(u - Au);
    const png_u2 avg = (png_u2)(Au + Bu) >> 1;
    const uint32_t pred = (a & M.sub) | (b & M.up) | (__builtin_bit_cast(uint32_t, avg) & M.avg) | (__builtin_bit_cast(uint32_t, paeth) & M.paeth);
    const png_u2 r = __builtin_bit_cast(png_u2, f) + __builtin_bit_cast(png_u2, pred);
    return __builtin_bit_cast(uint32_t, r) & 0x00ff00ffu;
}
__device__ __forceinline__ uint32_t png_spread_lo(uint32_t p) { return __builtin_amdgcn_perm(p, p, 0x0c010c00u); }   // c0 | c1 << 16
__device__ __forceinline__ uint32_t png_spread_hi(uint32_t p) { return __builtin_amdgcn_perm(p, p, 0x0c030c02u); }   // c2 | c3 << 16
__device__ __forceinline__ uint32_t png_gather(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x06040200u); }

__device__ __forceinline__ uint32_t png_swap_rb(uint32_t p) { return __builtin_amdgcn_perm(p, p, 0x03000102u); }

// the lane above (wave_shr:1); lane 0 keeps `edge`
__device__ __forceinline__ uint32_t png_from_above(uint32_t edge, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ int png_progress(int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// four finished pixels (B,G,R order already) of the row above a slice, as the lanes hand them on: one per word
template <int BPP>
__device__ __forceinline__ uint4 png_prev_group(const uint8_t* prev, int g) {
    const uint32_t* q = (const uint32_t*)(prev + (size_t)g * 4 * BPP);      // (rows are 4-byte aligned; 12 / 16 / 4 bytes per group)
    if constexpr (BPP == 4) return make_uint4(q[0], q[1], q[2], q[3]);
    else if constexpr (BPP == 3) {
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
        return make_uint4(d0 & 0xffffffu, __builtin_amdgcn_alignbit(d1, d0, 24) & 0xffffffu, __builtin_amdgcn_alignbit(d2, d1, 16) & 0xffffffu, d2 >> 8);
    } else {
        const uint32_t d0 = q[0];
        return make_uint4(d0 & 255u, (d0 >> 8) & 255u, (d0 >> 16) & 255u, d0 >> 24);
    }
}

template <int BPP>
__global__ __launch_bounds__(PNG_WAVES * 64) void k_png_unfilter(const PngJob J) {
    extern __shared__ uint4 s_png[];
    const int G = (J.w + 3) >> 2;                                    // groups of 4 pixels per row
    const int nbands = (J.h + 63) >> 6;
    uint4* s_edge = s_png;                                           // [PNG_WAVES][G]: the last row of a band, unpacked
    int* s_prog = (int*)(s_png + (size_t)PNG_WAVES * G);             // [nbands]: groups of that row which are in s_edge
    for (int i = threadIdx.x; i < nbands; i += blockDim.x) s_prog[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t rstride = (size_t)J.w * BPP + 1;
    constexpr int NW = BPP;                                          // words per group (4 pixels x BPP bytes)

    for (int band = wave; band < nbands; band += PNG_WAVES) {
        const int row = band * 64 + lane;
        const bool rowok = row < J.h;
        const uint8_t* rp = J.raw + (size_t)(rowok ? row : J.h - 1) * rstride;
        const int type = rowok ? (int)rp[0] : 0;
        const PngTypeMasks tm = png_type_masks(type);
        const uintptr_t a0 = (uintptr_t)(rp + 1);
        const uint32_t sh = (uint32_t)(a0 & 3) * 8;
        const uint32_t* wp = (const uint32_t*)(a0 & ~(uintptr_t)3);
        uint4* edge_out = s_edge + (size_t)(band % PNG_WAVES) * G;
        const uint4* edge_in = s_edge + (size_t)((band + PNG_WAVES - 1) % PNG_WAVES) * G;
        uint8_t* drow = J.dst + (size_t)(rowok ? row : 0) * J.step;
        // the slot this band writes its last row into was read by band - PNG_WAVES + 1: that band must be through
        if (band >= PNG_WAVES)
            while (png_progress(&s_prog[band - PNG_WAVES + 1]) < G) __builtin_amdgcn_s_sleep(8);
        int known = 0;                                               // groups of the row above this band known to be in LDS
        uint32_t left = 0, upleft = 0;
        uint32_t cur[4] = {0, 0, 0, 0};
        uint32_t nxt[NW + 1];
        {
            const int g0 = 0;                                        // every lane starts (or idles) on group 0
#pragma unroll
            for (int i = 0; i <= NW; i++) nxt[i] = wp[g0 * NW + i];
        }
        uint4 pe = make_uint4(0, 0, 0, 0);                           // band 0 of a slice: lane 0's next group of the row above
        if (band == 0 && J.prev != nullptr && lane == 0) pe = png_prev_group<BPP>(J.prev, 0);
        const int steps = G + 63;
        for (int S = 0; S < steps; S++) {
            const int g = S - lane;
            const bool act = rowok && g >= 0 && g < G;
            // the words of this lane's group, and the next group's on their way
            uint32_t wv[NW + 1];
#pragma unroll
            for (int i = 0; i <= NW; i++) wv[i] = nxt[i];
            {
                int gn = g + 1;
                gn = gn < 0 ? 0 : (gn > G - 1 ? G - 1 : gn);
#pragma unroll
                for (int i = 0; i <= NW; i++) nxt[i] = wp[gn * NW + i];
            }
            uint32_t d[NW];
#pragma unroll
            for (int i = 0; i < NW; i++) d[i] = __builtin_amdgcn_alignbit(wv[i + 1], wv[i], sh);
            uint32_t px[4];
            if constexpr (BPP == 4) {
#pragma unroll
                for (int i = 0; i < 4; i++) px[i] = png_swap_rb(d[i]);
            } else if constexpr (BPP == 3) {
                px[0] = d[0] & 0xffffffu;
                px[1] = __builtin_amdgcn_alignbit(d[1], d[0], 24) & 0xffffffu;
                px[2] = __builtin_amdgcn_alignbit(d[2], d[1], 16) & 0xffffffu;
                px[3] = d[2] >> 8;
#pragma unroll
                for (int i = 0; i < 4; i++) px[i] = png_swap_rb(px[i]) & 0xffffffu;
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) px[i] = (d[0] >> (8 * i)) & 255u;
            }
            // the row above: the previous band's last row for lane 0 (LDS; for the first band of a slice the row the slice before
            // left in the image, fetched one step ahead), the lane above for everyone else
            uint4 e = make_uint4(0, 0, 0, 0);
            if (band == 0 && J.prev != nullptr) {
                e = pe;
                if (lane == 0) pe = png_prev_group<BPP>(J.prev, S + 1 < G ? S + 1 : G - 1);
            }
            if (band > 0 && S < G) {
                while (known <= S) {
                    known = png_progress(&s_prog[band - 1]);
                    if (known <= S) __builtin_amdgcn_s_sleep(2);
                }
                if (lane == 0) e = edge_in[S];
            }
            uint32_t up[4];
            up[0] = png_from_above(e.x, cur[0]);
            up[1] = png_from_above(e.y, cur[1]);
            up[2] = png_from_above(e.z, cur[2]);
            up[3] = png_from_above(e.w, cur[3]);
            if (g == 0) { left = 0; upleft = 0; }
            if constexpr (BPP >= 3) {
                // channels in pairs: the pixel to the left stays in its spread form from one pixel to the next
                uint32_t a_lo = png_spread_lo(left), a_hi = png_spread_hi(left);
                uint32_t c_lo = png_spread_lo(upleft), c_hi = png_spread_hi(upleft);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t b_lo = png_spread_lo(up[i]), b_hi = png_spread_hi(up[i]);
                    a_lo = png_recon_pair(tm, png_spread_lo(px[i]), a_lo, b_lo, c_lo);
                    a_hi = png_recon_pair(tm, png_spread_hi(px[i]), a_hi, b_hi, c_hi);
                    cur[i] = png_gather(a_lo, a_hi);
                    c_lo = b_lo; c_hi = b_hi;
                }
            } else {
                cur[0] = png_recon<BPP>(type, px[0], left, up[0], upleft);
                cur[1] = png_recon<BPP>(type, px[1], cur[0], up[1], up[0]);
                cur[2] = png_recon<BPP>(type, px[2], cur[1], up[2], up[1]);
                cur[3] = png_recon<BPP>(type, px[3], cur[2], up[3], up[2]);
            }
            left = cur[3];
            upleft = up[3];
            if (act) {
                const int x0 = g * 4, n = J.w - x0 >= 4 ? 4 : J.w - x0;
                if (n == 4) {
                    if constexpr (BPP == 4) {
                        uint32_t* o = (uint32_t*)(drow + (size_t)x0 * 4);          // (rows are 4-byte, not 16-byte, aligned)
                        o[0] = cur[0]; o[1] = cur[1]; o[2] = cur[2]; o[3] = cur[3];
                    } else if constexpr (BPP == 3) {
                        uint32_t* o = (uint32_t*)(drow + (size_t)x0 * 3);
                        o[0] = cur[0] | (cur[1] << 24);
                        o[1] = (cur[1] >> 8) | (cur[2] << 16);
                        o[2] = (cur[2] >> 16) | (cur[3] << 8);
                    } else {
                        *(uint32_t*)(drow + x0) = cur[0] | (cur[1] << 8) | (cur[2] << 16) | (cur[3] << 24);
                    }
                } else {
                    for (int i = 0; i < n; i++)
                        for (int ch = 0; ch < BPP; ch++) drow[(size_t)(x0 + i) * BPP + ch] = (uint8_t)(cur[i] >> (8 * ch));
                }
                if (lane == 63) {
                    edge_out[g] = make_uint4(cur[0], cur[1], cur[2], cur[3]);
                    __hip_atomic_store(&s_prog[band], g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
}

static size_t png_lds_bytes(int w, int h) { return (size_t)PNG_WAVES * ((w + 3) / 4) * 16 + (size_t)((h + 63) / 64) * 4; }

static int launch_png_unfilter(const PngJob& job, int bpp, hipStream_t s) {
    const size_t lds = png_lds_bytes(job.w, job.h);
    hipError_t e = hipSuccess;
#define PNG_LAUNCH(BPP_)                                                                                                      \
    do {                                                                                                                      \
        e = lds_limit_once<k_png_unfilter<BPP_>>();  /* (once per process; the attribute is per function) */                 \
        if (e == hipSuccess) {                                                                                                \
            hipLaunchKernelGGL(k_png_unfilter<BPP_>, dim3(1), dim3(PNG_WAVES * 64), lds, s, job);                             \
            e = hipGetLastError();                                                                                            \
        }                                                                                                                     \
    } while (0)
    if (bpp == 4) PNG_LAUNCH(4);
    else if (bpp == 3) PNG_LAUNCH(3);
    else PNG_LAUNCH(1);
#undef PNG_LAUNCH
    if (e != hipSuccess) { set_error("k_png_unfilter", e); return IMP_ERROR_DEVICE; }
    return IMP_OK;
}

static thread_local double t_png_us[4] = {0, 0, 0, 0};

static double png_now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace imp

using namespace imp;

extern "C" {

int impgpu_png_stage_times(double* microseconds, int n) {
    if (!microseconds || n < 0) return IMP_ERROR_INVALID_ARGS;
    for (int i = 0; i < n; i++) microseconds[i] = i < 4 ? t_png_us[i] : 0.0;
    return IMP_OK;
}

int impgpu_image_decode_png(const unsigned char* blob, size_t size, impgpu_image** out) {
    if (!out) return IMP_ERROR_INVALID_ARGS;
    *out = nullptr;
    if (!env_ready()) { set_error_text("impgpu_env_start has not been called"); return IMP_ERROR_DEVICE; }
    PngHeader H;
    const double t0 = png_now_us();
    int rc = png_header(blob, size, &H);
    if (rc) return rc;
    if (!H.taken) return IMP_ERROR_UNSUPPORTED;
    TraceRange tr("IMP_STEP_DECODE");
    IMP_FAULT_POINT(IMP_STEP_DECODE);
    const size_t rstride = (size_t)H.w * H.bpp + 1, raw_bytes = rstride * H.h;
    // the scanlines can be no larger than zlib's best ratio lets the file hold (1032 : 1, zlib technical details):
    // a header that promises more is refused before anything is staged for it
    if (raw_bytes / 1032 > size) return IMP_ERROR_DECODE_FAILED;
    // Pinned memory for the scanlines, device memory for them and for the frame -- all before the inflate starts, because the
    // rows leave for the device WHILE it runs: every time another SLICE_ROWS rows are final (the inflate reports at the ends
    // of deflate blocks) they are uploaded and a k_png_unfilter launch takes them, its first row reading the row above from the
    // frame the slice before wrote.  The device's share of a decode (a quarter of it) then hides behind the host's.
    void *host = nullptr, *token = nullptr;
    rc = stage_begin(raw_bytes + PNG_RAW_SLACK, &host, &token);
    if (rc) return rc;
    impgpu_image* im = nullptr;
    rc = image_new(H.w, H.h, H.bpp, &im);
    if (rc) { (void)stage_upload(token, nullptr, 0); return rc; }
    void* dev = nullptr;
    rc = dev_alloc(raw_bytes + PNG_RAW_SLACK, &dev);
    if (rc) { (void)stage_upload(token, nullptr, 0); image_delete(im); return rc; }
    struct Feed {
        void* token; uint8_t* dev; impgpu_image* im; PngHeader H; size_t rstride; int sent; hipStream_t s;
        int upto(int rows, bool last) {                              // rows [sent, rows) -> device, unfiltered there
            if (rows <= sent) return last ? stage_upload_part(token, 0, nullptr, 0, true) : IMP_OK;
            const size_t off = (size_t)sent * rstride, bytes = (size_t)(rows - sent) * rstride + (last ? PNG_RAW_SLACK : 0);
            if (int rc = stage_upload_part(token, off, dev + off, bytes, last)) return rc;
            PngJob job{dev + off, im->d + (size_t)sent * im->step, sent ? im->d + (size_t)(sent - 1) * im->step : nullptr, H.w, rows - sent, im->step};
            if (int rc = launch_png_unfilter(job, H.bpp, s)) return rc;
            sent = rows;
            return IMP_OK;
        }
    } feed{token, (uint8_t*)dev, im, H, rstride, 0, env_stream()};
    constexpr int SLICE_ROWS = 128;                                  // two bands of the kernel; a launch costs ~10 us of the host's time
    std::memset((unsigned char*)host + raw_bytes, 0, PNG_RAW_SLACK);
    const double t1 = png_now_us();
    rc = png_scanlines(blob, size, H, (unsigned char*)host,
                       [](void* p, int complete) -> int {
                           Feed* f = (Feed*)p;
                           const int ready = complete & ~63;         // whole bands only: a slice starts on a band's first row
                           return ready - f->sent >= SLICE_ROWS ? f->upto(ready, false) : IMP_OK;
                       }, &feed);
    const double t2 = png_now_us();
    if (!rc) rc = feed.upto(H.h, true);
    else (void)stage_upload_part(token, 0, nullptr, 0, true);        // (pieces may be on their way: fence the buffer all the same)
    dev_free(dev);                                                   // (handed out again in lane-stream order)
    if (rc) { image_delete(im); return rc; }
    const double t3 = png_now_us();
    t_png_us[0] = t1 - t0; t_png_us[1] = t2 - t1; t_png_us[2] = t3 - t2; t_png_us[3] = (double)raw_bytes;
    *out = im;
    return IMP_OK;
}

}  // extern "C"
