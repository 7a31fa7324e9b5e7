// imp_blur.hip -- cvSmooth(image, image, CV_GAUSSIAN, 0, 0, sigma, 0) (reference filters.c:204).
//
// OpenCV 2.4.9 semantics for CV_8U (smooth.cpp / filter.cpp, x86-64 build): kernel size
// cvRound(6*sigma + 1) | 1, float Gaussian normalised in double, converted to 8-bit fixed point
// (x256) per axis; row pass in int32; column pass SymmColumnVec_32s8u = float accumulation of
// (row[+k] + row[-k]) * (ky[k] / 65536) with round-half-even for the first (w*cn & ~3)
// elements of a row, (sum + 2^15) >> 16 for the rest; BORDER_REPLICATE; in place.
//
// Two streaming kernels with an int32 intermediate in HBM.  Taps are wave-uniform (scalar
// loads); BGRA pixels move as dwords / 16-byte int4 rows so both passes are coalesced.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>
#include "imp_internal.h"

namespace imp {

__device__ __forceinline__ int clampb(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }
__device__ __forceinline__ int sat8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// row pass: int32 sum of u8 * kx[k]; BGRA: one thread per pixel
__global__ __launch_bounds__(256) void k_blur_row4(const uint8_t* __restrict__ src, long long stride, int w, int h, int step,
                                                   const int* __restrict__ kx, int ksize, int4* __restrict__ tmp) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)w * h) return;
    const int y = (int)(idx / w), x = (int)(idx - (long long)y * w);
    const uint32_t* row = (const uint32_t*)(src + (long long)blockIdx.y * stride + (size_t)y * step);
    const int r = ksize >> 1;
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int k = 0; k < ksize; k++) {
        const uint32_t p = row[clampb(x + k - r, w - 1)];
        const int c = kx[k];
        a0 += (int)(p & 0xff) * c; a1 += (int)((p >> 8) & 0xff) * c;
        a2 += (int)((p >> 16) & 0xff) * c; a3 += (int)(p >> 24) * c;
    }
    tmp[(long long)blockIdx.y * w * h + idx] = make_int4(a0, a1, a2, a3);
}

__global__ __launch_bounds__(256) void k_blur_col4(const int4* __restrict__ tmp, int w, int h,
                                                   const float* __restrict__ kyf, int ry,
                                                   uint8_t* __restrict__ dst, long long stride, int step) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)w * h) return;
    const int y = (int)(idx / w), x = (int)(idx - (long long)y * w);
    const int4* t = tmp + (long long)blockIdx.y * w * h;
    const int4 c = t[idx];
    const float f0 = kyf[0];
    float s0 = __fadd_rn(__fmul_rn((float)c.x, f0), 0.f), s1 = __fadd_rn(__fmul_rn((float)c.y, f0), 0.f);
    float s2 = __fadd_rn(__fmul_rn((float)c.z, f0), 0.f), s3 = __fadd_rn(__fmul_rn((float)c.w, f0), 0.f);
    for (int k = 1; k <= ry; k++) {
        const int4 a = t[(long long)clampb(y + k, h - 1) * w + x];
        const int4 b = t[(long long)clampb(y - k, h - 1) * w + x];
        const float f = kyf[k];
        s0 = __fadd_rn(s0, __fmul_rn((float)(a.x + b.x), f));
        s1 = __fadd_rn(s1, __fmul_rn((float)(a.y + b.y), f));
        s2 = __fadd_rn(s2, __fmul_rn((float)(a.z + b.z), f));
        s3 = __fadd_rn(s3, __fmul_rn((float)(a.w + b.w), f));
    }
    const uint32_t o = (uint32_t)sat8(__float2int_rn(s0)) | ((uint32_t)sat8(__float2int_rn(s1)) << 8) |
                       ((uint32_t)sat8(__float2int_rn(s2)) << 16) | ((uint32_t)sat8(__float2int_rn(s3)) << 24);
    *(uint32_t*)(dst + (long long)blockIdx.y * stride + (size_t)y * step + (size_t)x * 4) = o;
}

// 1- and 3-channel frames: one thread per row element e = x*cn + c
__global__ __launch_bounds__(256) void k_blur_row_any(const uint8_t* __restrict__ src, long long stride, int w, int h, int cn,
                                                      int step, const int* __restrict__ kx, int ksize, int* __restrict__ tmp) {
    const int roww = w * cn;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)roww * h) return;
    const int y = (int)(idx / roww), e = (int)(idx - (long long)y * roww);
    const int x = e / cn, c = e - x * cn;
    const uint8_t* row = src + (long long)blockIdx.y * stride + (size_t)y * step;
    const int r = ksize >> 1;
    int acc = 0;
    for (int k = 0; k < ksize; k++) acc += (int)row[clampb(x + k - r, w - 1) * cn + c] * kx[k];
    tmp[(long long)blockIdx.y * roww * h + idx] = acc;
}

__global__ __launch_bounds__(256) void k_blur_col_any(const int* __restrict__ tmp, int w, int h, int cn,
                                                      const float* __restrict__ kyf, const int* __restrict__ kyi, int ry,
                                                      uint8_t* __restrict__ dst, long long stride, int step) {
    const int roww = w * cn;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)roww * h) return;
    const int y = (int)(idx / roww), e = (int)(idx - (long long)y * roww);
    const int* t = tmp + (long long)blockIdx.y * roww * h;
    int out;
    if (e < (roww & ~3)) {
        float s = __fadd_rn(__fmul_rn((float)t[idx], kyf[0]), 0.f);
        for (int k = 1; k <= ry; k++) {
            const int v = t[(long long)clampb(y + k, h - 1) * roww + e] + t[(long long)clampb(y - k, h - 1) * roww + e];
            s = __fadd_rn(s, __fmul_rn((float)v, kyf[k]));
        }
        out = sat8(__float2int_rn(s));
    } else {
        int s = kyi[0] * t[idx];
        for (int k = 1; k <= ry; k++)
            s += kyi[k] * (t[(long long)clampb(y + k, h - 1) * roww + e] + t[(long long)clampb(y - k, h - 1) * roww + e]);
        int t = (s + (1 << 15)) >> 16;
        asm volatile("" : "+v"(t));     // keep shift and clamp apart (v_ashr_pk_u8_i32 hazard, see imp_resize.hip)
        out = sat8(t);
    }
    dst[(long long)blockIdx.y * stride + (size_t)y * step + e] = (uint8_t)out;
}

// ------------------------------------------------------------------ fused BGRA / BGR Gaussian (radius <= 16)
// One kernel, one read and one write of the frame: a 256-thread block owns a 64 x TH tile of the
// output.  (0) its (64+2r) x (TH+2r) source footprint goes to LDS (edge-replicated, coalesced dword
// loads, 12 in flight per lane); (1) the row pass runs once per (footprint row, column) out of
// LDS -- lane i reads dwords i..i+2r of its row, conflict free -- two taps per v_dot2_i32_i16 after a
// v_perm_b32 pairs the channel bytes, and parks the four sums as u16 in an LDS plane (they fit: the
// host checks 255 * sum(kx) <= 65535); (2) the column pass is SymmColumnVec_32s8u's float sequence
// on (row[+k] + row[-k]) read 8 bytes per lane from the plane, then a coalesced dword store.  Runs
// out of place (a tile's halo belongs to its neighbours), which also materialises a cropped view.
// CN = 3 (every JPEG): pixels are assembled from / scattered to three bytes, channel 3 is skipped, and the last
// (3 w) % 4 elements of a row take SymmColumnVec's scalar tail, (sum + 2^15) >> 16 on integers, like the CPU.
typedef short bl_short2 __attribute__((ext_vector_type(2)));

template <int TH, int CN>
__global__ __launch_bounds__(256) void k_blur_fused4(const uint8_t* __restrict__ src, long long sstride, int sstep, int w, int h,
                                                     uint8_t* __restrict__ dst, long long dstride, int dstep,
                                                     const int* __restrict__ kxp, const float* __restrict__ kyf,
                                                     const int* __restrict__ kyi, int rx, int ry) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int SW = 64 + 2 * rx + 2, SH = TH + 2 * ry;         // +2: the padded last tap pair reads one dword further
    const int npair = rx + 1;                                 // 2*rx+1 taps -> rx+1 pairs, last one (tap, 0)
    uint32_t* s_kx = (uint32_t*)smem;                         // packed (k[2j], k[2j+1]) as 2 x i16
    float* s_ky = (float*)(s_kx + ((npair + 3) & ~3));
    int* s_kyi = (int*)(s_ky + ((ry + 1 + 3) & ~3));
    uint32_t* s_src = (uint32_t*)(s_kyi + ((ry + 1 + 3) & ~3));
    uint2* s_pl = (uint2*)(s_src + ((SW * SH + 3) & ~3));     // [SH][64] x 4 x u16
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tx0 = blockIdx.x * 64, ty0 = blockIdx.y * TH;
    const uint8_t* S = src + (long long)blockIdx.z * sstride;
    for (int i = tid; i < npair; i += 256) s_kx[i] = (uint32_t)kxp[i];
    for (int i = tid; i <= ry; i += 256) { s_ky[i] = kyf[i]; s_kyi[i] = kyi[i]; }

    // (0) footprint -> LDS
    {
        int sxc[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int sx = tx0 - rx + lane + 64 * q;
            sxc[q] = (sx < 0 ? 0 : (sx > w - 1 ? w - 1 : sx)) * CN;
        }
        for (int r0 = wv; r0 < SH; r0 += 16) {
            uint32_t v[4][3];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = r0 + 4 * u;
                const int sy = ty0 - ry + r;
                const uint8_t* row = S + (size_t)(sy < 0 ? 0 : (sy > h - 1 ? h - 1 : sy)) * sstep;
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    uint32_t px = 0u;
                    if (r < SH && lane + 64 * q < SW) {
                        if (CN == 4) px = *(const uint32_t*)(row + sxc[q]);
                        else px = (uint32_t)row[sxc[q]] | ((uint32_t)row[sxc[q] + 1] << 8) | ((uint32_t)row[sxc[q] + 2] << 16);
                    }
                    v[u][q] = px;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = r0 + 4 * u;
#pragma unroll
                for (int q = 0; q < 3; q++)
                    if (r < SH && lane + 64 * q < SW) s_src[r * SW + lane + 64 * q] = v[u][q];
            }
        }
    }
    __syncthreads();

    // (1) row pass -> u16 plane
    for (int r = wv; r < SH; r += 4) {
        const uint32_t* row = s_src + r * SW + lane;
        int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int j = 0; j < npair; j++) {
            const uint32_t p0 = row[2 * j], p1 = row[2 * j + 1];
            bl_short2 kk;
            const uint32_t kw = s_kx[j];
            __builtin_memcpy(&kk, &kw, 4);
            bl_short2 c;
            uint32_t pr;
            pr = __builtin_amdgcn_perm(p1, p0, 0x0c040c00u); __builtin_memcpy(&c, &pr, 4); a0 = __builtin_amdgcn_sdot2(c, kk, a0, false);
            pr = __builtin_amdgcn_perm(p1, p0, 0x0c050c01u); __builtin_memcpy(&c, &pr, 4); a1 = __builtin_amdgcn_sdot2(c, kk, a1, false);
            pr = __builtin_amdgcn_perm(p1, p0, 0x0c060c02u); __builtin_memcpy(&c, &pr, 4); a2 = __builtin_amdgcn_sdot2(c, kk, a2, false);
            if (CN == 4) { pr = __builtin_amdgcn_perm(p1, p0, 0x0c070c03u); __builtin_memcpy(&c, &pr, 4); a3 = __builtin_amdgcn_sdot2(c, kk, a3, false); }
        }
        s_pl[r * 64 + lane] = make_uint2((uint32_t)a0 | ((uint32_t)a1 << 16), (uint32_t)a2 | ((uint32_t)a3 << 16));
    }
    __syncthreads();

    // (2) column pass
    const int x = tx0 + lane;
    if (x < w) {
        for (int yl = wv; yl < TH; yl += 4) {
            const int y = ty0 + yl;
            if (y >= h) break;
            const uint2* col = s_pl + (yl + ry) * 64 + lane;
            const uint2 c = col[0];
            const float f0 = s_ky[0];
            float s0 = __fadd_rn(__fmul_rn((float)(c.x & 0xffff), f0), 0.f), s1 = __fadd_rn(__fmul_rn((float)(c.x >> 16), f0), 0.f);
            float s2 = __fadd_rn(__fmul_rn((float)(c.y & 0xffff), f0), 0.f), s3 = __fadd_rn(__fmul_rn((float)(c.y >> 16), f0), 0.f);
            for (int k = 1; k <= ry; k++) {
                const uint2 a = col[k * 64], b = col[-k * 64];
                const float f = s_ky[k];
                s0 = __fadd_rn(s0, __fmul_rn((float)((a.x & 0xffff) + (b.x & 0xffff)), f));
                s1 = __fadd_rn(s1, __fmul_rn((float)((a.x >> 16) + (b.x >> 16)), f));
                s2 = __fadd_rn(s2, __fmul_rn((float)((a.y & 0xffff) + (b.y & 0xffff)), f));
                s3 = __fadd_rn(s3, __fmul_rn((float)((a.y >> 16) + (b.y >> 16)), f));
            }
            if (CN == 4) {
                const uint32_t o = (uint32_t)sat8(__float2int_rn(s0)) | ((uint32_t)sat8(__float2int_rn(s1)) << 8) |
                                   ((uint32_t)sat8(__float2int_rn(s2)) << 16) | ((uint32_t)sat8(__float2int_rn(s3)) << 24);
                *(uint32_t*)(dst + (long long)blockIdx.z * dstride + (size_t)y * dstep + (size_t)x * 4) = o;
            } else {
                int o[3] = {sat8(__float2int_rn(s0)), sat8(__float2int_rn(s1)), sat8(__float2int_rn(s2))};
                const int vec_end = (w * 3) & ~3;                   // SymmColumnVec_32s8u covers whole groups of 4 elements
                if (x * 3 + 2 >= vec_end) {                         // this pixel holds tail elements: integer form for those
                    int t0 = s_kyi[0] * (int)(c.x & 0xffff), t1 = s_kyi[0] * (int)(c.x >> 16), t2 = s_kyi[0] * (int)(c.y & 0xffff);
                    for (int k = 1; k <= ry; k++) {
                        const uint2 a = col[k * 64], b = col[-k * 64];
                        const int fk = s_kyi[k];
                        t0 += fk * (int)((a.x & 0xffff) + (b.x & 0xffff));
                        t1 += fk * (int)((a.x >> 16) + (b.x >> 16));
                        t2 += fk * (int)((a.y & 0xffff) + (b.y & 0xffff));
                    }
                    int ti[3] = {(t0 + (1 << 15)) >> 16, (t1 + (1 << 15)) >> 16, (t2 + (1 << 15)) >> 16};
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        asm volatile("" : "+v"(ti[ch]));            // keep shift and clamp apart (v_ashr_pk_u8_i32 hazard)
                        if (x * 3 + ch >= vec_end) o[ch] = sat8(ti[ch]);
                    }
                }
                uint8_t* q = dst + (long long)blockIdx.z * dstride + (size_t)y * dstep + (size_t)x * 3;
                q[0] = (uint8_t)o[0]; q[1] = (uint8_t)o[1]; q[2] = (uint8_t)o[2];
            }
        }
    }
}

// ------------------------------------------------------------------ large radii: column strips with an LDS ring
// radius 17 .. 60 after trimming zero taps (sigma up to ~25; `docs/03 - Usage.md:224`: "execution time is proportional to sigma").  The two-pass
// fallback parks an int32 plane in HBM and re-reads it 2r+1 times through L2 (sigma = 8: 147 us per 1080p frame, 1.4 % of
// the roofline).  Here the row sums never leave the CU: a 256-thread block owns 64 columns and walks down a strip of rows,
// four source rows per step (one per wave):
//   (a) a wave copies its source row's segment (64 + 2r pixels, edge-replicated) into LDS -- the pixels were requested
//       one step earlier, and the previous step's output pixel is stored right after that wait and BEFORE the next
//       request, so the wait (loads and stores share vmcnt) never covers a store younger than one whole step;
//   (b) row pass out of that segment: lane i reads dwords i .. i+2r, two taps per v_dot2_i32_i16 after v_perm_b32 pairs
//       the channel bytes (exact integers), and parks the four sums AS FLOATS (exact: < 2^16) in ring row (y mod RH);
//   (c) one barrier; then each wave runs SymmColumnVec_32s8u's float sequence for the output row whose last tap row
//       was just written: s = f0*c; s += f_k * (row[+k] + row[-k]) for k = 1..r -- the integer add of the SSE2 code is
//       exact in float too (< 2^17) -- channel pairs in packed FP32, then round-half-even + saturate (v_cvt_pk_u8_f32).
// The ring holds RH >= 2r + 8 rows (a power of two), so the rows a fast wave writes in the next step are never rows a
// slower wave still reads in this one: one barrier per step is enough.
typedef float bl_float2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t bl_cvt_pk_u8(float x, uint32_t acc, int byte) {   // saturate_u8(round-half-even(x)) into one byte
    uint32_t r;
    asm("v_cvt_pk_u8_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(byte), "v"(acc));
    return r;
}

// CN = 3 (every JPEG): pixels are assembled from / scattered to three bytes, channel 3 is not computed, and the last
// (3 w) % 4 elements of a row take SymmColumnVec's scalar tail, (sum + 2^15) >> 16 on integers, like the CPU.
// NO = output rows per wave and step (a step = 4 * NO source rows).  With NO = 2 a wave's two outputs are neighbours, so
// the ring rows y+k and y+1-k of one tap step are the rows y+1+(k-1) and y-(k-1) of the step before: two ring reads per
// tap instead of four, half the barriers per row -- the column pass was LDS-read bound (PMC: VALU 40 % busy, 49 x 16-byte
// ring reads per output pixel).  Ring rows needed: RH >= 2r + 8 * NO.  (Measured and dropped: the row pass on channel
// planes with v_dot4_u32_u8 -- half the VALU work, but byte-wise plane writes and per-plane window reads double the LDS
// instructions of a kernel that is LDS-bound: sigma = 8 went from 57 back to 67 us.)
// R16: the ring holds the row sums as u16 (8 bytes per column and row) instead of floats (16): a 128-row ring is then 64 KB
// and two workgroups fit a CU where one did (sigma = 16: 154 -> 116 us); at 64 rows the four conversions per read cost
// more than the second pair of workgroups brings (sigma = 8: 58 -> 64 us), so that ring stays float.  u16 needs
// 255 * sum(kx) <= 65535 (host-checked); the float ring takes any kernel.
template <int RH, int CN, int NO, bool R16>
__global__ __launch_bounds__(256) void k_blur_strip4(const uint8_t* __restrict__ src, long long sstride, int sstep, int w, int h,
                                                     uint8_t* __restrict__ dst, long long dstride, int dstep,
                                                     const int* __restrict__ kxp, const float* __restrict__ kyf,
                                                     const int* __restrict__ kyi, int r, int rows_per_block) {
    static_assert(NO == 1 || NO == 2 || (NO == 4 && R16), "one, two or -- u16 ring only -- four output rows per wave and step");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int npair = r + 1;                                   // 2r+1 taps -> r+1 pairs, the last one (tap, 0)
    const int SEGW = (64 + 2 * r + 2 + 3) & ~3;                // +2: the padded last pair reads one dword further
    uint32_t* s_kx = (uint32_t*)smem;                          // packed (k[2j], k[2j+1]) as 2 x i16
    float* s_ky = (float*)(s_kx + ((npair + 3) & ~3));          // r + 1 taps, then zeros up to the next multiple of four (+ 4)
    int* s_kyi = (int*)(s_ky + ((r + 1 + 3) & ~3) + 4);
    uint32_t* s_seg = (uint32_t*)(s_kyi + ((r + 1 + 3) & ~3));  // [4][SEGW]
    typedef typename std::conditional<R16, uint2, float4>::type ring_t;
    ring_t* s_ring = (ring_t*)(s_seg + 4 * SEGW);              // [RH][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * 64;
    const int y0 = blockIdx.y * rows_per_block, y1 = min(h, y0 + rows_per_block);
    const uint8_t* S = src + (long long)blockIdx.z * sstride;
    uint8_t* D = dst + (long long)blockIdx.z * dstride + (size_t)min(x0 + lane, w - 1) * CN;
    const bool live = x0 + lane < w;
    for (int i = tid; i < npair; i += 256) s_kx[i] = (uint32_t)kxp[i];
    for (int i = tid; i <= r; i += 256) { s_ky[i] = kyf[i]; s_kyi[i] = kyi[i]; }
    for (int i = r + 1 + tid; i < ((r + 1 + 3) & ~3) + 4; i += 256) s_ky[i] = 0.f;

    // this lane's (up to four) columns of a segment, clamped into the row
    int sxc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) sxc[q] = min(max(x0 - r + lane + 64 * q, 0), w - 1) * CN;
    const int nq = (SEGW + 63) / 64;                           // dwords per lane per segment (<= 4 for r <= 60)
    uint32_t* seg = s_seg + wv * SEGW;
    auto request = [&](int ys, uint32_t* v) {
        const uint8_t* row = S + (size_t)min(max(ys, 0), h - 1) * sstep;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (q < nq) {
                if (CN == 4) v[q] = *(const uint32_t*)(row + sxc[q]);
                else v[q] = (uint32_t)row[sxc[q]] | ((uint32_t)row[sxc[q] + 1] << 8) | ((uint32_t)row[sxc[q] + 2] << 16);
            }
    };
    auto emit = [&](int y, uint32_t px) {
        if (CN == 4) *(uint32_t*)(D + (size_t)y * dstep) = px;
        else { uint8_t* o = D + (size_t)y * dstep; o[0] = (uint8_t)px; o[1] = (uint8_t)(px >> 8); o[2] = (uint8_t)(px >> 16); }
    };
    const int vec_end = CN == 3 ? (w * 3) & ~3 : 0;            // SymmColumnVec_32s8u covers whole groups of 4 row elements
    const bool tail = CN == 3 && (x0 + lane) * 3 + 2 >= vec_end;
    auto ring = [&](int y) -> float4 {
        if constexpr (R16) {
            const uint2 q = s_ring[(y & (RH - 1)) * 64 + lane];
            return make_float4((float)(q.x & 0xffffu), (float)(q.x >> 16), (float)(q.y & 0xffffu), (float)(q.y >> 16));
        } else {
            return s_ring[(y & (RH - 1)) * 64 + lane];
        }
    };
    auto finish = [&](int yo, bl_float2 sxy, bl_float2 szw) -> uint32_t {   // round, pack; the CN = 3 row tail in integers
        uint32_t px = bl_cvt_pk_u8(sxy.x, 0u, 0);
        px = bl_cvt_pk_u8(sxy.y, px, 1);
        px = bl_cvt_pk_u8(szw.x, px, 2);
        if (CN == 4) px = bl_cvt_pk_u8(szw.y, px, 3);
        if (tail) {
            const float4 c = ring(yo);
            int t[3] = {s_kyi[0] * (int)c.x, s_kyi[0] * (int)c.y, s_kyi[0] * (int)c.z};
            for (int k = 1; k <= r; k++) {
                const float4 pa = ring(min(yo + k, h - 1)), pb = ring(max(yo - k, 0));
                const int fk = s_kyi[k];
                t[0] += fk * ((int)pa.x + (int)pb.x); t[1] += fk * ((int)pa.y + (int)pb.y); t[2] += fk * ((int)pa.z + (int)pb.z);
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                int ti = (t[ch] + (1 << 15)) >> 16;
                asm volatile("" : "+v"(ti));                    // keep shift and clamp apart (v_ashr_pk_u8_i32 hazard)
                if ((x0 + lane) * 3 + ch >= vec_end) px = (px & ~(0xffu << (8 * ch))) | ((uint32_t)sat8(ti) << (8 * ch));
            }
        }
        return px;
    };

    uint32_t cur[NO][4], out_px[NO];
    int out_y[NO];                                             // output rows whose pixels wait in out_px (-1: none)
#pragma unroll
    for (int i = 0; i < NO; i++) { out_y[i] = -1; out_px[i] = 0; }
    int base = y0 - r;                                         // first source row of the step; this wave's: base + wv * NO + i
#pragma unroll
    for (int i = 0; i < NO; i++) request(base + wv * NO + i, cur[i]);
    __syncthreads();                                           // taps are in LDS
    for (; base <= y1 - 1 + r; base += 4 * NO) {
#pragma unroll
        for (int i = 0; i < NO; i++) {
            const int ys = base + wv * NO + i;
            // (a) the segment of row ys -> LDS; last step's pixels out; the row of the next step in flight
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (q < nq && lane + 64 * q < SEGW) seg[lane + 64 * q] = cur[i][q];
            asm volatile("" ::: "memory");
            if (i == 0) {
#pragma unroll
                for (int o = 0; o < NO; o++) {
                    if (out_y[o] >= 0 && live) emit(out_y[o], out_px[o]);
                    out_y[o] = -1;
                }
            }
            request(ys + 4 * NO, cur[i]);
            // (b) row pass
            const uint32_t* win = seg + lane;
            int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (int j = 0; j < npair; j++) {
                const uint32_t p0 = win[2 * j], p1 = win[2 * j + 1];
                bl_short2 kk;
                const uint32_t kw = s_kx[j];
                __builtin_memcpy(&kk, &kw, 4);
                bl_short2 c;
                uint32_t pr;
                pr = __builtin_amdgcn_perm(p1, p0, 0x0c040c00u); __builtin_memcpy(&c, &pr, 4); a0 = __builtin_amdgcn_sdot2(c, kk, a0, false);
                pr = __builtin_amdgcn_perm(p1, p0, 0x0c050c01u); __builtin_memcpy(&c, &pr, 4); a1 = __builtin_amdgcn_sdot2(c, kk, a1, false);
                pr = __builtin_amdgcn_perm(p1, p0, 0x0c060c02u); __builtin_memcpy(&c, &pr, 4); a2 = __builtin_amdgcn_sdot2(c, kk, a2, false);
                if (CN == 4) { pr = __builtin_amdgcn_perm(p1, p0, 0x0c070c03u); __builtin_memcpy(&c, &pr, 4); a3 = __builtin_amdgcn_sdot2(c, kk, a3, false); }
            }
            if constexpr (R16) s_ring[(ys & (RH - 1)) * 64 + lane] = make_uint2((uint32_t)a0 | ((uint32_t)a1 << 16), (uint32_t)a2 | ((uint32_t)a3 << 16));
            else s_ring[(ys & (RH - 1)) * 64 + lane] = make_float4((float)a0, (float)a1, (float)a2, (float)a3);
        }
        __syncthreads();
        // (c) column pass for the rows centred r rows above this wave's source rows.  Rows replicate at the frame's edges
        // exactly like the CPU's BORDER_REPLICATE: clamp the ROW index, whose sums are in the ring whenever the clamped row
        // lies inside this block's source range.
        const int yo = base + wv * NO - r;
        if (yo + NO - 1 >= y0 && yo < y1) {                    // wave-uniform
            const float f0 = s_ky[0];
            if constexpr (NO == 1) {
                const float4 c = ring(yo);
                bl_float2 sxy = bl_float2{c.x, c.y} * f0 + 0.f, szw = bl_float2{c.z, c.w} * f0 + 0.f;
                for (int k = 1; k <= r; k++) {
                    const float4 pa = ring(min(yo + k, h - 1)), pb = ring(max(yo - k, 0));
                    const float f = s_ky[k];
                    sxy = sxy + (bl_float2{pa.x, pa.y} + bl_float2{pb.x, pb.y}) * f;
                    szw = szw + (bl_float2{pa.z, pa.w} + bl_float2{pb.z, pb.w}) * f;
                }
                out_px[0] = finish(yo, sxy, szw);
                out_y[0] = yo;
            } else if constexpr (NO == 4) {
                // outputs yo .. yo + 3.  Tap k of output j pairs rows yo+j+k and yo+j-k: from one tap to the next the four upper
                // rows move up by one and the four lower rows down by one, so TWO ring reads per tap serve four outputs (a quarter
                // of the LDS bytes per pixel of the one-row form).  The windows are circular in registers -- row yo+q sits in
                // up[q & 3], row yo-q in lo[(-q) & 3] -- and the taps run in groups of four so that every index is static; the
                // kernel's taps are padded with zeros to a multiple of four (a zero tap adds +0.0f to a non-negative sum: exact;
                // the rows it reads are u16 -> finite).
                float4 up[4], lo[4];
                bl_float2 sxy[4], szw[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    up[j] = lo[j] = ring(min(max(yo + j, 0), h - 1));
                    sxy[j] = bl_float2{up[j].x, up[j].y} * f0 + 0.f;
                    szw[j] = bl_float2{up[j].z, up[j].w} * f0 + 0.f;
                }
                for (int k = 1; k <= r; k += 4) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int kk = k + u;                               // kk = 1 + u (mod 4)
                        up[u & 3] = ring(min(yo + 3 + kk, h - 1));          // row yo + (kk + 3): slot (kk + 3) & 3 = u
                        lo[(3 - u) & 3] = ring(max(yo - kk, 0));            // row yo - kk: slot (-kk) & 3 = (3 - u) & 3
                        const float f = s_ky[kk];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const float4 a = up[(j + 1 + u) & 3], b = lo[(j + 3 - u) & 3];   // rows yo+j+kk and yo+j-kk
                            sxy[j] = sxy[j] + (bl_float2{a.x, a.y} + bl_float2{b.x, b.y}) * f;
                            szw[j] = szw[j] + (bl_float2{a.z, a.w} + bl_float2{b.z, b.w}) * f;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (yo + j >= y0 && yo + j < y1) { out_px[j] = finish(yo + j, sxy[j], szw[j]); out_y[j] = yo + j; }
            } else {
                // outputs yo and yo + 1: row yo+k of this tap is row (yo+1)+(k-1) of the last one, row (yo+1)-k is row yo-(k-1)
                const int ylo = max(yo, 0), yhi = min(yo + 1, h - 1);      // (clamped like every other ring row)
                float4 b0 = ring(ylo), a1 = ring(yhi);                     // k = 0: the two centre rows
                bl_float2 s0xy = bl_float2{b0.x, b0.y} * f0 + 0.f, s0zw = bl_float2{b0.z, b0.w} * f0 + 0.f;
                bl_float2 s1xy = bl_float2{a1.x, a1.y} * f0 + 0.f, s1zw = bl_float2{a1.z, a1.w} * f0 + 0.f;
                for (int k = 1; k <= r; k++) {
                    const float4 a0 = a1, b1 = b0;                          // slid from the tap before
                    a1 = ring(min(yo + 1 + k, h - 1));
                    b0 = ring(max(yo - k, 0));
                    const float f = s_ky[k];
                    s0xy = s0xy + (bl_float2{a0.x, a0.y} + bl_float2{b0.x, b0.y}) * f;
                    s0zw = s0zw + (bl_float2{a0.z, a0.w} + bl_float2{b0.z, b0.w}) * f;
                    s1xy = s1xy + (bl_float2{a1.x, a1.y} + bl_float2{b1.x, b1.y}) * f;
                    s1zw = s1zw + (bl_float2{a1.z, a1.w} + bl_float2{b1.z, b1.w}) * f;
                }
                if (yo >= y0) { out_px[0] = finish(yo, s0xy, s0zw); out_y[0] = yo; }
                if (yo + 1 < y1) { out_px[1] = finish(yo + 1, s1xy, s1zw); out_y[1] = yo + 1; }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < NO; o++)
        if (out_y[o] >= 0 && live) emit(out_y[o], out_px[o]);
}

// ---------------------------------------------------------------------------------------------------------------------
// The separable convolution on the matrix unit (round 4).  north_star says "no MFMA (there is no dense contraction here)",
// which is true of 2-8-tap resampling; a 49-151-tap 8-bit Gaussian IS a banded contraction, and the VALU forms above spend
// ~350 instructions per output row and wave on it (0.04 / 0.02 of the HBM roofline at sigma 8 / 16).  Both passes are
// C = A x B with v_mfma_i32_16x16x64_i8: one operand is a 16 x 64 window of bytes, the other a Toeplitz band of the taps.
//   * everything is integer and EXACT.  OpenCV's column pass is float, but (taps <= 127, tap sum <= 257, which the host
//     checks) every product (a + b) * ky[k] is below 2^24 and so is every partial sum below 256.0 in units of 2^-16: the float
//     accumulation never rounds until the final cvRound (a sum past 256.0 can, and saturates to 255 either way), so
//     dst = saturate(round_half_even(sum / 65536)) -- or (sum + 2^15) >> 16 for the last (w * cn) & 3 elements of a row, which
//     OpenCV's SSE2 loop leaves to the scalar template -- reproduces it bit for bit.
//   * u8 operands: pixels enter as p - 128 (one XOR), 128 * (tap sum) is added back; the 16-bit row sums are split into
//     two bytes for the column pass (two MFMAs, 256 * hi + lo).
//   * interleaved channels are NOT separated: the row pass convolves along the BYTES of a row with the taps cn bytes apart
//     (three of four band entries are zero -- the matrix unit has the time: 144 MFMAs per 64 x 64-byte tile are 2 us per
//     1080p frame), so operands are plain 16-byte runs of a row.
//   * k_blur_mfma_rows leaves the row sums TRANSPOSED ([byte column][row], two byte planes): the accumulator layout of the
//     MFMA (a lane holds four consecutive ROWS of one byte column) writes that as dwords, and k_blur_mfma_cols reads its
//     operands -- sixteen consecutive rows of one byte column -- as one 16-byte run again.
// Operand maps checked on hardware with one-hot data (tools/mfma_i8_probe.hip): lane l, byte b of A is A[l & 15][16 (l >> 4) + b],
// of B is B[16 (l >> 4) + b][l & 15]; C[4 (l >> 4) + reg][l & 15].
typedef int bm_v4i __attribute__((ext_vector_type(4)));
constexpr int BM_W = 64, BM_H = 64;                 // bytes across / rows down per workgroup tile
// Which tile a workgroup takes.  Workgroups go to the eight XCDs round-robin in launch order, so tiles that are neighbours in
// x -- whose windows overlap by the 2 * CN * r halo bytes (four of every five staged bytes at sigma = 8) -- would each pull
// that halo into a different L2: FETCH_SIZE showed the row pass reading 57 MB for an 8.3 MB frame.  Launch slot g is therefore
// mapped to tile (slots of XCD k) = one contiguous range of the launch's tiles -- in row-major order for the passes whose
// windows overlap along x (rows, fused), in column-major order for the column pass, whose windows overlap along y.
template <bool COLUMN_MAJOR>
__device__ __forceinline__ void bm_xcd_tile(int* bx, int* by, int* bz) {
    const unsigned nx = gridDim.x, T = gridDim.x * gridDim.y, N = T * gridDim.z;
    const unsigned g = blockIdx.x + blockIdx.y * nx + blockIdx.z * T;
    const unsigned k = g & 7, i = g >> 3, q = N >> 3, rem = N & 7;
    const unsigned n = k * q + (k < rem ? k : rem) + i;
    const unsigned t = n % T;
    *bz = (int)(n / T);
    if (COLUMN_MAJOR && !(nx & 7)) {                           // launch order already keeps a column of tiles on one XCD (slot + nx = same XCD) and walks rows of the frame
        *bx = (int)blockIdx.x; *by = (int)blockIdx.y; *bz = (int)blockIdx.z;
    } else if (COLUMN_MAJOR) { *bx = (int)(t / gridDim.y); *by = (int)(t % gridDim.y); }
    else { *by = (int)(t / nx); *bx = (int)(t % nx); }
}

constexpr int BM_MAXC = 8;                          // 64-byte chunks of a row window: 16 + 2 * cn * r <= 512

// the Toeplitz operand of chunk c as the lanes hold it (built on the host, once per call: 64 lanes x 16 bytes): byte b of lane l =
// tap[(64 c + 16 (l >> 4) + b - (l & 15)) / dil] where that is a whole number in 0 .. 2r, else 0
static void bm_band_host(const std::vector<int>& ik, int r, int dil, int nchunk, std::vector<int>* blob) {
    for (int c = 0; c < nchunk; c++)
        for (int lane = 0; lane < 64; lane++)
            for (int q = 0; q < 4; q++) {
                uint32_t word = 0;
                for (int b = 0; b < 4; b++) {
                    const int d = 64 * c + 16 * (lane >> 4) + 4 * q + b - (lane & 15);
                    const int m = d >= 0 ? d / dil : -1;
                    const int v = (d >= 0 && m * dil == d && m <= 2 * r) ? ik[(size_t)m] : 0;
                    word |= (uint32_t)(v & 0xff) << (8 * b);
                }
                blob->push_back((int)word);
            }
}

template <int CN, bool WIDE>
__global__ __launch_bounds__(256) void k_blur_mfma_rows(const uint8_t* __restrict__ src, long long sstride, int sstep, int w, int h,
                                                        uint8_t* __restrict__ planes, long long pstride, int hp, int roww_pad,
                                                        const bm_v4i* __restrict__ bands, int r, int nchunk, int pitch_s, int bias) {
    extern __shared__ __attribute__((aligned(16))) uint8_t bm_smem[];
    uint8_t* s_src = bm_smem;                                       // [BM_H][pitch_s]: source bytes - 128, window column 0 = byte x0b - CN r
    uint8_t* s_pl = bm_smem + (size_t)BM_H * pitch_s;               // [2 or 3][BM_W][BM_H + 16]: the row sums' low / high bytes - 128 (WIDE: and bit 16), transposed
    constexpr int NPL = WIDE ? 3 : 2;
    constexpr int PT = BM_H + 16;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int tbx, tby, tbz;
    bm_xcd_tile<false>(&tbx, &tby, &tbz);
    const int x0b = tbx * BM_W, y0 = tby * BM_H, roww = w * CN;
    const uint8_t* frame = src + (long long)tbz * sstride;
    // stage the rows: replicated borders; sixteen bytes at a time where they are inside the row (a byte-aligned address is fine)
    const int wbytes = BM_W + 2 * CN * r, wq = (wbytes + 15) >> 4;
    for (int idx = t; idx < BM_H * wq; idx += 256) {
        const int ry = idx / wq, bc = (idx - ry * wq) * 16;
        const uint8_t* row = frame + (size_t)min(y0 + ry, h - 1) * sstep;
        const int gb = x0b - CN * r + bc;
        uint32_t v[4];
        if (gb >= 0 && gb + 15 < roww) {
            __builtin_memcpy(v, row + gb, 16);
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = 0;
#pragma unroll
            for (int b = 0; b < 16; b++) {
                const int g = gb + b;
                int px = g >= 0 ? g / CN : -((-g + CN - 1) / CN);    // floor division
                const int ch = g - px * CN;
                px = min(max(px, 0), w - 1);
                v[b >> 2] |= (uint32_t)row[px * CN + ch] << (8 * (b & 3));
            }
        }
        *(uint4*)(s_src + (size_t)ry * pitch_s + bc) = make_uint4(v[0] ^ 0x80808080u, v[1] ^ 0x80808080u, v[2] ^ 0x80808080u, v[3] ^ 0x80808080u);
    }
    bm_v4i band[BM_MAXC];
#pragma unroll
    for (int c = 0; c < BM_MAXC; c++) band[c] = c < nchunk ? bands[c * 64 + lane] : bm_v4i{0, 0, 0, 0};
    __syncthreads();
    // C[row][byte] for 16 x 16 tiles: A = the rows' windows, B = the band
    for (int tile = wv; tile < (BM_W / 16) * (BM_H / 16); tile += 4) {
        const int xg = tile & 3, rg = tile >> 2;
        bm_v4i acc = {0, 0, 0, 0};
        const uint8_t* arow = s_src + (size_t)(rg * 16 + (lane & 15)) * pitch_s + xg * 16 + 16 * (lane >> 4);
#pragma unroll
        for (int c = 0; c < BM_MAXC; c++)
            if (c < nchunk) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(*(const bm_v4i*)(arow + 64 * c), band[c], acc, 0, 0, 0);
        uint32_t lo = 0, hi = 0, top = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t S = (uint32_t)(acc[i] + bias);            // 0 .. 65535 (WIDE: below 2^17): the row sum of OpenCV's int32 row pass
            lo |= (S & 255u) << (8 * i);
            hi |= ((S >> 8) & 255u) << (8 * i);
            top |= (S >> 16) << (8 * i);
        }
        uint8_t* col = s_pl + (size_t)(xg * 16 + (lane & 15)) * PT + rg * 16 + 4 * (lane >> 4);
        *(uint32_t*)col = lo ^ 0x80808080u;
        *(uint32_t*)(col + (size_t)BM_W * PT) = hi ^ 0x80808080u;
        if constexpr (WIDE) *(uint32_t*)(col + 2 * (size_t)BM_W * PT) = top;          // 0 / 1 as it is
    }
    __syncthreads();
    // the tile's planes out, 64 contiguous bytes (64 rows) per byte column
    uint8_t* pl = planes + (long long)tbz * pstride;
    for (int idx = t; idx < NPL * BM_W * 4; idx += 256) {
        const int q = idx & 3, xb = (idx >> 2) & (BM_W - 1), p = idx >> 8;
        const uint4 v = *(const uint4*)(s_pl + (size_t)(p * BM_W + xb) * PT + q * 16);
        *(uint4*)(pl + ((size_t)p * roww_pad + x0b + xb) * hp + y0 + q * 16) = v;
    }
}

template <int CN, bool WIDE>
__global__ __launch_bounds__(256) void k_blur_mfma_cols(const uint8_t* __restrict__ planes, long long pstride, int hp, int roww_pad,
                                                        uint8_t* __restrict__ dst, long long dstride, int dstep, int w, int h,
                                                        const bm_v4i* __restrict__ bands, int r, int nchunk, int pitch_p, int bias) {
    extern __shared__ __attribute__((aligned(16))) uint8_t bm_smem[];
    uint8_t* s_pl = bm_smem;                                        // [2 or 3][BM_W][pitch_p]: row 0 = image row y0 - r
    constexpr int NPL = WIDE ? 3 : 2;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int tbx, tby, tbz;
    bm_xcd_tile<true>(&tbx, &tby, &tbz);
    const int x0b = tbx * BM_W, y0 = tby * BM_H, roww = w * CN;
    const uint8_t* pl = planes + (long long)tbz * pstride;
    const int nrows = BM_H + 2 * r, n16 = (nrows + 15) >> 4;
    for (int idx = t; idx < NPL * BM_W * n16; idx += 256) {
        const int piece = idx % n16, xb = (idx / n16) & (BM_W - 1), p = idx / (n16 * BM_W);
        const uint8_t* colp = pl + ((size_t)p * roww_pad + x0b + xb) * hp;
        const int ya = y0 - r + piece * 16;
        uint4 v;
        if (ya >= 0 && ya + 15 < h) {
            __builtin_memcpy(&v, colp + ya, 16);                    // (a byte-aligned address is fine)
        } else {
            uint32_t a[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 16; i++) a[i >> 2] |= (uint32_t)colp[min(max(ya + i, 0), h - 1)] << (8 * (i & 3));
            v = make_uint4(a[0], a[1], a[2], a[3]);
        }
        *(uint4*)(s_pl + (size_t)(p * BM_W + xb) * pitch_p + piece * 16) = v;
    }
    bm_v4i band[3];
#pragma unroll
    for (int c = 0; c < 3; c++) band[c] = c < nchunk ? bands[c * 64 + lane] : bm_v4i{0, 0, 0, 0};
    __syncthreads();
    uint8_t* out = dst + (long long)tbz * dstride;
    const int vec_end = roww & ~3;                                  // OpenCV's SSE2 column loop; behind it the scalar template
    for (int tile = wv; tile < (BM_W / 16) * (BM_H / 16); tile += 4) {
        const int xg = tile & 3, og = tile >> 2;
        bm_v4i al = {0, 0, 0, 0}, ah = {0, 0, 0, 0}, at = {0, 0, 0, 0};
        const uint8_t* acol = s_pl + (size_t)(xg * 16 + (lane & 15)) * pitch_p + og * 16 + 16 * (lane >> 4);
#pragma unroll
        for (int c = 0; c < 3; c++)
            if (c < nchunk) {
                al = __builtin_amdgcn_mfma_i32_16x16x64_i8(*(const bm_v4i*)(acol + 64 * c), band[c], al, 0, 0, 0);
                ah = __builtin_amdgcn_mfma_i32_16x16x64_i8(*(const bm_v4i*)(acol + (size_t)BM_W * pitch_p + 64 * c), band[c], ah, 0, 0, 0);
                if constexpr (WIDE) at = __builtin_amdgcn_mfma_i32_16x16x64_i8(*(const bm_v4i*)(acol + 2 * (size_t)BM_W * pitch_p + 64 * c), band[c], at, 0, 0, 0);
            }
        const int y = y0 + og * 16 + (lane & 15), xb = x0b + xg * 16 + 4 * (lane >> 4);
        if (y >= h || xb >= roww) continue;
        uint32_t px = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t T = (uint32_t)(al[i] + 256 * ah[i] + 65536 * at[i] + bias);        // the column sum in units of 2^-16
            uint32_t q = T >> 16;
            const uint32_t rem = T & 0xffffu;
            if (xb + i < vec_end) q += (rem > 0x8000u || (rem == 0x8000u && (q & 1u))) ? 1u : 0u;   // cvRound: half to even
            else q += rem >= 0x8000u ? 1u : 0u;                                                          // (sum + 2^15) >> 16
            px |= min(q, 255u) << (8 * i);
        }
        uint8_t* o = out + (size_t)y * dstep + xb;
        if (xb + 3 < roww) *(uint32_t*)o = px;
        else for (int i = 0; xb + i < roww; i++) o[i] = (uint8_t)(px >> (8 * i));
    }
}

// Both passes in ONE launch for the radii whose tile fits LDS twice per compute unit: the rows of the tile AND its 2r halo rows
// are reduced into the transposed planes in LDS (never in memory), then the columns.  The halo rows are reduced once per tile
// row they border (x (64 + 2r) / 64 of the row pass: the matrix unit has the time), the row sums never leave the chip.
template <int CN, bool WIDE>
__global__ __launch_bounds__(256) void k_blur_mfma_fused(const uint8_t* __restrict__ src, long long sstride, int sstep, int w, int h,
                                                         uint8_t* __restrict__ dst, long long dstride, int dstep,
                                                         const bm_v4i* __restrict__ bands_r, const bm_v4i* __restrict__ bands_c, int r, int nrc, int ncc,
                                                         int pitch_s, int pitch_p, int bias_r, int bias_c) {
    extern __shared__ __attribute__((aligned(16))) uint8_t bm_smem[];
    const int nr16 = (BM_H + 2 * r + 15) & ~15;                     // staged rows: image rows y0 - r .. (replicated past the borders)
    uint8_t* s_src = bm_smem;                                       // [nr16][pitch_s]
    uint8_t* s_pl = bm_smem + (size_t)nr16 * pitch_s;               // [2 or 3][BM_W][pitch_p], row 0 = image row y0 - r
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int tbx, tby, tbz;
    bm_xcd_tile<false>(&tbx, &tby, &tbz);
    const int x0b = tbx * BM_W, y0 = tby * BM_H, roww = w * CN;
    const uint8_t* frame = src + (long long)tbz * sstride;
    const int wbytes = BM_W + 2 * CN * r, wq = (wbytes + 15) >> 4;
    for (int idx = t; idx < nr16 * wq; idx += 256) {
        const int ry = idx / wq, bc = (idx - ry * wq) * 16;
        const uint8_t* row = frame + (size_t)min(max(y0 - r + ry, 0), h - 1) * sstep;
        const int gb = x0b - CN * r + bc;
        uint32_t v[4];
        if (gb >= 0 && gb + 15 < roww) {
            __builtin_memcpy(v, row + gb, 16);
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = 0;
#pragma unroll
            for (int b = 0; b < 16; b++) {
                const int g = gb + b;
                int px = g >= 0 ? g / CN : -((-g + CN - 1) / CN);    // floor division
                const int ch = g - px * CN;
                px = min(max(px, 0), w - 1);
                v[b >> 2] |= (uint32_t)row[px * CN + ch] << (8 * (b & 3));
            }
        }
        *(uint4*)(s_src + (size_t)ry * pitch_s + bc) = make_uint4(v[0] ^ 0x80808080u, v[1] ^ 0x80808080u, v[2] ^ 0x80808080u, v[3] ^ 0x80808080u);
    }
    bm_v4i band[BM_MAXC];
#pragma unroll
    for (int c = 0; c < BM_MAXC; c++) band[c] = c < nrc ? bands_r[c * 64 + lane] : bm_v4i{0, 0, 0, 0};
    __syncthreads();
    for (int tile = wv; tile < (BM_W / 16) * (nr16 / 16); tile += 4) {
        const int xg = tile & 3, rg = tile >> 2;
        bm_v4i acc = {0, 0, 0, 0};
        const uint8_t* arow = s_src + (size_t)(rg * 16 + (lane & 15)) * pitch_s + xg * 16 + 16 * (lane >> 4);
#pragma unroll
        for (int c = 0; c < BM_MAXC; c++)
            if (c < nrc) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(*(const bm_v4i*)(arow + 64 * c), band[c], acc, 0, 0, 0);
        uint32_t lo = 0, hi = 0, top = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t S = (uint32_t)(acc[i] + bias_r);
            lo |= (S & 255u) << (8 * i);
            hi |= ((S >> 8) & 255u) << (8 * i);
            top |= (S >> 16) << (8 * i);
        }
        uint8_t* col = s_pl + (size_t)(xg * 16 + (lane & 15)) * pitch_p + rg * 16 + 4 * (lane >> 4);
        *(uint32_t*)col = lo ^ 0x80808080u;
        *(uint32_t*)(col + (size_t)BM_W * pitch_p) = hi ^ 0x80808080u;
        if constexpr (WIDE) *(uint32_t*)(col + 2 * (size_t)BM_W * pitch_p) = top;
    }
    bm_v4i bandc[3];
#pragma unroll
    for (int c = 0; c < 3; c++) bandc[c] = c < ncc ? bands_c[c * 64 + lane] : bm_v4i{0, 0, 0, 0};
    __syncthreads();
    uint8_t* out = dst + (long long)tbz * dstride;
    const int vec_end = roww & ~3;
    for (int tile = wv; tile < (BM_W / 16) * (BM_H / 16); tile += 4) {
        const int xg = tile & 3, og = tile >> 2;
        bm_v4i al = {0, 0, 0, 0}, ah = {0, 0, 0, 0}, at = {0, 0, 0, 0};
        const uint8_t* acol = s_pl + (size_t)(xg * 16 + (lane & 15)) * pitch_p + og * 16 + 16 * (lane >> 4);
#pragma unroll
        for (int c = 0; c < 3; c++)
            if (c < ncc) {
                al = __builtin_amdgcn_mfma_i32_16x16x64_i8(*(const bm_v4i*)(acol + 64 * c), bandc[c], al, 0, 0, 0);
                ah = __builtin_amdgcn_mfma_i32_16x16x64_i8(*(const bm_v4i*)(acol + (size_t)BM_W * pitch_p + 64 * c), bandc[c], ah, 0, 0, 0);
                if constexpr (WIDE) at = __builtin_amdgcn_mfma_i32_16x16x64_i8(*(const bm_v4i*)(acol + 2 * (size_t)BM_W * pitch_p + 64 * c), bandc[c], at, 0, 0, 0);
            }
        const int y = y0 + og * 16 + (lane & 15), xb = x0b + xg * 16 + 4 * (lane >> 4);
        if (y >= h || xb >= roww) continue;
        uint32_t px = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t T = (uint32_t)(al[i] + 256 * ah[i] + 65536 * at[i] + bias_c);
            uint32_t q = T >> 16;
            const uint32_t rem = T & 0xffffu;
            if (xb + i < vec_end) q += (rem > 0x8000u || (rem == 0x8000u && (q & 1u))) ? 1u : 0u;
            else q += rem >= 0x8000u ? 1u : 0u;
            px |= min(q, 255u) << (8 * i);
        }
        uint8_t* o = out + (size_t)y * dstep + xb;
        if (xb + 3 < roww) *(uint32_t*)o = px;
        else for (int i = 0; xb + i < roww; i++) o[i] = (uint8_t)(px >> (8 * i));
    }
}

// The two launches above.  IMP_ERROR_UNSUPPORTED when the exactness conditions (taps <= 127, sum <= 257) or the size limits
// do not hold: the caller goes on to the VALU kernels.
static int launch_gaussian_mfma(const Frames& f, const std::vector<int>& ik, int r, hipStream_t s) {
    const View& v = f.v;
    long long sum = 0;
    int top = 0;
    for (int k : ik) { sum += k; top = std::max(top, k); }
    // (taps are operands of the i8 MFMA; 2 * 255 * sum * top < 2^24 keeps every product of OpenCV's float column pass exact)
    if (top > 127 || sum > 511 || 2LL * 255 * sum * top >= (1LL << 24) || r < 1 || 2 * r + 1 > 250) return IMP_ERROR_UNSUPPORTED;
    // Tap sums of 258 .. 260 (eleven of the sigmas 0.5 .. 25.0 in steps of 0.1, sigma = 4 among them) make row sums of 17
    // bits: a third byte plane holds bit 16 and the column pass takes a third MFMA per chunk (round 5; those sigmas ran on
    // the VALU kernels before).  The column sum is exact all the same: a partial sum can only round in float once it is past
    // 256.0, and then the pixel saturates to 255 whatever the rounding did.
    const bool wide = sum > 257;
    const int npl = wide ? 3 : 2;
    const int cn = v.c, roww = v.w * cn;
    const int nrc = (16 + 2 * cn * r + 63) / 64, ncc = (16 + 2 * r + 63) / 64;
    if (nrc > BM_MAXC || ncc > 3) return IMP_ERROR_UNSUPPORTED;
    if ((f.dstep & 3) || ((uintptr_t)f.dst & 3) || (f.dst_stride & 3)) return IMP_ERROR_UNSUPPORTED;
    const int roww_pad = (roww + BM_W - 1) / BM_W * BM_W, hp = (v.h + BM_H - 1) / BM_H * BM_H;
    const int pitch_s = 64 * nrc + 64 + 16, pitch_p = 64 * ncc + 64 + 16;   // (+16: sixteen rows / byte columns start in sixteen different 16-byte slots of the banks)
    const size_t lds_r = (size_t)BM_H * pitch_s + (size_t)npl * BM_W * (BM_H + 16), lds_c = (size_t)npl * BM_W * pitch_p;
    const long long pstride = (long long)npl * roww_pad * hp;
    void *dev_k = nullptr, *planes = nullptr;
    std::vector<int> blob;                                   // the two passes' band operands, lane by lane
    bm_band_host(ik, r, cn, nrc, &blob);
    const size_t off_c = blob.size();
    bm_band_host(ik, r, 1, ncc, &blob);
    if (int rc = upload_small(blob.data(), blob.size() * 4, &dev_k, s)) return rc;
    // one launch for the small radii (the halo rows are cheap there and four tiles fit a compute unit); IMPGPU_BLUR_MFMA2=1: always two (A/B)
    const int nr16 = (BM_H + 2 * r + 15) & ~15;
    const size_t lds_f = (size_t)nr16 * pitch_s + (size_t)npl * BM_W * pitch_p;
    static const bool two = ab_env("IMPGPU_BLUR_MFMA2") != nullptr;
    if (lds_f <= (size_t)ab_env_int("IMPGPU_BLUR_FUSE_KB", wide ? 50 : 40) * 1024 && !two && f.count <= 65535) {        // (measured at 1080p BGRA: sigma 2 19 us against 24 in two launches, sigma 8 32 against 30, sigma 12 48 against 35)
        hipError_t e = hipSuccess;
        const dim3 grid((unsigned)(roww_pad / BM_W), (unsigned)(hp / BM_H), (unsigned)f.count);
        const bm_v4i* br = (const bm_v4i*)dev_k;
        const bm_v4i* bc = (const bm_v4i*)((const int*)dev_k + off_c);
#define IMP_BLUR_FUSED(CN_, W_)                                                                                                      \
    do {                                                                                                                             \
        e = lds_limit_once<k_blur_mfma_fused<CN_, W_>>();                                                                            \
        if (e == hipSuccess)                                                                                                         \
            hipLaunchKernelGGL((k_blur_mfma_fused<CN_, W_>), grid, dim3(256), lds_f, s, f.src, f.src_stride, v.step, v.w, v.h, f.dst, f.dst_stride, \
                               f.dstep, br, bc, r, nrc, ncc, pitch_s, pitch_p, (int)(128 * sum), (int)(128 * sum * 257));             \
    } while (0)
        if (wide) { if (cn == 4) IMP_BLUR_FUSED(4, true); else if (cn == 3) IMP_BLUR_FUSED(3, true); else IMP_BLUR_FUSED(1, true); }
        else { if (cn == 4) IMP_BLUR_FUSED(4, false); else if (cn == 3) IMP_BLUR_FUSED(3, false); else IMP_BLUR_FUSED(1, false); }
#undef IMP_BLUR_FUSED
        if (e == hipSuccess) e = hipGetLastError();
        dev_free_on(dev_k, s);
        if (e != hipSuccess) { set_error("k_blur_mfma_fused", e); return IMP_ERROR_DEVICE; }
        return IMP_OK;
    }
    int chunk = (int)std::min<long long>(f.count, std::max<long long>(1, (1LL << 30) / pstride));
    if (int rc = dev_alloc_on((size_t)pstride * chunk, &planes, s)) { dev_free_on(dev_k, s); return rc; }
    hipError_t e = hipSuccess;
    const int bias_r = (int)(128 * sum), bias_c = (int)(128 * sum * 257);
    for (int f0 = 0; f0 < f.count && e == hipSuccess; f0 += chunk) {
        const int n = std::min(chunk, f.count - f0);
        const dim3 grid((unsigned)(roww_pad / BM_W), (unsigned)(hp / BM_H), (unsigned)n);
        const uint8_t* src = f.src + (long long)f0 * f.src_stride;
        uint8_t* dst = f.dst + (long long)f0 * f.dst_stride;
#define IMP_BLUR_MFMA(CN_, W_)                                                                                                              \
    do {                                                                                                                                    \
        e = lds_limit_once<k_blur_mfma_rows<CN_, W_>>();                                                                                    \
        if (e == hipSuccess) e = lds_limit_once<k_blur_mfma_cols<CN_, W_>>();                                                               \
        if (e == hipSuccess) {                                                                                                              \
            hipLaunchKernelGGL((k_blur_mfma_rows<CN_, W_>), grid, dim3(256), lds_r, s, src, f.src_stride, v.step, v.w, v.h, (uint8_t*)planes, pstride, hp, \
                               roww_pad, (const bm_v4i*)dev_k, r, nrc, pitch_s, bias_r);                                                       \
            hipLaunchKernelGGL((k_blur_mfma_cols<CN_, W_>), grid, dim3(256), lds_c, s, (const uint8_t*)planes, pstride, hp, roww_pad, dst, f.dst_stride, \
                               f.dstep, v.w, v.h, (const bm_v4i*)((const int*)dev_k + off_c), r, ncc, pitch_p, bias_c);                                              \
        }                                                                                                                                   \
    } while (0)
        if (wide) { if (cn == 4) IMP_BLUR_MFMA(4, true); else if (cn == 3) IMP_BLUR_MFMA(3, true); else IMP_BLUR_MFMA(1, true); }
        else { if (cn == 4) IMP_BLUR_MFMA(4, false); else if (cn == 3) IMP_BLUR_MFMA(3, false); else IMP_BLUR_MFMA(1, false); }
#undef IMP_BLUR_MFMA
        if (e == hipSuccess) e = hipGetLastError();
    }
    dev_free_on(planes, s);
    dev_free_on(dev_k, s);
    if (e != hipSuccess) { set_error("k_blur_mfma", e); return IMP_ERROR_DEVICE; }
    return IMP_OK;
}

// src view -> dst (same size, BGRA, separate buffers).  IMP_ERROR_UNSUPPORTED when the fused form does not apply
// (other channel counts, radius > 16, fixed-point taps summing above 257, 1-pixel axes): callers fall back to
// launch_gaussian.  sigma must give ksize > 1.
int launch_gaussian_fused(const Frames& f, double sigma, hipStream_t s) {
    const View& v = f.v;
    if ((v.c != 4 && v.c != 3) || f.count <= 0 || f.count > 65535 || f.dw != v.w || f.dh != v.h || v.w < 2 || v.h < 2) return IMP_ERROR_UNSUPPORTED;
    if (f.src == f.dst) return IMP_ERROR_UNSUPPORTED;
    if (v.c == 4 && (((uintptr_t)f.src | (uintptr_t)f.dst | (uintptr_t)v.step | (uintptr_t)f.dstep | (uintptr_t)f.src_stride | (uintptr_t)f.dst_stride) & 3))
        return IMP_ERROR_UNSUPPORTED;
    const int ks0 = gaussian_ksize(sigma);
    if (ks0 <= 1 || ks0 > 4096) return IMP_ERROR_UNSUPPORTED;
    std::vector<int> ik;
    gaussian_kernel_fixed(ks0, sigma, &ik);
    long long sum = 0;
    for (int k : ik) sum += k;
    const bool fits16 = sum <= 257;                           // row sums fit 16 bits (the fused kernel's plane, the u16 ring)
    if (sum > 65536) return IMP_ERROR_UNSUPPORTED;            // (float ring: 255 * sum < 2^24 stays exact)
    // The 8-bit fixed-point taps of a wide Gaussian are ZERO towards both ends (sigma = 8: 49 taps, the outer 4 + 4 round
    // to 0/256).  A zero tap adds 0 to the integer row sum and +0.0f to the non-negative float column sum -- both exact
    // no-ops -- and a replicated border pixel under a zero tap is irrelevant, so the kernels run on the non-zero core only.
    int r = ks0 / 2;
    {
        const int r0 = r;
        while (r > 1 && ik[r0 + r] == 0 && ik[r0 - r] == 0) r--;
        ik = std::vector<int>(ik.begin() + (r0 - r), ik.begin() + (r0 + r + 1));
    }
    const int ks = 2 * r + 1;
    if (r >= 3) {                                           // the matrix-unit form (exact when its conditions hold; else the VALU kernels below)
        const int rc = launch_gaussian_mfma(f, ik, r, s);
        if (rc != IMP_ERROR_UNSUPPORTED) return rc;
    }
    if (r > 60) return IMP_ERROR_UNSUPPORTED;
    std::vector<int> blob;
    for (int j = 0; j <= r; j++) {                           // pairs (k[2j], k[2j+1]); the pair past the end is (k[2r], 0)
        const int lo = ik[2 * j], hi = (2 * j + 1 < ks) ? ik[2 * j + 1] : 0;
        blob.push_back((lo & 0xffff) | (hi << 16));
    }
    while (blob.size() % 4) blob.push_back(0);
    const size_t off_f = blob.size();
    for (int k = 0; k <= r; k++) {
        float fk = (float)(ik[r + k] * (1. / 65536));
        int bits;
        std::memcpy(&bits, &fk, 4);
        blob.push_back(bits);
    }
    const size_t off_i = blob.size();
    for (int k = 0; k <= r; k++) blob.push_back(ik[r + k]);
    void* dev_k = nullptr;
    if (int rc = upload_small(blob.data(), blob.size() * 4, &dev_k, s)) return rc;
    if (r > 16 || !fits16) {   // column strips with an LDS ring (k_blur_strip4); also any radius whose rounded taps sum above 257
        const int RH0 = 2 * r + 8 <= 64 ? 64 : 128;          // (sigma = 4: 258 -- those kernels used to take the two-pass fallback: 89 us)
        static const bool one_row = ab_env("IMPGPU_BLUR_NO1") != nullptr;     // A/B: one output row per wave and step
        // Four output rows per wave and step (a 128-row ring of u16 sums: radius <= 48, taps summing to at most 257) is built,
        // bit-exact and OFF: two ring reads per tap then serve four outputs -- a quarter of the one-row form's LDS bytes -- but
        // the eight u16 -> float conversions per tap and 162 VGPRs cost more than the reads saved (sigma = 8: 70 us against 57,
        // sigma = 16: 124 against 114).  The column pass is bound by its packed-FP32 issue, not by LDS.  In the binary only with
        // -DIMPGPU_AB_SWITCHES (then IMPGPU_BLUR_FOUR=1 selects it).
        static const bool want_four = ab_env("IMPGPU_BLUR_FOUR") != nullptr;
        const bool four = want_four && !one_row && fits16 && 2 * r + 32 <= 128;
        const int RH = four ? 128 : RH0;
        const bool r16 = RH == 128 && fits16;
        const int NO = four ? 4 : ((!one_row && 2 * r + 16 <= RH) ? 2 : 1);
        const int SEGW = (64 + 2 * r + 2 + 3) & ~3;
        const size_t lds = (size_t)(((r + 1 + 3) & ~3) * 3 + 4 + 4 * SEGW) * 4 + (size_t)RH * 64 * (r16 ? 8 : 16);
        const int nbx = (v.w + 63) / 64;
        int rpb = 256;                                         // taller strips recompute fewer halo rows; shorter ones fill the chip
        while (rpb > 64 && (long long)nbx * ((v.h + rpb - 1) / rpb) * f.count < 1024) rpb /= 2;
        const dim3 sgrid((unsigned)nbx, (unsigned)((v.h + rpb - 1) / rpb), (unsigned)f.count);
        hipError_t e = hipSuccess;
#define IMP_BLUR_STRIP(RH_, CN_, NO_, R16_)                                                                                        \
    do {                                                                                                                           \
        e = lds_limit_once<k_blur_strip4<RH_, CN_, NO_, R16_>>();                                                                  \
        if (e == hipSuccess)                                                                                                       \
            hipLaunchKernelGGL((k_blur_strip4<RH_, CN_, NO_, R16_>), sgrid, dim3(256), lds, s, f.src, f.src_stride, v.step, v.w, v.h,     \
                               f.dst, f.dst_stride, f.dstep, (const int*)dev_k, (const float*)((const int*)dev_k + off_f),          \
                               (const int*)dev_k + off_i, r, rpb);                                                                  \
    } while (0)
#ifdef IMPGPU_AB_SWITCHES
        if (NO == 4) {
            if (v.c == 4) IMP_BLUR_STRIP(128, 4, 4, true); else IMP_BLUR_STRIP(128, 3, 4, true);
        } else
#endif
        switch ((RH == 64 ? 0 : 4) + (v.c == 4 ? 0 : 2) + (NO == 2 ? 0 : 1) + (r16 ? 8 : 0)) {
            case 0: IMP_BLUR_STRIP(64, 4, 2, false); break;
            case 1: IMP_BLUR_STRIP(64, 4, 1, false); break;
            case 2: IMP_BLUR_STRIP(64, 3, 2, false); break;
            case 3: IMP_BLUR_STRIP(64, 3, 1, false); break;
            case 4: IMP_BLUR_STRIP(128, 4, 2, false); break;
            case 5: IMP_BLUR_STRIP(128, 4, 1, false); break;
            case 6: IMP_BLUR_STRIP(128, 3, 2, false); break;
            case 7: IMP_BLUR_STRIP(128, 3, 1, false); break;
            case 12: IMP_BLUR_STRIP(128, 4, 2, true); break;
            case 13: IMP_BLUR_STRIP(128, 4, 1, true); break;
            case 14: IMP_BLUR_STRIP(128, 3, 2, true); break;
            default: IMP_BLUR_STRIP(128, 3, 1, true); break;
        }
#undef IMP_BLUR_STRIP
        if (e == hipSuccess) e = hipGetLastError();
        dev_free_on(dev_k, s);
        if (e != hipSuccess) { set_error("k_blur_strip4", e); return IMP_ERROR_DEVICE; }
        return IMP_OK;
    }
    const int TH = 32;
    const int SW = 64 + 2 * r + 2, SH = TH + 2 * r;
    const size_t lds = (size_t)(((r + 1 + 3) & ~3) * 3 + ((SW * SH + 3) & ~3)) * 4 + (size_t)SH * 64 * 8;
    const dim3 grid((v.w + 63) / 64, (v.h + TH - 1) / TH, f.count), block(256);
    if (grid.y > 65535) { dev_free_on(dev_k, s); return IMP_ERROR_UNSUPPORTED; }
    if (v.c == 4)
        hipLaunchKernelGGL((k_blur_fused4<32, 4>), grid, block, lds, s, f.src, f.src_stride, v.step, v.w, v.h, f.dst, f.dst_stride, f.dstep,
                           (const int*)dev_k, (const float*)((const int*)dev_k + off_f), (const int*)dev_k + off_i, r, r);
    else
        hipLaunchKernelGGL((k_blur_fused4<32, 3>), grid, block, lds, s, f.src, f.src_stride, v.step, v.w, v.h, f.dst, f.dst_stride, f.dstep,
                           (const int*)dev_k, (const float*)((const int*)dev_k + off_f), (const int*)dev_k + off_i, r, r);
    hipError_t e = hipGetLastError();
    dev_free_on(dev_k, s);
    if (e != hipSuccess) { set_error("k_blur_fused4", e); return IMP_ERROR_DEVICE; }
    return IMP_OK;
}

int launch_gaussian(uint8_t* d, long long stride, int w, int h, int c, int step, int count, double sigma, hipStream_t s) {
    if (count <= 0 || !(sigma > 0)) return IMP_OK;
    int kxs = gaussian_ksize(sigma), kys = kxs;
    if (h == 1) kys = 1;        // GaussianBlur: single-row / single-column images drop that axis
    if (w == 1) kxs = 1;
    if (kxs == 1 && kys == 1) return IMP_OK;
    if (kxs > 32767 || kys > 32767) return IMP_ERROR_INVALID_ARGS;
    if (c == 4 && (((uintptr_t)d | (uintptr_t)step | (uintptr_t)stride) & 3)) return IMP_ERROR_INVALID_ARGS;
    std::vector<int> ikx, iky;
    gaussian_kernel_fixed(kxs, sigma, &ikx);
    gaussian_kernel_fixed(kys, sigma, &iky);
    const int ry = kys / 2;
    // device blob: kx ints | ky float halves | ky int halves
    std::vector<int> blob(ikx);
    const size_t off_f = blob.size();
    for (int k = 0; k <= ry; k++) {
        float f = (float)(iky[ry + k] * (1. / 65536));
        int bits;
        std::memcpy(&bits, &f, 4);
        blob.push_back(bits);
    }
    const size_t off_i = blob.size();
    for (int k = 0; k <= ry; k++) blob.push_back(iky[ry + k]);
    void* dev_k = nullptr;
    if (int rc = upload_small(blob.data(), blob.size() * 4, &dev_k, s)) return rc;
    const int* dkx = (const int*)dev_k;
    const float* dkyf = (const float*)((const int*)dev_k + off_f);
    const int* dkyi = (const int*)dev_k + off_i;

    // int32 intermediate: bounded chunks of frames
    const size_t per_frame = (size_t)w * h * c * 4;
    int chunk = (int)((size_t(1) << 31) / per_frame);
    if (chunk < 1) chunk = 1;
    if (chunk > count) chunk = count;
    if (chunk > 65535) chunk = 65535;
    void* tmp = nullptr;
    if (int rc = dev_alloc_on(per_frame * chunk, &tmp, s)) { dev_free_on(dev_k, s); return rc; }
    int rc = IMP_OK;
    for (int f0 = 0; f0 < count && rc == IMP_OK; f0 += chunk) {
        const int n = count - f0 < chunk ? count - f0 : chunk;
        uint8_t* base = d + (long long)f0 * stride;
        if (c == 4) {
            const dim3 grid((unsigned)(((long long)w * h + 255) / 256), (unsigned)n), block(256);
            hipLaunchKernelGGL(k_blur_row4, grid, block, 0, s, base, stride, w, h, step, dkx, kxs, (int4*)tmp);
            hipLaunchKernelGGL(k_blur_col4, grid, block, 0, s, (const int4*)tmp, w, h, dkyf, ry, base, stride, step);
        } else {
            const dim3 grid((unsigned)(((long long)w * c * h + 255) / 256), (unsigned)n), block(256);
            hipLaunchKernelGGL(k_blur_row_any, grid, block, 0, s, base, stride, w, h, c, step, dkx, kxs, (int*)tmp);
            hipLaunchKernelGGL(k_blur_col_any, grid, block, 0, s, (const int*)tmp, w, h, c, dkyf, dkyi, ry, base, stride, step);
        }
        if (hipGetLastError() != hipSuccess) rc = IMP_ERROR_DEVICE;
    }
    dev_free_on(tmp, s);
    dev_free_on(dev_k, s);
    return rc;
}

}  // namespace imp
