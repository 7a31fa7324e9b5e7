// imp_geom.hip -- exact byte movers: the copy behind Crop (cvSetImageROI + cvCopy,
// bridge.c:130-135), cvFlip (filters.c:93-99, :126), cvTranspose + cvFlip as one rotation
// (filters.c:116-119) and cvCvtColor(GRAY2BGR) (bridge.c:613-618).  Bit-exact by nature.
//
// Rotation by 90 / 270 of BGRA goes through a 32x32-pixel LDS tile (33-dword row pitch, so
// the transposed read is bank-conflict free): global reads run along source rows, global
// writes along destination rows, both coalesced, in one pass instead of the reference's
// transpose + flip pair.
#include "imp_internal.h"

namespace imp {

struct GArgs {
    const uint8_t* src; long long src_stride; int sstep, sw, sh;
    uint8_t* dst; long long dst_stride; int dstep, dw, dh;
};

// ---- copy: `unit` bytes per thread (4 when everything is dword aligned, else 1) ----
template <int UNIT>
__global__ __launch_bounds__(256) void k_copy(GArgs a, int row_units) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)row_units * a.dh) return;
    const int y = (int)(idx / row_units), u = (int)(idx - (long long)y * row_units);
    const uint8_t* s = a.src + (long long)blockIdx.y * a.src_stride + (size_t)y * a.sstep + (size_t)u * UNIT;
    uint8_t* d = a.dst + (long long)blockIdx.y * a.dst_stride + (size_t)y * a.dstep + (size_t)u * UNIT;
    if (UNIT == 4) *(uint32_t*)d = *(const uint32_t*)s;
    else *d = *s;
}

// ---- cvFlip: mode 0 = vertical (around x axis), > 0 = horizontal, < 0 = both ----
template <int CN>
__global__ __launch_bounds__(256) void k_flip(GArgs a, int mode) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.dw * a.dh) return;
    const int y = (int)(idx / a.dw), x = (int)(idx - (long long)y * a.dw);
    const int sy = mode <= 0 ? a.sh - 1 - y : y;
    const int sx = mode != 0 ? a.sw - 1 - x : x;
    const uint8_t* s = a.src + (long long)blockIdx.y * a.src_stride + (size_t)sy * a.sstep + (size_t)sx * CN;
    uint8_t* d = a.dst + (long long)blockIdx.y * a.dst_stride + (size_t)y * a.dstep + (size_t)x * CN;
    if (CN == 4) *(uint32_t*)d = *(const uint32_t*)s;
    else {
#pragma unroll
        for (int c = 0; c < CN; c++) d[c] = s[c];
    }
}

// ---- rotate 90 (clockwise): R[i][j] = S[H-1-j][i];  270: R[i][j] = S[j][W-1-i]  (filters.c:116-119) ----
__global__ __launch_bounds__(256) void k_rotate_bgra(GArgs a, int amount) {
    __shared__ uint32_t tile[32][33];
    const int tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 32;   // destination tile origin (x along dw = sh)
    const uint8_t* S = a.src + (long long)blockIdx.z * a.src_stride;
    uint8_t* D = a.dst + (long long)blockIdx.z * a.dst_stride;
    const int li = threadIdx.x;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int lj = threadIdx.y + 8 * r;
        const int j = tx0 + lj, i = ty0 + li;                  // destination (x = j, y = i)
        if (j < a.dw && i < a.dh) {
            const int srow = amount == 90 ? a.sh - 1 - j : j;
            const int scol = amount == 90 ? i : a.sw - 1 - i;
            tile[lj][li] = *(const uint32_t*)(S + (size_t)srow * a.sstep + (size_t)scol * 4);
        }
    }
    __syncthreads();
    const int lj = threadIdx.x;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int l = threadIdx.y + 8 * r;
        const int j = tx0 + lj, i = ty0 + l;
        if (j < a.dw && i < a.dh) *(uint32_t*)(D + (size_t)i * a.dstep + (size_t)j * 4) = tile[lj][l];
    }
}

// The same for 3-channel frames (every JPEG): a 32 x 32 pixel tile is 32 source row segments of 96 bytes.  Full,
// dword-aligned tiles move as dwords on both sides (24 per row segment) and the quarter turn happens in the byte
// gather out of LDS; border tiles and odd alignments take bytes on the global side.  LDS row pitch 100 bytes.
__global__ __launch_bounds__(256) void k_rotate_bgr(GArgs a, int amount) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[32][100];
    const int tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 32;   // destination tile origin (x along dw = sh, y along dh = sw)
    const uint8_t* S = a.src + (long long)blockIdx.z * a.src_stride;
    uint8_t* D = a.dst + (long long)blockIdx.z * a.dst_stride;
    const int tid = threadIdx.y * 32 + threadIdx.x;
    const bool full = tx0 + 32 <= a.dw && ty0 + 32 <= a.dh;
    // source tile: rows srow0 .. srow0+31, columns scol0 .. scol0+31 (tile[r][3*c + ch])
    const int srow0 = amount == 90 ? a.sh - 32 - tx0 : tx0;
    const int scol0 = amount == 90 ? ty0 : a.sw - 32 - ty0;
    const bool vec = full && !(((uintptr_t)S | (uintptr_t)D | (uintptr_t)a.sstep | (uintptr_t)a.dstep | (uintptr_t)(scol0 * 3)) & 3);
    if (vec) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int e = tid + 256 * q;                       // 768 dwords: 32 rows x 24
            const int r = e / 24, d = e - r * 24;
            *(uint32_t*)&tile[r][4 * d] = *(const uint32_t*)(S + (size_t)(srow0 + r) * a.sstep + (size_t)scol0 * 3 + 4 * d);
        }
    } else {
        for (int e = tid; e < 32 * 32; e += 256) {
            const int r = e >> 5, c = e & 31;
            const int sr = srow0 + r, sc = scol0 + c;
            if (sr >= 0 && sr < a.sh && sc >= 0 && sc < a.sw) {
                const uint8_t* p = S + (size_t)sr * a.sstep + (size_t)sc * 3;
                tile[r][3 * c] = p[0]; tile[r][3 * c + 1] = p[1]; tile[r][3 * c + 2] = p[2];
            }
        }
    }
    __syncthreads();
    // destination pixel (row ty0 + li, column tx0 + lj) = tile[31 - lj][li] (90)  or  tile[lj][31 - li] (270)
    if (vec) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int e = tid + 256 * q;
            const int li = e / 24, d = e - li * 24;
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int byte = 4 * d + b, lj = byte / 3, ch = byte - 3 * lj;
                const uint8_t v = amount == 90 ? tile[31 - lj][3 * li + ch] : tile[lj][3 * (31 - li) + ch];
                o |= (uint32_t)v << (8 * b);
            }
            *(uint32_t*)(D + (size_t)(ty0 + li) * a.dstep + (size_t)tx0 * 3 + 4 * d) = o;
        }
    } else {
        for (int e = tid; e < 32 * 32; e += 256) {
            const int li = e >> 5, lj = e & 31;
            if (ty0 + li < a.dh && tx0 + lj < a.dw) {
                const uint8_t* t = amount == 90 ? &tile[31 - lj][3 * li] : &tile[lj][3 * (31 - li)];
                uint8_t* p = D + (size_t)(ty0 + li) * a.dstep + (size_t)(tx0 + lj) * 3;
                p[0] = t[0]; p[1] = t[1]; p[2] = t[2];
            }
        }
    }
}

template <int CN>
__global__ __launch_bounds__(256) void k_rotate_any(GArgs a, int amount) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.dw * a.dh) return;
    const int i = (int)(idx / a.dw), j = (int)(idx - (long long)i * a.dw);
    const int srow = amount == 90 ? a.sh - 1 - j : j;
    const int scol = amount == 90 ? i : a.sw - 1 - i;
    const uint8_t* s = a.src + (long long)blockIdx.y * a.src_stride + (size_t)srow * a.sstep + (size_t)scol * CN;
    uint8_t* d = a.dst + (long long)blockIdx.y * a.dst_stride + (size_t)i * a.dstep + (size_t)j * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) d[c] = s[c];
}

__global__ __launch_bounds__(256) void k_gray2bgr(GArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.dw * a.dh) return;
    const int y = (int)(idx / a.dw), x = (int)(idx - (long long)y * a.dw);
    const uint8_t v = a.src[(long long)blockIdx.y * a.src_stride + (size_t)y * a.sstep + x];
    uint8_t* d = a.dst + (long long)blockIdx.y * a.dst_stride + (size_t)y * a.dstep + (size_t)x * 3;
    d[0] = v; d[1] = v; d[2] = v;
}

// IplToFI32 / IplToFI24 (advancedio.c:65-101): vertical flip + channel repack into FreeImage's row pitch
template <int SC, int DC>
__global__ __launch_bounds__(256) void k_pack_fi(const uint8_t* __restrict__ src, int sstep, int w, int h, uint8_t* __restrict__ dst, int dpitch) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)w * h) return;
    const int y = (int)(idx / w), x = (int)(idx - (long long)y * w);
    const uint8_t* s = src + (size_t)(h - 1 - y) * sstep + (size_t)x * SC;
    uint8_t* d = dst + (size_t)y * dpitch + (size_t)x * DC;
    if (SC == 4 && DC == 4) { *(uint32_t*)d = *(const uint32_t*)s; return; }
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
    if (DC == 4) d[3] = SC == 4 ? s[3] : 255;
}

int launch_pack_fi(const View& v, int bpp, uint8_t* dst, int dpitch, hipStream_t s) {
    if (v.c < 3 || (bpp != 24 && bpp != 32)) return IMP_ERROR_INVALID_ARGS;
    const dim3 grid((unsigned)(((long long)v.w * v.h + 255) / 256)), block(256);
    if (v.c == 4 && bpp == 32) {
        if (((uintptr_t)v.d | (uintptr_t)v.step | (uintptr_t)dst | (uintptr_t)dpitch) & 3) return IMP_ERROR_INVALID_ARGS;
        hipLaunchKernelGGL((k_pack_fi<4, 4>), grid, block, 0, s, v.d, v.step, v.w, v.h, dst, dpitch);
    } else if (v.c == 4) hipLaunchKernelGGL((k_pack_fi<4, 3>), grid, block, 0, s, v.d, v.step, v.w, v.h, dst, dpitch);
    else if (bpp == 32) hipLaunchKernelGGL((k_pack_fi<3, 4>), grid, block, 0, s, v.d, v.step, v.w, v.h, dst, dpitch);
    else hipLaunchKernelGGL((k_pack_fi<3, 3>), grid, block, 0, s, v.d, v.step, v.w, v.h, dst, dpitch);
    IMP_HIP(hipGetLastError());
    return IMP_OK;
}

// LoadGIF's compositing loop (advancedio.c:204-247), one lane per canvas pixel walking the pages in order: the
// `master` index canvas of the destructive mode is that pixel's register.  Same definitions as the oracle where the
// reference is undefined (read one byte past the frame row at x == left + w; index < 0 -> colour 0).
__global__ __launch_bounds__(256) void k_gif_compose(const uint8_t* __restrict__ blob, const GifPageDev* __restrict__ pg,
                                                     uint8_t* const* __restrict__ outs, int npages, int cw, int ch,
                                                     int ostep, int destructive, int only) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)cw * ch) return;
    const int y = (int)(idx / cw), x = (int)(idx - (long long)y * cw);
    int master = 0;
    for (int f = 0; f < npages; f++) {
        const GifPageDev p = pg[f];
        const int rowidx = p.h + p.top - y - 1;
        int coloridx;
        if (rowidx < 0 || x < p.left || y < p.top || x > p.left + p.w || y > p.top + p.h) {
            coloridx = p.key;
        } else {
            const long long o = (long long)rowidx * p.pitch + (x - p.left);
            coloridx = o < (long long)p.pitch * p.h ? (int)blob[p.idx_off + o] : p.key;
        }
        if (destructive) {
            if (p.dispose == 2) {
                if (coloridx == p.key) coloridx = 0;
                else master = coloridx;
            } else {
                if (coloridx == p.key && f > 0) coloridx = master;
                else master = coloridx;
            }
        }
        if (only >= 0 && f != only) continue;
        uint32_t px = 0;
        if (coloridx >= 0 && coloridx < 256) {
            const uint8_t* q = blob + p.pal_off + 4 * coloridx;
            px = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16);
        }
        if (coloridx != p.key) px |= 0xff000000u;
        *(uint32_t*)(outs[only >= 0 ? 0 : f] + (size_t)y * ostep + (size_t)x * 4) = px;
    }
}

int launch_gif_compose(const uint8_t* blob, const GifPageDev* pages, uint8_t* const* outs, int npages, int cw, int ch,
                       int ostep, int destructive, int only, hipStream_t s) {
    const dim3 grid((unsigned)(((long long)cw * ch + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_gif_compose, grid, block, 0, s, blob, pages, outs, npages, cw, ch, ostep, destructive, only);
    IMP_HIP(hipGetLastError());
    return IMP_OK;
}

static GArgs gargs(const Frames& f) {
    return GArgs{f.src, f.src_stride, f.v.step, f.v.w, f.v.h, f.dst, f.dst_stride, f.dstep, f.dw, f.dh};
}
static bool dword_ok(const Frames& f) {
    return !(((uintptr_t)f.src | (uintptr_t)f.dst | (uintptr_t)f.v.step | (uintptr_t)f.dstep |
              (uintptr_t)f.src_stride | (uintptr_t)f.dst_stride) & 3);
}
static unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

int launch_copy(const Frames& f, hipStream_t s) {
    if (f.count <= 0) return IMP_OK;
    if (f.dw != f.v.w || f.dh != f.v.h || f.count > 65535) return IMP_ERROR_INVALID_ARGS;
    GArgs a = gargs(f);
    const int rowbytes = f.dw * f.v.c;
    if (dword_ok(f) && rowbytes % 4 == 0) {
        const int units = rowbytes / 4;
        hipLaunchKernelGGL((k_copy<4>), dim3(blocks_for((long long)units * f.dh), f.count), dim3(256), 0, s, a, units);
    } else {
        hipLaunchKernelGGL((k_copy<1>), dim3(blocks_for((long long)rowbytes * f.dh), f.count), dim3(256), 0, s, a, rowbytes);
    }
    IMP_HIP(hipGetLastError());
    return IMP_OK;
}

int launch_flip(const Frames& f, int mode, hipStream_t s) {
    if (f.count <= 0) return IMP_OK;
    if (f.dw != f.v.w || f.dh != f.v.h || f.count > 65535) return IMP_ERROR_INVALID_ARGS;
    GArgs a = gargs(f);
    const dim3 grid(blocks_for((long long)f.dw * f.dh), f.count), block(256);
    if (f.v.c == 4 && dword_ok(f)) hipLaunchKernelGGL((k_flip<4>), grid, block, 0, s, a, mode);
    else if (f.v.c == 4) return IMP_ERROR_INVALID_ARGS;
    else if (f.v.c == 3) hipLaunchKernelGGL((k_flip<3>), grid, block, 0, s, a, mode);
    else hipLaunchKernelGGL((k_flip<1>), grid, block, 0, s, a, mode);
    IMP_HIP(hipGetLastError());
    return IMP_OK;
}

int launch_rotate(const Frames& f, int amount, hipStream_t s) {
    if (f.count <= 0) return IMP_OK;
    if (amount == 180) return launch_flip(f, -1, s);
    if ((amount != 90 && amount != 270) || f.dw != f.v.h || f.dh != f.v.w || f.count > 65535) return IMP_ERROR_INVALID_ARGS;
    GArgs a = gargs(f);
    if (f.v.c == 4) {
        if (!dword_ok(f)) return IMP_ERROR_INVALID_ARGS;
        const dim3 grid((f.dw + 31) / 32, (f.dh + 31) / 32, f.count), block(32, 8);
        if (grid.y > 65535) return IMP_ERROR_INVALID_ARGS;
        hipLaunchKernelGGL(k_rotate_bgra, grid, block, 0, s, a, amount);
    } else if (f.v.c == 3 && (f.dh + 31) / 32 <= 65535) {
        const dim3 grid((f.dw + 31) / 32, (f.dh + 31) / 32, f.count), block(32, 8);
        hipLaunchKernelGGL(k_rotate_bgr, grid, block, 0, s, a, amount);
    } else {
        const dim3 grid(blocks_for((long long)f.dw * f.dh), f.count), block(256);
        if (f.v.c == 3) hipLaunchKernelGGL((k_rotate_any<3>), grid, block, 0, s, a, amount);
        else hipLaunchKernelGGL((k_rotate_any<1>), grid, block, 0, s, a, amount);
    }
    IMP_HIP(hipGetLastError());
    return IMP_OK;
}

int launch_gray2bgr(const Frames& f, hipStream_t s) {
    if (f.count <= 0) return IMP_OK;
    if (f.v.c != 1 || f.dw != f.v.w || f.dh != f.v.h || f.count > 65535) return IMP_ERROR_INVALID_ARGS;
    GArgs a = gargs(f);
    hipLaunchKernelGGL(k_gray2bgr, dim3(blocks_for((long long)f.dw * f.dh), f.count), dim3(256), 0, s, a);
    IMP_HIP(hipGetLastError());
    return IMP_OK;
}

}  // namespace imp
