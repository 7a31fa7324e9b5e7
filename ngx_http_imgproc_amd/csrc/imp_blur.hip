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
#include <cmath>
#include <cstring>
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
    if (int rc = dev_alloc(per_frame * chunk, &tmp)) { dev_free(dev_k); return rc; }
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
    if (s != env_stream()) (void)hipStreamSynchronize(s);
    dev_free(tmp);
    dev_free(dev_k);
    return rc;
}

}  // namespace imp
