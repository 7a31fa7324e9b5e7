// Can v_fma_mix_f32 stand in for v_cvt_f32_ubyteN + v_mul_f32?  A byte b zero-extended into a 16-bit half IS the f16
// denormal b * 2^-24, and fma_mix(f16 b, f32 w * 2^24, 0) rounds b * w once -- the same single rounding as
// v_mul_f32(float(b), w) (scaling by a power of two commutes with rounding while nothing under/overflows).
//   hipcc --offload-arch=gfx950 -O3 tools/mix_probe.hip -o /tmp/mix_probe && /tmp/mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void check(const float* w, int nw, unsigned* bad) {
    const int b = threadIdx.x;                       // byte value
    for (int i = blockIdx.x; i < nw; i += gridDim.x) {
        const float ref = __fmul_rn((float)b, w[i]);
        const float ws = w[i] * 16777216.f;          // exact
        const unsigned lo = (unsigned)b, hi = (unsigned)b << 16;
        float m0, m1;
        asm volatile("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(m0) : "v"(lo), "v"(ws));
        asm volatile("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(m1) : "v"(hi), "v"(ws));
        if (__float_as_uint(m0) != __float_as_uint(ref) || __float_as_uint(m1) != __float_as_uint(ref)) atomicAdd(bad, 1u);
    }
}

template <int OP>
__global__ __launch_bounds__(256) void rate(unsigned* out, int iters, unsigned seed) {
    unsigned a[8];
    for (int i = 0; i < 8; i++) a[i] = (seed * (threadIdx.x + 1) + i) & 0x00ff00ffu;
    float b = 1.5f + seed;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (OP == 0) asm volatile("v_fma_mix_f32 %0, %0, %1, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 1) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]));
                if constexpr (OP == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 3) asm volatile("v_fma_mix_f32 %0, %0, %1, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 4) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[i]));
            }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static void run(const char* name, unsigned* d) {
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks * 4 waves * iters * 64 / 1024 SIMDs
    const double per_simd = (double)blocks * 4 * iters * 64 / 1024.0;
    std::printf("%-28s %8.3f ms  ~%.2f cycles/wave-instr/SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
}

int main() {
    std::vector<float> w;
    for (int i = 1; i <= 4000; i++) { w.push_back(1.0f / i); w.push_back((float)(1.0 / (i * 0.731 + 0.004))); w.push_back(i * 1e-4f); }
    w.push_back(1.f); w.push_back(0.001f); w.push_back(1e-3f * 1.0001f); w.push_back(0.99999994f);
    float* dw; unsigned* bad; unsigned* out;
    hipMalloc(&dw, w.size() * 4); hipMalloc(&bad, 4); hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(check, dim3(512), dim3(256), 0, 0, dw, (int)w.size(), bad);
    unsigned hb = 1;
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    std::printf("fma_mix(byte as f16 denormal, w * 2^24, 0) == float(byte) * w : %s (%u mismatches over %zu weights x 256 bytes x 2 halves)\n",
                hb ? "NO" : "yes, bit for bit", hb, w.size());
    run<0>("v_fma_mix_f32 (f16, f32, 0)", out);
    run<3>("v_fma_mix_f32 (f16 hi, f32, f32)", out);
    run<1>("v_cvt_f32_ubyte0", out);
    run<2>("v_mul_f32", out);
    run<4>("v_cvt_f32_f16", out);
    return hb ? 1 : 0;
}
