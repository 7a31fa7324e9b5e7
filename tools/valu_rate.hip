// Microbenchmark: sustained issue rate of the integer VALU ops the resize kernels lean on (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed) {
    unsigned a[8];
    for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + 1) + i * 0x9e3779b9u;
    unsigned b = seed ^ 0x1234567u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) a[i] = a[i] * 3u + b;                                        // v_mad_u32_u24 / mul_lo+add (compiler's choice)
                else if (OP == 1) a[i] = __builtin_amdgcn_perm(a[i], b, 0x0c040c00u);
                else if (OP == 2) { s2 x, y; __builtin_memcpy(&x, &a[i], 4); __builtin_memcpy(&y, &b, 4); a[i] = __builtin_amdgcn_sdot2(x, y, (int)a[i], false); }
                else if (OP == 3) a[i] = (unsigned)__mul24((int)a[i], (int)b) + 7u;
                else if (OP == 4) a[i] = a[i] + b;                                        // v_add_u32
                else if (OP == 5) { float f = __uint_as_float(a[i]); f = __fmul_rn(f, 1.0001f); a[i] = __float_as_uint(f); }
            }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, unsigned* d) {
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters * 64;      // lane-ops
    double waveinstr = ops / 64;
    // 1024 SIMDs; report cycles per wave-instruction per SIMD at 2.1 GHz (approx)
    printf("%-22s %8.3f ms  %7.2f Tlane-ops/s  ~%.2f cycles/wave-instr/SIMD @2.1GHz\n", name, ms, ops / ms / 1e9,
           ms * 1e-3 * 2.1e9 * 1024 / waveinstr);
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<4>("v_add_u32", d); run<5>("v_mul_f32", d); run<0>("a*3+b (int)", d); run<1>("v_perm_b32", d);
    run<2>("v_dot2c_i32_i16", d); run<3>("mul24+add", d);
    return 0;
}
