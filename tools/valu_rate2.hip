// Microbenchmark 2: sustained issue rate of individual gfx950 VALU instructions (inline asm, so the
// instruction measured is exactly the one named).  8 independent chains per lane, 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate2.hip -o /tmp/valu_rate2 && /tmp/valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>

#define OPS(X)                                                                                                  \
    X(0, "v_add_u32", "v_add_u32 %0, %0, %1")                                                                   \
    X(1, "v_add_u32_sdwa B1", "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1") \
    X(2, "v_mul_i32_i24_sdwa B1", "v_mul_i32_i24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") \
    X(3, "v_mad_i32_i24", "v_mad_i32_i24 %0, %0, %1, %0")                                                       \
    X(4, "v_mul_i32_i24", "v_mul_i32_i24 %0, %0, %1")                                                           \
    X(5, "v_add3_u32", "v_add3_u32 %0, %0, %1, %1")                                                             \
    X(6, "v_perm_b32", "v_perm_b32 %0, %0, %1, %1")                                                             \
    X(7, "v_dot2c_i32_i16", "v_dot2c_i32_i16 %0, %1, %1")                                                       \
    X(8, "v_mad_i32_i16", "v_mad_i32_i16 %0, %0, %1, %0 op_sel:[1,0,0,0]")                                      \
    X(9, "v_mad_u32_u16", "v_mad_u32_u16 %0, %0, %1, %0 op_sel:[1,0,0,0]")                                      \
    X(10, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1")                                                            \
    X(11, "v_pk_lshrrev_b16", "v_pk_lshrrev_b16 %0, 8, %0")                                                     \
    X(12, "v_and_b32", "v_and_b32 %0, %0, %1")                                                                  \
    X(13, "v_bfe_u32", "v_bfe_u32 %0, %0, 8, 8")                                                                \
    X(14, "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 %0, %0")                                                        \
    X(15, "v_fma_f32", "v_fma_f32 %0, %0, %1, %0")                                                              \
    X(16, "v_mul_f32", "v_mul_f32 %0, %0, %1")                                                                  \
    X(17, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %0")                                                      \
    X(18, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1")                                                            \
    X(19, "v_pk_mad_u16", "v_pk_mad_u16 %0, %0, %1, %0")                                                        \
    X(20, "v_pk_mul_lo_u16", "v_pk_mul_lo_u16 %0, %0, %1")                                                      \
    X(21, "v_dot4_u32_u8", "v_dot4_u32_u8 %0, %0, %1, %0")                                                      \
    X(22, "v_dot2_i32_i16", "v_dot2_i32_i16 %0, %0, %1, %0")                                                    \
    X(23, "v_med3_i32", "v_med3_i32 %0, %0, %1, %1")                                                            \
    X(24, "v_lshrrev_b32", "v_lshrrev_b32 %0, 3, %0")                                                           \
    X(25, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %0")                                                              \
    X(26, "v_sad_u8", "v_sad_u8 %0, %0, %1, %0")                                                                \
    X(27, "v_mov_b32", "v_mov_b32 %0, %1")                                                                      \
    X(28, "v_mul_u32_u24_sdwa B1W0", "v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:WORD_0") \
    X(29, "v_lshlrev_b32_sdwa B1", "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")

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
#define X(ID, NAME, ASM) if constexpr (OP == ID) asm volatile(ASM : "+v"(a[i]) : "v"(b));
                OPS(X)
#undef X
            }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// packed fp32 (64-bit operands)
template <int OP>
__global__ __launch_bounds__(256) void k2(unsigned* out, int iters, unsigned seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8];
    for (int i = 0; i < 8; i++) { a[i].x = (float)(seed * (threadIdx.x + 1) + i); a[i].y = a[i].x + 1.f; }
    f2 b; b.x = 1.0001f; b.y = 0.9999f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (OP == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
                else if constexpr (OP == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
            }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)s;
}

static void report(const char* name, float ms, int blocks, int iters) {
    const double waveinstr = (double)blocks * 4 * iters * 64;
    printf("%-26s %8.3f ms  ~%.2f cycles/wave-instr/SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / waveinstr);
}
template <typename F>
static float timeit(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(10); hipDeviceSynchronize();
    hipEventRecord(e0); launch(2000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    const int blocks = 256 * 8;
    unsigned* d; hipMalloc(&d, blocks * 256 * 4);
#define X(ID, NAME, ASM) report(NAME, timeit([&](int it) { hipLaunchKernelGGL((k<ID>), dim3(blocks), dim3(256), 0, 0, d, it, 3u); }), blocks, 2000);
    OPS(X)
#undef X
    report("v_pk_fma_f32", timeit([&](int it) { hipLaunchKernelGGL((k2<0>), dim3(blocks), dim3(256), 0, 0, d, it, 3u); }), blocks, 2000);
    report("v_pk_mul_f32", timeit([&](int it) { hipLaunchKernelGGL((k2<1>), dim3(blocks), dim3(256), 0, 0, d, it, 3u); }), blocks, 2000);
    report("v_lshl_add_u64", timeit([&](int it) { hipLaunchKernelGGL((k2<2>), dim3(blocks), dim3(256), 0, 0, d, it, 3u); }), blocks, 2000);
    return 0;
}
