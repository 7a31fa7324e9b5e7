// What does v_ashr_pk_u8_i32 do on gfx950, exactly?  (The compiler's own use of it assumed the upper half is cleared.)
//   hipcc --offload-arch=gfx950 -O3 tools/pk_probe.hip -o _build/pk_probe && _build/pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const int* in, unsigned* out, int n, int sh) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
    unsigned lo_only = 0xdeadbeefu, both = 0xdeadbeefu;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "+v"(lo_only) : "v"(a), "v"(b), "v"(sh));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "+v"(both) : "v"(a), "v"(b), "v"(sh));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(both) : "v"(c), "v"(d), "v"(sh));
    out[2 * i] = lo_only;
    out[2 * i + 1] = both;
}
static unsigned sat(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
int main() {
    const int n = 1 << 16, sh = 22;
    std::vector<int> h(4 * n);
    srand(1);
    for (int& v : h) {
        const int r = rand() % 4;
        v = r == 0 ? (rand() % 512 - 128) << sh : r == 1 ? (int)((unsigned)rand() * 2654435761u) : r == 2 ? -(rand() % (1 << 30)) : rand();
    }
    int* din; unsigned* dout;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 2 * n * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, din, dout, n, sh);
    std::vector<unsigned> o(2 * n);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    long bad_lo = 0, bad_keep = 0, bad_both = 0;
    for (int i = 0; i < n; i++) {
        const unsigned lo = sat(h[4 * i] >> sh) | (sat(h[4 * i + 1] >> sh) << 8);
        const unsigned hi = sat(h[4 * i + 2] >> sh) | (sat(h[4 * i + 3] >> sh) << 8);
        if ((o[2 * i] & 0xffffu) != lo) bad_lo++;
        if ((o[2 * i] >> 16) != 0xdeadu) bad_keep++;
        if (o[2 * i + 1] != (lo | (hi << 16))) bad_both++;
    }
    printf("low half = {sat(S1>>sh), sat(S0>>sh)}: %ld mismatches; upper half preserved: %ld mismatches; "
           "op_sel hi writes the upper half: %ld mismatches (of %d)\n", bad_lo, bad_keep, bad_both, n);
    return (bad_lo || bad_keep || bad_both) ? 1 : 0;
}
