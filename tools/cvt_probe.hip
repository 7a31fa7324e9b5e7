// Does v_cvt_pk_u8_f32 round to nearest even and saturate (what cvRound + saturate_cast<uchar> do)?  Checked over a grid
// of values incl. exact .5 ties, negatives, > 255, and random floats.  hipcc --offload-arch=gfx950 tools/cvt_probe.hip -o /tmp/cvt_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const float* in, unsigned* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned r = 0;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(r) : "v"(in[i]), "v"(0u));   // byte 1
    out[i] = r;
}
int main() {
    std::vector<float> v;
    for (int i = -1200; i <= 1200; i++) v.push_back(i * 0.25f);
    for (int i = 0; i < 256; i++) { v.push_back(i + 0.5f); v.push_back(std::nextafterf(i + 0.5f, 0.f)); v.push_back(std::nextafterf(i + 0.5f, 1e9f)); }
    srand(1);
    for (int i = 0; i < 100000; i++) v.push_back((float)rand() / RAND_MAX * 300.f - 20.f);
    float *di; unsigned* dout; int n = (int)v.size();
    hipMalloc(&di, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(di, v.data(), n * 4, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(di, dout, n);
    std::vector<unsigned> o(n);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        long w = lrintf(v[i]);
        w = w < 0 ? 0 : (w > 255 ? 255 : w);
        if (o[i] != ((unsigned)w << 8)) { if (bad < 10) printf("x=%.9g got 0x%x want %ld\n", v[i], o[i], w); bad++; }
    }
    printf("v_cvt_pk_u8_f32: %d of %d differ from sat_u8(lrintf(x)) placed in byte 1\n", bad, n);
    return 0;
}
