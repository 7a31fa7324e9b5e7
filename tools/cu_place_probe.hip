// Where do the workgroups of SMALL kernels land when several streams launch them at the same time -- on the same compute
// units, or spread over the device?  And what does a CU mask on the stream (hipExtStreamCreateWithCUMask) do to that?
// (Round 5: the broker's lanes run latency-bound chains of small kernels side by side; each chain gets slower with every
// other chain in flight.)  Every workgroup records (XCC id, SE, CU) and spins ~20 us so that the kernels overlap.
//   hipcc --offload-arch=gfx950 -O3 tools/cu_place_probe.hip -o /tmp/cu_place_probe && /tmp/cu_place_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <set>
#include <map>

__global__ __launch_bounds__(256) void k_where(unsigned* out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}

static void report(const char* what, const std::vector<std::vector<unsigned>>& res) {
    std::map<unsigned, int> per_cu;                       // (xcc, se, sh, cu) -> workgroups
    for (size_t s = 0; s < res.size(); s++) {
        std::set<unsigned> cus, xccs;
        for (size_t b = 0; b * 2 < res[s].size(); b++) {
            const unsigned xcc = res[s][2 * b] & 15, hw = res[s][2 * b + 1];
            const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            cus.insert(key); xccs.insert(xcc); per_cu[key]++;
        }
        printf("  %s stream %zu: %zu workgroups on %zu distinct CUs of %zu XCDs\n", what, s, res[s].size() / 2, cus.size(), xccs.size());
    }
    int shared = 0, most = 0;
    for (auto& kv : per_cu) { if (kv.second > 1) shared++; if (kv.second > most) most = kv.second; }
    printf("  %s: %zu CUs used in all, %d of them by more than one workgroup (most on one CU: %d)\n", what, per_cu.size(), shared, most);
}

int main() {
    const int NS = 4, WG = 64, SPIN = 2000;              // 100 MHz clock: 20 us
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("%s: %d CUs\n", prop.name, prop.multiProcessorCount);
    unsigned* d[NS];
    for (int s = 0; s < NS; s++) hipMalloc(&d[s], WG * 8);
    for (int mode = 0; mode < 3; mode++) {
        hipStream_t st[NS];
        for (int s = 0; s < NS; s++) {
            if (mode == 0) hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
            else {
                // 256 CUs = 8 words of mask.  mode 1: a contiguous quarter of the bits; mode 2: every bit whose index is s mod 4
                unsigned mask[8];
                memset(mask, 0, sizeof mask);
                for (int i = 0; i < 256; i++) {
                    const bool mine = mode == 1 ? (i / 64 == s) : (i % 4 == s);
                    if (mine) mask[i / 32] |= 1u << (i % 32);
                }
                if (hipExtStreamCreateWithCUMask(&st[s], 8, mask) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); return 1; }
            }
        }
        for (int rep = 0; rep < 3; rep++)
            for (int s = 0; s < NS; s++) hipLaunchKernelGGL(k_where, dim3(WG), dim3(256), 0, st[s], d[s], SPIN);
        std::vector<std::vector<unsigned>> res(NS, std::vector<unsigned>(WG * 2));
        for (int s = 0; s < NS; s++) { hipStreamSynchronize(st[s]); hipMemcpy(res[s].data(), d[s], WG * 8, hipMemcpyDeviceToHost); }
        const char* names[3] = {"no mask", "mask = contiguous quarter", "mask = every 4th bit"};
        printf("%s, %d streams x %d workgroups of 256 at the same time:\n", names[mode], NS, WG);
        report(names[mode], res);
        // per stream: which XCDs
        for (int s = 0; s < NS; s++) {
            std::map<unsigned, int> x;
            for (int b = 0; b < WG; b++) x[res[s][2 * b] & 15]++;
            printf("    stream %d XCDs:", s);
            for (auto& kv : x) printf(" %u:%d", kv.first, kv.second);
            printf("\n");
        }
        for (int s = 0; s < NS; s++) hipStreamDestroy(st[s]);
    }
    return 0;
}
