// Which lane holds which element of the A and B operands of v_mfma_i32_16x16x64_i8 (gfx950)?  The guide gives the bf16 maps and
// says "other dtypes: check with exact integer data".  One-hot A against B[k][j] = k tells (row, k) of every (lane, byte) of A;
// one-hot B against A[i][k] = k tells (k, column) of every (lane, byte) of B.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_i8_probe.hip -o tools/_build/mfma_i8_probe && tools/_build/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k_probe(const uint8_t* a_bytes, const uint8_t* b_bytes, int* c_out) {
    const int l = threadIdx.x;
    v4i a, b, c = {0, 0, 0, 0};
    memcpy(&a, a_bytes + l * 16, 16);
    memcpy(&b, b_bytes + l * 16, 16);
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) c_out[l * 4 + r] = c[r];
}

int main() {
    uint8_t *da, *db; int* dc;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dc, 1024);
    std::vector<uint8_t> A(1024), B(1024);
    std::vector<int> C(256);
    auto run = [&]() { hipMemcpy(da, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 1024, hipMemcpyHostToDevice);
                       hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, dc); hipMemcpy(C.data(), dc, 1024, hipMemcpyDeviceToHost); };
    // C layout (guide): C[row = 4*(lane>>4) + reg][col = lane & 15]
    auto Cat = [&](int row, int col) { return C[(col + 16 * (row >> 2)) * 4 + (row & 3)]; };
    // hypothesis for both operands: lane l, byte j <-> (index l & 15, k = 16 * (l >> 4) + j)
    // step 1: B under the hypothesis with B[k][col] = k for every col; A one-hot -> C[row][*] = k
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int j = 0; j < 16; j++) B[l * 16 + j] = (uint8_t)(16 * (l >> 4) + j + 1);        // (k + 1: k = 0 must show too)
    printf("A operand: (lane, byte) -> (row, k)\n");
    for (int l = 0; l < 64; l++) {
        for (int j = 0; j < 16; j++) {
            std::fill(A.begin(), A.end(), 0);
            A[l * 16 + j] = 1;
            run();
            int row = -1, k = -1, hits = 0;
            for (int r = 0; r < 16; r++) { bool any = false; for (int c = 0; c < 16; c++) if (Cat(r, c)) any = true; if (any) { row = r; k = Cat(r, 0); hits++; } }
            const bool ok = hits == 1 && row == (l & 15) && k - 1 == 16 * (l >> 4) + j;
            if (!ok) { bad++; if (bad < 20) printf("  lane %d byte %d -> row %d k %d (rows hit %d)\n", l, j, row, k, hits); }
        }
    }
    printf("A map %s the hypothesis row = lane & 15, k = 16 * (lane >> 4) + byte (given B's)\n", bad ? "DIFFERS from" : "matches");
    // step 2: A[row][k] = k under the hypothesis; B one-hot -> C[*][col] = k
    int bad2 = 0;
    for (int l = 0; l < 64; l++) for (int j = 0; j < 16; j++) A[l * 16 + j] = (uint8_t)(16 * (l >> 4) + j + 1);
    for (int l = 0; l < 64; l++) {
        for (int j = 0; j < 16; j++) {
            std::fill(B.begin(), B.end(), 0);
            B[l * 16 + j] = 1;
            run();
            int col = -1, k = -1, hits = 0;
            for (int c = 0; c < 16; c++) { bool any = false; for (int r = 0; r < 16; r++) if (Cat(r, c)) any = true; if (any) { col = c; k = Cat(0, c); hits++; } }
            const bool ok = hits == 1 && col == (l & 15) && k - 1 == 16 * (l >> 4) + j;
            if (!ok) { bad2++; if (bad2 < 20) printf("  B lane %d byte %d -> col %d k %d (cols hit %d)\n", l, j, col, k, hits); }
        }
    }
    printf("B map %s the hypothesis col = lane & 15, k = 16 * (lane >> 4) + byte\n", bad2 ? "DIFFERS from" : "matches");
    // step 3: signed bytes: (-128) * 127 summed over one k
    std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0);
    A[0] = 0x80; B[0] = 0x7f;
    run();
    printf("signed check: (-128) * 127 = %d (C[0][0])\n", Cat(0, 0));
    return bad || bad2;
}
